#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "gemm_tn or split_bf16" 2>&1 | tail -2
UR_TEST=tn_split_ts=64 timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "gemm_tn or split_bf16" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_gpu.py -q -x 2>&1 | tail -2
for m in 0 6; do
  UR_MFMA_ARITH=$m python bench.py --all-configs --no-cpu-baseline --no-gather-bench --no-extra-legs --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('UR_MFMA_ARITH=$m headline', j['ms_per_step'], ' '.join(k+'='+str(v['ms_per_step'])+' tn='+str(v['kernel_time_ms_per_step'].get('gemm_tn')) for k,v in j['other_configs'].items()))"
done
