#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_gpu_parity.py tests/test_full_size_gpu.py tests/test_trainer_gpu.py tests/test_edge_cases_gpu.py -q -x -m gpu 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('default ms', j['ms_per_step'], j['mfma_arith']['terms'], j['roofline']['kernel'], j['roofline']['frac'], j['mfma_classes_warmup'].get('gemm_tn'))"
UR_MFMA_ARITH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('exact ms', j['ms_per_step'], j['mfma_arith']['terms'], j['roofline']['kernel'], j['roofline']['frac'], j['mfma_classes_warmup'].get('gemm_tn'))"
