#!/bin/bash
# round 6, stage A (GPU box): split-bf16 dW kernel -- error table, unit tests under the split, headline A/B, per-kernel durations
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/split_bf16_stage_a.py 21248 $out/r06a_stage_a.json > $out/r06a_stage_a.txt 2>&1
tail -50 $out/r06a_stage_a.txt
for m in 6 9; do
  echo "=== tests/test_gemm_gpu.py -k gemm_tn under UR_TEST=tn_split=$m"
  UR_TEST=tn_split=$m timeout 600 python -m pytest tests/test_gemm_gpu.py -k gemm_tn -q -x 2>&1 | tail -5
done
for m in 0 6 0 6; do
  UR_TEST=tn_split=$m python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('tn_split=$m headline ms', j['ms_per_step'], 'fit', j.get('trainer_fit',{}).get('ms_per_step'), 'roofline', j['roofline'].get('class'), j['roofline'].get('frac'))"
done
bash tools/kstats.sh r06a_k0 UR_TEST=tn_split=0 -- 2>&1 | head -14
bash tools/kstats.sh r06a_k6 UR_TEST=tn_split=6 -- 2>&1 | head -14
