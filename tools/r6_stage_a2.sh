#!/bin/bash
# round 6, stage A second pass (GPU box): 8-wave split kernel -- group timing, unit tests, headline A/B (prefetch depth 1 / 2), kernel stats
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/split_bf16_stage_a.py 21248 $out/r06b_stage_a.json > $out/r06b_stage_a.txt 2>&1
head -8 $out/r06b_stage_a.txt; grep " 6 \| 0 " $out/r06b_stage_a.txt | grep randn
echo "=== tests/test_gemm_gpu.py -k gemm_tn under UR_TEST=tn_split=6"
UR_TEST=tn_split=6 timeout 600 python -m pytest tests/test_gemm_gpu.py -k gemm_tn -q -x 2>&1 | tail -3
for m in "tn_split=0" "tn_split=6" "tn_split=6,tn_split_pf=1" "tn_split=6,tn_split_target=384" "tn_split=0" "tn_split=6"; do
  UR_TEST=$m python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$m headline ms', j['ms_per_step'], 'fit', j.get('trainer_fit',{}).get('ms_per_step'))"
done
bash tools/kstats.sh r06b_k6 UR_TEST=tn_split=6 -- 2>&1 | head -14
