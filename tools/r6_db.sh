#!/bin/bash
cd $GRAFT_REPO_ROOT
UR_TEST=tn_split_db=1 timeout 600 python -m pytest tests/test_gemm_gpu.py -k "gemm_tn or split_bf16" -q -x 2>&1 | tail -2
for v in "" "tn_split_db=1" "tn_split_db=1,tn_split_target=384" "tn_split_db=1,tn_split_target=512"; do
  UR_TEST=$v python tools/tn_group_bench.py 6 40 2>&1 | tail -1
done
UR_TEST=tn_split_db=1,tn_split_trace=2 python tools/tn_group_bench.py 6 8 2>&1 | grep -A14 "tn_split trace" | cut -c1-230
for m in "" "tn_split_db=1" "" "tn_split_db=1"; do
  UR_TEST=$m python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('UR_TEST=$m headline ms', j['ms_per_step'])"
done
