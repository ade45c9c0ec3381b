#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "tn_split_procost=100" "tn_split_procost=135" "tn_split_procost=150" "tn_split_procost=135,tn_split_proprio=1" "tn_split_procost=100,tn_split_proprio=1" "tn_split_procost=135,tn_split_target=448"; do
  UR_TEST=$v python tools/tn_group_bench.py 6 40 2>&1 | tail -1
done
for m in "tn_split=6,tn_split_procost=100" "tn_split=6" "tn_split=6,tn_split_proprio=1" "tn_split=0"; do
  UR_TEST=$m python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$m headline ms', j['ms_per_step'])"
done
