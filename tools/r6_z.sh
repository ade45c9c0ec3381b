#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in "" "dw_early=1" "dw_early=2" "" "dw_early=1"; do
  UR_TEST=$m python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('UR_TEST=$m headline ms', j['ms_per_step'])"
done
