#!/bin/bash
# usage (GPU box): bash tools/r6_ab2.sh "<hook=value> <hook=value> ..."   -> headline ms per step under each UR_TEST setting, two rounds
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-gather-bench --no-extra-legs --no-prof"
for r in 1 2; do
  for h in $1; do
    echo -n "$h headline ms "; UR_TEST=$h $B 2>&1 | grep -o '"ms_per_step": [0-9.]*' | tail -1
  done
done
