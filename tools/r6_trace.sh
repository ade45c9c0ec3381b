#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "--- standalone group"; UR_TEST=tn_split_trace=2 python tools/tn_group_bench.py 6 8 2>&1 | tail -27
echo "--- in situ (4th dW launch of the bench = a top-layer launch)"; UR_TEST=tn_split=6,tn_split_trace=2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>&1 | grep -A26 "tn_split trace" | head -60
