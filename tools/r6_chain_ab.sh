#!/bin/bash
# usage (GPU box): bash tools/r6_chain_ab.sh   -> row-chain kernels in split-bf16 arithmetic (default) against the exact stream (chain_split=0)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-gather-bench --no-extra-legs --no-prof"
for r in 1 2; do
  for h in 1 0; do
    echo -n "chain_split=$h headline ms "; UR_TEST=chain_split=$h $B 2>&1 | grep -o '"ms_per_step": [0-9.]*' | tail -1
  done
done
