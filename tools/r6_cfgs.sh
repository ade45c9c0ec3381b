#!/bin/bash
# usage (GPU box): bash tools/r6_cfgs.sh   -> C2 / C3 / B=4096 / dropout step times with the row chains in split (default) and exact (chain_split=0) arithmetic
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 200 --warmup 20"
P='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["ms_per_step"], j["value"])'
for h in 1 0; do
  export UR_TEST=chain_split=$h
  echo "== chain_split=$h"
  python bench.py $F --n-items 60000 --d 64 2>/dev/null | python -c "$P" C2
  python bench.py $F --n-items 2000000 --seq-len 200 --negatives 1000 --loss softmax --batch 128 2>/dev/null | python -c "$P" C3
  python bench.py $F --batch 4096 2>/dev/null | python -c "$P" C5_B4096
  python bench.py $F --dropout 0.5 2>/dev/null | python -c "$P" C5_dropout0.5
done
