#!/bin/bash
# usage (GPU box): bash tools/r6_gru2.sh   -> C4 encoder (H = 128 / 768) step time with the one-product-per-launch GEMMs split (default) / exact (nt_split=0)
cd $GRAFT_REPO_ROOT
for h in 1 0; do
  echo "== nt_split=$h"
  UR_TEST=nt_split=$h python tools/gru_bench.py --steps 100 2>/dev/null | tail -1
  UR_TEST=nt_split=$h python tools/gru_bench.py --hidden 768 --steps 30 2>/dev/null | tail -1
done
