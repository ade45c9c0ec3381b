#!/bin/bash
# usage (GPU box): bash tools/r6_gru4.sh   -> kernel trace of the H = 768 GRU encoder step, split / exact step kernels
cd $GRAFT_REPO_ROOT
for h in 1 0; do
  echo "== gru_step_split=$h"
  UR_TEST=gru_step_split=$h CMD="python $GRAFT_REPO_ROOT/tools/gru_bench.py --hidden 768 --steps 30" MARK=ids_time_major_kernel TAILN=30 bash tools/timeline.sh 2>&1 | grep -v "^  *[0-9.]* us/step\|amdgpu.ids" | tail -24 | cut -c1-150
done
