#!/bin/bash
# usage (GPU box): bash tools/r6_gru3.sh   -> GRU encoder step time by hidden size with the step kernels exact (0) / split in both sweeps (1) / forward only (2) / backward only (3)
cd $GRAFT_REPO_ROOT
for H in 768 512 256; do
  for h in 0 1 2 3; do
    echo -n "H=$H gru_step_split=$h  "
    UR_TEST=gru_step_split=$h,gru_step_split_hmin=128 python tools/gru_bench.py --hidden $H --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], 'gru', j['kernel_ms_and_launches_per_step']['gru'][0])"
  done
done
