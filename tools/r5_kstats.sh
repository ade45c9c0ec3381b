#!/bin/bash
# round 5: per-kernel durations (rocprofv3 kernel trace) of the step with the 64 x 64 / the 128 x 128 weight-gradient kernel, side stream on / off
mkdir -p gpurun_out
for spec in "0 1" "1 1" "0 0" "1 0"; do
  set -- $spec
  echo "== UR_TN_BIG=$1 UR_SASREC_SIDE=$2"
  bash tools/kstats.sh r5_big$1_side$2 UR_TN_BIG=$1 UR_SASREC_SIDE=$2 -- --no-extra-legs 2>&1 | grep -E "ms_per_step|gemm_tn|reduce_batch|sum of"
done > gpurun_out/r5_kstats.txt 2>&1
cat gpurun_out/r5_kstats.txt
