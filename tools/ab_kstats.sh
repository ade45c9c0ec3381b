#!/bin/bash
# per-kernel average durations of two builds on one box (tree's .so against unirec_amd/libunirec_amd.so.base), side stream on and off
# usage (GPU box): bash tools/ab_kstats.sh [kernel-name filter, default chain]
f=${1:-chain}
cp unirec_amd/libunirec_amd.so /tmp/new.so
for side in ${SIDES:-1 0}; do
  for v in base new; do
    if [ $v = base ]; then cp unirec_amd/libunirec_amd.so.base unirec_amd/libunirec_amd.so; else cp /tmp/new.so unirec_amd/libunirec_amd.so; fi
    echo "== $v SIDE=$side"
    bash tools/kstats.sh ${v}_s$side UR_SASREC_SIDE=$side -- --no-extra-legs 2>&1 | grep -E "ms_per_step|$f"
  done
done
cp /tmp/new.so unirec_amd/libunirec_amd.so
