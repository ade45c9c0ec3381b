#!/bin/bash
# the bottom layer's four products as ONE product of the same tile count (R = 384 + 128 + 512 + 512 = 1536, C = 128): the group launch, isolated
mkdir -p gpurun_out
{
for big in 0 1; do
  for tgt in 0 128 192 256 384; do
    [ $big = 0 ] && [ $tgt != 0 ] && continue
    echo "== UR_TN_BIG=$big UR_TN_TARGET=$tgt"
    UR_TN_BIG=$big UR_TN_TARGET=$tgt python - <<'P'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import gemm_bench as g
for T in (21248,):
    g.tn(T, 1536, 128, 0)
    g.tn(T, 512, 128, 0)
    g.tn(T, 256, 128, 0)
P
  done
done
} 2>&1 | grep -v "^NT" > gpurun_out/r5_tn_group_iso.txt
cat gpurun_out/r5_tn_group_iso.txt
