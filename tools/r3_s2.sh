#!/bin/bash
mkdir -p gpurun_out
{
for b in 288 576 864; do echo "== BLOCKS=$b"; UR_TN_BLOCKS=$b timeout 900 bash tools/ab_env.sh UR_SASREC_HOLD "0 1 2 3" 2; done
echo "== LDS (hold 0, blocks 288)"; timeout 600 bash tools/ab_env.sh UR_TN_LDS_KB "32 52 68 84" 2
echo "== LDS (hold 0, blocks 576)"; UR_TN_BLOCKS=576 timeout 600 bash tools/ab_env.sh UR_TN_LDS_KB "32 52 68 84" 2
echo "== BLOCKS hold 0"; timeout 600 bash tools/ab_env.sh UR_TN_BLOCKS "576 768 1152 1536" 2
} > gpurun_out/s2_ab.txt 2>&1
cat gpurun_out/s2_ab.txt
