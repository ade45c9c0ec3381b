#!/bin/bash
# round-3 GPU session 1: parity of the grouped weight-gradient launch, then A/B runs (writes gpurun_out/s1_*.txt)
mkdir -p gpurun_out
echo skip-tests
cat gpurun_out/s1_tests.txt
{ timeout 600 bash tools/ab_env.sh UR_TN_GROUP "0 1" 3; } > gpurun_out/s1_ab_group.txt 2>&1
cat gpurun_out/s1_ab_group.txt
{ timeout 600 bash tools/ab_env.sh UR_LIB_VARIANT "base prio" 3; } > gpurun_out/s1_ab_prio.txt 2>&1
cat gpurun_out/s1_ab_prio.txt
{ timeout 600 bash tools/ab_env.sh UR_TN_BLOCKS "192 288 384 576" 2; } > gpurun_out/s1_ab_blocks.txt 2>&1
cat gpurun_out/s1_ab_blocks.txt
TAILN=90 timeout 600 bash tools/timeline.sh > gpurun_out/s1_timeline.txt 2>&1
tail -70 gpurun_out/s1_timeline.txt
