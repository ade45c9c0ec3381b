#!/bin/bash
mkdir -p gpurun_out
{
echo "== chain 63"; UR_SASREC_CHAIN=63 timeout 900 bash tools/ab_env.sh UR_SASREC_HOLD "0 2 1 3" 2
echo "== chain 63 blocks"; UR_SASREC_CHAIN=63 timeout 900 bash tools/ab_env.sh UR_TN_BLOCKS "576 864 1152 1728" 2
} > gpurun_out/s6_ab.txt 2>&1
awk '{print $1,$2,$3}' gpurun_out/s6_ab.txt
UR_SASREC_CHAIN=63 TAILN=75 bash tools/timeline.sh > gpurun_out/s6_timeline_c63.txt 2>&1
sed -n '/per step, by queue/,$p' gpurun_out/s6_timeline_c63.txt
