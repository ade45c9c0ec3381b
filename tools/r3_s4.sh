#!/bin/bash
mkdir -p gpurun_out
{
for b in 576 864; do echo "== EARLY_FORK=1 BLOCKS=$b"; UR_SASREC_EARLY_FORK=1 UR_TN_BLOCKS=$b timeout 900 bash tools/ab_env.sh UR_SASREC_HOLD "0 2 3" 2; done
echo "== ref"; UR_TN_BLOCKS=864 timeout 900 bash tools/ab_env.sh UR_SASREC_HOLD "2" 2
} > gpurun_out/s4_ab.txt 2>&1
awk '{print $1,$2,$3}' gpurun_out/s4_ab.txt
UR_SASREC_EARLY_FORK=1 UR_SASREC_HOLD=2 UR_TN_BLOCKS=864 TAILN=75 bash tools/timeline.sh > gpurun_out/s4_timeline_ef_h2_b864.txt 2>&1
sed -n '/per step, by queue/,/one step/p' gpurun_out/s4_timeline_ef_h2_b864.txt
