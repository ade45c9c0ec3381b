#!/bin/bash
mkdir -p gpurun_out
UR_SASREC_HOLD=2 UR_TN_BLOCKS=864 TAILN=75 bash tools/timeline.sh > gpurun_out/s3_timeline_h2_b864.txt 2>&1
UR_SASREC_HOLD=0 UR_TN_BLOCKS=576 TAILN=75 bash tools/timeline.sh > gpurun_out/s3_timeline_h0_b576.txt 2>&1
tail -48 gpurun_out/s3_timeline_h2_b864.txt
