#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15
TAILN=60 bash tools/timeline.sh > gpurun_out/r06_d_timeline.txt 2>&1; head -40 gpurun_out/r06_d_timeline.txt
