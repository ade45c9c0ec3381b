#!/bin/bash
# usage (GPU box): tools/ramp_trace.sh   -> per-step breakdown of the driver's protocol (20 steps after 5 warm-ups)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 20 --warmup 5 > /tmp/prof_r.out 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_r.out | tail -1
python $GRAFT_REPO_ROOT/tools/ramp_trace.py /tmp/prof_r/r_kernel_trace.csv
