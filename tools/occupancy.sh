#!/bin/bash
# usage (GPU box): bash tools/occupancy.sh [bench args]   -> tools/occupancy.py over a kernel trace of the default bench
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_o
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -o o -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 200 --warmup 20 "$@" > /tmp/prof_o.out 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_o.out | tail -1
head -1 /tmp/prof_o/o_kernel_trace.csv
python $GRAFT_REPO_ROOT/tools/occupancy.py /tmp/prof_o/o_kernel_trace.csv
