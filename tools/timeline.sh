#!/bin/bash
# usage (GPU box): [CMD="python tools/x.py"] [SHOW=kernel-substring] [TAILN=60] tools/timeline.sh   -> busy fraction, per-queue kernel time, gaps
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_g
CMD=${CMD:-"python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 200 --warmup 20"}
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o g -- $CMD > /tmp/prof_g.out 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_g.out | tail -1
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py /tmp/prof_g/g_kernel_trace.csv | tail -${TAILN:-60}
