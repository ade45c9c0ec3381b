cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --steps 200 --warmup 20 > /tmp/prof_g.out 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_g.out | tail -1
ls /tmp/prof_g; head -1 /tmp/prof_g/g_kernel_trace.csv
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py /tmp/prof_g/g_kernel_trace.csv | tail -${TAILN:-60}
