#!/bin/bash
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_w1 -o w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 30 --warmup 5 --sharded-w1 > /tmp/prof_w1.out 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_w1/w_kernel_stats.csv")))
tot=0; n=0
for r in rows:
    per=int(r["Calls"])/35
    if per>=0.9:
        n+=per; print("%6.2f/step %8.1f us avg  %s" % (per, float(r["AverageNs"])/1000, r["Name"][:100]))
print("kernel launches per step ~", round(n,1))
import os
p="/tmp/prof_w1/w_memory_copy_stats.csv"
if os.path.exists(p):
    for r in csv.DictReader(open(p)): print("memcpy", r)
PY
