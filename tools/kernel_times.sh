#!/bin/bash
# usage: tools/kernel_times.sh <tag> <bench.py args...>   -> per-kernel average duration (rocprofv3 kernel trace) of the ur:: kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --steps 30 --warmup 5 "$@" > /tmp/prof_$tag.out 2>&1
grep '^{"metric"' /tmp/prof_$tag.out | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'ex/s', j['value'])"
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_$tag/${tag}_kernel_stats.csv")))
for r in rows:
    if "ur::" in r["Name"] and any(k in r["Name"] for k in "${KFILTER:-attn}".split(",")): print("%8.1f us x %5.1f  %s" % (float(r["AverageNs"])/1000, int(r["Calls"])/35, r["Name"][:70]))
PY
