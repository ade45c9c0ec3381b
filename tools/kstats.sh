#!/bin/bash
# usage (GPU box): tools/kstats.sh <tag> [ENV=VAL ...] -- <bench.py args>   -> gpurun_out/<tag>_kernel_stats.csv + a per-step table of the ur:: kernels
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ -n "$1" ]; do envs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --steps 30 --warmup 5 "$@" > /tmp/prof_$tag.out 2>&1
cp /tmp/prof_$tag/${tag}_kernel_stats.csv $out/${tag}_kernel_stats.csv
grep '^{"metric"' /tmp/prof_$tag.out | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tag ms_per_step', j['ms_per_step'], 'ex/s', j['value'])"
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_$tag/${tag}_kernel_stats.csv")))
tot=0
for r in rows:
    if "ur::" in r["Name"]:
        per=float(r["TotalDurationNs"])/35/1000
        tot+=per
        print("%8.1f us avg x %5.2f/step = %7.1f us/step  %s" % (float(r["AverageNs"])/1000, int(r["Calls"])/35, per, r["Name"][:90]))
print("sum of ur:: kernels per step: %.1f us" % tot)
PY
