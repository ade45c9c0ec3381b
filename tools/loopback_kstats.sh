#!/bin/bash
# usage (GPU box): tools/loopback_kstats.sh <tag> -> gpurun_out/<tag>_w8_loopback_kstats.txt: per-kernel durations (rocprofv3 --kernel-trace --stats)
# of the multi-GPU step at true W = 8 shapes (bench.py --loopback 8) next to the same path at world 1 (--sharded-w1) and the plain step:
# which kernels the sharded path ADDS per rank-step, and what they cost at 8-rank shapes (8-run merge plan, cap = 4416, cap2 = 576)
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_w8_loopback_kstats.txt; mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, ranks, args...
  name=$1; ranks=$2; shift 2
  rm -rf /tmp/lk_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lk_$name -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-extra-legs --no-prof --steps 60 --warmup 10 "$@" > /tmp/lk_$name.out 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/lk_$name/k_kernel_stats.csv")))
steps = 70.0 * $ranks
print("== $name: per RANK-STEP (%d ranks x 70 steps)" % $ranks)
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if not ("ur::" in n or "kernel" in n and ("plan" in n or "rows_" in n or "shard" in n or "loop_" in n)):
        continue
    per = float(r["TotalDurationNs"]) / steps / 1000
    tot += per
    if per >= 0.5:
        print("%8.1f us avg x %5.2f/step = %7.1f us/step  %s" % (float(r["AverageNs"]) / 1000, int(r["Calls"]) / steps, per, n[:100]))
print("sum of the library's kernels per rank-step: %.1f us" % tot)
PY
}
{
run plain 1
run sharded_w1 1 --sharded-w1
run loopback8 8 --loopback 8
grep -h '"loopback"' /tmp/lk_loopback8.out | cut -c1-400
} > $out 2>&1
cat $out
