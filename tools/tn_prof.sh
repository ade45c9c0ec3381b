#!/bin/bash
# usage (GPU box): bash tools/tn_prof.sh <tag> [ENV=VAL ...]   -> per-launch durations of the gemm_tn kernels at the C5 shapes
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/tn_bench.py > /tmp/prof_$tag.out 2>&1
python - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("/tmp/prof_$tag/${tag}_kernel_trace.csv")) if "gemm_tn" in r["Kernel_Name"] or "reduce_splits" in r["Kernel_Name"]]
tn = [r for r in rows if "gemm_tn" in r["Kernel_Name"]]
rd = [r for r in rows if "reduce_splits" in r["Kernel_Name"]]
shapes = ["T x 128 x 512", "T x 512 x 128", "T x 128 x 128", "T x 384 x 128", "T x 256 x 128", "512 x 128 x 512", "512 x 128 x 128"]
for i, s in enumerate(shapes):
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tn[i::7])
    e = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rd[i::7])
    print("$tag %-16s gemm_tn median %6.1f us (min %6.1f)   reduce median %5.1f us   WGs %s" % (s, d[len(d)//2] / 1e3, d[0] / 1e3, e[len(e)//2] / 1e3, tn[i].get("Grid_Size_X", "?")))
PY
