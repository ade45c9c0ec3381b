cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o gp -- python $GRAFT_REPO_ROOT/tools/probe/gru_h768_leg.py > /tmp/gp.out 2>&1
grep h768 /tmp/gp.out
python - <<PY
import csv
for r in csv.DictReader(open("/tmp/gp/gp_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print("%8.1f us x %6d  %5.1f%%  %s" % (float(r["AverageNs"])/1000, int(r["Calls"]), float(r["Percentage"]), r["Name"][:90]))
PY
