#!/bin/bash
# usage (GPU box): tools/gather_pmc.sh <tag> -> gpurun_out/<tag>_gather_pmc.json  (one rocprofv3 --pmc pass per counter group; kernel trace only)
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/gather_pmc.py"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/gp_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/gp_$i -o g -- $CMD > /tmp/gp_$i.out 2>&1 || echo "pass '$grp' failed: $(tail -2 /tmp/gp_$i.out)"
  i=$((i+1))
done
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/gp_*/g_counter_collection.csv")):
    seen = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gather_kernel" in k or "scorer_loss_fwd" in k:
            seen[(k[:60], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in seen.items():
        for c, v in cs.items():
            res[k][c].append(v)
out = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in res.items()}   # median over the launches of a kernel
import hashlib, os
root = os.path.join("$GRAFT_REPO_ROOT", "unirec_amd", "csrc")
h = hashlib.sha256()
for f in sorted(os.listdir(root)):
    if f.endswith((".hip", ".h", ".cpp")):
        h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
json.dump({"csrc_digest": h.hexdigest()[:16], "source": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python tools/gather_pmc.py (median over 3 launches per kernel)",
           "units": "FETCH_SIZE / WRITE_SIZE in KB (gfx950: HBM read bytes = 2 x FETCH_SIZE, MI355X_MICROARCH.md)", "per_kernel": out},
          open("$out/${tag}_gather_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
