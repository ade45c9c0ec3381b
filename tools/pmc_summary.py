#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv -> JSON.
usage: pmc_summary.py <counter_collection.csv> [<out.json>]
For SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE passes it also prints the MFMA-pipe utilisation of each kernel:
`mfma_util` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the counters are summed over the 8 XCD instances (GRBM) and over
all SIMDs (SQ), so this is the fraction of the chip's 1024 MFMA pipes' cycles that were busy while the kernel ran."""
import collections
import csv
import json
import sys


def main():
    path = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            launches[k] += 1
    out = []
    for k, v in agg.items():
        n = max(launches[k], 1)
        e = {"kernel": k[:140], "launches": n}
        for c, x in v.items():
            e[c + "_per_launch"] = round(x / n, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            e["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128.0), 4)
        out.append(e)
    out.sort(key=lambda e: -e.get("GRBM_GUI_ACTIVE_per_launch", 0) * e["launches"])
    js = json.dumps({"source": path, "kernels": out}, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js)
    for e in out[:25]:
        print("%-70s n=%5d %s" % (e["kernel"][:70], e["launches"], {k: v for k, v in e.items() if k not in ("kernel", "launches")}))


if __name__ == "__main__":
    main()
