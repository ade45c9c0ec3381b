#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md HBM
section) -> the JSON bench.py reads for `roofline.traffic`.
usage: pmc_hbm.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command that was profiled>"
Counter values are KB; gfx950 correction: HBM read bytes = 2 x FETCH_SIZE (the guide's note on 64-B vs 128-B requests)."""
import collections, csv, hashlib, json, os, sys


def csrc_digest():
    """what the summary was measured ON: sha256 over the kernel sources (the same function as bench.py's: it refuses a summary whose
    digest is not the tree's)"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unirec_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    tot, n, seen = collections.Counter(), collections.Counter(), set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); n[k] += 1
    return tot, n


def cls(name):
    if "gemm_tn" in name:      # gemm_tn_group_kernel and (round 6) gemm_tn_split_kernel: the weight-gradient class
        return "gemm_tn"
    if "_split_kernel" in name or "lastrow_" in name:
        return "row_chain_last"
    for c in ("chain_ffn_fwd_kernel", "chain_ffn_bwd_kernel", "chain_proj_bwd_kernel", "chain_embed_proj_kernel"):
        if c in name:
            return "row_chain"      # bench.py's class of the same name (PC_CHAIN): the four full-sequence row-chain launches of a step
    for c in ("gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "attn_last", "ln_bwd", "sparse_adam", "rows_reduce", "reduce_batch", "scorer_loss", "gru_seq4"):
        if c in name:
            return c
    return None


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
kern, klass = [], collections.defaultdict(lambda: [0.0, 0.0, 0])
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k] + write[k])):
    if "ur::" not in k:
        continue
    n = max(nf[k], nw[k], 1)
    kern.append({"kernel": k[:150], "launches": n, "FETCH_SIZE_KB_per_launch": round(fetch[k] / max(nf[k], 1), 1),
                 "WRITE_SIZE_KB_per_launch": round(write[k] / max(nw[k], 1), 1),
                 "hbm_bytes_per_launch_corrected": int((2 * fetch[k] / max(nf[k], 1) + write[k] / max(nw[k], 1)) * 1024)})
    c = cls(k)
    if c:
        klass[c][0] += 2 * fetch[k] * 1024; klass[c][1] += write[k] * 1024; klass[c][2] += n
steps = max([v[2] for c, v in klass.items() if c == "scorer_loss"] + [1])     # one scorer launch per training step
# (the library's kernels only: the process also fills 150 GB of tables and optimizer state once, which is not a step's traffic)
total = sum((2 * fetch[k] + write[k]) * 1024 for k in set(fetch) | set(write) if "ur::" in k)
out = {"csrc_digest": csrc_digest(), "steps_profiled": steps, "hbm_bytes_per_step_all_kernels": int(total / steps),
       "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {sys.argv[4]}",
       "units": "counter values are KB; gfx950 correction: HBM read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section)",
       "per_class": {c: {"launches": v[2], "hbm_bytes_per_launch": int((v[0] + v[1]) / max(v[2], 1)), "read_bytes_per_launch": int(v[0] / max(v[2], 1)),
                         "write_bytes_per_launch": int(v[1] / max(v[2], 1))} for c, v in klass.items()},
       "per_kernel": kern}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for c, v in out["per_class"].items():
    print(c, v)
