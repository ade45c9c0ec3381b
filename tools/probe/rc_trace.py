#!/usr/bin/env python3
"""Phase stamps (s_memtime) of chain_ffn_bwd's workgroups, last launch of a run.  Needs a library built with the RC_T instrumentation."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from unirec_amd import _lib
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec

sys.argv += ["--n-items", "2000000"]
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=40)
def step(b, nxt):
    opt.zero_grad()
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    opt.step(late_join=True)
for i in range(30):
    step(batches[i], batches[i + 1])
torch.cuda.synchronize()
buf = np.zeros(1024 * 24, dtype=np.uint64)
_lib.lib.ur_debug_rc_trace.argtypes = [C.c_void_p]
rc = _lib.lib.ur_debug_rc_trace(buf.ctypes.data)
full = buf.reshape(1024, 24).astype(np.int64)
t = full[:, :17]
full = full[(t[:, 0] > 0) & (t[:, 16] > 0)]
t = t[t[:, 0] > 0]
t = t[t[:, 16] > 0]
print("workgroups with stamps:", len(t), "rc", rc)
base = t[:, 0].min()
names = ["ld gy,yhat + LNbwd", "colsum+bar", "GEMM g_tf W2 c0", "bar", "epi h1 c0 + bar", "GEMM W1 c0", "GEMM W2 c1", "bar", "epi h1 c1 + bar", "GEMM W1 c1",
         "acc->tile + bar", "epi LN1 bwd (ahat)", "bar + colsum", "GEMM Wo", "bar", "store g_ctx"]
d = np.diff(t, axis=1)
print(f"start spread: p50 {np.percentile(t[:,0]-base,50):.0f} p90 {np.percentile(t[:,0]-base,90):.0f} max {(t[:,0]-base).max()}   end: p50 {np.percentile(t[:,16]-base,50):.0f} max {(t[:,16]-base).max()}")
print(f"workgroup duration: mean {(t[:,16]-t[:,0]).mean():.0f}  p10 {np.percentile(t[:,16]-t[:,0],10):.0f} p90 {np.percentile(t[:,16]-t[:,0],90):.0f}")
for i, n in enumerate(names):
    print(f"  {i:2d} {n:24s} mean {d[:, i].mean():8.0f}  p10 {np.percentile(d[:, i],10):8.0f}  p90 {np.percentile(d[:, i],90):8.0f}")

print("10 -> 17 (after last GEMM stamp to before acc_to_tile):", np.percentile(full[:,17]-full[:,10],[10,50,90]))
print("17 -> 18 (acc_to_tile):", np.percentile(full[:,18]-full[:,17],[10,50,90]))
print("18 -> 11 (barrier):", np.percentile(full[:,11]-full[:,18],[10,50,90]))
