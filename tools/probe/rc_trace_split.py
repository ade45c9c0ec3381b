#!/usr/bin/env python3
"""Phase stamps (s_memtime) of chain_ffn_fwd_split's workgroups, last launch of a run.  Needs a library built with RC_T stamps in that kernel."""
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
_lib.lib.ur_debug_rc_trace(buf.ctypes.data)
t = buf.reshape(1024, 24).astype(np.int64)[:64]
names = ["stage ctx + bar", "GEMM Wo + bar", "LN epi + bar", "GEMM W1 + bar", "act epi + bar", "GEMM W2 + bar", "put partial + bar", "counter + bar", "last: sum + LN"]
d = np.diff(t[:, :10], axis=1)
last = t[:, 9] > t[:, 8]
print("workgroups:", len(t), "last-arrivers:", int(last.sum()))
for i, n in enumerate(names):
    col = d[:, i] if i < 8 else d[last, i]
    print(f"  {i} {n:20s} mean {col.mean():8.0f} p10 {np.percentile(col,10):8.0f} p90 {np.percentile(col,90):8.0f}")
print("total to stamp 8: mean", (t[:, 8] - t[:, 0]).mean(), " last-arriver total:", (t[last, 9] - t[last, 0]).mean())
