#!/usr/bin/env python3
"""Phase stamps (s_memtime, thread 0 of every workgroup) of the last-row layer's two kernels (csrc/lastrow.hip), last launch of a run of
training steps at the headline shape.  usage (GPU box): python tools/probe/lr_trace.py [bench args]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from unirec_amd import _lib
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec

if "--n-items" not in sys.argv:
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


f = _lib.lib.ur_debug_lr_trace
f.argtypes = [C.c_int, C.c_void_p]
for i in range(20):
    step(batches[i], batches[i + 1])
torch.cuda.synchronize()
f(1, None)
for i in range(20, 30):
    step(batches[i], batches[i + 1])
torch.cuda.synchronize()
buf = np.zeros(2 * 1024 * 32, dtype=np.uint64)
f(0, buf.ctypes.data)
full = buf.reshape(2, 1024, 32).astype(np.int64)
NAMES = [
    ["stage x rows (Wq, K/V requested)", "q mma + put (+ Wo requested)", "barrier", "q epilogue + barrier", "attention math", "W1 requested + barrier",
     "softmax merge + barrier", "out-proj mma + barrier", "LN1 epilogue + barrier", "FFN1 mma + put", "W2 requested + barrier", "h1 epilogue + barrier",
     "FFN2 mma + barrier", "LN2 epilogue"],
    ["LNbwd 2 (W2 requested) + barrier", "colsum + FFN2^T mma + put", "W1 requested + barrier", "g_h1 epilogue + barrier", "FFN1^T mma + put",
     "Wo, K/V, q, ctx requested + barrier", "LNbwd 1 epilogue + barrier", "colsum + Wo^T mma... ", "g_ctx epilogue + barrier", "attention backward", "Wq, Wk/Wv units requested + barrier",
     "dq epilogue + barrier", "dq Wq + unit mma + barrier", "xl epilogue + barrier", "S V + row stores"],
]
for k, (what, last) in enumerate((("lastrow_fwd", 14), ("lastrow_bwd", 15))):
    t = full[k][:, : last + 1]
    t = t[(t[:, 0] > 0) & (t[:, last] > 0)]
    if not len(t):
        print(what, ": no stamps")
        continue
    base = t[:, 0].min()
    d = np.diff(t, axis=1)
    print(f"{what}: {len(t)} workgroups; start spread p50 {np.percentile(t[:, 0] - base, 50):.0f} max {(t[:, 0] - base).max()};"
          f" end p50 {np.percentile(t[:, last] - base, 50):.0f} max {(t[:, last] - base).max()} cycles")
    print(f"  workgroup duration: mean {(t[:, last] - t[:, 0]).mean():.0f}  p10 {np.percentile(t[:, last] - t[:, 0], 10):.0f}  p90 {np.percentile(t[:, last] - t[:, 0], 90):.0f}")
    for i, n in enumerate(NAMES[k]):
        print(f"  {i:2d} {n:48s} mean {d[:, i].mean():8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
