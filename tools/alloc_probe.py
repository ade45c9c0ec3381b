#!/usr/bin/env python3
"""Does the caching allocator reach a steady state in the first steps?  Per step: reserved bytes, number of segments, and the addresses of
a few per-step buffers (the next batch's plan workspace, the row-gradient buffer)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=120)
def step(b, nxt):
    opt.zero_grad()
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    opt.step(late_join=True)
gc.collect(); gc.disable()
seen = set()
for k in range(40):
    step(batches[k], batches[k + 1])
    st = torch.cuda.memory_stats()
    ptrs = []
    pre = opt._prefetched
    if pre is not None:
        for name, bufs in pre[3][1].items():
            for t in (bufs if isinstance(bufs, (list, tuple)) else [bufs]):
                if torch.is_tensor(t): ptrs.append(t.data_ptr())
    x = torch.empty(1 << 20, device=dev); px = x.data_ptr(); del x
    new = [p for p in ptrs if p not in seen]
    seen.update(ptrs)
    print(f"step {k:2d}: reserved {st['reserved_bytes.all.current']/2**20:9.1f} MiB  segments {st['segment.all.current']}  allocs so far {st['allocation.all.allocated']}  "
          f"new plan-buffer addresses {len(new)}/{len(ptrs)}  probe {px:#x}")
