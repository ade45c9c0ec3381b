#!/usr/bin/env python3
"""Host-side time of each call of the training step (no synchronisation inside the loop), and how long after a synchronisation the first
main-stream kernel of a step is enqueued (the pipeline-fill latency a short timed region pays once)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("UR_SPIN"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)))
import bench
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec

a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=80)
names = ["zero_grad", "plan_batch", "prefetch_plan", "forward_backward", "opt.step"]
acc = [0.0] * 5

def step(b, nxt, rec):
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"]); t.append(time.perf_counter())
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"]); t.append(time.perf_counter())
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"]); t.append(time.perf_counter())
    opt.step(late_join=True); t.append(time.perf_counter())
    if rec:
        for i in range(5): acc[i] += t[i + 1] - t[i]

for i in range(10):
    step(batches[i], batches[i + 1], False)
torch.cuda.synchronize()
gc.collect(); gc.disable()
K = 50
t0 = time.perf_counter()
for i in range(K):
    step(batches[10 + i], batches[11 + i], True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/K:.3f} ms/step; wall {1e3*(t2-t0)/K:.3f} ms/step")
for n, v in zip(names, acc): print(f"  {n:18s} {1e6*v/K:7.1f} us")
# 20-step regions like the driver's: wall per step, and the same with the GPU kept busy across the region start (no sync before)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(batches[10 + i], batches[11 + i], False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"20 steps after a sync: {1e3*(t1-t0)/20:.4f} ms/step")
# an empty region: sync, sync
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"synchronize() on an idle device: {1e6*(t1-t0):.1f} us")
x = torch.zeros(1, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); x.add_(1); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"one tiny kernel + synchronize(): {1e6*(t1-t0):.1f} us")
