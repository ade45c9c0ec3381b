#!/usr/bin/env python3
"""What does the first 20-step region after 5 warm-up steps pay for?  MODE=none | busy (30 ms of unrelated device work first) |
steps (20 more training steps first) | sleep (30 ms of host sleep first)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec
mode = os.environ.get("MODE", "none")
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=320)
def step(b, nxt):
    opt.zero_grad()
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    opt.step(late_join=True)
k = 0
W = int(os.environ.get("W", "5"))
if mode.startswith("gcfirst"):
    torch.cuda.synchronize(); gc.collect(); gc.disable()
    if mode in ("gcfirst_sleep", "gcfirst_busy", "gcfirst_busy_async"): time.sleep(0.2)
    if mode.startswith("gcfirst_busy"):
        x = torch.empty(64 << 20, device=dev)
        ms = float(os.environ.get("BUSY_MS", "30"))
        if mode == "gcfirst_busy":
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < ms * 1e-3:
                x.add_(1.0)
                torch.cuda.synchronize()
        else:   # queued back to back, no host sync in between: ~0.1 ms per pass over 256 MB
            for _ in range(int(ms * 10)): x.add_(1.0)
if mode == "gcfirst_mm":
    ms = float(os.environ.get("BUSY_MS", "50"))
    A = torch.randn(4096, 4096, device=dev); Bm = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < ms * 1e-3:
        for _ in range(4): A @ Bm
        torch.cuda.synchronize()
if mode == "gcfirst_touch":
    stride = int(os.environ.get("TOUCH", "16384"))
    for st in opt.tables.values():
        for key in ("w", "m", "v"):
            t = st[key]
            if t is not None: t.view(-1)[::stride].sum()
for i in range(W):
    step(batches[k], batches[k + 1]); k += 1
torch.cuda.synchronize()
if mode == "busy":
    x = torch.empty(64 << 20, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        x.add_(1.0)
        torch.cuda.synchronize()
elif mode == "steps":
    for i in range(20):
        step(batches[k], batches[k + 1]); k += 1
elif mode == "sleep":
    time.sleep(0.03)
if not mode.startswith("gcfirst"):
    gc.collect(); gc.disable()
if os.environ.get("IDLE_MS"):     # idle device right in front of the first region (after W warm-up steps)
    torch.cuda.synchronize(); time.sleep(float(os.environ["IDLE_MS"]) * 1e-3)
if os.environ.get("EMPTY_CACHE"):   # give every cached block back: the allocator starts over (new addresses, unsettled reuse pattern)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    step(batches[k], batches[k + 1]); k += 1     # (one untimed step pays the hipMallocs)
    torch.cuda.synchronize()
if os.environ.get("POST_BUSY_MS"):   # unrelated device work between the idle period and the first region
    ms = float(os.environ["POST_BUSY_MS"])
    A = torch.randn(4096, 4096, device=dev); Bm = torch.randn(4096, 4096, device=dev); x = torch.empty(64 << 20, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < ms * 1e-3:
        for _ in range(4): A @ Bm; x.add_(1.0)
        torch.cuda.synchronize()
out = []
for rep in range(int(os.environ.get("REGIONS", "4"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        step(batches[k], batches[k + 1]); k += 1
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out.append(f"{1e3*(t1-t0)/20:.4f}")
print(mode, " ".join(out))
