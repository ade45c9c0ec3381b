import os, sys, time
PS = time.perf_counter()
sys.path.insert(0, "/root/repo")
import torch, gc
import bench
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=160)
def step(b, nxt):
    opt.zero_grad()
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    opt.step()
T0 = time.perf_counter()
hs = []
for i in range(150):
    t = time.perf_counter()
    step(batches[i], batches[i + 1])
    if i % 10 == 9: torch.cuda.synchronize()
    hs.append((time.perf_counter() - t) * 1e3)
    if hs[-1] > 20 and i > 0: print("stall at step", i, "ms", hs[-1], "since loop start", time.perf_counter() - T0, "since process start", time.perf_counter() - PS)
print("host ms per step:", " ".join(f"{x:.2f}" for x in hs))
print("gc counts", gc.get_count(), "elapsed", time.perf_counter() - T0)
