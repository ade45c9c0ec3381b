#!/usr/bin/env python3
"""Host-side profile (cProfile) of the multi-GPU step's code path at world 1: where the Python / ctypes time of a step goes.
usage (GPU box): python tools/host_profile_sharded.py [--plain]"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
plain = "--plain" in sys.argv
sys.argv = [a for a in sys.argv if a != "--plain"] + ["--n-items", "4000000"]
import torch
import bench
from unirec_amd.facility.distributed import ShardedSparseDenseAdam
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=64)
if plain:
    opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
    def step(b, nxt):
        opt.zero_grad()
        opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
        opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
        model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
        opt.step(late_join=True)
else:
    opt = ShardedSparseDenseAdam(model, 0, 1, lr=1e-3, table_mode=a.table_mode, full_rows={"item_embedding": a.n_items})
    def step(b, nxt):
        opt.train_step(b, nxt)
for i in range(20):
    step(batches[i], batches[i + 1])
torch.cuda.synchronize()
N = 300
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(N):
    step(batches[i % 60], batches[i % 60 + 1])
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host {1e3*(t1-t0)/N:.4f} ms/step (under cProfile)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
