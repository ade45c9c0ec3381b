#!/usr/bin/env python3
"""Micro-benchmark of ur_lazy_adam_catchup: rows never updated (last_step = 0) vs rows with k missed steps of momentum to replay."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unirec_amd import ops

dev = torch.device("cuda:0")
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 128
w = torch.zeros(N, d, device=dev); m = torch.zeros(N, d, device=dev); v = torch.zeros(N, d, device=dev)
last = torch.zeros(N, dtype=torch.int32, device=dev)
g = torch.Generator(device=dev).manual_seed(0)


def run(tag, k, touched):
    ids = torch.randint(1, N, (28160,), device=dev, generator=g, dtype=torch.int32)
    pl = ops.rows_plan(ids, None, N)
    rows = pl.uniq_idx[: int(pl.n_uniq.item())].long()
    if touched:
        m[rows] = 0.01; v[rows] = 1e-4; last[rows] = 1000 - k
    cfg = ops.adam_cfg(1e-3, 1001)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lazy_adam_catchup(cfg, w, m, v, last, pl)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) * 1e3:.1f} us")


for rep in range(3):
    run("fresh rows (last_step = 0)", 0, False)
for k in (1, 8, 64, 192):
    for rep in range(2):
        run(f"rows with {k} missed steps", k, True)
