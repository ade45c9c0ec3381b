#!/usr/bin/env python3
"""The north-star gather (100 M x 128 fp32 table, uniform random ids) alone, for rocprofv3 --pmc passes: the gather that writes the
rows back (ur_embedding_gather_f32) and the fused gather-dot the training path uses for candidates (scorer_loss_fwd).
  tools/gather_pmc.sh <tag>   collects FETCH_SIZE / WRITE_SIZE / TCC hit-miss / TLB-walk proxies in separate passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unirec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, d = 100_000_000, 128
table = torch.empty(N, d, device=dev).normal_(0, 0.02)
n = 8 * 1024 * 1024
idx = torch.randint(1, N, (n,), device=dev)
for _ in range(3):
    out = ops.embedding_gather(table, idx)
    del out
B, G = 4096, 1001
cfg = ops.loss_cfg(B, G, d, "softmax")
cfg.loss_type = -1
user = torch.randn(B, d, device=dev)
ids = torch.randint(1, N, (B, G), device=dev)
for _ in range(3):
    ops.gather_dot_loss_fwd(cfg, user, table, ids)
torch.cuda.synchronize()
print("gather lookups", n, "gather-dot lookups", B * G, "row bytes", d * 4)
