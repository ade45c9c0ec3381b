#!/usr/bin/env python3
"""End-to-end Trainer.fit throughput INCLUDING the input pipeline: interaction pairs + CSR history in
HBM, rows built on the device, SASRec C5 shape.  Compare with bench.py (pre-built batches) and with the host builder."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR, HostRowBuilder  # noqa: E402
from unirec_amd.facility.trainer import DeviceBatchLoader, Trainer  # noqa: E402
from unirec_amd.utils.argument_parser import parse_arguments  # noqa: E402
from unirec_amd.utils.general import get_class_instance, init_seed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-items", type=int, default=10_000_000)
    ap.add_argument("--n-users", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dropout", type=float, default=0.0, help="hidden / attention dropout (reference yaml default: 0.5)")
    ap.add_argument("--host", action="store_true", help="build rows with the native HOST builder instead")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    lens = np.clip(np.exp(rng.normal(4.25, 1.0, a.n_users)).astype(np.int64), 5, 1000)
    ptr = np.zeros(a.n_users + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    items = rng.integers(1, a.n_items, int(ptr[-1])).astype(np.int32)
    csr = HistoryCSR.__new__(HistoryCSR)
    csr.ptr, csr.items, csr.n_users, csr._dev = ptr, items, a.n_users, None
    csr.sorted = items.copy()
    for u in range(a.n_users):
        csr.sorted[ptr[u]:ptr[u + 1]].sort()
    n_pairs = a.batch * a.steps
    users = rng.integers(0, a.n_users, n_pairs)
    pos = items[ptr[users] + (rng.random(n_pairs) * lens[users]).astype(np.int64)]
    pairs = np.stack([users, pos.astype(np.int64)], 1)
    cfg = parse_arguments(dict(model="SASRec", n_users=a.n_users, n_items=a.n_items, device="cuda:0", loss_type="bpr", embedding_size=128,
                               hidden_size=128, inner_size=512, n_heads=16, n_layers=2, max_seq_len=50, epochs=1, batch_size=a.batch, seed=1,
                               hidden_dropout_prob=a.dropout, attn_dropout_prob=a.dropout))
    init_seed(1)
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    tr = Trainer(cfg, model)
    if a.host:
        from unirec_amd.facility.trainer import BatchLoader

        class DS:   # minimal dataset over the host builder
            def __init__(self):
                self.b = HostRowBuilder(a.n_users, a.n_items, 4, 50, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=1)
            def __len__(self):
                return len(pairs)
            def get_batch(self, idx):
                return self.b.build(pairs[idx, 0], pairs[idx, 1])
        loader = BatchLoader(DS(), a.batch, device="cuda:0")
    else:
        bld = DeviceRowBuilder(a.n_users, a.n_items, 4, 50, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=1)
        loader = DeviceBatchLoader(pairs, bld, a.batch)
    it = iter(loader)
    for _ in range(10):   # warm-up steps outside the clock
        tr.train_step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    cur = next(it, None)
    while cur is not None:
        nxt = next(it, None)
        tr.train_step(cur, nxt)
        cur = nxt
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"pipeline": "host builder + H2D" if a.host else "device-resident", "steps": n, "ms_per_step": round(dt / n * 1e3, 4),
                      "examples_per_s": round(n * a.batch / dt, 1), "n_items": a.n_items}))


if __name__ == "__main__":
    main()
