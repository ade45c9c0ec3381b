#!/usr/bin/env python3
"""Input-pipeline throughput (rows/s): native host row builder (MT19937 stream of the reference) vs the device-resident
builder.  The reference's own __getitem__ path measured 1.25 K rows/s per worker at K = 1000."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR, HostRowBuilder  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    n_users, n_items = 20000, 2_000_000
    lens = np.clip(np.exp(rng.normal(4.25, 1.0, n_users)).astype(np.int64), 5, 1000)
    u2h = np.empty(n_users, dtype=object)
    for u in range(n_users):
        u2h[u] = rng.integers(1, n_items, lens[u]).astype(np.int32)
    csr = HistoryCSR(u2h)
    for (B, L, K) in ((512, 50, 4), (128, 200, 1000)):
        user = rng.integers(0, n_users, B).astype(np.int64)
        pos = np.array([int(u2h[u][rng.integers(0, len(u2h[u]))]) for u in user], dtype=np.int64)
        host = HostRowBuilder(n_users, n_items, K, L, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=1)
        host.build(user, pos)
        t0 = time.perf_counter(); reps = 20
        for _ in range(reps):
            rows = host.build(user, pos)
            _ = {k: torch.from_numpy(v).cuda(non_blocking=True) for k, v in rows.items()}
        torch.cuda.synchronize()
        t_host = (time.perf_counter() - t0) / reps
        dev = DeviceRowBuilder(n_users, n_items, K, L, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=1)
        ud, pd_ = torch.from_numpy(user).cuda(), torch.from_numpy(pos).cuda()
        dev.build(ud, pd_)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 200
        for _ in range(reps):
            dev.build(ud, pd_)
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
        print(json.dumps({"B": B, "L": L, "K": K, "host_rows_per_s": round(B / t_host), "device_rows_per_s": round(B / t_dev),
                          "host_ms_per_batch": round(t_host * 1e3, 3), "device_ms_per_batch": round(t_dev * 1e3, 3)}))


if __name__ == "__main__":
    main()
