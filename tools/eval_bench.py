#!/usr/bin/env python3
"""Micro-benchmark of ur_full_rank (one_vs_all evaluation) on one GPU.
Prints one JSON line per (n_items, batch): users/s, fp32-MFMA TFLOP/s (2*B*N*d / t) and table read GB/s (N*d*4 / t)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unirec_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-items", type=int, nargs="+", default=[2_000_000, 100_000_000])
    ap.add_argument("--batch", type=int, nargs="+", default=[512, 4096])
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--hist", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for N in a.n_items:
        table = torch.empty(N, a.d, device=dev).normal_(0, 0.05)
        for B in a.batch:
            g = torch.Generator(device=dev).manual_seed(1)
            ue = torch.empty(B, a.d, device=dev).normal_(0, 0.05, generator=g)
            tgt = torch.randint(1, N, (B,), device=dev, generator=g)
            uid = torch.arange(B, device=dev)
            hp = (torch.arange(B + 1, device=dev) * a.hist).to(torch.int64)
            hs = torch.randint(1, N, (B, a.hist), device=dev, generator=g).sort(1).values.to(torch.int32).reshape(-1).contiguous()
            ops.full_rank(ue, table, tgt, uid, hp, hs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                r, _ = ops.full_rank(ue, table, tgt, uid, hp, hs)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            print(json.dumps({"op": "ur_full_rank", "n_items": N, "batch": B, "d": a.d, "ms": round(ms, 3),
                              "users_per_s": round(B / ms * 1e3, 1), "tflops": round(2.0 * B * N * a.d / ms / 1e9, 2),
                              "table_read_GBps": round(N * a.d * 4 / ms / 1e6, 1), "mean_rank": float(r.float().mean())}))
            if B <= 512:
                ops.full_topk(ue, table, 100, uid, hp, hs)
                torch.cuda.synchronize()
                e0.record()
                sc, ids = ops.full_topk(ue, table, 100, uid, hp, hs)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                print(json.dumps({"op": "ur_full_topk", "k": 100, "n_items": N, "batch": B, "d": a.d, "ms": round(ms, 3),
                                  "users_per_s": round(B / ms * 1e3, 1), "best_score_mean": float(sc[:, 0].mean())}))
        del table


if __name__ == "__main__":
    main()
