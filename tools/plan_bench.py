#!/usr/bin/env python3
"""Micro-benchmark of ur_rows_plan (the batch's id sort + segment heads).  UR_TEST=plan_multi forces the multi-launch path."""
import os
import sys
import json

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unirec_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for (B, L, G, N) in ((512, 50, 5, 100_000_000), (512, 50, 5, 60_000), (128, 200, 1001, 2_000_000)):
        g = torch.Generator(device=dev).manual_seed(0)
        a = torch.randint(1, N, (B * L,), device=dev, generator=g, dtype=torch.int32)
        b = torch.randint(1, N, (B * G,), device=dev, generator=g, dtype=torch.int64)
        for _ in range(3):
            pl = ops.rows_plan(a, b, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            pl = ops.rows_plan(a, b, N)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"op": "ur_rows_plan", "n": B * (L + G), "n_rows": N, "us": round(e0.elapsed_time(e1) / reps * 1e3, 1),
                          "n_uniq": int(pl.n_uniq.item()), "multi": "plan_multi" in os.environ.get("UR_TEST", "")}))


if __name__ == "__main__":
    main()
