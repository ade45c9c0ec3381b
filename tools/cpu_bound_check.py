#!/usr/bin/env python3
"""Is the training step host-bound?  Times the host-side enqueue of K steps (no sync) against the synchronised wall time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from unirec_amd.facility.optimizer import SparseDenseAdam  # noqa: E402
from unirec_amd.model.sequential.sasrec import SASRec  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    a = bench.parse()
    dev = torch.device("cuda:0")
    cfg = bench.model_config(a, "cuda:0")
    model = SASRec(cfg)
    opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
    model.train()
    batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=64)

    def step(b, nxt):
        opt.zero_grad()
        opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
        if not a.no_prefetch:
            opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
        if a.autograd:
            loss, _, _, _ = model(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
            loss.backward()
        else:
            model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
        opt.step(late_join=True)

    for i in range(10):
        step(batches[i % 64], batches[(i + 1) % 64])
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter()
    for i in range(K):
        step(batches[i % 64], batches[(i + 1) % 64])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, synchronised {1e3 * (t2 - t0) / K:.3f} ms/step")


if __name__ == "__main__":
    main()
