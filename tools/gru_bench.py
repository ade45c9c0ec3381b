#!/usr/bin/env python3
"""BASELINE config C4's encoder on ONE GPU: GRU4Rec, n_items = 10 M, d = H = 128, L = 50, B = 512, 4 negatives (the 8-way sharded
run is the driver's; this is the per-GPU step)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from unirec_amd import _lib  # noqa: E402
from unirec_amd.facility.optimizer import SparseDenseAdam  # noqa: E402
from unirec_amd.model.sequential.gru import GRU  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-items", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--hidden", type=int, default=128, help="768 = the reference yaml's default width")
    x = ap.parse_args()
    sys.argv = [sys.argv[0], "--n-items", str(x.n_items)]
    a = bench.parse()
    dev = torch.device("cuda:0")
    cfg = bench.model_config(a, "cuda:0")
    cfg.update(model="GRU", hidden_size=x.hidden, loss_type="softmax")
    model = GRU(cfg)
    opt = SparseDenseAdam(model, lr=1e-3, table_mode="lazy_dense")
    model.train()
    batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=x.steps + 12)

    def step(b, nxt):
        opt.zero_grad()
        opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
        opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
        model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
        opt.step()

    for i in range(10):
        step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    _lib.lib.ur_prof_reset(); _lib.lib.ur_prof_enable(1)
    for i in range(10, 12):
        step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    import ctypes as C
    n = _lib.lib.ur_prof_num_classes()
    ms, cnt, work = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
    _lib.lib.ur_prof_read(ms, cnt, work)
    _lib.lib.ur_prof_enable(0)
    _lib.lib.ur_prof_class_name.restype = C.c_char_p
    classes = {_lib.lib.ur_prof_class_name(i).decode(): (round(ms[i] / 2, 4), cnt[i] // 2) for i in range(n) if cnt[i]}
    t0 = time.perf_counter()
    for i in range(12, 12 + x.steps - 1):
        step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (x.steps - 1)
    print(json.dumps({"workload": f"GRU n_items={x.n_items} d=128 H={x.hidden} L=50 B=512 K=4 softmax", "ms_per_step": round(dt * 1e3, 4),
                      "examples_per_s": round(a.batch / dt, 1), "kernel_ms_and_launches_per_step": classes}))


if __name__ == "__main__":
    main()
