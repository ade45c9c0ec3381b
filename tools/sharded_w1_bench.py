#!/usr/bin/env python3
"""Overhead of the row-sharded step's code path itself: ShardedSasrecStep with world = 1 (no collectives) against the plain
single-GPU step on the same batches.  What differs: the (owner, row) plan, the compact table of fetched rows, the owner-side
plan / reduction, the separate dense-Adam call."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from unirec_amd import _lib  # noqa: E402
from unirec_amd.sharded import ShardedSasrecStep  # noqa: E402


def main():
    sys.argv = [sys.argv[0], "--n-items", "10000000"] + sys.argv[1:]
    a = bench.parse()
    dev = torch.device("cuda:0")
    cfg = bench.model_config(a, "cuda:0")
    st = ShardedSasrecStep(cfg, dev, 0, 1, table_mode=a.table_mode)
    batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=80)
    for i in range(10):
        st.step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    _lib.lib.ur_prof_reset(); _lib.lib.ur_prof_enable(1)
    for i in range(10, 14):
        st.step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    import ctypes as C
    n = _lib.lib.ur_prof_num_classes()
    ms, cnt, work = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
    _lib.lib.ur_prof_read(ms, cnt, work)
    _lib.lib.ur_prof_enable(0)
    _lib.lib.ur_prof_class_name.restype = C.c_char_p
    classes = {_lib.lib.ur_prof_class_name(i).decode(): (round(ms[i] / 4, 4), cnt[i] // 4) for i in range(n) if cnt[i]}
    t0 = time.perf_counter()
    K = 60
    for i in range(14, 14 + K):
        st.step(batches[i], batches[i + 1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"path": "ShardedSasrecStep world=1", "ms_per_step": round(dt * 1e3, 4), "examples_per_s": round(a.batch / dt, 1),
                      "kernel_ms_and_launches_per_step": classes}))


if __name__ == "__main__":
    main()
