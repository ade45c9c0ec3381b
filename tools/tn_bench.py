#!/usr/bin/env python3
"""gemm_tn shapes of one C5 step, one launch each per repetition (kernel durations are read from rocprofv3 --kernel-trace --stats)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unirec_amd._lib import lib, check

dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 21360
shapes = [(T, 128, 512), (T, 512, 128), (T, 128, 128), (T, 384, 128), (T, 256, 128), (512, 128, 512), (512, 128, 128)]
bufs = []
for (t, R, Cc) in shapes:
    P = torch.randn(t, R, device=dev); Q = torch.randn(t, Cc, device=dev); out = torch.empty(R, Cc, device=dev); bo = torch.empty(R, device=dev)
    ws = torch.empty(lib.ur_gemm_tn_workspace_floats(t, R, Cc), device=dev)
    bufs.append((t, R, Cc, P, Q, out, bo, ws))
for rep in range(20):
    for (t, R, Cc, P, Q, out, bo, ws) in bufs:
        check(lib.ur_gemm_tn(p(P), R, p(Q), Cc, t, R, Cc, 0, 2, p(out), Cc, p(bo), p(ws), st()))
        torch.cuda.synchronize()
