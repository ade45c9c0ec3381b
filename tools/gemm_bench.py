#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM kernels at the SASRec shapes (M = B*L tokens). Prints TFLOP/s."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unirec_amd._lib import lib, check

dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def time_it(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def nt(M, N, K, pro=0, epi=1):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cc = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev); aux = torch.randn(M, N, device=dev)
    g = torch.ones(N, device=dev); b = torch.zeros(N, device=dev); xh = torch.empty(M, N, device=dev); rs = torch.empty(M, device=dev)
    f = lambda: check(lib.ur_gemm_nt(p(A), K, p(W), K, p(Cc), N, M, N, K, pro, epi, 2, p(bias), p(aux), N, p(g), p(b), 1e-10, p(xh), p(rs), st()))
    us = time_it(f)
    print(f"NT M={M:6d} N={N:4d} K={K:4d} pro={pro} epi={epi}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")


def tn(T, R, Cc_, act=0):
    P = torch.randn(T, R, device=dev); Q = torch.randn(T, Cc_, device=dev); out = torch.empty(R, Cc_, device=dev); bo = torch.empty(R, device=dev)
    ws = torch.empty(lib.ur_gemm_tn_workspace_floats(T, R, Cc_), device=dev)
    f = lambda: check(lib.ur_gemm_tn(p(P), R, p(Q), Cc_, T, R, Cc_, act, 2, p(out), Cc_, p(bo), p(ws), st()))
    us = time_it(f)
    print(f"TN T={T:6d} R={R:4d} C={Cc_:4d} act={act}: {us:8.1f} us  {2.0*T*R*Cc_/us/1e6:7.1f} TF/s")


if __name__ == "__main__":
    if os.environ.get("NT_ONLY"):   # the one-product-per-launch shapes of the GRU (H = 128 / 768) in both arithmetics
        for arith in (0, 6):
            check(lib.ur_set_mfma_arith(arith))
            print(f"--- mfma_arith = {arith}")
            for (M, N, K, epi) in [(25600, 384, 128, 1), (25600, 128, 384, 0), (25600, 2304, 128, 1), (25600, 128, 2304, 0), (25600, 512, 128, 1),
                                   (25600, 128, 512, 4), (512, 2304, 768, 0)]:
                nt(M, N, K, 0, epi)
        check(lib.ur_set_mfma_arith(0))
        sys.exit(0)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
    for (N, K, pro, epi) in [(384, 128, 0, 1), (256, 128, 0, 1), (128, 128, 0, 2), (512, 128, 0, 1), (128, 512, 1, 2), (512, 128, 0, 3),
                             (128, 512, 0, 4), (128, 128, 0, 0), (128, 384, 0, 4), (128, 256, 0, 0)]:
        nt(M, N, K, pro, epi)
    for (R, Cc_, act) in [(128, 512, 1), (512, 128, 0), (128, 128, 0), (384, 128, 0), (256, 128, 0)]:
        tn(M, R, Cc_, act)
    nt(512, 128, 128, 0, 2); nt(512, 512, 128, 0, 1); tn(512, 128, 512, 1)
    nt(204800, 384, 128, 0, 1); nt(204800, 128, 512, 1, 2); tn(204800, 128, 512, 1)
