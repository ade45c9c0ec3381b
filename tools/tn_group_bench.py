#!/usr/bin/env python3
"""The bottom layer's weight-gradient products of one C5 backward pass as ONE ur_gemm_tn_group launch, `reps` times in the arithmetic
given (for rocprofv3 --kernel-trace / --pmc):  python tools/tn_group_bench.py <arith 0|6|9> [reps] [T]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from unirec_amd._lib import check, lib  # noqa: E402

arith = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 21248
dev = torch.device("cuda:0")
pro = os.environ.get("TN_PRO", "0001")   # which of the four products apply the activation to Q (the real layer: the last one)
shapes = [(384, 128, int(pro[0])), (128, 128, int(pro[1])), (512, 128, int(pro[2])), (128, 512, int(pro[3]))]
g = torch.Generator(device=dev).manual_seed(T)
Ps = [torch.randn(T, R, device=dev, generator=g) for R, _, _ in shapes]
Qs = [torch.randn(T, Cc, device=dev, generator=g) for _, Cc, _ in shapes]
outs = [torch.empty(R, Cc, device=dev) for R, Cc, _ in shapes]
bos = [torch.empty(R, device=dev) for R, _, _ in shapes]
wss = [torch.empty(lib.ur_gemm_tn_workspace_floats(T, R, Cc), device=dev) for R, Cc, _ in shapes]
n = len(shapes)
arr_p = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
arr_i = lambda vs: (C.c_int * n)(*vs)  # noqa: E731
args = (n, arr_p(Ps), arr_i([R for R, _, _ in shapes]), arr_p(Qs), arr_i([c for _, c, _ in shapes]), arr_i([T] * n),
        arr_i([R for R, _, _ in shapes]), arr_i([c for _, c, _ in shapes]), arr_i([pa for _, _, pa in shapes]), 2, arr_p(outs),
        arr_i([c for _, c, _ in shapes]), arr_p(bos), arr_p(wss))
check(lib.ur_set_mfma_arith(arith), "ur_set_mfma_arith")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.ur_gemm_tn_group(*args, st), "ur_gemm_tn_group")
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"arith {arith} UR_TEST={os.environ.get('UR_TEST', '')}: group launch + reduction {ts[len(ts) // 2]:.1f} us median, {ts[0]:.1f} min  "
      f"({sum(2.0 * T * R * c for R, c, _ in shapes) / ts[len(ts) // 2] / 1e6:.1f} TF/s)")
