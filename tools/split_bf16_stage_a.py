#!/usr/bin/env python3
"""Round 6, stage A of the split-bf16 study: the weight-gradient product out[R,C] = P[T,R]^T Q[T,C] at the C5 shapes in every
arithmetic the library offers (ur_set_mfma_arith: 0 = exact fp32 MFMA, 6 / 9 = bf16 split terms, 3 = narrower-than-fp32 control),
error against an fp64 product of the SAME fp32 operands and time per call (kernel + deferred reduction, HIP events).

  err_rel  = max over outputs of |out - ref| / sum_t |p q|      (the fp32-roundoff yardstick: ~1e-7 for an fmaf chain)
  err_rms  = rms  over outputs of the same ratio
  err_out  = max |out - ref| / max |ref|                        (what tests/test_gemm_gpu.py bounds by 1e-4)

Gate: err_rel of the split <= 1.5 x err_rel of the exact-fp32 kernel."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from unirec_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def operands(kind, T, R, Cc, g):
    if kind == "randn":
        return torch.randn(T, R, device=dev, generator=g), torch.randn(T, Cc, device=dev, generator=g)
    if kind == "grad-like":   # small gradients with column scales over four decades x activations of mixed sign and scale
        sc = torch.exp(torch.randn(R, device=dev, generator=g) * 2.3) * 1e-5
        P = torch.randn(T, R, device=dev, generator=g) * sc
        x = torch.randn(T, Cc, device=dev, generator=g) * 3
        return P, x * torch.sigmoid(x)
    if kind == "cancelling":  # a product whose exact sum nearly cancels: the yardstick sum |p q| is ~1e3 x the result
        P = torch.randn(T, R, device=dev, generator=g)
        Q = torch.randn(T, Cc, device=dev, generator=g)
        P[T // 2:] = -P[:T - T // 2] * (1 + 1e-3 * torch.randn(T - T // 2, R, device=dev, generator=g))
        Q[T // 2:] = Q[:T - T // 2]
        return P, Q
    raise ValueError(kind)


def run(T, R, Cc, kind, reps=30):
    g = torch.Generator(device=dev).manual_seed(T + 7 * R + 13 * Cc)
    P, Q = operands(kind, T, R, Cc, g)
    ref = P.double().T @ Q.double()
    yard = P.double().abs().T @ Q.double().abs()
    bref = P.double().sum(0)
    ws = torch.empty(lib.ur_gemm_tn_workspace_floats(T, R, Cc), device=dev)
    out, bo = torch.empty(R, Cc, device=dev), torch.empty(R, device=dev)
    rows = []
    for arith in (0, 6, 9, 3):
        check(lib.ur_set_mfma_arith(arith), "ur_set_mfma_arith")
        out.fill_(float("nan")); bo.fill_(float("nan"))
        check(lib.ur_gemm_tn(p(P), R, p(Q), Cc, T, R, Cc, 0, 0, p(out), Cc, p(bo), p(ws), st()), "ur_gemm_tn")
        torch.cuda.synchronize()
        err = (out.double() - ref).abs()
        ratio = err / yard.clamp_min(1e-300)
        berr = float((bo.double() - bref).abs().max() / bref.abs().max())
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.ur_gemm_tn(p(P), R, p(Q), Cc, T, R, Cc, 0, 0, p(out), Cc, p(bo), p(ws), st())
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        rows.append(dict(arith=arith, err_rel=float(ratio.max()), err_rms=float((ratio ** 2).mean().sqrt()),
                         err_out=float(err.max() / ref.abs().max()), bias_err=berr, us_med=ts[len(ts) // 2], us_min=ts[0],
                         tflops_med=2.0 * T * R * Cc / ts[len(ts) // 2] / 1e6))
    check(lib.ur_set_mfma_arith(0), "ur_set_mfma_arith")
    return rows


def group(T, reps=30):
    """the bottom layer's weight-gradient products of one C5 backward pass in ONE launch (ur_gemm_tn_group; the in-situ shape)"""
    shapes = [(384, 128, 0), (128, 128, 0), (512, 128, 0), (128, 512, 1)]
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
    flop = sum(2.0 * T * R * Cc for R, Cc, _ in shapes)
    rows = []
    for arith in (0, 6, 9):
        check(lib.ur_set_mfma_arith(arith), "ur_set_mfma_arith")
        check(lib.ur_gemm_tn_group(*args, st()), "ur_gemm_tn_group")
        torch.cuda.synchronize()
        x = Qs[3].double()
        errs = []
        for (R, Cc, pa), P, Q, o in zip(shapes, Ps, Qs, outs):
            Q64 = Q.double() * torch.sigmoid(Q.double()) if pa else Q.double()
            ref, yard = P.double().T @ Q64, P.double().abs().T @ Q64.abs()
            errs.append(float(((o.double() - ref).abs() / yard).max()))
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.ur_gemm_tn_group(*args, st())
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        rows.append(dict(arith=arith, err_rel_max=max(errs), us_med=ts[len(ts) // 2], us_min=ts[0], tflops_med=flop / ts[len(ts) // 2] / 1e6))
        print(f"group of 4 (T={T}: 384x128, 128x128, 512x128, 128x512 with swish on Q)  arith {arith}: err_rel {max(errs):.3e}  "
              f"{ts[len(ts) // 2]:.1f} us med {ts[0]:.1f} min  {rows[-1]['tflops_med']:.1f} TF/s (launch + reduction)")
    check(lib.ur_set_mfma_arith(0), "ur_set_mfma_arith")
    return rows


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 21248
    res = [dict(group=group(T))]
    print(f"{'shape':>18} {'operands':>11} {'arith':>5} {'err_rel':>10} {'err_rms':>10} {'err_out':>10} {'bias_err':>9} {'us med':>8} {'us min':>8} {'TF/s':>7}  gate")
    for (R, Cc) in ((384, 128), (128, 512), (512, 128), (128, 128)):
        for kind in ("randn", "grad-like", "cancelling"):
            rows = run(T, R, Cc, kind)
            base = rows[0]["err_rel"]
            for r in rows:
                gate = "" if r["arith"] == 0 else ("PASS" if r["err_rel"] <= 1.5 * base else "FAIL")
                print(f"{T:>6}x{R:>4}x{Cc:>4}   {kind:>11} {r['arith']:>5} {r['err_rel']:>10.3e} {r['err_rms']:>10.3e} {r['err_out']:>10.3e} "
                      f"{r['bias_err']:>9.1e} {r['us_med']:>8.1f} {r['us_min']:>8.1f} {r['tflops_med']:>7.1f}  {gate}")
                res.append(dict(T=T, R=R, C=Cc, operands=kind, gate=gate, **r))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
