"""The two fp32-MFMA GEMM kernels behind nn.Linear (ur_gemm_nt / ur_gemm_tn) against an fp64 restatement, every
prologue / epilogue, ragged and strided shapes, small and large M.
Tolerance: 1e-4 relative to the largest output magnitude (fp32 accumulation of K <= 1024 products)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ACT = 2   # UR_ACT_SWISH


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _swish(x):
    return x * torch.sigmoid(x)


def _dswish(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def _close(got, ref, tol=1e-4):
    ref = ref.to(torch.float64)
    scale = max(1e-6, float(ref.abs().max()))
    err = float((got.to(torch.float64).cpu() - ref.cpu()).abs().max()) / scale
    assert err < tol, err


@pytest.fixture
def arith(request):
    """the arithmetic of the weight-gradient products for ONE test (ur_set_mfma_arith is process-wide): 0 = exact fp32-input MFMA,
    6 / 9 = split-bf16 terms; the raw hooks take the split kernel at every shape"""
    from unirec_amd._lib import check, lib
    check(lib.ur_set_mfma_arith(request.param), "ur_set_mfma_arith")
    assert lib.ur_get_mfma_arith() == request.param
    yield request.param
    check(lib.ur_set_mfma_arith(0), "ur_set_mfma_arith")


ARITHS = pytest.mark.parametrize("arith", [0, 6, 9], indirect=True)


@ARITHS
@pytest.mark.parametrize("M,N,K", [(25600, 384, 128), (700, 128, 512), (512, 512, 128), (512, 128, 128), (33, 100, 36), (1, 128, 128),
                                   (1025, 256, 64), (2000, 132, 260), (64, 200, 512)])
@pytest.mark.parametrize("pro,epi", [(0, 0), (0, 1), (0, 2), (1, 2), (0, 3), (0, 4)])
def test_gemm_nt(M, N, K, pro, epi, arith):
    """arith 6 / 9: the plain-epilogue products (epi 0, 1, 3, 4) run the six-term split-bf16 K loop (round 6d); the LayerNorm epilogue stays exact"""
    from unirec_amd._lib import check, lib
    if epi == 2 and N > 256:
        pytest.skip("fused LayerNorm epilogue: N <= 256")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M * 7 + N * 3 + K)
    lda, ldw, ldaux = K + 8, K + 4, N + 12          # strided operands
    A = torch.randn(M, lda, device=dev, generator=g)
    W = torch.randn(N, ldw, device=dev, generator=g) * 0.1
    aux = torch.randn(M, ldaux, device=dev, generator=g)
    bias = torch.randn(N, device=dev, generator=g)
    gamma, beta = torch.randn(N, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
    Cc = torch.full((M, N), float("nan"), device=dev)
    xhat, rstd = torch.empty(M, N, device=dev), torch.empty(M, device=dev)
    check(lib.ur_gemm_nt(_p(A), lda, _p(W), ldw, _p(Cc), N, M, N, K, pro, epi, ACT, _p(bias), _p(aux), ldaux, _p(gamma), _p(beta),
                         1e-10, _p(xhat), _p(rstd), _st()), "ur_gemm_nt")
    A64, W64, aux64 = A[:, :K].double(), W[:, :K].double(), aux[:, :N].double()
    acc = (_swish(A64) if pro else A64) @ W64.T
    if epi == 0:
        ref = acc
    elif epi == 1:
        ref = acc + bias.double()
    elif epi == 2:
        t = acc + bias.double() + aux64
        mu = t.mean(1, keepdim=True)
        var = ((t - mu) ** 2).mean(1, keepdim=True)
        xh = (t - mu) / torch.sqrt(var + 1e-10)
        ref = xh * gamma.double() + beta.double()
        _close(xhat, xh, 2e-4)
        _close(rstd, 1 / torch.sqrt(var + 1e-10).squeeze(1), 2e-4)
    elif epi == 3:
        ref = acc * _dswish(aux64)
    else:
        ref = acc + aux64
    _close(Cc, ref, 2e-4 if epi == 2 else 1e-4)




@ARITHS
@pytest.mark.parametrize("T,R,Cc_", [(25600, 128, 512), (25600, 384, 128), (512, 128, 128), (700, 100, 36), (5, 128, 128), (3000, 260, 132)])
@pytest.mark.parametrize("act_on_q", [0, 1])
def test_gemm_tn(T, R, Cc_, act_on_q, arith):
    from unirec_amd._lib import check, lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T + R + Cc_)
    ldp, ldq = R + 4, Cc_ + 8
    P = torch.randn(T, ldp, device=dev, generator=g)
    Q = torch.randn(T, ldq, device=dev, generator=g)
    out = torch.full((R, Cc_), float("nan"), device=dev)
    bo = torch.full((R,), float("nan"), device=dev)
    ws = torch.empty(int(lib.ur_gemm_tn_workspace_floats(T, R, Cc_)), device=dev)
    for _ in range(2):   # twice: the second run must be bit-identical (fixed-order split reduction)
        check(lib.ur_gemm_tn(_p(P), ldp, _p(Q), ldq, T, R, Cc_, act_on_q, ACT, _p(out), Cc_, _p(bo), _p(ws), _st()), "ur_gemm_tn")
        if _ == 0:
            first = (out.clone(), bo.clone())
    assert torch.equal(first[0], out) and torch.equal(first[1], bo)
    P64, Q64 = P[:, :R].double(), Q[:, :Cc_].double()
    ref = P64.T @ (_swish(Q64) if act_on_q else Q64)
    _close(out, ref)
    _close(bo, P64.sum(0))


_ACTS = {0: lambda x: 0.5 * x * (1 + torch.erf(x * 0.7071067811865476)), 1: torch.relu, 2: _swish, 3: torch.tanh, 4: torch.sigmoid}


@pytest.mark.parametrize("T", [1, 31, 33, 64, 65, 1000, 4097, 21248])
@pytest.mark.parametrize("R,Cc_", [(128, 128), (256, 128), (128, 512), (512, 128), (384, 128)])
@ARITHS
@pytest.mark.parametrize("act", [-1, 0, 1, 2, 3, 4])
def test_gemm_tn_128_tile_shapes(T, R, Cc_, act, arith):
    """The d = 128-class weight-gradient shapes (R, Cc multiples of 128): every activation on the Q operand, token counts around the
    32-token stage and the split boundaries, strided operands -- in the exact fp32 arithmetic (64 x 64 tiles) and in the split-bf16
    arithmetic (round 6: 128 x 128 tiles, 32-token stages, ragged last stage peeled)."""
    from unirec_amd._lib import check, lib
    if act >= 0 and (T, R) not in ((33, 128), (1000, 128), (21248, 128), (4097, 256)):
        pytest.skip("activations: a subset of the shapes")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T * 3 + R + Cc_ + act)
    ldp, ldq = R + 4, Cc_ + 8
    P = torch.randn(T, ldp, device=dev, generator=g)
    Q = torch.randn(T, ldq, device=dev, generator=g)
    out = torch.full((R, Cc_), float("nan"), device=dev)
    bo = torch.full((R,), float("nan"), device=dev)
    ws = torch.empty(int(lib.ur_gemm_tn_workspace_floats(T, R, Cc_)), device=dev)
    for rep in range(2):
        check(lib.ur_gemm_tn(_p(P), ldp, _p(Q), ldq, T, R, Cc_, 1 if act >= 0 else 0, max(act, 0), _p(out), Cc_, _p(bo), _p(ws), _st()), "ur_gemm_tn")
        if rep == 0:
            first = (out.clone(), bo.clone())
    assert torch.equal(first[0], out) and torch.equal(first[1], bo)
    P64, Q64 = P[:, :R].double(), Q[:, :Cc_].double()
    ref = P64.T @ (_ACTS[act](Q64) if act >= 0 else Q64)
    _close(out, ref)
    _close(bo, P64.sum(0))


@pytest.mark.parametrize("kind", ["randn", "scaled", "cancelling"])
@pytest.mark.parametrize("M,N,K", [(25600, 384, 128), (25600, 128, 384), (4096, 2304, 128), (25600, 128, 2304)])
def test_split_bf16_nt_products_are_fp32_equivalent(M, N, K, kind):
    """the same gate for ur_gemm_nt's split K loop (the GRU's input projection and its gradient at H = 128 / 768): error against an
    fp64 product of the same fp32 operands, measured in units of sum_k |a w|, at most 1.5 x the exact fp32-MFMA loop's"""
    from unirec_amd._lib import check, lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A, W = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    if kind == "scaled":          # column scales over four decades on both operands
        A = A * torch.exp(torch.randn(K, device=dev, generator=g) * 2.3)
        W = W * torch.exp(torch.randn(K, device=dev, generator=g) * 2.3) * 1e-3
    elif kind == "cancelling":    # the exact sum nearly cancels
        h = K // 2
        A[:, h:2 * h] = -A[:, :h] * (1 + 1e-3 * torch.randn(M, h, device=dev, generator=g))
        W[:, h:2 * h] = W[:, :h]
    ref = A.double() @ W.double().T
    yard = A.double().abs() @ W.double().abs().T
    err = {}
    try:
        for a in (0, 6):
            check(lib.ur_set_mfma_arith(a), "ur_set_mfma_arith")
            out = torch.full((M, N), float("nan"), device=dev)
            check(lib.ur_gemm_nt(_p(A), K, _p(W), K, _p(out), N, M, N, K, 0, 0, ACT, None, None, 0, None, None, 1e-10, None, None, _st()), "ur_gemm_nt")
            err[a] = float(((out.double() - ref).abs() / yard).max())
    finally:
        check(lib.ur_set_mfma_arith(0), "ur_set_mfma_arith")
    assert err[0] < 5e-6, err                 # (the worst of M x N outputs, each an fmaf chain over K products)
    assert err[6] <= 1.5 * err[0], err


@pytest.mark.parametrize("kind", ["randn", "grad-like", "cancelling"])
@pytest.mark.parametrize("R,Cc_", [(384, 128), (128, 512)])
def test_split_bf16_products_are_fp32_equivalent(R, Cc_, kind):
    """VERDICT r5 item 1's gate, kept as a test: against an fp64 product of the SAME fp32 operands, the error of the six- and nine-term
    split-bf16 arithmetic (max over outputs of |err| / sum_t |p q|, the fp32-roundoff yardstick) is at most 1.5 x the exact fp32-MFMA
    kernel's -- and the three-term split, which IS narrower than fp32, fails that by an order of magnitude (so the yardstick can tell)."""
    from unirec_amd._lib import check, lib
    dev = torch.device("cuda:0")
    T = 21248
    g = torch.Generator(device=dev).manual_seed(R + Cc_)
    if kind == "randn":
        P, Q = torch.randn(T, R, device=dev, generator=g), torch.randn(T, Cc_, device=dev, generator=g)
    elif kind == "grad-like":     # small gradients with column scales over four decades x swish activations
        P = torch.randn(T, R, device=dev, generator=g) * (torch.exp(torch.randn(R, device=dev, generator=g) * 2.3) * 1e-5)
        x = torch.randn(T, Cc_, device=dev, generator=g) * 3
        Q = x * torch.sigmoid(x)
    else:                         # the exact sum nearly cancels: the yardstick is ~1e3 x the result
        P, Q = torch.randn(T, R, device=dev, generator=g), torch.randn(T, Cc_, device=dev, generator=g)
        P[T // 2:] = -P[:T - T // 2] * (1 + 1e-3 * torch.randn(T - T // 2, R, device=dev, generator=g))
        Q[T // 2:] = Q[:T - T // 2]
    ref = P.double().T @ Q.double()
    yard = P.double().abs().T @ Q.double().abs()
    ws = torch.empty(int(lib.ur_gemm_tn_workspace_floats(T, R, Cc_)), device=dev)
    err = {}
    try:
        for a in (0, 6, 9, 3):
            check(lib.ur_set_mfma_arith(a), "ur_set_mfma_arith")
            out = torch.full((R, Cc_), float("nan"), device=dev)
            check(lib.ur_gemm_tn(_p(P), R, _p(Q), Cc_, T, R, Cc_, 0, 0, _p(out), Cc_, None, _p(ws), _st()), "ur_gemm_tn")
            err[a] = float(((out.double() - ref).abs() / yard).max())
    finally:
        check(lib.ur_set_mfma_arith(0), "ur_set_mfma_arith")
    assert err[0] < 1e-7                      # an fmaf chain over 21 248 products, split 32 ways
    assert err[6] <= 1.5 * err[0], err
    assert err[9] <= 1.5 * err[0], err
    assert err[3] > 5 * err[0], err


@ARITHS
def test_gemm_tn_group_equals_one_product_at_a_time(arith):
    """ur_gemm_tn_group (how the encoders' backward passes issue the products queued at one fork) == ur_gemm_tn per product up to the
    summation order of the token splits (a group gives every product fewer splits), bias gradients included."""
    from unirec_amd._lib import check, lib
    dev = torch.device("cuda:0")
    T = 5000
    shapes = [(384, 128, 0), (128, 128, 0), (512, 128, 0), (128, 512, 1), (68, 36, 0)]
    g = torch.Generator(device=dev).manual_seed(7)
    n = len(shapes)
    Ps = [torch.randn(T, R, device=dev, generator=g) for R, _, _ in shapes]
    Qs = [torch.randn(T, c, device=dev, generator=g) for _, c, _ in shapes]
    outs = [torch.full((R, c), float("nan"), device=dev) for R, c, _ in shapes]
    bos = [torch.full((R,), float("nan"), device=dev) for R, _, _ in shapes]
    wss = [torch.empty(int(lib.ur_gemm_tn_workspace_floats(T, R, c)), device=dev) for R, c, _ in shapes]
    ap = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    ai = lambda vs: (C.c_int * n)(*vs)  # noqa: E731
    check(lib.ur_gemm_tn_group(n, ap(Ps), ai([R for R, _, _ in shapes]), ap(Qs), ai([c for _, c, _ in shapes]), ai([T] * n),
                               ai([R for R, _, _ in shapes]), ai([c for _, c, _ in shapes]), ai([pa for _, _, pa in shapes]), ACT, ap(outs),
                               ai([c for _, c, _ in shapes]), ap(bos), ap(wss), _st()), "ur_gemm_tn_group")
    for (R, c, pa), P, Q, o, b in zip(shapes, Ps, Qs, outs, bos):
        ref = P.double().T @ (_swish(Q.double()) if pa else Q.double())
        _close(o, ref)
        _close(b, P.double().sum(0))

