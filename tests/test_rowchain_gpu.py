"""Row-chain kernels (csrc/rowchain.hip) against the one-GEMM-per-launch path they replace, on identical inputs.

The chain kernels keep the K order of the stand-alone GEMMs, so the FORWARD agrees to fp32 rounding of the LayerNorm sums (1e-6
of the output scale; the MFMA sums themselves are the same sequence); the backward sums the LayerNorm-affine partials over a different workgroup partition,
so gradients are compared at 2e-6 of each tensor's scale (pure fp32 re-association).  Shapes: the three supported widths
(d = 32 / 64 / 128), ragged row counts (padding skipped: the row count lives on the device), 1-3 layers with and without the
last-row specialisation, hidden dropout (forward only: the backward then takes the unfused path), BASELINE's C5 shape.
Parity with the reference itself is covered by the golden tests, which run through the same kernels (d = 32, inner = 64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cfg_kw, B, seed, chain, train_drop=0.0):
    from unirec_amd import _lib, ops
    dev = torch.device("cuda:0")
    d, L, I, H, nl = cfg_kw["d"], cfg_kw["L"], cfg_kw["inner"], cfg_kw["heads"], cfg_kw["layers"]
    N = 5000
    prev = _lib.lib.ur_sasrec_set_chain(int(chain) if not isinstance(chain, bool) else (127 if chain else 0))   # bit mask: 1 fwd, 2 bwd, 4 projection, 8 / 16 fwd / bwd of the last-row layer (row-chain kernels), 32 input block, 64 the last-row layer as two launches (lastrow.hip)
    try:
        cfg = ops.sasrec_cfg(B, L, d, H, I, nl, cfg_kw.get("act", "swish"), True, 1e-10, last_only=cfg_kw.get("last_only", 1),
                             skip_padding=cfg_kw.get("skip_padding", 1), p_hidden=train_drop, p_attn=0.0, drop_seed=7, drop_step=3)
        offs, total = ops.sasrec_param_layout(cfg)
        g = torch.Generator(device=dev).manual_seed(seed)
        dense = torch.randn(total, device=dev, generator=g) * 0.08
        table = torch.randn(N, d, device=dev, generator=g) * 0.1
        table[0] = 0
        seq = torch.randint(1, N, (B, L), device=dev, generator=g, dtype=torch.int32)
        lens = torch.randint(0, L + 1, (B,), device=dev, generator=g)
        lens[0] = L
        if B > 1:
            lens[1] = 0                    # an all-padding row
        seq = torch.where(torch.arange(L, device=dev).unsqueeze(0) >= (L - lens).unsqueeze(1), seq, torch.zeros_like(seq)).contiguous()
        d_user = torch.randn(B, d, device=dev, generator=g)
        ws = ops.sasrec_workspace(cfg, dev)
        ws.zero_()
        ue = ops.sasrec_fwd(cfg, table, dense, seq, ws).clone()
        dense_grad, d_rows = ops.sasrec_bwd(cfg, table, dense, seq, d_user, ws)
        torch.cuda.synchronize()
        return ue, dense_grad.clone(), d_rows.clone()
    finally:
        _lib.lib.ur_sasrec_set_chain(prev)


def _close(a, b, tol, what):
    scale = max(1e-12, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err < tol, (what, err)


@pytest.mark.parametrize("kw", [
    dict(d=128, L=50, inner=512, heads=16, layers=2),                       # C5 layer shape
    dict(d=128, L=50, inner=512, heads=16, layers=2, last_only=0),
    dict(d=128, L=20, inner=128, heads=8, layers=3),
    dict(d=64, L=50, inner=256, heads=16, layers=2),                        # C2 width
    dict(d=64, L=12, inner=64, heads=4, layers=1, last_only=0),
    dict(d=32, L=10, inner=64, heads=2, layers=2),
    dict(d=32, L=10, inner=96, heads=4, layers=3, last_only=0, act="gelu"),
    dict(d=128, L=50, inner=512, heads=16, layers=2, skip_padding=0),       # padded rows: host-side row count
    dict(d=128, L=200, inner=256, heads=16, layers=2),                      # C3 sequence length
    dict(d=128, L=30, inner=256, heads=16, layers=1),                       # ONE layer that is the last-row layer: the input block projects K, V only
    dict(d=64, L=20, inner=128, heads=8, layers=1, act="gelu"),
])
@pytest.mark.parametrize("B", [1, 37, 512])
def test_chain_equals_unfused(kw, B):
    if B == 512 and kw["L"] == 200:
        B = 128
    ue1, dg1, dr1 = _run(kw, B, 11, chain=True)
    ue0, dg0, dr0 = _run(kw, B, 11, chain=False)
    assert torch.isfinite(ue1).all() and torch.isfinite(dg1).all() and torch.isfinite(dr1).all()
    _close(ue1, ue0, 1e-6, "user_emb")        # same K order; only the compiler's fma contraction of the LayerNorm sums may differ
    _close(dr1, dr0, 2e-6, "d_emb_rows")
    from unirec_amd import ops
    cfg = ops.sasrec_cfg(B, kw["L"], kw["d"], kw["heads"], kw["inner"], kw["layers"], kw.get("act", "swish"), True, 1e-10)
    offs, total = ops.sasrec_param_layout(cfg)
    bounds = list(offs) + [total]
    for j in range(len(offs)):               # every parameter tensor at its own scale
        if j >= 3 and (j - 3) % 16 == 4:     # key.bias: analytically zero (softmax shift invariance), both sides hold rounding noise
            continue
        a, b = dg1[bounds[j]:bounds[j + 1]], dg0[bounds[j]:bounds[j + 1]]
        if b.numel() and float(b.abs().max()) > 1e-9:
            _close(a, b, 3e-4, f"param {j}")   # B = 1: a few dozen rows, rounding differences of the activations are not averaged out


@pytest.mark.parametrize("mask", [1, 2, 4, 5, 8, 16, 24, 25, 32, 57, 64, 65, 103, 63])
def test_chain_phases_are_independent_switches(mask):
    """forward / backward / projection-gradient chains and the last-row layer's two can be switched on one by one (ur_sasrec_set_chain bit mask)"""
    kw = dict(d=128, L=50, inner=512, heads=16, layers=2)
    ue1, dg1, dr1 = _run(kw, 64, 3, chain=mask)
    ue0, dg0, dr0 = _run(kw, 64, 3, chain=0)
    _close(ue1, ue0, 1e-6, "user_emb")
    _close(dr1, dr0, 5e-6, "d_emb_rows")
    _close(dg1, dg0, 3e-4, "dense_grad")


@pytest.mark.parametrize("d,inner", [(128, 512), (64, 128), (32, 64)])
def test_chain_with_hidden_dropout_matches_the_unfused_path(d, inner):
    kw = dict(d=d, L=30, inner=inner, heads=4, layers=2)
    ue1, dg1, dr1 = _run(kw, 64, 5, chain=True, train_drop=0.3)
    ue0, dg0, dr0 = _run(kw, 64, 5, chain=False, train_drop=0.3)
    _close(ue1, ue0, 1e-6, "user_emb")
    _close(dr1, dr0, 1e-5, "d_emb_rows")                       # hidden dropout: the chain backward re-evaluates the forward's masks
    _close(dg1, dg0, 1e-5, "dense_grad")                       # (round 3; until then both backward passes were the unfused one)


def test_unsupported_widths_take_the_unfused_path():
    kw = dict(d=96, L=10, inner=96, heads=12, layers=2)
    ue1, dg1, dr1 = _run(kw, 9, 2, chain=True)
    ue0, dg0, dr0 = _run(kw, 9, 2, chain=False)
    assert torch.equal(ue1, ue0), float((ue1 - ue0).abs().max())
    assert torch.equal(dr1, dr0), float((dr1 - dr0).abs().max())
    assert torch.equal(dg1, dg0), (float((dg1 - dg0).abs().max()), torch.nonzero(dg1 != dg0).flatten()[:8].tolist())


def test_last_row_split_kernels_are_race_free_over_repeated_steps():
    """The last-row layer's chain kernels split the inner dimension over workgroups and hand partial tiles to the workgroup of the row
    block that finishes last (device-scope atomics + a counter per block, no device-scope fence).  A hand-off that is merely USUALLY in
    order shows up only when the partial buffers carry the previous step's values: 30 steps with fresh inputs on ONE workspace, twice
    -- the two runs must agree bit for bit (fixed summation order), and with the unsplit launches to fp32 rounding.  (The first version
    used device-scope stores: 6 of 40 backward passes summed a stale partial.)"""
    from unirec_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, L, d, I, H, nl, N = 512, 50, 128, 512, 16, 2, 20000
    cfg = ops.sasrec_cfg(B, L, d, H, I, nl, "swish", True, 1e-10, last_only=1, skip_padding=1, p_hidden=0.0, p_attn=0.0, drop_seed=7, drop_step=3)
    _, total = ops.sasrec_param_layout(cfg)

    def run(mask):
        prev = _lib.lib.ur_sasrec_set_chain(mask)
        try:
            g = torch.Generator(device=dev).manual_seed(0)
            ws = ops.sasrec_workspace(cfg, dev)
            ws.zero_()
            outs = []
            for _ in range(30):
                dense = torch.randn(total, device=dev, generator=g) * 0.08
                table = torch.randn(N, d, device=dev, generator=g) * 0.1
                seq = torch.randint(1, N, (B, L), device=dev, generator=g, dtype=torch.int32)
                du = torch.randn(B, d, device=dev, generator=g)
                ue = ops.sasrec_fwd(cfg, table, dense, seq, ws).clone()
                dg, dr = ops.sasrec_bwd(cfg, table, dense, seq, du, ws)
                outs.append((ue, dg.clone(), dr.clone()))
            torch.cuda.synchronize()
            return outs
        finally:
            _lib.lib.ur_sasrec_set_chain(prev)

    a, b, ref = run(57), run(57), run(33)        # 33: forward chain + input block, the last-row layer through the stand-alone GEMMs
    for step, (x, y, z) in enumerate(zip(a, b, ref)):
        for k, tol in ((0, 5e-6), (1, 3e-4), (2, 5e-6)):
            assert torch.equal(x[k], y[k]), (step, k)
            _close(x[k], z[k], tol, (step, k))
