"""Row-chain kernels (csrc/rowchain.hip) against the one-GEMM-per-launch path they replace, on identical inputs.

These comparisons run in the EXACT arithmetic (mfma_arith = 0: the fp32-input MFMA on both sides); the split-bf16 form of the chains
(the default, round 6) is held against an fp64 evaluation of the oracle instead -- its error may not exceed 1.5 x the exact chains'
(test_split_row_chains_are_fp32_equivalent, the yardstick of tools/split_bf16_stage_a.py) -- and against the goldens and the oracle
at the model level (tests/test_gpu_parity.py, tests/test_full_size_gpu.py: mfma_arith = 6 / 9 / 0 at the same tolerances).

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


def _run(cfg_kw, B, seed, chain, train_drop=0.0, arith=0):
    from unirec_amd import _lib, ops
    dev = torch.device("cuda:0")
    d, L, I, H, nl = cfg_kw["d"], cfg_kw["L"], cfg_kw["inner"], cfg_kw["heads"], cfg_kw["layers"]
    N = 5000
    prev = _lib.lib.ur_sasrec_set_chain(int(chain) if not isinstance(chain, bool) else (127 if chain else 0))   # bit mask: 1 fwd, 2 bwd, 4 projection, 8 / 16 fwd / bwd of the last-row layer (row-chain kernels), 32 input block, 64 the last-row layer as two launches (lastrow.hip)
    try:
        cfg = ops.sasrec_cfg(B, L, d, H, I, nl, cfg_kw.get("act", "swish"), True, 1e-10, last_only=cfg_kw.get("last_only", 1),
                             skip_padding=cfg_kw.get("skip_padding", 1), p_hidden=train_drop, p_attn=0.0, drop_seed=7, drop_step=3,
                             mfma_arith=arith)
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


def _l2(a, ref):
    return float((a.double() - ref.double()).norm() / max(1e-300, float(ref.double().norm())))


@pytest.mark.parametrize("d,L,heads,B,extra", [(128, 50, 16, 256, {}), (64, 50, 16, 256, {}), (32, 20, 4, 128, {}), (128, 200, 16, 32, {}),
                                               (64, 20, 8, 96, dict(n_layers=3, last_row_only=0)),      # every layer a full-sequence chain
                                               (128, 30, 16, 64, dict(n_layers=4, skip_padding=0, hidden_act="gelu"))])   # 48 copies: the batch's limit
def test_split_row_chains_are_fp32_equivalent(d, L, heads, B, extra):
    """The row-chain kernels in split-bf16 arithmetic (mfma_arith = 6, the default: six piece products per product of two fp32 values on
    the bf16 matrix pipe) against the same kernels on the fp32-input MFMA (mfma_arith = 0), both measured against the ORACLE EVALUATED
    IN fp64: the relative L2 error of the user vectors, of every dense gradient and of the item-table gradient in split arithmetic may
    not exceed 1.5 x the exact arithmetic's (2 x for the d-element vectors).  UR_ERROR_TABLE_OUT=<file>: the table is appended there (profiles/r06_k_chain_split_error.txt)."""
    import os
    from oracle import model_ref
    from unirec_amd.model.sequential.sasrec import SASRec
    from test_gpu_parity import _dense_table_grad
    N, K = 3000, 4
    cfg = dict(model="SASRec", n_users=10, n_items=N, device="cuda:0", loss_type="bpr", embedding_size=d, hidden_size=d, dropout_prob=0.0,
               init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, has_user_bias=False, has_item_bias=False,
               distance_type="dot", tau=1.0, train_file_format="user-item", exp_name="t", n_layers=2, n_heads=heads, inner_size=4 * d,
               hidden_dropout_prob=0.0, attn_dropout_prob=0.0, hidden_act="swish", layer_norm_eps=1e-10, max_seq_len=L,
               use_position_emb=True, seed=2022)
    cfg.update(extra)
    g = torch.Generator().manual_seed(d + L)
    seq = torch.randint(1, N, (B, L), generator=g, dtype=torch.int64).to(torch.int32)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[::4] = L
    seq = torch.where(torch.arange(L)[None, :] >= (L - lens)[:, None], seq, torch.zeros_like(seq)).contiguous()
    item_id = torch.randint(1, N, (B, K + 1), generator=g)
    label = torch.zeros(B, K + 1, dtype=torch.int32)
    label[:, 0] = 1
    got = {}
    sd = None
    for arith in (6, 0):
        torch.manual_seed(1)
        m = SASRec(dict(cfg, mfma_arith=arith))
        if sd is None:
            sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            for k in sd:                       # trained-looking weights: LayerNorm affine and biases off their initial 1 / 0
                if sd[k].dim() == 1:
                    sd[k] += 0.1 * torch.randn(sd[k].shape, generator=g)
        m.load_state_dict(sd)
        m.check_views()
        m.train()
        loss, _, ue, _ = m(item_id=item_id.cuda(), label=label.cuda(), item_seq=seq.cuda(), return_loss_only=False)
        loss.backward()
        torch.cuda.synchronize()
        out = {"user_emb": ue.detach().cpu(), "item_embedding.weight": torch.from_numpy(_dense_table_grad(m, "item_embedding", N, d))}
        for k, p in m.named_parameters():
            if k == "item_embedding.weight" or not p.requires_grad:
                continue
            off = (p.data_ptr() - m.dense_flat.data_ptr()) // 4
            out[k] = m.dense_flat.grad[off:off + p.numel()].view(p.shape).cpu().clone()
        got[arith] = out
    P64 = {k: v.double() for k, v in sd.items()}
    _, _, ue64, G64 = model_ref.grads_of(P64, dict(item_seq=seq, item_id=item_id, label=label, user_id=torch.ones(B, dtype=torch.int64)), cfg)
    ref = dict(G64)
    ref["user_emb"] = ue64
    lines, worst = [], 0.0
    for k in got[6]:
        if k.endswith("key.bias"):             # analytically zero (softmax shift invariance): rounding noise on both sides
            continue
        e6, e0 = _l2(got[6][k], ref[k]), _l2(got[0][k], ref[k])
        lines.append(f"  {k:48s} split {e6:.3e}   exact {e0:.3e}   ratio {e6 / max(e0, 1e-30):.2f}")
        worst = max(worst, e6 / max(e0, 1e-30))
        gate = 1.5 if ref[k].numel() >= 1024 else 2.0   # (a d-element vector: few samples, the L2 statistic itself scatters by ~ 30 %)
        assert e6 <= gate * e0 + 1e-9, (k, e6, e0)
    text = f"d={d} L={L} heads={heads} B={B} inner={4 * d} {cfg['n_layers']} layers {extra}: relative L2 error vs the fp64 oracle; worst ratio {worst:.2f}\n" + "\n".join(lines)
    print(text)
    if os.environ.get("UR_ERROR_TABLE_OUT"):
        with open(os.environ["UR_ERROR_TABLE_OUT"], "a") as f:
            f.write(text + "\n")
