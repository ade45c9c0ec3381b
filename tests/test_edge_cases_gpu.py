"""Smallest and oddest shapes through the whole HIP path (model forward/backward + one optimizer step) against the oracle:
B = 1, L = 1, one candidate, d = 4, a two-row catalogue, every sequence empty, heads of 2 / 32 / 64 dims, 8 layers."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _dense_table_grad, _dev

pytestmark = pytest.mark.gpu


def _run(model_name, cfg_kw, B, L, G, N, seq_fn=None, loss="softmax", rtol=2e-4):
    from oracle import model_ref
    from unirec_amd.model.cf.mf import MF
    from unirec_amd.model.sequential.gru import GRU
    from unirec_amd.model.sequential.sasrec import SASRec
    dev = _dev()
    d = cfg_kw.get("embedding_size", 32)
    cfg = dict(model=model_name, n_users=7, n_items=N, device="cuda:0", loss_type=loss, embedding_size=d, hidden_size=d, dropout_prob=0.0,
               init_method="normal", init_mean=0.0, init_std=0.1, has_user_emb=model_name == "MF", has_user_bias=False, has_item_bias=False,
               distance_type="dot", tau=1.0, train_file_format="user-item", exp_name="t", n_layers=2, n_heads=2, inner_size=16,
               hidden_dropout_prob=0.0, attn_dropout_prob=0.0, hidden_act="gelu", layer_norm_eps=1e-10, max_seq_len=L, use_position_emb=True)
    cfg.update(cfg_kw)
    torch.manual_seed(B * 100 + L)
    m = {"SASRec": SASRec, "GRU": GRU, "MF": MF}[model_name](cfg)
    g = torch.Generator().manual_seed(5)
    seq = torch.randint(1, N, (B, L), generator=g, dtype=torch.int32)
    if seq_fn is not None:
        seq = seq_fn(seq)
    item_id = torch.randint(1, N, (B, G), generator=g)
    label = torch.zeros(B, G, dtype=torch.int32)
    label[:, 0] = 1
    uid = torch.randint(1, 7, (B,), generator=g)
    batch = dict(item_seq=seq, item_id=item_id, label=label, user_id=uid)
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    loss_r, scores_r, ue_r, G_r = model_ref.grads_of(P, batch, cfg)
    m.train()
    kw = dict(item_id=item_id.to(dev), label=label.to(dev), user_id=uid.to(dev))
    if model_name != "MF":
        kw["item_seq"] = seq.to(dev)
    out_loss, scores, ue, _ = m(return_loss_only=False, **kw)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), scores_r.numpy(), rtol=rtol, atol=1e-5)
    np.testing.assert_allclose(float(out_loss), float(loss_r), rtol=rtol)
    out_loss.backward()
    named = dict(m.named_parameters())
    for k, ref in G_r.items():
        if k in ("item_embedding.weight", "user_embedding.weight"):
            got = _dense_table_grad(m, k.split(".")[0], ref.shape[0], ref.shape[1])
        elif k in ("user_bias", "item_bias"):
            got = named[k].grad.cpu().numpy()
        else:
            p = named[k]
            off = (p.data_ptr() - m.dense_flat.data_ptr()) // 4
            got = m.dense_flat.grad[off:off + p.numel()].view(p.shape).cpu().numpy()
        if k.endswith("key.bias"):
            continue
        # (L = 1: the softmax over one key is constant, so the q / k gradients are analytically zero -- exactly 0 in the reference,
        #  rounding noise of ~1e-9 here; the user-bias gradient of BPR cancels the same way: the scale floor keeps noise from
        #  being compared with noise relatively)
        scale = max(1e-3, float(np.abs(ref.numpy()).max()))
        np.testing.assert_allclose(got / scale, ref.numpy() / scale, rtol=max(1e-3, rtol), atol=max(5e-5, rtol / 10), err_msg=k)
    m.sparse_grads.clear()


@pytest.mark.parametrize("B,L,G,N", [(1, 1, 2, 2), (1, 7, 1, 9), (3, 1, 4, 50), (2, 64, 3, 40), (1, 65, 2, 40), (513, 2, 2, 3)])
def test_sasrec_tiny_and_boundary_shapes(B, L, G, N):
    _run("SASRec", {}, B, L, G, N, loss="bce" if G == 1 else "softmax")


@pytest.mark.parametrize("d,heads", [(4, 1), (4, 2), (8, 4), (64, 2), (64, 1), (128, 2), (256, 16)])
def test_sasrec_head_dims_from_2_to_64(d, heads):
    _run("SASRec", dict(embedding_size=d, hidden_size=d, n_heads=heads, inner_size=max(16, d)), 5, 9, 3, 60)


def test_sasrec_eight_layers_and_every_sequence_empty():
    _run("SASRec", dict(n_layers=8), 4, 6, 3, 30)
    _run("SASRec", {}, 4, 6, 3, 30, seq_fn=lambda s: torch.zeros_like(s), rtol=5e-3)      # literal -10000 path on every row
    _run("SASRec", dict(use_position_emb=False), 4, 6, 3, 30, seq_fn=lambda s: torch.cat([torch.zeros_like(s[:, :4]), s[:, 4:]], 1))


# 24: gemm_nt + cell kernels per step; 384 / 192 / 768 / 512: the fused step kernels (H % 64 == 0 beyond the persistent kernels' range) with
# 16, 32 and 48 units per workgroup (the wider tiles need >= 200 workgroups: B >= 416), B not a multiple of the 32-row tile
@pytest.mark.parametrize("B,L,H", [(1, 1, 16), (2, 3, 32), (17, 5, 64), (3, 2, 128), (4, 4, 24), (5, 4, 384), (37, 3, 192), (420, 3, 768),
                                   (421, 2, 512), (33, 2, 768)])
def test_gru_small_shapes_both_recurrence_paths(B, L, H):
    _run("GRU", dict(hidden_size=H, embedding_size=16), B, L, 3, 40)


def test_mf_single_row_and_two_item_catalogue():
    _run("MF", {}, 1, 1, 2, 2, loss="bpr")
    _run("MF", dict(has_user_bias=True, has_item_bias=True, tau=0.5), 6, 1, 5, 11, loss="bpr")


def test_torch_ops_dispatch_to_the_same_kernels():
    """torch.ops.unirec_amd.* (unirec_amd/torch_ops.py) are the ctypes calls behind a dispatcher schema: same results, bit for bit,
    and they survive torch.compile's tracing (fullgraph: no graph break at the custom ops)."""
    import unirec_amd.torch_ops  # noqa: F401
    from unirec_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    N, d, B, L, G, I = 500, 64, 9, 12, 5, 128
    table = torch.randn(N, d, device=dev, generator=g) * 0.1
    table[0] = 0
    ids = torch.randint(1, N, (B, G), device=dev, generator=g)
    assert torch.equal(torch.ops.unirec_amd.embedding_gather(table, ids), table[ids])
    cfg = ops.sasrec_cfg(B, L, d, 4, I, 2, "swish", True, 1e-10)
    offs, total = ops.sasrec_param_layout(cfg)
    dense = torch.randn(total, device=dev, generator=g) * 0.05
    seq = torch.randint(1, N, (B, L), device=dev, generator=g, dtype=torch.int32)
    seq[:, :3] = 0
    ws = torch.ops.unirec_amd.sasrec_workspace(seq, d, 4, I, 2, 0.0)
    ue = torch.ops.unirec_amd.sasrec_fwd(table, dense, seq, ws, 4, I, 2, "swish", True, 1e-10)
    want = ops.sasrec_fwd(cfg, table, dense, seq, ops.sasrec_workspace(cfg, dev))
    assert torch.equal(ue, want)
    sc, _, lo = torch.ops.unirec_amd.gather_dot_loss_fwd(ue, table, ids, None, None, None, None, "bpr")
    sc2, _, lo2 = ops.gather_dot_loss_fwd(ops.loss_cfg(B, G, d, "bpr"), ue, table, ids)
    assert torch.equal(sc, sc2) and torch.equal(lo[:1], lo2[:1])

    @torch.compile(fullgraph=True, backend="eager")
    def f(t, i):
        return torch.ops.unirec_amd.embedding_gather(t, i) * 2.0
    assert torch.equal(f(table, ids), table[ids] * 2.0)


def test_parameter_init_statistics_follow_the_reference_rules():
    """SURVEY.md 8 a15 (unirec/model/base/reco_abc.py:19-58, 210-218): normal(init_mean, init_std) for embeddings and linear weights,
    zeros for biases, ones / zeros for LayerNorm, the padding row of every table zero; nn.GRU keeps torch's U(-1/sqrt(H), 1/sqrt(H))."""
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.general import get_class_instance, init_seed
    base = dict(n_users=300, n_items=20000, device="cuda:0", embedding_size=64, hidden_size=64, inner_size=128, n_heads=4, n_layers=2,
                max_seq_len=20, init_method="normal", init_mean=0.0, init_std=0.02, hidden_dropout_prob=0.0, attn_dropout_prob=0.0, seed=1)
    init_seed(1)
    m = get_class_instance("SASRec", "unirec_amd/model")(parse_arguments(dict(base, model="SASRec")))
    sd = m.state_dict()
    tab = sd["item_embedding.weight"]
    assert not tab[0].any() and abs(float(tab[1:].mean())) < 5e-4 and abs(float(tab[1:].std()) - 0.02) < 5e-4
    for k, v in sd.items():
        if k.endswith("LayerNorm.weight"):
            assert torch.equal(v, torch.ones_like(v)), k
        elif k.endswith(".bias"):
            assert not v.any(), k
        elif k.endswith(".weight") and v.dim() == 2 and "embedding" not in k:
            assert abs(float(v.mean())) < 3e-3 and abs(float(v.std()) - 0.02) < 2e-3, (k, float(v.std()))
    assert abs(float(sd["position_embedding.weight"].std()) - 0.02) < 3e-3
    init_seed(1)
    g = get_class_instance("GRU", "unirec_amd/model")(parse_arguments(dict(base, model="GRU", hidden_size=64)))
    bound = 1.0 / 8.0
    for k, v in g.state_dict().items():
        if k.startswith("gru_layers."):
            assert float(v.abs().max()) <= bound and abs(float(v.std()) - bound / 3 ** 0.5) < 0.01, k     # U(-b, b): std = b / sqrt(3)
    x = get_class_instance("SASRec", "unirec_amd/model")(parse_arguments(dict(base, model="SASRec", init_method="xavier_normal")))
    w = x.state_dict()["trm_encoder.layer.0.feed_forward.dense_1.weight"]      # [128, 64]: std = sqrt(2 / (128 + 64))
    assert abs(float(w.std()) - (2.0 / 192) ** 0.5) < 5e-3


@pytest.mark.parametrize("loss,G,biases", [("bpr", 5, False), ("bpr", 2, True), ("bce", 7, True), ("ccl", 5, False), ("bpr", 60, False)])
def test_fused_loss_step_equals_forward_plus_backward(loss, G, biases):
    """ur_gather_dot_loss_fwd_bwd (the training step's scorer + loss + gradient as one launch, batch loss finished by the last
    workgroup) against ur_gather_dot_loss_fwd followed by ur_gather_dot_loss_bwd: scores, loss and coefficients to fp32 rounding
    (same arithmetic per element), d_user to re-association of the sum over the candidates; twice in a row (the completion counter resets itself); softmax is not fusable and must take the two-launch path."""
    from unirec_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, d, N = 300, 64, 5000
    g = torch.Generator(device=dev).manual_seed(G)
    ue = torch.randn(B, d, device=dev, generator=g) * 0.3
    table = torch.randn(N, d, device=dev, generator=g) * 0.3
    ids = torch.randint(1, N, (B, G), device=dev, generator=g)
    lab = torch.zeros(B, G, dtype=torch.int32, device=dev)
    lab[:, 0] = 1
    ub = torch.randn(50, device=dev, generator=g) * 0.1 if biases else None
    ib = torch.randn(N, device=dev, generator=g) * 0.1 if biases else None
    uid = torch.randint(1, 50, (B,), device=dev, generator=g) if biases else None
    cfg = ops.loss_cfg(B, G, d, loss, 0.7, 3.0 if loss == "bce" else -1.0, 0.5, 0.2)
    assert _lib.lib.ur_gather_dot_loss_fused_supported(cfg)
    s0, _, lo0 = ops.gather_dot_loss_fwd(cfg, ue, table, ids, lab, ub, ib, uid)
    c0, du0, dub0 = ops.gather_dot_loss_bwd(cfg, ue, table, ids, lab, s0, lo0, None, want_user_bias=biases)
    for _ in range(2):
        s1, lo1, c1, du1, dub1 = ops.gather_dot_loss_fwd_bwd(cfg, ue, table, ids, lab, ub, ib, uid, want_user_bias=biases)
        torch.cuda.synchronize()
        assert torch.allclose(s0, s1, rtol=2e-6, atol=1e-6)      # (the compiler contracts the dot products of the two kernels differently)
        assert torch.allclose(lo0[:3], lo1[:3], rtol=2e-6, atol=0), (lo0, lo1)
        assert torch.allclose(c0, c1, rtol=1e-5, atol=1e-9)
        assert torch.allclose(du0, du1, rtol=1e-5, atol=1e-8)
        if biases:
            assert torch.allclose(dub0, dub1, rtol=1e-5, atol=1e-9)
    cfg_s = ops.loss_cfg(B, G, d, "softmax", 0.7, -1.0)
    assert not _lib.lib.ur_gather_dot_loss_fused_supported(cfg_s)
    s2, lo2, c2, du2, _ = ops.gather_dot_loss_fwd_bwd(cfg_s, ue, table, ids, lab, None, None, None)
    s3, _, lo3 = ops.gather_dot_loss_fwd(cfg_s, ue, table, ids, lab)
    assert torch.equal(s2, s3) and torch.equal(lo2[:3], lo3[:3])
    big = ops.loss_cfg(8, 1001, 128, "bpr", 1.0, -1.0)          # 1001 candidates of 512 bytes do not fit the LDS budget
    assert not _lib.lib.ur_gather_dot_loss_fused_supported(big)
    # a NaN score reaches the update guard through the fused path as well
    ue_bad = ue.clone()
    ue_bad[3, 0] = float("nan")
    _, lo4, _, _, _ = ops.gather_dot_loss_fwd_bwd(cfg, ue_bad, table, ids, lab, ub, ib, uid, want_user_bias=biases)
    if loss != "bce":                                           # (bce runs with score_clip here: fmin / fmax clip the NaN away, in both paths)
        assert float(lo4[2]) == -1.0
    _, lo5, _, _, _ = ops.gather_dot_loss_fwd_bwd(cfg, ue, table, ids, lab, ub, ib, uid, want_user_bias=biases)
    assert float(lo5[2]) == 1.0 and torch.allclose(lo5[:3], lo0[:3], rtol=2e-6, atol=0)


def test_roctx_ranges_behind_their_switch():
    """UR_ROCTX=1: every C-ABI entry that enqueues device work pushes / pops a roctx range named after itself (csrc/common.h: TraceScope;
    SURVEY.md 5 "roctx ranges per op").  The library resolves roctx at run time: with the switch on the ops still give the right answer and
    ur_trace_ranges_pushed counts their ranges; with it off (the default) nothing is pushed."""
    import os
    import subprocess
    import sys
    code = ("import torch\n"
            "from unirec_amd import ops\n"
            "t = torch.arange(200 * 32, device='cuda:0', dtype=torch.float32).reshape(200, 32)\n"
            "idx = torch.tensor([3, 0, 199], device='cuda:0')\n"
            "assert torch.equal(ops.embedding_gather(t, idx).cpu(), t[idx].cpu())\n"
            "from unirec_amd._lib import lib\n"
            "print('RANGES_PUSHED' if lib.ur_trace_ranges_pushed() > 0 else 'RANGES_NONE')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env, want in (({"UR_ROCTX": "1"}, "RANGES_PUSHED"), ({}, "RANGES_NONE")):
        e = {k: v for k, v in os.environ.items() if k != "UR_ROCTX"}
        r = subprocess.run([sys.executable, "-c", code], env=dict(e, **env), cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and want in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


def test_backward_on_a_workspace_of_another_forward_is_refused():
    """ur_sasrec_bwd reads the K-major weight copies, row maps and activations the LAST ur_sasrec_fwd on its workspace left there:
    a backward pass handed a workspace whose last forward saw other ids or other weights is an error, not a silent stale read."""
    from unirec_amd import _lib, ops
    dev = torch.device("cuda:0")
    cfg = ops.sasrec_cfg(8, 10, 32, 4, 64, 2, "gelu", True, 1e-10)
    _, total = ops.sasrec_param_layout(cfg)
    table = torch.randn(50, 32, device=dev)
    dense, dense2 = torch.randn(total, device=dev) * 0.1, torch.randn(total, device=dev) * 0.1
    seq = torch.randint(1, 50, (8, 10), dtype=torch.int32, device=dev)
    seq2 = torch.randint(1, 50, (8, 10), dtype=torch.int32, device=dev)
    ws = ops.sasrec_workspace(cfg, dev)
    ue = ops.sasrec_fwd(cfg, table, dense, seq, ws)
    ops.sasrec_bwd(cfg, table, dense, seq, torch.ones_like(ue), ws)            # the pair as intended
    ops.sasrec_fwd(cfg, table, dense, seq2, ws)                                 # another forward on the same workspace ...
    with pytest.raises(_lib.UnirecAmdError, match="last ur_sasrec_fwd on this workspace"):
        ops.sasrec_bwd(cfg, table, dense, seq, torch.ones_like(ue), ws)        # ... and a backward for the first one
    ops.sasrec_fwd(cfg, table, dense2, seq, ws)
    with pytest.raises(_lib.UnirecAmdError, match="last ur_sasrec_fwd on this workspace"):
        ops.sasrec_bwd(cfg, table, dense, seq, torch.ones_like(ue), ws)
    # round 6: the forward pass pre-splits the weights for both passes' row chains when its mfma_arith names a split form -- a backward
    # pass in the other arithmetic would stream copies that were never made
    cfg6 = ops.sasrec_cfg(8, 10, 32, 4, 64, 2, "gelu", True, 1e-10, mfma_arith=6)
    cfg0 = ops.sasrec_cfg(8, 10, 32, 4, 64, 2, "gelu", True, 1e-10, mfma_arith=0)
    for a, b in ((cfg0, cfg6), (cfg6, cfg0)):
        ops.sasrec_fwd(a, table, dense, seq, ws)
        with pytest.raises(_lib.UnirecAmdError, match="mfma_arith"):
            ops.sasrec_bwd(b, table, dense, seq, torch.ones_like(ue), ws)
        ops.sasrec_fwd(b, table, dense, seq, ws)
        ops.sasrec_bwd(b, table, dense, seq, torch.ones_like(ue), ws)
    torch.cuda.synchronize()
