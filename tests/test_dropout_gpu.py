"""Training-time dropout of the SASRec encoder (sasrec.py:69; modules.py:307,313,352) on the HIP path.

The keep masks are a counter-based hash evaluated inside the kernels (csrc/common.h DropSpec), never stored; the oracle
(oracle/dropout_ref.py) restates the hash and replays the multipliers through the reference formulas, whose placement of
the dropouts is pinned by the g17 fixtures (tests/test_oracle_golden.py).  Tolerances: 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _dense_table_grad, _dev

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _cfg(d, H, I, L, p_h, p_a, **kw):
    cfg = dict(n_users=10, n_items=3000, device="cuda:0", loss_type="softmax", embedding_size=d, hidden_size=d, dropout_prob=0.0,
               init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, distance_type="dot", tau=1.0,
               train_file_format="user-item", exp_name="t", n_layers=2, n_heads=H, inner_size=I, hidden_dropout_prob=p_h,
               attn_dropout_prob=p_a, hidden_act="swish", layer_norm_eps=1e-10, max_seq_len=L, use_position_emb=True, model="SASRec",
               seed=2022)
    cfg.update(kw)
    return cfg


def _batch(B, L, G, seed=1, all_pad=False):
    g = torch.Generator().manual_seed(seed)
    seq = torch.randint(1, 3000, (B, L), generator=g, dtype=torch.int32)
    for b in range(B):
        seq[b, : (b * 3) % L] = 0
    if all_pad:
        seq[2] = 0                 # an all-padding sequence (literal -10000 path)
    if L > 8:
        seq[5, L - 4] = 0          # interior zero ('unorder' masking)
    item_id = torch.randint(1, 3000, (B, G), generator=g)
    label = torch.zeros(B, G, dtype=torch.int32)
    label[:, 0] = 1
    return dict(item_seq=seq, item_id=item_id, label=label, user_id=torch.ones(B, dtype=torch.int64))


# (d, heads, inner, L): head dim 8 / 4 / 16 on the MFMA kernels (L <= 64), head dim 8 on the register-broadcast kernels
# (L > 64), head dim 32 on the scalar-load kernels
SHAPES = [(32, 4, 64, 10), (64, 16, 128, 50), (64, 4, 128, 33), (32, 4, 64, 70), (64, 2, 128, 12)]


@pytest.mark.parametrize("all_pad", [False, True])
@pytest.mark.parametrize("p_h,p_a", [(0.3, 0.2), (0.5, 0.0), (0.0, 0.5)])
@pytest.mark.parametrize("last_row_only,skip_padding", [(1, 1), (0, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("shape", SHAPES)
def test_sasrec_dropout_forward_backward_vs_oracle(shape, last_row_only, skip_padding, p_h, p_a, all_pad):
    from oracle import dropout_ref, model_ref
    from unirec_amd.model.sequential.sasrec import SASRec
    d, H, I, L = shape
    if (p_h, p_a) != (0.3, 0.2) and not (last_row_only and skip_padding):
        pytest.skip("single-site variants only on the default layout")
    if all_pad and (p_h, p_a) != (0.3, 0.2):
        pytest.skip("the all-padding sequence only with both dropouts on")
    # A sequence with no item takes the reference's literal path: scores + (-10000.0) in fp32 (ulp 1e-3) before the softmax,
    # so ITS attention weights carry ~1e-3 of rounding noise on both sides (with or without dropout); gradients are then
    # compared at 1e-3 of each tensor's scale instead of 2e-5.
    g_atol = 1e-3 if all_pad else 2e-5
    dev = _dev()
    cfg = _cfg(d, H, I, L, p_h, p_a, last_row_only=last_row_only, skip_padding=skip_padding,
               use_position_emb=not (shape == SHAPES[2]))          # one shape without the causal mask
    torch.manual_seed(d + H + L)
    m = SASRec(cfg)
    B, G = 13, 7
    batch = _batch(B, L, G, all_pad=all_pad)
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    m.train()
    step0 = m._drop_step
    loss, scores, ue, _ = m(item_id=batch["item_id"].to(dev), label=batch["label"].to(dev), item_seq=batch["item_seq"].to(dev),
                            return_loss_only=False)
    assert m._drop_step == step0 + 1
    ob = dict(batch)
    ob["drop_masks"] = dropout_ref.sasrec_masks(B, L, d, H, cfg["n_layers"], p_h, p_a, cfg["seed"], m._drop_step)
    loss_r, scores_r, ue_r, G_r = model_ref.grads_of(P, ob, cfg)
    np.testing.assert_allclose(ue.detach().cpu().numpy(), ue_r.numpy(), rtol=RTOL, atol=1e-4 if all_pad else 1e-5)
    np.testing.assert_allclose(float(loss), float(loss_r), rtol=RTOL)
    loss.backward()
    named = dict(m.named_parameters())
    for k, ref in G_r.items():
        if k == "item_embedding.weight":
            got = _dense_table_grad(m, "item_embedding", 3000, d)
        else:
            p = named[k]
            off = (p.data_ptr() - m.dense_flat.data_ptr()) // 4
            got = m.dense_flat.grad[off:off + p.numel()].view(p.shape).cpu().numpy()
        if k.endswith("key.bias"):
            continue   # analytically zero without dropout; with it, tiny: covered by the scaled comparison of the others
        if k == "position_embedding.weight" and not cfg["use_position_emb"]:
            continue
        scale = max(1e-8, float(np.abs(ref.numpy()).max()))
        np.testing.assert_allclose(got / scale, ref.numpy() / scale, rtol=2e-4, atol=g_atol, err_msg=k)
    m.sparse_grads.clear()


def test_dropout_is_off_in_eval_and_fresh_every_training_step():
    from unirec_amd.model.sequential.sasrec import SASRec
    dev = _dev()
    cfg = _cfg(64, 16, 128, 50, 0.5, 0.5)
    torch.manual_seed(0)
    m = SASRec(cfg)
    cfg0 = dict(cfg, hidden_dropout_prob=0.0, attn_dropout_prob=0.0)
    m0 = SASRec(cfg0)
    m0.load_state_dict(m.state_dict())
    seq = _batch(9, 50, 3)["item_seq"].to(dev)
    m.eval(); m0.eval()
    with torch.no_grad():
        e, e0 = m.forward_user_emb(item_seq=seq), m0.forward_user_emb(item_seq=seq)
    assert torch.equal(e, e0)                                   # evaluation: bit-identical to the dropout-free model
    m.train()
    a = m.forward_user_emb(item_seq=seq).detach().clone()
    b = m.forward_user_emb(item_seq=seq).detach().clone()
    assert not torch.allclose(a, b) and not torch.allclose(a, e)  # a new mask stream on every training forward
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


def test_mean_of_dropped_forward_is_unbiased_at_the_first_site():
    """E[dropout(x)] = x: averaged over many steps the embedded-input dropout leaves LN0's output unchanged.  Checked on the
    C entry point with p_attn = 0 and a model reduced to the input block's effect: ur_sasrec_fwd's workspace holds x0."""
    from unirec_amd import ops
    dev = _dev()
    B, L, d = 4, 16, 32
    g = torch.Generator().manual_seed(3)
    table = torch.randn(100, d, generator=g).to(dev)
    seq = torch.randint(1, 100, (B, L), generator=g, dtype=torch.int32).to(dev)
    base = ops.sasrec_cfg(B, L, d, 4, 64, 1, "gelu", True, 1e-10, last_only=0, skip_padding=0)
    offs, total = ops.sasrec_param_layout(base)
    dense = (torch.randn(total, generator=g) * 0.05).to(dev)
    dense[offs[1]:offs[1] + d] = 1.0
    ws = ops.sasrec_workspace(ops.sasrec_cfg(B, L, d, 4, 64, 1, "gelu", True, 1e-10, last_only=0, skip_padding=0, p_hidden=0.5), dev)
    ops.sasrec_fwd(base, table, dense, seq, ws)
    x0 = ws.view(torch.float32)[: B * L * d].clone()
    acc = torch.zeros_like(x0)
    n = 400
    kept = 0.0
    for step in range(n):
        c = ops.sasrec_cfg(B, L, d, 4, 64, 1, "gelu", True, 1e-10, last_only=0, skip_padding=0, p_hidden=0.5, drop_seed=11, drop_step=step)
        ops.sasrec_fwd(c, table, dense, seq, ws)
        xs = ws.view(torch.float32)[: B * L * d]
        acc += xs
        kept += float((xs != 0).float().mean())
        assert torch.all((xs == 0) | torch.isclose(xs, 2 * x0, rtol=1e-6, atol=0))
    assert abs(kept / n - 0.5) < 0.01
    err = (acc / n - x0).abs().max() / x0.abs().max()
    assert err < 0.25, float(err)      # 400 Bernoulli(1/2) draws per element: sd of the mean = |x| / 20


@pytest.mark.parametrize("model_name", ["ConvFormer", "FASTConvFormer"])
@pytest.mark.parametrize("padding_mode,seq_merge,K", [("circular", False, 10), ("reflect", True, 4), ("constant", False, 7)])
def test_convformer_dropout_forward_backward_vs_oracle(model_name, padding_mode, seq_merge, K):
    """hidden dropout of ConvFormer / FASTConvFormer (convformer.py:59,97,115; fastconvformer.py:58): the embedded input, the
    mixer output and the feed-forward output, each before its residual LayerNorm."""
    from oracle import dropout_ref, model_ref
    from unirec_amd.model.sequential.convformer import ConvFormer
    from unirec_amd.model.sequential.fastconvformer import FASTConvFormer
    dev = _dev()
    d, I, L, B, G, p = 32, 64, 10, 9, 5, 0.4
    cfg = dict(n_users=10, n_items=3000, device="cuda:0", loss_type="softmax", embedding_size=d, hidden_size=d, dropout_prob=0.0,
               init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, distance_type="dot", tau=1.0,
               train_file_format="user-item", exp_name="t", n_layers=2, inner_size=I, hidden_dropout_prob=p, hidden_act="gelu",
               layer_norm_eps=1e-9, max_seq_len=L, model=model_name, seed=77, conv_size=K, padding_mode=padding_mode, seq_merge=seq_merge,
               seq_decay=-0.3, init_ratio=0.05)
    torch.manual_seed(K)
    m = (ConvFormer if model_name == "ConvFormer" else FASTConvFormer)(cfg)
    batch = _batch(B, L, G)
    batch["item_seq_len"] = (batch["item_seq"] > 0).sum(1).to(torch.int64)
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    m.train()
    loss, scores, ue, _ = m(item_id=batch["item_id"].to(dev), label=batch["label"].to(dev), item_seq=batch["item_seq"].to(dev),
                            item_seq_len=batch["item_seq_len"].to(dev), return_loss_only=False)
    assert m._drop_step == 1
    ob = dict(batch)
    ob["drop_masks"] = dropout_ref.convformer_masks(B, L, d, 2, p, cfg["seed"], 1)
    loss_r, scores_r, ue_r, G_r = model_ref.grads_of(P, ob, cfg)
    np.testing.assert_allclose(ue.detach().cpu().numpy(), ue_r.numpy(), rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(float(loss), float(loss_r), rtol=RTOL)
    loss.backward()
    named = dict(m.named_parameters())
    for k, ref in G_r.items():
        if k == "item_embedding.weight":
            got = _dense_table_grad(m, "item_embedding", 3000, d)
        else:
            pp = named[k]
            if not pp.requires_grad:
                continue
            off = (pp.data_ptr() - m.dense_flat.data_ptr()) // 4
            got = m.dense_flat.grad[off:off + pp.numel()].view(pp.shape).cpu().numpy()
        scale = max(1e-8, float(np.abs(ref.numpy()).max()))
        np.testing.assert_allclose(got / scale, ref.numpy() / scale, rtol=2e-4, atol=2e-5, err_msg=k)
    m.sparse_grads.clear()
    m.eval()
    with torch.no_grad():
        e1 = m.forward_user_emb(item_seq=batch["item_seq"].to(dev), item_seq_len=batch["item_seq_len"].to(dev))
        e2 = m.forward_user_emb(item_seq=batch["item_seq"].to(dev), item_seq_len=batch["item_seq_len"].to(dev))
    assert torch.equal(e1, e2)      # evaluation: no dropout


@pytest.mark.parametrize("model_name", ["GRU", "AttHist"])
def test_gru_and_atthist_dropout_vs_oracle(model_name):
    """config dropout_prob: GRU drops the gathered embeddings (gru.py:29), AttHist the pooled output (modules.py:242)."""
    from oracle import dropout_ref, model_ref
    from unirec_amd.model.sequential.atthist import AttHist
    from unirec_amd.model.sequential.gru import GRU
    dev = _dev()
    d, L, B, G, p = 32, 12, 11, 5, 0.35
    cfg = dict(n_users=10, n_items=3000, device="cuda:0", loss_type="softmax", embedding_size=d, hidden_size=24 if model_name == "GRU" else d,
               dropout_prob=p, init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, distance_type="dot", tau=1.0,
               train_file_format="user-item", exp_name="t", max_seq_len=L, model=model_name, seed=5)
    torch.manual_seed(3)
    m = (GRU if model_name == "GRU" else AttHist)(cfg)
    batch = _batch(B, L, G)
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    m.train()
    loss, scores, ue, _ = m(item_id=batch["item_id"].to(dev), label=batch["label"].to(dev), item_seq=batch["item_seq"].to(dev),
                            return_loss_only=False)
    assert m._drop_step == 1
    ob = dict(batch)
    ob["drop_masks"] = dropout_ref.gru_masks(B, L, d, p, 5, 1) if model_name == "GRU" else dropout_ref.atthist_masks(B, d, p, 5, 1)
    loss_r, scores_r, ue_r, G_r = model_ref.grads_of(P, ob, cfg)
    np.testing.assert_allclose(ue.detach().cpu().numpy(), ue_r.numpy(), rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(float(loss), float(loss_r), rtol=RTOL)
    loss.backward()
    named = dict(m.named_parameters())
    for k, ref in G_r.items():
        if k == "item_embedding.weight":
            got = _dense_table_grad(m, "item_embedding", 3000, d)
        else:
            pp = named[k]
            off = (pp.data_ptr() - m.dense_flat.data_ptr()) // 4
            got = m.dense_flat.grad[off:off + pp.numel()].view(pp.shape).cpu().numpy()
        scale = max(1e-8, float(np.abs(ref.numpy()).max()))
        np.testing.assert_allclose(got / scale, ref.numpy() / scale, rtol=2e-4, atol=2e-5, err_msg=k)
    m.sparse_grads.clear()
    m.eval()
    with torch.no_grad():
        e1 = m.forward_user_emb(item_seq=batch["item_seq"].to(dev))
    ob.pop("drop_masks")
    np.testing.assert_allclose(e1.cpu().numpy(), model_ref.grads_of(P, ob, cfg)[2].numpy(), rtol=RTOL, atol=1e-5)   # evaluation: none
