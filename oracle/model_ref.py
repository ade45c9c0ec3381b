"""Torch fp32 CPU restatement of the reference model math (oracle; test infrastructure only).

Every function cites the reference lines it restates (paths relative to /root/reference).
Parameters are passed as a flat ``dict[str, Tensor]`` keyed by the *reference state_dict names*
(SURVEY.md section 8b), so a golden state_dict can be fed in unchanged.

Backward passes come from torch autograd over these explicit forward formulas; that is exactly
how the reference obtains its gradients (it has no hand-written backward).
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS = 1e-8  # unirec/constants/global_variables.py:4

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------- activations
def activation(x: Tensor, name: str) -> Tensor:
    """unirec/model/modules.py:337-345 (ACT2FN table)."""
    if name == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))  # nn.GELU(approximate='none')
    if name == "relu":
        return torch.clamp_min(x, 0.0)
    if name == "swish":
        return x * torch.sigmoid(x)  # nn.SiLU
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise KeyError(name)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """torch.nn.LayerNorm over the last dim: biased variance, eps inside the sqrt, affine."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear: weight layout [out, in]."""
    y = x @ w.t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------- embeddings
def embedding(table: Tensor, idx: Tensor) -> Tensor:
    """nn.Embedding forward = row gather (unirec/model/base/recommender.py:67,137).

    padding_idx=0 only affects the gradient (row 0 receives none): reco_abc.py:170."""
    return table[idx.long()]


# ----------------------------------------------------------------------------- SASRec
def sasrec_attention_mask(item_seq: Tensor, use_pos_emb: bool = True) -> Tensor:
    """unirec/model/sequential/sasrec.py:40-57.  Returns additive mask [B,1,L,L] of {0,-10000}."""
    att = (item_seq > 0).long().unsqueeze(1).unsqueeze(2)  # [B,1,1,L]
    if use_pos_emb:
        L = item_seq.shape[-1]
        sub = torch.triu(torch.ones((1, L, L)), diagonal=1)
        sub = (sub == 0).unsqueeze(1).long()
        att = att * sub
    att = att.to(torch.float32)
    return (1.0 - att) * -10000.0


def multi_head_attention(x: Tensor, mask: Tensor, P: Params, pre: str, n_heads: int, eps: float,
                         collect: Optional[dict] = None, drop_attn: Optional[Tensor] = None, drop_out: Optional[Tensor] = None) -> Tensor:
    """unirec/model/modules.py:284-316.  drop_attn [B,h,L,L] / drop_out [B,L,d]: dropout multipliers (0 or 1/(1-p)) of
    attn_dropout (:307, on the probabilities) and out_dropout (:313, on the dense output before the residual); None = p 0."""
    B, L, d = x.shape
    hd = d // n_heads
    q = linear(x, P[pre + "query.weight"], P[pre + "query.bias"])
    k = linear(x, P[pre + "key.weight"], P[pre + "key.bias"])
    v = linear(x, P[pre + "value.weight"], P[pre + "value.bias"])

    def split(t):  # transpose_for_scores, modules.py:279-282
        return t.view(B, L, n_heads, hd).permute(0, 2, 1, 3)

    s = torch.matmul(split(q), split(k).transpose(-1, -2))
    s = s / math.sqrt(hd)
    s = s + mask
    p = torch.softmax(s, dim=-1)
    if drop_attn is not None:
        p = p * drop_attn
    ctx = torch.matmul(p, split(v)).permute(0, 2, 1, 3).contiguous().view(B, L, d)
    h = linear(ctx, P[pre + "dense.weight"], P[pre + "dense.bias"])
    if drop_out is not None:
        h = h * drop_out
    out = layer_norm(h + x, P[pre + "LayerNorm.weight"], P[pre + "LayerNorm.bias"], eps)
    if collect is not None:
        collect[pre + "ctx"] = ctx
    return out


def feed_forward(x: Tensor, P: Params, pre: str, act: str, eps: float, drop: Optional[Tensor] = None) -> Tensor:
    """unirec/model/modules.py:347-355; drop [B,L,d]: multipliers of the dropout at :352 (after dense_2, before the residual)."""
    h = activation(linear(x, P[pre + "dense_1.weight"], P[pre + "dense_1.bias"]), act)
    h = linear(h, P[pre + "dense_2.weight"], P[pre + "dense_2.bias"])
    if drop is not None:
        h = h * drop
    return layer_norm(h + x, P[pre + "LayerNorm.weight"], P[pre + "LayerNorm.bias"], eps)


def sasrec_user_emb(P: Params, item_seq: Tensor, cfg: dict, collect: Optional[dict] = None, drop: Optional[dict] = None) -> Tensor:
    """unirec/model/sequential/sasrec.py:59-76 -> [B,d].  drop: None (evaluation / p = 0) or the dropout multiplier tensors
    {"embed" [B,L,d] (:69), "attn{i}" [B,h,L,L], "out{i}" [B,L,d], "ffn{i}" [B,L,d]} of a training forward."""
    dm = (lambda k: None) if drop is None else (lambda k: torch.as_tensor(drop[k]) if k in drop else None)
    eps = float(cfg["layer_norm_eps"])
    use_pos = bool(cfg.get("use_position_emb", True))
    x = embedding(P["item_embedding.weight"], item_seq)
    if use_pos:
        L = item_seq.shape[1]
        x = x + P["position_embedding.weight"][:L].unsqueeze(0)
    x = layer_norm(x, P["LayerNorm.weight"], P["LayerNorm.bias"], eps)
    if dm("embed") is not None:
        x = x * dm("embed")
    mask = sasrec_attention_mask(item_seq, use_pos)
    if collect is not None:
        collect["mask"] = mask
        collect["x0"] = x
    for i in range(int(cfg["n_layers"])):
        pre = f"trm_encoder.layer.{i}."
        x = multi_head_attention(x, mask, P, pre + "multi_head_attention.", int(cfg["n_heads"]), eps, collect, dm(f"attn{i}"), dm(f"out{i}"))
        x = feed_forward(x, P, pre + "feed_forward.", cfg["hidden_act"], eps, dm(f"ffn{i}"))
        if collect is not None:
            collect[f"layer{i}"] = x
    return x[:, -1, :]


# ----------------------------------------------------------------------------- GRU
def gru_user_emb(P: Params, item_seq: Tensor, cfg: dict, collect: Optional[dict] = None, drop: Optional[dict] = None) -> Tensor:
    """unirec/model/sequential/gru.py:27-35.  torch.nn.GRU (1 layer, batch_first, h0=0) restated
    with the published gate equations (gate order r,z,n in weight_ih_l0/weight_hh_l0 rows):

        r = sigma(W_ir x + b_ir + W_hr h + b_hr)
        z = sigma(W_iz x + b_iz + W_hz h + b_hz)
        n = tanh (W_in x + b_in + r * (W_hn h + b_hn))
        h' = (1 - z) * n + z * h
    All L steps run, including the left padding (zero vectors)."""
    x = embedding(P["item_embedding.weight"], item_seq)  # [B,L,d]
    if drop is not None:
        x = x * torch.as_tensor(drop["embed"])           # emb_dropout (gru.py:29), multipliers [B,L,d]
    w_ih, w_hh = P["gru_layers.weight_ih_l0"], P["gru_layers.weight_hh_l0"]
    b_ih, b_hh = P["gru_layers.bias_ih_l0"], P["gru_layers.bias_hh_l0"]
    H = w_hh.shape[1]
    B, L, _ = x.shape
    h = torch.zeros(B, H, dtype=x.dtype)
    gi_all = linear(x, w_ih, b_ih)  # [B,L,3H]
    for t in range(L):
        gi = gi_all[:, t]
        gh = linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
    if collect is not None:
        collect["h_last"] = h
    return linear(h, P["dense.weight"], P["dense.bias"])


# ----------------------------------------------------------------------------- MF
def mf_user_emb(P: Params, user_id: Tensor) -> Tensor:
    """unirec/model/base/recommender.py:42-44."""
    return embedding(P["user_embedding.weight"], user_id)


# ----------------------------------------------------------------------------- scorer + loss
def scores_dot(user_emb: Tensor, items_emb: Tensor, P: Params, user_id: Optional[Tensor],
               item_id: Tensor, cfg: dict) -> Tensor:
    """InnerProductScorer (unirec/model/modules.py:49-67) through _predict_layer
    (unirec/model/base/recommender.py:76-96): [B,d],[B,G,d] -> [B,G]; biases; /tau; clamp."""
    if items_emb.dim() == user_emb.dim():
        if items_emb.shape[0] == user_emb.shape[0]:
            s = (user_emb * items_emb).sum(-1)
        else:
            s = user_emb @ items_emb.t()
    else:
        s = torch.matmul(items_emb, user_emb.unsqueeze(-1)).squeeze(-1)
    if cfg.get("has_user_bias", False):
        ub = P["user_bias"][user_id.long()]
        if ub.shape != s.shape:
            ub = ub.unsqueeze(1).expand_as(s)
        s = s + ub
    if cfg.get("has_item_bias", False):
        s = s + P["item_bias"][item_id.long()]
    s = s / float(cfg.get("tau", 1.0))
    clip = float(cfg.get("score_clip_value", -1) or -1)
    if clip > 0:
        s = torch.clamp(s, min=-clip, max=clip)
    return s


def loss_from_scores(scores: Tensor, label: Optional[Tensor], cfg: dict, reduction: bool = True) -> Tensor:
    """AbstractRecommender._cal_loss (unirec/model/base/reco_abc.py:220-272) and
    bpr_loss / ccl_loss (unirec/model/modules.py:15-35).  The 10% label sanity check (:239-246)
    has no numerical effect and is not restated."""
    lt = cfg["loss_type"]
    gs = int(cfg.get("group_size", -1) or -1)
    if gs > 0:
        scores = scores.view(-1, gs)
        if label is not None:
            label = label.view(-1, gs)
    if lt == "bce":
        logits = torch.clamp(torch.sigmoid(scores), min=-EPS, max=1 - EPS)
        lab = label.float()
        return F.binary_cross_entropy(logits, lab, reduction="mean" if reduction else "none").mean(dim=-1)
    if lt == "bpr":
        neg = scores[:, 1:]
        pos = scores[:, 0].unsqueeze(1).expand_as(neg)
        l = -torch.log(EPS + torch.sigmoid(pos - neg))
        return l.mean() if reduction else l.mean(dim=-1)
    if lt == "ccl":
        neg = scores[:, 1:]
        pos = scores[:, 0]
        l = 1 - pos + float(cfg["ccl_w"]) * torch.clamp(neg - float(cfg["ccl_m"]), min=0).mean(dim=-1)
        return l.mean() if reduction else l
    if lt == "softmax":
        l = -F.log_softmax(scores, dim=-1)
        l = l[label > 0]
        return l.mean() if reduction else l
    if lt == "fullsoftmax":
        pos = torch.gather(scores, 1, label.reshape(-1, 1).long()).squeeze(-1)
        l = torch.logsumexp(scores, dim=-1) - pos
        return l.mean() if reduction else l
    raise KeyError(lt)


def model_forward(P: Params, batch: dict, cfg: dict, reduction: bool = True, collect: Optional[dict] = None):
    """BaseRecommender.forward in training mode (unirec/model/base/recommender.py:46-64).
    Returns (loss, scores, user_emb, items_emb)."""
    model = cfg["model"]
    item_id = batch["item_id"]
    label = batch.get("label")
    if cfg["loss_type"] == "fullsoftmax":
        label = item_id
        in_item_id = torch.arange(P["item_embedding.weight"].shape[0])
    else:
        in_item_id = item_id
    items_emb = embedding(P["item_embedding.weight"], in_item_id)
    if model == "SASRec":
        user_emb = sasrec_user_emb(P, batch["item_seq"], cfg, collect, drop=batch.get("drop_masks"))
    elif model == "GRU":
        user_emb = gru_user_emb(P, batch["item_seq"], cfg, collect, drop=batch.get("drop_masks"))
    elif model == "MF":
        user_emb = mf_user_emb(P, batch["user_id"])
    elif model == "AttHist":
        user_emb = atthist_user_emb(P, batch["item_seq"], drop=batch.get("drop_masks"))
    elif model in ("ConvFormer", "FASTConvFormer"):
        user_emb = convformer_user_emb(P, batch["item_seq"], batch.get("item_seq_len"), cfg, fast=model == "FASTConvFormer",
                                       drop=batch.get("drop_masks"))
    elif model in ("AvgHist", "SVDPlusPlus"):
        dst = "item_dst_embedding.weight" if (model == "SVDPlusPlus" or cfg.get("asymmetric", True)) else "item_embedding.weight"
        user_emb = pooled_user_emb(P, batch["item_seq"], batch["item_seq_len"], float(cfg.get("user_sequence_alpha", 0.5)), dst,
                                   batch["user_id"] if model == "SVDPlusPlus" else None)
    else:
        raise KeyError(model)
    scores = scores_dot(user_emb, items_emb, P, batch.get("user_id"), in_item_id, cfg)
    loss = loss_from_scores(scores, label, cfg, reduction)
    return loss, scores, user_emb, items_emb


# ----------------------------------------------------------------------------- step
def grads_of(P: Params, batch: dict, cfg: dict):
    """loss and dense gradients of every parameter, reference semantics:
    dense [N,d] embedding gradient with the padding row forced to zero (nn.Embedding padding_idx=0,
    reco_abc.py:168,170; position_embedding has no padding index, sasrec.py:25)."""
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss, scores, user_emb, _ = model_forward(Q, batch, cfg)
    loss.backward()
    G = {}
    for k, v in Q.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        if k in ("item_embedding.weight", "user_embedding.weight", "item_dst_embedding.weight"):
            g = g.clone()
            g[0].zero_()
        G[k] = g
    return loss.detach(), scores.detach(), user_emb.detach(), G


def clip_grad_norm_(G: Params, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (L2, all params) as called at unirec/facility/trainer.py:347-348:
    coef = max_norm / (total_norm + 1e-6), applied only when < 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in G.values())).float()
    coef = max_norm / (float(total) + 1e-6)
    if coef < 1.0:
        for g in G.values():
            g.mul_(coef)
    return float(total)


def adam_step_(P: Params, G: Params, state: dict, lr: float, wd: float = 0.0,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam (unirec/facility/trainer.py:136,349): dense update of EVERY element,
    weight_decay as L2 added to the gradient; bias-corrected; denom = sqrt(v)/sqrt(bc2) + eps."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k, p in P.items():
        g = G[k]
        if wd != 0.0:
            g = g + wd * p
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def optimizer_step_(P: Params, G: Params, state: dict, algo: str, lr: float, wd: float = 0.0) -> None:
    """The torch.optim rule Trainer._build_optimizer constructs for config['optimizer'] (unirec/facility/trainer.py:134-152; only
    lr and weight_decay are passed, the rest are torch's defaults), dense over EVERY element, restated from the published
    update rules:  adam / adamw (betas 0.9, 0.999, eps 1e-8; L2-in-gradient vs decoupled decay), sgd (momentum 0),
    adagrad (eps 1e-10, lr_decay 0, accumulator 0), rmsprop (alpha 0.99, eps 1e-8, momentum 0, not centered)."""
    if algo == "adam":
        return adam_step_(P, G, state, lr, wd)
    t = state["t"] = state.get("t", 0) + 1
    for k, w in P.items():
        g = G[k]
        st = state.setdefault(k, {"m": torch.zeros_like(w), "v": torch.zeros_like(w)})
        if algo == "adamw":
            w.mul_(1.0 - lr * wd)
            st["m"].mul_(0.9).add_(g, alpha=0.1)
            st["v"].mul_(0.999).addcmul_(g, g, value=0.001)
            bc1, bc2 = 1.0 - 0.9 ** t, 1.0 - 0.999 ** t
            w.addcdiv_(st["m"], (st["v"].sqrt() / math.sqrt(bc2)).add_(1e-8), value=-lr / bc1)
            continue
        if wd != 0:
            g = g + wd * w
        if algo == "sgd":
            w.add_(g, alpha=-lr)
        elif algo == "adagrad":
            st["v"].addcmul_(g, g, value=1.0)
            w.addcdiv_(g, st["v"].sqrt().add_(1e-10), value=-lr)
        elif algo == "rmsprop":
            st["v"].mul_(0.99).addcmul_(g, g, value=0.01)
            w.addcdiv_(g, st["v"].sqrt().add_(1e-8), value=-lr)
        else:
            raise KeyError(algo)


def train_step(P: Params, state: dict, batch: dict, cfg: dict, lr: float = 1e-3, wd: float = 0.0,
               grad_clip: Optional[float] = None, algo: str = "adam") -> float:
    """One iteration of the Trainer.fit loop body (unirec/facility/trainer.py:340-349)."""
    loss, _, _, G = grads_of(P, batch, cfg)
    if grad_clip is not None and grad_clip > 0:
        clip_grad_norm_(G, grad_clip)
    with torch.no_grad():
        optimizer_step_(P, G, state, algo, lr, wd)
    return float(loss)


def pooled_user_emb(P: Params, item_seq: Tensor, item_seq_len: Tensor, alpha: float, dst_key: str,
                    user_id: Optional[Tensor] = None) -> Tensor:
    """AvgHist (unirec/model/sequential/avghist.py:35-42) and, with user_id, SVD++ (svdplusplus.py:32-40):
    [U[user] +] (len + 1)^(-alpha) * sum_l E_dst[item_seq[:, l]]."""
    emb = embedding(P[dst_key], item_seq.long())
    coeff = torch.pow((item_seq_len + 1).float(), -alpha).unsqueeze(1)
    out = coeff * emb.sum(1)
    if user_id is not None:
        out = embedding(P["user_embedding.weight"], user_id.long()) + out
    return out


def atthist_user_emb(P: Params, item_seq: Tensor, drop: Optional[dict] = None) -> Tensor:
    """AttHist (unirec/model/sequential/atthist.py:17-23) = AttentionMergeLayer (unirec/model/modules.py:236-244) on the
    gathered history: z = E[seq] W^T + b; p = softmax_l(z . h) (no mask); sum_l p_l z_l."""
    z = linear(embedding(P["item_embedding.weight"], item_seq.long()), P["attention.dense.weight"], P["attention.dense.bias"])
    p = torch.softmax(torch.matmul(z, P["attention.h"]).squeeze(-1), dim=-1)
    out = torch.matmul(p.unsqueeze(-1).transpose(-1, -2), z).squeeze(1)
    if drop is not None:
        out = out * torch.as_tensor(drop["out"])         # emb_dropout on the pooled vector (modules.py:242), multipliers [B,d]
    return out


def convformer_user_emb(P: Params, item_seq: Tensor, item_seq_len: Optional[Tensor], cfg: dict, fast: bool = False,
                        drop: Optional[dict] = None) -> Tensor:
    """ConvFormer (unirec/model/sequential/convformer.py:52-72, 88-129) and FASTConvFormer (fastconvformer.py:47-62).
    drop: None or the training dropout multipliers {"embed" (:59), "out{i}" (:97 / fastconvformer.py:58), "ffn{i}" (:115)}, each [B,L,d]."""
    dm = (lambda k: None) if drop is None else (lambda k: torch.as_tensor(drop[k]) if k in drop else None)
    L, eps = cfg["max_seq_len"], float(cfg["layer_norm_eps"])
    K = cfg["conv_size"]
    act = "gelu" if fast else cfg.get("hidden_act", "gelu")
    x = embedding(P["item_embedding.weight"], item_seq.long()) + P["position_embedding.weight"][:L].unsqueeze(0)
    x = layer_norm(x, P["LayerNorm.weight"], P["LayerNorm.bias"], eps)
    if dm("embed") is not None:
        x = x * dm("embed")
    for i in range(cfg["n_layers"]):
        pre = f"encoder.{i}."
        if fast:
            w = torch.cat([P[pre + "filterlayer.conv_weight"], torch.zeros(1, L - K, x.shape[-1])], dim=1)
            h = torch.fft.irfft(torch.fft.rfft(x, dim=1, norm="ortho") * torch.fft.rfft(w, dim=1, norm="ortho"), n=L, dim=1, norm="ortho")
        else:
            xt = x.transpose(1, 2)
            pad = K - 1
            mode = cfg.get("padding_mode", "circular")
            if mode == "circular":
                xt = torch.cat((xt[:, :, L - pad:] if pad else xt[:, :, :0], xt), dim=2)
            elif mode == "reflect":
                xt = torch.cat((torch.flip(xt, dims=[2])[:, :, 0:pad], xt), dim=2)
            else:
                xt = torch.cat((torch.zeros(xt.size(0), xt.size(1), pad), xt), dim=2)
            h = torch.nn.functional.conv1d(xt, P[pre + "filterlayer.conv.depthwise_conv.weight"], P[pre + "filterlayer.conv.depthwise_conv.bias"],
                                           groups=x.shape[-1]).transpose(1, 2)
        if dm(f"out{i}") is not None:
            h = h * dm(f"out{i}")
        y1 = layer_norm(h + x, P[pre + "filterlayer.LayerNorm.weight"], P[pre + "filterlayer.LayerNorm.bias"], eps)
        hh = activation(linear(y1, P[pre + "intermediate.dense_1.weight"], P[pre + "intermediate.dense_1.bias"]), act)
        hh = linear(hh, P[pre + "intermediate.dense_2.weight"], P[pre + "intermediate.dense_2.bias"])
        if dm(f"ffn{i}") is not None:
            hh = hh * dm(f"ffn{i}")
        x = layer_norm(hh + y1, P[pre + "intermediate.LayerNorm.weight"], P[pre + "intermediate.LayerNorm.bias"], eps)
    if cfg.get("seq_merge", False):
        decay = torch.logspace(float(cfg.get("seq_decay", -0.3)), 0, steps=L).unsqueeze(0).unsqueeze(-1)
        return (x * decay).sum(1) / (item_seq_len.unsqueeze(-1) + 1).pow(0.5)
    return x[:, -1, :]
