#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (microsoft/UniRec) as a CPU oracle.

Runs only in the build container, where /root/reference exists:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/capture_goldens.py

Nothing from the reference is copied: the fixtures are data (seeded inputs, the state_dict the
reference initialised, and the outputs / gradients / updated parameters it computed).  The GPU box
never sees /root/reference; tests there read only the committed .npz files.

Fixture list follows SURVEY.md section 8c (G1..G10).
"""
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.environ.get("UR_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _shims():
    """stub optional imports that never touch hot-path arithmetic."""
    import accelerate  # noqa: F401  (must be imported before stubbing wandb)
    for name in ("feather", "wandb", "cvxpy"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sp = types.ModuleType("setproctitle")
    sp.setproctitle = lambda *a, **k: None
    sys.modules.setdefault("setproctitle", sp)
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # noqa: D401
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.modules.setdefault("torch.utils.tensorboard.writer", tb)
    np.Inf = np.inf


def base_cfg(**kw):
    cfg = dict(n_users=40, n_items=200, device="cpu", loss_type="bpr", embedding_size=32, hidden_size=32,
               dropout_prob=0.0, init_method="normal", init_mean=0.0, init_std=0.02, has_user_emb=False,
               has_user_bias=False, has_item_bias=False, distance_type="dot", tau=1.0,
               train_file_format="user-item", exp_name="golden", n_layers=2, n_heads=2, inner_size=64,
               hidden_dropout_prob=0.0, attn_dropout_prob=0.0, hidden_act="swish", layer_norm_eps=1e-10,
               max_seq_len=10, use_position_emb=True)
    cfg.update(kw)
    return cfg


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def make_batch(rng, B, L, G, n_items, n_users, pad_counts=None):
    item_seq = rng.integers(1, n_items, size=(B, L)).astype(np.int32)
    if pad_counts is None:
        pad_counts = rng.integers(0, L, size=B)
    for b, p in enumerate(pad_counts):
        item_seq[b, :p] = 0
    item_id = rng.integers(1, n_items, size=(B, G)).astype(np.int64)
    # force some duplicates between rows and between seq and candidates
    item_id[1, 1] = item_id[0, 0]
    item_seq[2, -1] = int(item_id[0, 0])
    label = np.zeros((B, G), dtype=np.int32)
    label[:, 0] = 1
    user_id = rng.integers(1, n_users, size=(B,)).astype(np.int64)
    seq_len = (item_seq > 0).sum(1).astype(np.int64)
    return dict(user_id=user_id, item_id=item_id, label=label, item_seq=item_seq, item_seq_len=seq_len)


def tt(batch):
    return {k: torch.from_numpy(v) for k, v in batch.items()}


def run_model(model, batch, want_layers=False):
    """loss, scores, user_emb + dense grads from the reference model in train mode."""
    model.train()
    model.zero_grad()
    tb = tt(batch)
    loss, scores, user_emb, items_emb = model(user_id=tb["user_id"], item_id=tb["item_id"], label=tb["label"],
                                              item_seq=tb["item_seq"], item_seq_len=tb["item_seq_len"],
                                              return_loss_only=False)
    loss.backward()
    out = {"out.loss": loss.detach().numpy().copy(), "out.scores": scores.detach().numpy().copy(),
           "out.user_emb": user_emb.detach().numpy().copy()}
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out["grad." + k] = g.detach().numpy().copy()
    return out


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]   # e.g. `capture_goldens.py g11` rewrites only g11_* files


def save(name, **arrs):
    if ONLY and not any(name.startswith(p) for p in ONLY):
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.1f} KiB")


def pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


def main():
    sys.path.insert(0, REF)
    _shims()
    from unirec.data.transform.addnegsamples import AddNegSamples
    from unirec.data.transform.adduserhistory import AddUserHistory
    from unirec.model.cf.mf import MF
    from unirec.model.sequential.gru import GRU
    from unirec.model.sequential.sasrec import SASRec

    # ---------------------------------------------------------------- G1 sampler known answers
    g1 = {}
    random.seed(2022)
    t = AddNegSamples(5, 60000, 4, user2history=np.array([None, np.array([5, 6, 7, 8]), np.array([9, 10])], dtype=object))
    rows = [t(np.array([1, 7], dtype=object))[1] for _ in range(3)]
    g1["kat_seed2022_n60000_k4"] = np.stack(rows)
    # small catalogue: rejections + exhaustion (every id is in the history -> id 0 after 100 tries)
    random.seed(7)
    hist = np.empty(3, dtype=object)
    hist[0] = None
    hist[1] = np.arange(1, 12)
    hist[2] = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])  # 12 free ids out of N=24
    t2 = AddNegSamples(3, 24, 6, user2history=hist)
    g1["small_rows_user"] = np.array([1, 2, 1, 2, 1], dtype=np.int64)
    g1["small_rows_pos"] = np.array([3, 5, 7, 9, 11], dtype=np.int64)
    g1["small_out"] = np.stack([t2(np.array([u, p], dtype=object))[1] for u, p in zip(g1["small_rows_user"], g1["small_rows_pos"])])
    random.seed(11)
    full = np.empty(2, dtype=object)
    full[0] = None
    full[1] = np.arange(1, 8)
    t3 = AddNegSamples(2, 8, 3, user2history=full)
    g1["exhaust_out"] = t3(np.array([1, 3], dtype=object))[1]
    g1["after_exhaust_getrandbits32"] = np.array([random.getrandbits(32)], dtype=np.int64)  # stream position check
    # popularity (alias) sampler
    random.seed(5)
    pop = np.arange(30, dtype=np.float64)
    t4 = AddNegSamples(3, 30, 8, item_popularity=pop, neg_by_pop_alpha=0.5)
    g1["pop"] = pop
    g1["pop_out"] = np.stack([t4(np.array([1, 4], dtype=object))[1] for _ in range(4)])
    save("g1_sampler", **g1)

    # ---------------------------------------------------------------- G2 history / padding
    g2 = {}
    h = np.empty(4, dtype=object)
    h[0] = None
    h[1] = np.array([5, 6, 7, 8], dtype=np.int32)
    h[2] = np.array([9, 10, 9, 11, 9, 12], dtype=np.int32)
    h[3] = np.arange(1, 30, dtype=np.int32)
    g2["h1"], g2["h2"], g2["h3"] = h[1], h[2], h[3]
    for mode in ("autoregressive", "unorder", "autoagressive"):
        for seq_last in (0, 1):
            random.seed(3)
            tr = AddUserHistory(h, mode, seq_last=seq_last)
            outs = []
            for (u, it) in [(1, 7), (2, 9), (2, 9), (2, 9), (3, 15), (7, 3), (1, np.array([7, 99, 6]))]:
                hist_o, ln, _ = tr((u, it))
                outs.append(np.concatenate([[ln], np.asarray(hist_o, dtype=np.int64)]))
            g2[f"{mode}_sl{seq_last}"] = np.array(outs, dtype=object)
    # object arrays are not portable; flatten to (lens, concat)
    flat = {}
    for k, v in list(g2.items()):
        if v.dtype == object:
            flat[k + ".lens"] = np.array([len(x) for x in v], dtype=np.int64)
            flat[k + ".cat"] = np.concatenate([np.asarray(x, dtype=np.int64) for x in v])
        else:
            flat[k] = v
    # padding rule (SeqRecDataset._padding) evaluated through a bare instance
    from unirec.data.dataset.seqrecdataset import SeqRecDataset
    ds = SeqRecDataset.__new__(SeqRecDataset)
    ds.config = {"max_seq_len": 6}
    for nm, x in (("short", [4, 5]), ("exact", [1, 2, 3, 4, 5, 6]), ("long", list(range(1, 10))), ("one", [0])):
        flat["pad_" + nm] = ds._padding(np.array(x, dtype=np.int32))
    save("g2_history", **flat)

    # ---------------------------------------------------------------- G5/G6 SASRec fwd + bwd
    rng = np.random.default_rng(2022)
    for tag, kw in {
        "sasrec_h2_swish_bpr": dict(n_heads=2, hidden_act="swish", loss_type="bpr"),
        "sasrec_h16_gelu_softmax": dict(n_heads=16, hidden_act="gelu", loss_type="softmax", tau=0.7),
        "sasrec_h4_relu_bpr_nopos": dict(n_heads=4, hidden_act="relu", loss_type="bpr", use_position_emb=False),
        "sasrec_h2_tanh_bce_bias": dict(n_heads=2, hidden_act="tanh", loss_type="bce", has_item_bias=True,
                                        has_user_bias=True),
        "sasrec_h2_sigmoid_ccl": dict(n_heads=2, hidden_act="sigmoid", loss_type="ccl", ccl_w=150, ccl_m=0.4),
        "sasrec_d64_1layer_fullsoftmax": dict(n_heads=4, embedding_size=64, hidden_size=64, n_layers=1,
                                              loss_type="fullsoftmax", inner_size=96),
    }.items():
        cfg = base_cfg(model="SASRec", **kw)
        torch.manual_seed(1234)
        m = SASRec(cfg)
        # move LayerNorm / bias params off their trivial init so their gradients are exercised
        with torch.no_grad():
            for k, p in m.named_parameters():
                if "LayerNorm" in k or k.endswith(".bias"):
                    p.add_(0.05 * torch.randn_like(p))
        B, L, G = 6, cfg["max_seq_len"], 5
        batch = make_batch(rng, B, L, G, cfg["n_items"], cfg["n_users"], pad_counts=[0, 3, 9, 10, 5, 1])
        if cfg["loss_type"] == "fullsoftmax":  # one positive id per row, no sampled negatives (recommender.py:47-50)
            batch["item_id"] = batch["item_id"][:, 0].copy()
            batch["label"] = batch["label"][:, 0].copy()
        out = run_model(m, batch)
        # intermediate activations for debugging parity (mask, x0, per-layer outputs)
        m.eval()
        with torch.no_grad():
            seq = torch.from_numpy(batch["item_seq"])
            out["mid.mask"] = m._get_attention_mask(seq).numpy()
        save("g5_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}),
             **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G7 GRU
    for tag, kw in {"gru_h32_bpr": dict(hidden_size=32, loss_type="bpr"),
                    "gru_h48_softmax": dict(hidden_size=48, loss_type="softmax")}.items():
        cfg = base_cfg(model="GRU", **kw)
        torch.manual_seed(99)
        m = GRU(cfg)
        B, L, G = 5, cfg["max_seq_len"], 4
        batch = make_batch(rng, B, L, G, cfg["n_items"], cfg["n_users"], pad_counts=[0, 4, 9, 10, 2])
        out = run_model(m, batch)
        save("g7_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}),
             **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G8 MF (+ scorer/loss with biases)
    cfg = base_cfg(model="MF", has_user_emb=True, has_user_bias=True, has_item_bias=True, loss_type="bpr", tau=0.5,
                   embedding_size=16, hidden_size=16)
    torch.manual_seed(5)
    m = MF(cfg)
    batch = make_batch(rng, 7, 4, 6, cfg["n_items"], cfg["n_users"])
    out = run_model(m, batch)
    save("g8_mf_bpr_bias", **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}),
         **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G8b / G5b: `user-item-label` rows (group_size > 0)
    # ONE (user, item, label) triple per row; _cal_loss views the [rows] scores as [-1, group_size] (reco_abc.py:233-236).  Rows of a
    # group share the user and the history, as the rank data sets deliver them -- except the last group, whose rows differ.
    def group_rows_batch(rng_, n_groups, gs, L, n_items, n_users, two_positives):
        b = make_batch(rng_, n_groups * gs, L, 2, n_items, n_users)
        b["item_id"] = b["item_id"][:, 0].copy()
        for gi in range(n_groups - 1):
            for k in ("user_id", "item_seq", "item_seq_len"):
                b[k][gi * gs:(gi + 1) * gs] = b[k][gi * gs]
        lab = np.zeros((n_groups, gs), dtype=np.int32)
        lab[:, 0] = 1
        if two_positives:
            lab[1, 3] = 1
        b["label"] = lab.reshape(-1)
        return b

    for tag, cls, kw, two in (("g8_mf_grouprows_softmax", MF, dict(model="MF", has_user_emb=True, loss_type="softmax", tau=0.7), True),
                              ("g8_mf_grouprows_bpr_bias", MF, dict(model="MF", has_user_emb=True, has_user_bias=True, has_item_bias=True,
                                                                    loss_type="bpr"), False),
                              ("g5_sasrec_grouprows_bce", SASRec, dict(model="SASRec", loss_type="bce"), True),
                              ("g5_sasrec_grouprows_ccl", SASRec, dict(model="SASRec", loss_type="ccl", ccl_w=150, ccl_m=0.4), False)):
        cfg = base_cfg(train_file_format="user-item-label", group_size=5, embedding_size=16, hidden_size=16, **kw)
        torch.manual_seed(31)
        m = cls(cfg)
        batch = group_rows_batch(np.random.default_rng(77), 4, 5, cfg["max_seq_len"], cfg["n_items"], cfg["n_users"], two)
        out = run_model(m, batch)
        assert out["out.scores"].shape == (20,)
        save(tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G9 optimizer: 3 dense-Adam steps
    for tag, wd, clip in (("wd0", 0.0, None), ("wd1e-6_clip", 1e-6, 0.1)):
        cfg = base_cfg(model="SASRec", n_items=50, n_heads=2, loss_type="bpr", max_seq_len=6)
        torch.manual_seed(77)
        m = SASRec(cfg)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=wd)
        arrs = pack("cfg.", {k: np.array(v) for k, v in cfg.items()})
        arrs.update(pack("sd0.", sd_np(m)))
        arrs["hp.wd"] = np.array(wd)
        arrs["hp.clip"] = np.array(-1.0 if clip is None else clip)
        for step in range(3):
            batch = make_batch(rng, 4, 6, 3, cfg["n_items"], cfg["n_users"])
            m.train()
            tb = tt(batch)
            loss, _, _, _ = m(user_id=tb["user_id"], item_id=tb["item_id"], label=tb["label"],
                              item_seq=tb["item_seq"], item_seq_len=tb["item_seq_len"])
            opt.zero_grad()
            loss.backward()
            if clip is not None:
                torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
            opt.step()
            arrs.update(pack(f"in{step}.", batch))
            arrs[f"loss{step}"] = loss.detach().numpy().copy()
            arrs.update(pack(f"sd{step + 1}.", sd_np(m)))
        save("g9_adam_" + tag, **arrs)

    # ---------------------------------------------------------------- G11 one_vs_all full-item evaluation
    # OnePositiveEvaluator.evaluate_with_full_items run on the reference's own models; get_rank's output is recorded.
    import unirec.facility.evaluation.onepos as onepos

    class _Acc:   # the two Accelerate calls the evaluator makes, single process
        device, is_local_main_process = "cpu", True
        unwrap_model = staticmethod(lambda m: m)
        gather_for_metrics = staticmethod(lambda t: t)

    class _Data(list):
        pass

    rng11 = np.random.default_rng(1111)
    for tag, mk in {"mf_bias_tau": lambda: (MF, base_cfg(model="MF", n_items=517, has_user_emb=True, has_user_bias=True,
                                                       has_item_bias=True, tau=0.5, embedding_size=16, hidden_size=16)),
                    "sasrec": lambda: (SASRec, base_cfg(model="SASRec", n_items=640, n_heads=2))}.items():
        cls, cfg = mk()
        torch.manual_seed(11)
        m = cls(cfg)
        if cfg["has_item_bias"]:
            with torch.no_grad():
                m.item_bias.normal_(0, 0.02)
                m.user_bias.normal_(0, 0.02)
        n_users, n_items, L = cfg["n_users"], cfg["n_items"], cfg["max_seq_len"]
        u2h = np.empty(n_users, dtype=object)
        for u in range(n_users):
            u2h[u] = None if u % 7 == 0 else rng11.integers(0 if u % 5 == 0 else 1, n_items, rng11.integers(1, 60)).astype(np.int64)
        batches, raw = [], []
        for bi in range(2):
            B = 9
            user = rng11.integers(0, n_users + (3 if bi else 0), size=B).astype(np.int64)   # ids >= len(history): no history
            user = np.minimum(user, n_users - 1) if cfg["has_user_emb"] else user
            item = rng11.integers(1, n_items, size=B).astype(np.int64)
            if u2h[user[0]] is not None:
                item[0] = u2h[user[0]][0] if u2h[user[0]][0] > 0 else item[0]                 # a target that is also in the history
            seq = rng11.integers(1, n_items, size=(B, L)).astype(np.int64)
            for b in range(B):
                seq[b, : rng11.integers(0, L)] = 0
            raw.append((user, item, seq))
            batches.append([torch.from_numpy(user), torch.from_numpy(item), torch.from_numpy(seq),
                            torch.from_numpy((seq > 0).sum(1))])
        data = _Data(batches)
        data.dataset = types.SimpleNamespace(return_key_2_index={"user_id": 0, "item_id": 1, "item_seq": 2, "item_seq_len": 3},
                                             config={"data_format": "user-item"})
        ranks = []
        orig = onepos.get_rank
        onepos.get_rank = lambda S: ranks.append(orig(S).copy()) or ranks[-1]
        ev = onepos.OnePositiveEvaluator("['hit@1;5;10', 'ndcg@5;10', 'mrr', 'group_auc']", -1, cfg, _Acc())
        res = ev.evaluate_with_full_items(data, m, u2h)
        onepos.get_rank = orig
        arrs = pack("cfg.", {k: np.array(v) for k, v in cfg.items()})
        arrs.update(pack("sd.", sd_np(m)))
        arrs["hist.ptr"] = np.cumsum([0] + [0 if h is None else len(h) for h in u2h]).astype(np.int64)
        arrs["hist.items"] = np.concatenate([h for h in u2h if h is not None]).astype(np.int64)
        for bi, (user, item, seq) in enumerate(raw):
            arrs[f"in{bi}.user_id"], arrs[f"in{bi}.item_id"], arrs[f"in{bi}.item_seq"] = user, item, seq
            arrs[f"out{bi}.rank"] = ranks[bi].astype(np.int64)
        for k, v in res.items():
            arrs["metric." + k] = np.array(v, dtype=np.float64)
        save("g11_fullrank_" + tag, **arrs)

    # ---------------------------------------------------------------- G12 on-disk formats (prepared-dataset directory)
    # small pickled frames in the reference's formats + what the reference's own loaders make of them
    if not ONLY or any("g12".startswith(p) or p.startswith("g12") for p in ONLY):
        import pandas as pd
        from unirec.utils.general import load_user_history
        from unirec.data.dataset.basedataset import BaseDataset as RefBaseDataset
        ddir = os.path.join(OUT, "g12_dataset")
        os.makedirs(ddir, exist_ok=True)
        r12 = np.random.default_rng(1212)
        n_users, n_items = 30, 90
        users = r12.integers(1, n_users, 400)
        users = users[users != 7]                                    # user 7 has no interactions
        items = r12.integers(1, n_items, len(users))
        full = pd.DataFrame({"user_id": users.astype(np.int64), "item_id": items.astype(np.int64)})
        full.iloc[:300].to_pickle(os.path.join(ddir, "train.pkl"))
        full.iloc[300:].reset_index(drop=True).to_pickle(os.path.join(ddir, "valid.pkl"))
        full.to_pickle(os.path.join(ddir, "user_history.pkl"))      # 'user-item' history: one interaction per row
        seq_df = full.groupby("user_id")["item_id"].apply(lambda x: np.array(x, dtype=np.int32)).to_frame().reset_index()
        seq_df.columns = ["user_id", "item_seq"]
        seq_df.to_pickle(os.path.join(ddir, "user_history_seq.pkl"))  # 'user-item_seq' history: one row per user
        with open(os.path.join(ddir, "data.info"), "w") as f:
            import json
            json.dump({"n_users": n_users, "n_items": n_items, "train_file_format": "user-item", "user_history_file_format": "user-item"}, f)
        h1, _ = load_user_history(ddir, "user_history", n_users=n_users, format="user-item")
        h5, _ = load_user_history(ddir, "user_history_seq", n_users=n_users, format="user-item_seq")
        h_inf, _ = load_user_history(ddir, "user_history", n_users=None, format="user-item")
        ds = RefBaseDataset.__new__(RefBaseDataset)
        import logging
        ds.logger = logging.getLogger("g12")
        train = ds.load_data(ddir, "train")
        arrs = {"n_users": np.array(n_users), "inferred_n_users": np.array(len(h_inf)),
                "train": train[["user_id", "item_id"]].values.astype(np.int64)}
        for tag, h in (("h1", h1), ("h5", h5)):
            arrs[tag + ".ptr"] = np.cumsum([0] + [0 if x is None else len(x) for x in h]).astype(np.int64)
            arrs[tag + ".items"] = np.concatenate([np.asarray(x) for x in h if x is not None]).astype(np.int64)
            arrs[tag + ".isnone"] = np.array([x is None for x in h])
        save("g12_on_disk_expected", **arrs)

    # ---------------------------------------------------------------- G13 full-item top-k (BaseRecommender.topk)
    r13 = np.random.default_rng(1313)
    for tag, mk in {"mf_bias_tau": lambda: (MF, base_cfg(model="MF", n_items=517, has_user_emb=True, has_user_bias=True,
                                                       has_item_bias=True, tau=0.5, embedding_size=16, hidden_size=16)),
                    "sasrec": lambda: (SASRec, base_cfg(model="SASRec", n_items=640, n_heads=2))}.items():
        cls, cfg = mk()
        torch.manual_seed(13)
        m = cls(cfg)
        m.eval()
        if cfg["has_item_bias"]:
            with torch.no_grad():
                m.item_bias.normal_(0, 0.02)
                m.user_bias.normal_(0, 0.02)
        B, H, L, k = 11, 25, cfg["max_seq_len"], 10
        user = r13.integers(1, cfg["n_users"], B).astype(np.int64)
        seq = r13.integers(1, cfg["n_items"], (B, L)).astype(np.int64)
        for b in range(B):
            seq[b, : r13.integers(0, L)] = 0
        hist = r13.integers(1, cfg["n_items"], (B, H)).astype(np.int64)
        for b in range(B):
            hist[b, : r13.integers(1, H)] = 0          # every row keeps >= 1 padding zero (so item 0 is masked, as here)
        inter = {"user_id": torch.from_numpy(user), "item_seq": torch.from_numpy(seq), "item_seq_len": torch.from_numpy((seq > 0).sum(1))}
        with torch.no_grad():
            sc, ids = m.topk(inter, k, user_hist=torch.from_numpy(hist))
        arrs = pack("cfg.", {kk: np.array(v) for kk, v in cfg.items()})
        arrs.update(pack("sd.", sd_np(m)))
        arrs.update({"in.user_id": user, "in.item_seq": seq, "in.user_hist": hist, "out.scores": sc.numpy().copy(), "out.ids": ids.numpy().copy(),
                     "k": np.array(k)})
        save("g13_topk_" + tag, **arrs)

    # ---------------------------------------------------------------- G14 pooled-history encoders (AvgHist, SVD++)
    from unirec.model.sequential.avghist import AvgHist
    from unirec.model.sequential.svdplusplus import SVDPlusPlus
    r14 = np.random.default_rng(1414)
    for tag, cls, kw in (("avghist_asym_bpr", AvgHist, dict(model="AvgHist", asymmetric=True, user_sequence_alpha=0.5, loss_type="bpr")),
                         ("avghist_sym_softmax", AvgHist, dict(model="AvgHist", asymmetric=False, user_sequence_alpha=0.3, loss_type="softmax")),
                         ("svdpp_bce_bias", SVDPlusPlus, dict(model="SVDPlusPlus", user_sequence_alpha=0.5, loss_type="bce", has_user_emb=True,
                                                              has_user_bias=True, has_item_bias=True, tau=0.8))):
        cfg = base_cfg(**kw)
        torch.manual_seed(14)
        m = cls(cfg)
        if kw.get("asymmetric", True):   # make the two tables differ (they are equal copies at construction)
            with torch.no_grad():
                m.item_dst_embedding.weight.add_(torch.randn_like(m.item_dst_embedding.weight) * 0.01)
                m.item_dst_embedding.weight[0].zero_()
        B, L, G = 9, cfg["max_seq_len"], 5
        batch = make_batch(r14, B, L, G, cfg["n_items"], cfg["n_users"])
        out = run_model(m, batch)
        save("g14_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G15 AttHist
    from unirec.model.sequential.atthist import AttHist
    r15 = np.random.default_rng(1515)
    for tag, kw in (("atthist_bpr", dict(loss_type="bpr")), ("atthist_softmax_bias", dict(loss_type="softmax", has_item_bias=True, tau=0.7))):
        cfg = base_cfg(model="AttHist", **kw)
        torch.manual_seed(15)
        m = AttHist(cfg)
        with torch.no_grad():
            m.attention.h.mul_(0.3)      # keep the softmax away from saturation
        batch = make_batch(r15, 8, cfg["max_seq_len"], 4, cfg["n_items"], cfg["n_users"])
        out = run_model(m, batch)
        save("g15_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G16 ConvFormer / FASTConvFormer
    from unirec.model.sequential.convformer import ConvFormer
    from unirec.model.sequential.fastconvformer import FASTConvFormer
    r16 = np.random.default_rng(1616)
    common = dict(n_layers=2, inner_size=64, hidden_dropout_prob=0.0, hidden_act="gelu", layer_norm_eps=1e-9, seq_decay=-0.3, init_ratio=0.05)
    for tag, cls, kw in (("convformer_circ_full", ConvFormer, dict(model="ConvFormer", conv_size=10, padding_mode="circular", seq_merge=False, loss_type="bpr")),
                         ("convformer_reflect_k4_merge", ConvFormer, dict(model="ConvFormer", conv_size=4, padding_mode="reflect", seq_merge=True, loss_type="softmax", hidden_act="swish")),
                         ("convformer_const_k7", ConvFormer, dict(model="ConvFormer", conv_size=7, padding_mode="constant", seq_merge=False, loss_type="bce")),
                         ("fastconvformer_k6", FASTConvFormer, dict(model="FASTConvFormer", conv_size=6, padding_mode=0, seq_merge=False, loss_type="bpr")),
                         ("fastconvformer_full_merge", FASTConvFormer, dict(model="FASTConvFormer", conv_size=10, padding_mode=0, seq_merge=True, loss_type="softmax"))):
        cfg = base_cfg(**dict(common, **kw))
        torch.manual_seed(16)
        m = cls(cfg)
        batch = make_batch(r16, 7, cfg["max_seq_len"], 4, cfg["n_items"], cfg["n_users"])
        out = run_model(m, batch)
        save("g16_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch), **out)

    # ---------------------------------------------------------------- G17 SASRec in training mode WITH dropout
    # Pins WHERE the reference drops and how it scales (sasrec.py:69; modules.py:307,313,352).  torch's nn.Dropout draws from
    # torch's generator, which the device does not reproduce; so every nn.Dropout module of the reference model gets its
    # forward replaced by "multiply with a recorded Bernoulli(1-p)/(1-p) tensor" (= F.dropout's definition), the tensors are
    # stored in the fixture and the oracle replays them.
    r17 = np.random.default_rng(1717)
    for tag, kw in (("sasrec_dropout_bpr", dict(loss_type="bpr", hidden_dropout_prob=0.3, attn_dropout_prob=0.2, n_heads=4)),
                    ("sasrec_dropout_softmax_nopos", dict(loss_type="softmax", hidden_dropout_prob=0.5, attn_dropout_prob=0.5, n_heads=2,
                                                          use_position_emb=False))):
        cfg = base_cfg(model="SASRec", **kw)
        torch.manual_seed(17)
        m = SASRec(cfg)
        m.train()
        gen = torch.Generator().manual_seed(171717)
        masks = {}

        def patch(mod, name):
            def fwd(x, _mod=mod, _name=name):
                mk = (torch.rand(x.shape, generator=gen) >= _mod.p).float() / (1.0 - _mod.p)
                masks[_name] = mk.numpy().copy()
                return x * mk
            mod.forward = fwd
        patch(m.dropout, "embed")
        for i, layer in enumerate(m.trm_encoder.layer):
            patch(layer.multi_head_attention.attn_dropout, f"attn{i}")
            patch(layer.multi_head_attention.out_dropout, f"out{i}")
            patch(layer.feed_forward.dropout, f"ffn{i}")
        batch = make_batch(r17, 6, cfg["max_seq_len"], 4, cfg["n_items"], cfg["n_users"])
        out = run_model(m, batch)
        assert len(masks) == 1 + 3 * cfg["n_layers"], sorted(masks)
        save("g17_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch),
             **pack("mask.", masks), **out)

    # ---------------------------------------------------------------- G18 ConvFormer / FASTConvFormer in training mode WITH dropout
    # (recorded-mask replay as G17; sites convformer.py:59,97,115 and fastconvformer.py:58)
    r18 = np.random.default_rng(1818)
    for tag, cls, kw in (("convformer_dropout", ConvFormer, dict(model="ConvFormer", conv_size=5, padding_mode="reflect", seq_merge=True, loss_type="softmax")),
                         ("fastconvformer_dropout", FASTConvFormer, dict(model="FASTConvFormer", conv_size=6, padding_mode=0, seq_merge=False, loss_type="bpr"))):
        cfg = base_cfg(**dict(common, **kw))
        cfg["hidden_dropout_prob"] = 0.4
        torch.manual_seed(18)
        m = cls(cfg)
        m.train()
        gen = torch.Generator().manual_seed(181818)
        masks = {}

        def patch(mod, name):
            def fwd(x, _mod=mod, _name=name):
                mk = (torch.rand(x.shape, generator=gen) >= _mod.p).float() / (1.0 - _mod.p)
                masks[_name] = mk.numpy().copy()
                return x * mk
            mod.forward = fwd
        patch(m.dropout, "embed")
        for i, layer in enumerate(m.encoder):
            patch(layer.filterlayer.out_dropout, f"out{i}")
            patch(layer.intermediate.dropout, f"ffn{i}")
        batch = make_batch(r18, 6, cfg["max_seq_len"], 4, cfg["n_items"], cfg["n_users"])
        out = run_model(m, batch)
        assert len(masks) == 1 + 2 * cfg["n_layers"], sorted(masks)
        save("g18_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch),
             **pack("mask.", masks), **out)

    # ---------------------------------------------------------------- G19 GRU / AttHist in training mode WITH dropout_prob
    # (recorded-mask replay as G17; sites gru.py:29 and modules.py:242)
    from unirec.model.sequential.atthist import AttHist
    r19 = np.random.default_rng(1919)
    for tag, cls, kw, mods in (("gru_dropout", GRU, dict(model="GRU", loss_type="softmax", hidden_size=24), lambda m: {"embed": m.emb_dropout}),
                               ("atthist_dropout", AttHist, dict(model="AttHist", loss_type="bpr"), lambda m: {"out": m.attention.emb_dropout})):
        cfg = base_cfg(dropout_prob=0.3, **kw)
        torch.manual_seed(19)
        m = cls(cfg)
        m.train()
        gen = torch.Generator().manual_seed(191919)
        masks = {}

        def patch(mod, name):
            def fwd(x, _mod=mod, _name=name):
                mk = (torch.rand(x.shape, generator=gen) >= _mod.p).float() / (1.0 - _mod.p)
                masks[_name] = mk.numpy().copy()
                return x * mk
            mod.forward = fwd
        for name, mod in mods(m).items():
            patch(mod, name)
        batch = make_batch(r19, 6, cfg["max_seq_len"], 4, cfg["n_items"], cfg["n_users"])
        out = run_model(m, batch)
        assert len(masks) == 1, sorted(masks)
        save("g19_" + tag, **pack("cfg.", {k: np.array(v) for k, v in cfg.items()}), **pack("sd.", sd_np(m)), **pack("in.", batch),
             **pack("mask.", masks), **out)


if __name__ == "__main__":
    main()


def capture_g3_g10():
    """G3: the first batches the reference's own DataLoader builds (unirec/main/main.py:121-204 get_data_loader ->
    SeqRecDataset.__getitem__, seqrecdataset.py:38-57: AddNegSamples, AddUserHistory, _padding, default collate) from the committed
    g12_dataset directory.  G10: per-step losses and final parameters of the reference's own Trainer.fit
    (unirec/facility/trainer.py:234-357) over those batches, 2 epochs, Adam, gradient clipping on.  The per-step loss is read
    through Trainer._check_nan (called with every step's loss, trainer.py:343); nothing of the reference is modified."""
    sys.path.insert(0, REF)
    _shims()
    import json
    import logging
    import accelerate
    from unirec.data.dataset.seqrecdataset import SeqRecDataset
    from unirec.data.transform.adduserhistory import AddUserHistory
    from unirec.facility.trainer import Trainer
    from unirec.main import main as refmain
    from unirec.model.sequential.sasrec import SASRec
    from unirec.utils.general import load_user_history
    ddir = os.path.join(OUT, "g12_dataset")
    info = json.load(open(os.path.join(ddir, "data.info")))
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format="user-item")
    cfg = base_cfg(model="SASRec", n_users=info["n_users"], n_items=info["n_items"], n_heads=4, loss_type="softmax", max_seq_len=8,
                   dataset_path=ddir, train_file_format="user-item", num_workers=0, n_sample_neg_train=4, batch_size=64,
                   history_mask_mode="autoregressive", seq_last=0, shuffle_train=0, pin_memory=False, persistent_workers=False,
                   dataloader="SeqRecDataset", use_features=0, time_seq=0, output_path="/tmp/g10_out", checkpoint_dir="ck",
                   optimizer="adam", scheduler="off", scheduler_factor=0.1, learning_rate=2e-3, weight_decay=0, grad_clip_value=0.5,
                   use_tensorboard=0, use_wandb=0, freeze=0, epochs=2, early_stop=0, metrics="['hit@5']", key_metric="hit@5",
                   verbose=0, seed=21, task="train")
    logging.getLogger(cfg["exp_name"]).setLevel(logging.ERROR)

    def loader():
        return refmain.get_data_loader(cfg, "train", AddUserHistory, SeqRecDataset, ddir, "train", user2history=u2h)

    # ---- G3: the batches, as the loop body sees them (trainer.py:328)
    random.seed(21)
    ld = loader()
    keys = ld.dataset.return_key_2_index
    g3 = {"n_batches": np.array(len(ld)), "batch_size": np.array(cfg["batch_size"]), "seed": np.array(21)}
    for e in range(2):                      # two epochs: the sampler stream runs on across epochs
        for i, b in enumerate(ld):
            for k, v in keys.items():
                g3[f"e{e}.b{i}.{k}"] = b[v].numpy().copy()
    save("g3_dataloader_batches", **g3)

    # ---- G10: Trainer.fit over the same stream
    losses = []
    orig = Trainer._check_nan

    def rec(self, loss):
        losses.append(float(loss.detach()))
        return orig(self, loss)
    Trainer._check_nan = rec
    try:
        torch.manual_seed(33)
        m = SASRec(cfg)
        sd0 = sd_np(m)
        tr = Trainer(cfg, m, accelerate.Accelerator(cpu=True))
        tr.evaluate = lambda *a, **k: {cfg["key_metric"]: 0.0}     # no validation set here: fit() evaluates before every epoch
        random.seed(21)
        tr.fit(loader(), valid_data=None, save_model=False, verbose=2)   # (verbose != 2 trips over len(enumerate) at trainer.py:330)
    finally:
        Trainer._check_nan = orig
    keep = {k: (np.array(v) if not isinstance(v, (list, dict)) else np.array(str(v))) for k, v in cfg.items()}
    arrs = pack("cfg.", keep)
    arrs.update(pack("sd0.", sd0))
    arrs.update(pack("sd1.", sd_np(m)))
    arrs["step_losses"] = np.array(losses, dtype=np.float64)
    save("g10_trainer_fit", **arrs)
    print("G10: %d steps, first %.6f last %.6f" % (len(losses), losses[0], losses[-1]))


def capture_g10_gru_mf():
    """G10 for the other two model families SURVEY.md 8(c) names: the reference's own Trainer.fit losses and final parameters for
    GRU (tests/golden/g12_dataset) and for MF + BPR at BASELINE configs[0]'s shape (C1: ML-100K-shaped synthetic data written by
    tests/ml100k_shaped.py, 943 users x 1 682 items, d = 64 per unirec/config/model/MF.yaml:2, batch 400, one epoch = 250 steps)."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    _shims()
    import json
    import logging
    import accelerate
    import ml100k_shaped
    from unirec.data.dataset.basedataset import BaseDataset
    from unirec.data.dataset.seqrecdataset import SeqRecDataset
    from unirec.data.transform.adduserhistory import AddUserHistory
    from unirec.facility.trainer import Trainer
    from unirec.main import main as refmain
    from unirec.model.cf.mf import MF
    from unirec.model.sequential.gru import GRU
    from unirec.utils.general import load_user_history

    def fit(cfg, model, mk_loader, seed):
        losses, orig = [], Trainer._check_nan

        def rec(self, loss):
            losses.append(float(loss.detach()))
            return orig(self, loss)
        Trainer._check_nan = rec
        try:
            tr = Trainer(cfg, model, accelerate.Accelerator(cpu=True))
            tr.evaluate = lambda *a, **k: {cfg["key_metric"]: 0.0}
            random.seed(seed)
            tr.fit(mk_loader(), valid_data=None, save_model=False, verbose=2)
        finally:
            Trainer._check_nan = orig
        return np.array(losses, dtype=np.float64)

    common = dict(train_file_format="user-item", num_workers=0, n_sample_neg_train=4, shuffle_train=0, pin_memory=False,
                  persistent_workers=False, use_features=0, time_seq=0, output_path="/tmp/g10_out", checkpoint_dir="ck", optimizer="adam",
                  scheduler="off", scheduler_factor=0.1, weight_decay=0, use_tensorboard=0, use_wandb=0, freeze=0, early_stop=0,
                  metrics="['hit@5']", key_metric="hit@5", verbose=0, task="train")
    # ---- GRU on the small data set
    ddir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g12_dataset")
    info = json.load(open(os.path.join(ddir, "data.info")))
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format="user-item")
    cfg = base_cfg(model="GRU", n_users=info["n_users"], n_items=info["n_items"], loss_type="bpr", max_seq_len=8, n_layers=1, dataset_path=ddir,
                   batch_size=64, history_mask_mode="autoregressive", seq_last=0, dataloader="SeqRecDataset", learning_rate=2e-3,
                   grad_clip_value=0.5, epochs=2, seed=23, **common)
    logging.getLogger(cfg["exp_name"]).setLevel(logging.ERROR)
    torch.manual_seed(34)
    m = GRU(cfg)
    sd0 = sd_np(m)
    losses = fit(cfg, m, lambda: refmain.get_data_loader(cfg, "train", AddUserHistory, SeqRecDataset, ddir, "train", user2history=u2h), 23)
    arrs = pack("cfg.", {k: (np.array(v) if not isinstance(v, (list, dict)) else np.array(str(v))) for k, v in cfg.items() if k != "dataset_path"})
    arrs.update(pack("sd0.", sd0))
    arrs.update(pack("sd1.", sd_np(m)))
    arrs["step_losses"] = losses
    save("g10_trainer_fit_gru", **arrs)
    print("G10 GRU: %d steps, first %.6f last %.6f" % (len(losses), losses[0], losses[-1]))
    # ---- MF + BPR at C1's shape
    ddir = ml100k_shaped.write("/tmp/g10_ml100k_shaped")
    u2h, _ = load_user_history(ddir, "user_history", n_users=ml100k_shaped.N_USERS, format="user-item")
    cfg = base_cfg(model="MF", n_users=ml100k_shaped.N_USERS, n_items=ml100k_shaped.N_ITEMS, has_user_emb=True, loss_type="bpr", embedding_size=64,
                   hidden_size=64, dataset_path=ddir, batch_size=400, history_mask_mode="unorder", seq_last=0, dataloader="BaseDataset",
                   learning_rate=1e-3, grad_clip_value=-1, epochs=1, seed=25, **common)
    m = MF(cfg)
    sd0 = ml100k_shaped.initial_state()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    losses = fit(cfg, m, lambda: refmain.get_data_loader(cfg, "train", None, BaseDataset, ddir, "train", user2history=u2h), 25)
    arrs = pack("cfg.", {k: (np.array(v) if not isinstance(v, (list, dict)) else np.array(str(v))) for k, v in cfg.items() if k != "dataset_path"})
    arrs.update(pack("sd1_every8.", {k: v[::8].copy() for k, v in sd_np(m).items()}))      # (sd0 = ml100k_shaped.initial_state())
    arrs["step_losses"] = losses
    save("g10_trainer_fit_mf_c1", **arrs)
    print("G10 MF C1: %d steps, first %.6f last %.6f" % (len(losses), losses[0], losses[-1]))


if __name__ == "__main__" and (not ONLY or any(p in ("g3", "g10") for p in ONLY)):
    capture_g3_g10()
if __name__ == "__main__" and (not ONLY or any(p.startswith("g10") for p in ONLY)):
    capture_g10_gru_mf()
