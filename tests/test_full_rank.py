"""one_vs_all full-item ranking: oracle/eval_ref.py vs the reference's recorded ranks (CPU), and
ur_full_rank / Trainer.evaluate_full_items vs both (GPU).  Ranks are integers: the bar is exact equality, except that
two fp32 summation orders may disagree on items whose score ties the target's to ~1e-6 -- the GPU tests bound the
difference by the number of such near-ties counted in fp64 (zero on the golden fixtures)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import eval_ref, model_ref

FIX = ["g11_fullrank_mf_bias_tau", "g11_fullrank_sasrec"]
TIE_MARGIN = 2e-6


def _hist(g):
    ptr, items = g["hist"]["ptr"], g["hist"]["items"]
    u2h = np.empty(len(ptr) - 1, dtype=object)
    for u in range(len(u2h)):
        u2h[u] = items[ptr[u]:ptr[u + 1]] if ptr[u + 1] > ptr[u] else None
    return u2h


def _user_emb_cpu(cfg, P, g, bi):
    if cfg["model"] == "MF":
        return model_ref.mf_user_emb(P, torch.from_numpy(g[f"in{bi}"]["user_id"])).numpy()
    return model_ref.sasrec_user_emb(P, torch.from_numpy(g[f"in{bi}"]["item_seq"]), cfg).numpy()


@pytest.mark.parametrize("name", FIX)
def test_oracle_rank_matches_reference(name):
    cfg, g = load_golden(name)
    P = {k: torch.from_numpy(v) for k, v in g["sd"].items()}
    u2h = _hist(g)
    ranks = []
    for bi in range(2):
        uid, tgt = g[f"in{bi}"]["user_id"], g[f"in{bi}"]["item_id"]
        ue = _user_emb_cpu(cfg, P, g, bi)
        ub = g["sd"]["user_bias"][uid] if cfg["has_user_bias"] else None
        ib = g["sd"]["item_bias"] if cfg["has_item_bias"] else None
        s = eval_ref.full_scores(ue, g["sd"]["item_embedding.weight"], ub, ib, cfg["tau"])
        r, _ = eval_ref.full_rank(s, uid, tgt, u2h)
        assert np.array_equal(r, g[f"out{bi}"]["rank"])
        ranks.append(r)
    m = eval_ref.metrics_from_rank(np.concatenate(ranks), cfg["n_items"])
    for k, v in g["metric"].items():
        np.testing.assert_allclose(m[k], float(v), rtol=1e-12, err_msg=k)


# ------------------------------------------------------------------------------------------------ GPU
def _gpu_rank(ue, table, tgt, uid, u2h, ub, ib, tau):
    from unirec_amd import ops
    from unirec_amd.data.rows import HistoryCSR
    dev = "cuda:0"
    hp = hs = None
    if u2h is not None:
        hp, hs = HistoryCSR(u2h).to_device(dev)
    t = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    r, ts = ops.full_rank(t(ue, torch.float32), t(table, torch.float32), t(tgt, torch.int64), t(uid, torch.int64), hp, hs,
                          t(ub, torch.float32), t(ib, torch.float32), tau)
    return r.cpu().numpy(), ts.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIX)
def test_gpu_rank_matches_reference_golden(name):
    cfg, g = load_golden(name)
    P = {k: torch.from_numpy(v) for k, v in g["sd"].items()}
    u2h = _hist(g)
    for bi in range(2):
        uid, tgt = g[f"in{bi}"]["user_id"], g[f"in{bi}"]["item_id"]
        ue = _user_emb_cpu(cfg, P, g, bi)
        r, _ = _gpu_rank(ue, g["sd"]["item_embedding.weight"], tgt, uid, u2h,
                         g["sd"]["user_bias"] if cfg["has_user_bias"] else None,
                         g["sd"]["item_bias"] if cfg["has_item_bias"] else None, float(cfg["tau"]))
        assert np.array_equal(r, g[f"out{bi}"]["rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,B,bias,hist", [(1017, 32, 77, False, True), (128, 64, 5, True, False), (100, 16, 3, True, True),
                                             (60001, 64, 300, True, True), (40960, 128, 513, False, True),
                                             (5000, 200, 64, True, True), (3000, 32, 4200, True, True), (700, 100, 130, False, True)])
def test_gpu_rank_matches_oracle(N, d, B, bias, hist):
    rng = np.random.default_rng(N + d)
    table = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    ue = rng.normal(0, 0.1, (B, d)).astype(np.float32)
    n_users = 50
    uid = rng.integers(0, n_users + 5, B).astype(np.int64)          # some users beyond the history table
    tgt = rng.integers(1, N, B).astype(np.int64)
    u2h = None
    if hist:
        u2h = np.empty(n_users, dtype=object)
        for u in range(n_users):
            h = rng.integers(0, N, rng.integers(0, 400))
            u2h[u] = None if len(h) == 0 else np.concatenate([h, h[:3]])     # duplicates; item 0 appears for some users
        for b in range(0, B, 4):                                             # targets that are also in the history
            if uid[b] < n_users and u2h[uid[b]] is not None and u2h[uid[b]][0] > 0:
                tgt[b] = u2h[uid[b]][0]
    tgt[-1] = N - 1                                                          # a target in the GEMM's N tail
    ub = rng.normal(0, 0.1, n_users + 5).astype(np.float32) if bias else None
    ib = rng.normal(0, 0.1, N).astype(np.float32) if bias else None
    tau = 0.7 if bias else 1.0
    r, ts = _gpu_rank(ue, table, tgt, uid, u2h, ub, ib, tau)
    s64 = eval_ref.full_scores(ue, table, None if ub is None else ub[uid], ib, tau, dtype=np.float64)
    ties = eval_ref.near_ties(s64, uid, tgt, TIE_MARGIN)
    r_ref, ts_ref = eval_ref.full_rank(s64.copy(), uid, tgt, u2h)
    np.testing.assert_allclose(ts, ts_ref, rtol=1e-4, atol=1e-6)
    assert (np.abs(r.astype(np.int64) - r_ref) <= ties).all(), (r[:8], r_ref[:8], ties[:8])
    assert (r == r_ref).mean() > 0.98


@pytest.mark.gpu
def test_trainer_one_vs_all_matches_oracle():
    from unirec_amd.facility.trainer import Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.general import get_class_instance, init_seed
    rng = np.random.default_rng(8)
    n_users, n_items, L = 60, 2000, 12
    cfg = parse_arguments(dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, model="SASRec", n_users=n_users, n_items=n_items, device="cuda:0", loss_type="softmax", embedding_size=32,
                               hidden_size=32, inner_size=64, n_heads=4, max_seq_len=L, epochs=0, batch_size=64, seed=5))
    init_seed(5)
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    tr = Trainer(cfg, model)
    u2h = np.empty(n_users, dtype=object)
    for u in range(n_users):
        u2h[u] = rng.integers(1, n_items, rng.integers(1, 80)).astype(np.int64)
    tr.set_user_history(u2h)
    tr.reset_evaluator("user-item", "one_vs_all")
    batches = []
    for _ in range(3):
        B = 50
        seq = rng.integers(1, n_items, (B, L)).astype(np.int32)
        for b in range(B):
            seq[b, : rng.integers(0, L)] = 0
        batches.append({"user_id": torch.from_numpy(rng.integers(0, n_users, B)).cuda(), "item_id": torch.from_numpy(rng.integers(1, n_items, B)).cuda(),
                        "item_seq": torch.from_numpy(seq).cuda(), "item_seq_len": torch.from_numpy((seq > 0).sum(1)).cuda()})
    res = tr.evaluate(batches, load_best_model=False)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ranks = []
    for b in batches:
        ue = model_ref.sasrec_user_emb(P, b["item_seq"].cpu().long(), cfg).numpy()
        s = eval_ref.full_scores(ue, P["item_embedding.weight"].numpy(), None, None, 1.0, dtype=np.float64)
        ranks.append(eval_ref.full_rank(s, b["user_id"].cpu().numpy(), b["item_id"].cpu().numpy(), u2h)[0])
    ref = eval_ref.metrics_from_rank(np.concatenate(ranks), n_items)
    for k in ("mrr", "group_auc", "hit@10", "ndcg@10"):
        np.testing.assert_allclose(res[k], ref[k], rtol=2e-3, atol=1e-4, err_msg=k)


# ------------------------------------------------------------------------------------------------ top-k (8 f1, second half)
TOPK_FIX = ["g13_topk_mf_bias_tau", "g13_topk_sasrec"]


def _topk_inputs(name):
    cfg, g = load_golden(name)
    P = {k: torch.from_numpy(v) for k, v in g["sd"].items()}
    uid, hist = g["in"]["user_id"], g["in"]["user_hist"]
    if cfg["model"] == "MF":
        ue = model_ref.mf_user_emb(P, torch.from_numpy(uid)).numpy()
    else:
        ue = model_ref.sasrec_user_emb(P, torch.from_numpy(g["in"]["item_seq"]), cfg).numpy()
    return cfg, g, ue, uid, hist


@pytest.mark.parametrize("name", TOPK_FIX)
def test_oracle_topk_matches_reference(name):
    cfg, g, ue, uid, hist = _topk_inputs(name)
    s = eval_ref.full_scores(ue, g["sd"]["item_embedding.weight"], g["sd"]["user_bias"][uid] if cfg["has_user_bias"] else None,
                             g["sd"]["item_bias"] if cfg["has_item_bias"] else None, cfg["tau"])
    sc, ids = eval_ref.full_topk(s, int(g["k"][""]), [h for h in hist])
    assert np.array_equal(ids, g["out"]["ids"])
    np.testing.assert_allclose(sc, g["out"]["scores"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", TOPK_FIX)
def test_gpu_topk_matches_reference_golden(name):
    from unirec_amd.utils.general import get_class_instance
    cfg, g, ue, uid, hist = _topk_inputs(name)
    cfg = dict(cfg, device="cuda:0")
    m = get_class_instance(cfg["model"], "unirec_amd/model")(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g["sd"].items()}, strict=False)
    m.check_views()
    m.eval()
    inter = {"user_id": torch.from_numpy(uid).cuda(), "item_seq": torch.from_numpy(g["in"]["item_seq"]).cuda()}
    sc, ids = m.topk(inter, int(g["k"][""]), user_hist=torch.from_numpy(hist).cuda())
    assert np.array_equal(ids.cpu().numpy(), g["out"]["ids"])
    np.testing.assert_allclose(sc.cpu().numpy(), g["out"]["scores"], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,B,k,bias", [(1017, 32, 40, 10, True), (5003, 64, 33, 100, False), (300, 16, 7, 400, True),
                                          (2_300_001, 32, 9, 50, True), (40960, 128, 130, 1024, False),
                                          (3_200_003, 128, 24, 20, False), (2_200_000, 64, 5, 1000, True),
                                          (300_001, 96, 33, 10, True), (1_048_576, 128, 3, 200, False)])   # N >= 256 K: pruned path
def test_gpu_topk_matches_oracle(N, d, B, k, bias):
    from unirec_amd import ops
    from unirec_amd.data.rows import HistoryCSR
    rng = np.random.default_rng(N + k)
    table = rng.normal(0, 0.1, (N, d)).astype(np.float32)
    ue = rng.normal(0, 0.1, (B, d)).astype(np.float32)
    n_users = 20
    uid = rng.integers(0, n_users + 3, B).astype(np.int64)
    u2h = np.empty(n_users, dtype=object)
    for u in range(n_users):
        h = rng.integers(0, N, rng.integers(0, 300))
        u2h[u] = None if len(h) == 0 else np.concatenate([h, h[:2]])
    if N == 300:
        u2h[1] = np.arange(1, 290)                      # fewer than k admissible items for this user
    ub = rng.normal(0, 0.1, n_users + 3).astype(np.float32) if bias else None
    ib = rng.normal(0, 0.1, N).astype(np.float32) if bias else None
    tau = 0.7 if bias else 1.0
    dev = "cuda:0"
    hp, hs = HistoryCSR(u2h).to_device(dev)
    t = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    sc, ids = ops.full_topk(t(ue, torch.float32), t(table, torch.float32), k, t(uid, torch.int64), hp, hs, t(ub, torch.float32),
                            t(ib, torch.float32), tau)
    sc, ids = sc.cpu().numpy(), ids.cpu().numpy()
    s64 = eval_ref.full_scores(ue, table, None if ub is None else ub[uid], ib, tau, dtype=np.float64)
    rows = [u2h[u] if u < n_users else None for u in uid]
    ref_sc, ref_ids = eval_ref.full_topk(s64.copy(), k, rows)
    fin = np.isfinite(ref_sc)
    assert np.array_equal(np.isfinite(sc), fin) and (ids[~fin] == -1).all()
    np.testing.assert_allclose(sc[fin], ref_sc[fin], rtol=1e-4, atol=1e-6)
    assert (np.diff(sc, axis=1)[fin[:, 1:]] <= 0).all()                                   # best first
    for b in range(B):                                                                     # same set up to fp32 near-ties at the cut
        got, want = set(ids[b][fin[b]].tolist()), set(ref_ids[b][fin[b]].tolist())
        odd = got ^ want
        if odd:
            cut = ref_sc[b][fin[b]][-1]
            assert all(abs(s64[b, n] - cut) < 2e-6 for n in odd), (b, odd)
        h = rows[b]
        assert 0 not in got and (h is None or not (got & set(np.asarray(h).tolist())))


@pytest.mark.gpu
@pytest.mark.parametrize("d,W,with_bias", [(64, 2, False), (64, 3, True), (160, 2, True), (128, 4, False)])
def test_sharded_two_phase_count_equals_the_whole_catalogue(d, W, with_bias):
    """ur_full_rank_shard: W shards of a row-sharded table (item i -> shard i % W, local row i // W + 1), the two
    phases run shard by shard in ONE process and the all-reduces replaced by sums == ur_full_rank on the whole table (exact)."""
    from unirec_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(d + W)
    N, B, n_users = 2777, 37, 20
    table = torch.randn(N, d, generator=g) * 0.3
    table[0] = 0
    bias = (torch.randn(N, generator=g) * 0.1) if with_bias else None
    ue = torch.randn(B, d, generator=g)
    tgt = torch.randint(1, N, (B,), generator=g)
    uid = torch.randint(0, n_users + 2, (B,), generator=g)
    lens = torch.randint(0, 60, (n_users,), generator=g)
    hp = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    raw = torch.randint(1, N, (int(lens.sum()),), generator=g, dtype=torch.int32)
    hs = torch.cat([raw[int(hp[u]):int(hp[u + 1])].sort().values for u in range(n_users)])
    want, _ = ops.full_rank(ue.to(dev), table.to(dev), tgt.to(dev), user_id=uid.to(dev), hist_ptr=hp.to(dev), hist_sorted=hs.to(dev),
                            item_bias=bias.to(dev) if with_bias else None)
    n_local = (N + W - 1) // W + 1
    shards, sbias, args = [], [], []
    for r in range(W):
        t = torch.full((n_local, d), 7.0)          # unused rows hold garbage that WOULD count: they must be ignored
        b = torch.full((n_local,), 3.0)
        ids = torch.arange(1, N)[torch.arange(1, N) % W == r]
        t[ids // W + 1] = table[ids]
        b[ids // W + 1] = bias[ids] if with_bias else 0.0
        t[0] = 0
        b[0] = bias[0] if with_bias else 0.0
        shards.append(t.to(dev)); sbias.append(b.to(dev) if with_bias else None)
        own = hs % W == r
        csum = torch.cat([torch.zeros(1, dtype=torch.int64), own.to(torch.int64).cumsum(0)])
        n_rows, excl = ((N - 1) // W + 2, 1) if r == 0 else (((N - 1 - r) // W + 1) + 1, -1)
        ltgt = torch.where(tgt % W == r, tgt // W + 1, torch.full_like(tgt, -1))
        args.append((ltgt.to(dev), csum[hp].contiguous().to(dev), (hs[own] // W + 1).to(torch.int32).to(dev), n_rows, excl))
    thr = sum(ops.full_rank_shard(1, ue.to(dev), shards[r], args[r][0], item_bias_local=sbias[r], n_rows=args[r][3]) for r in range(W))
    got = sum(ops.full_rank_shard(2, ue.to(dev), shards[r], args[r][0], thr=thr.contiguous(), user_id=uid.to(dev), hist_ptr=args[r][1],
                                  hist_sorted_local=args[r][2], item_bias_local=sbias[r], n_rows=args[r][3], excl_row=args[r][4])
              for r in range(W))
    assert torch.equal(got.cpu().to(torch.int32), want.cpu()), (got.cpu(), want.cpu())


@pytest.mark.gpu
def test_topk_candidate_overflow_falls_back_to_the_chunked_path():
    """UR_TEST=topk_cap=16 forces every row's candidate list of the pruned path to overflow; the result must still be the oracle's."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, UR_TEST="topk_cap=16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "test_gpu_topk_matches_oracle and 3200003"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
