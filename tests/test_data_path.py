"""Data path: the native host row builder is bit-exact against the golden vectors captured
from the imported reference and against the oracle; the device sampler is bit-exact against oracle/philox_ref.py and
passes the distribution checks."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import data_ref, philox_ref
from unirec_amd.data.rows import HistoryCSR, HostRowBuilder


def _u2h(*hists):
    a = np.empty(len(hists), dtype=object)
    for i, h in enumerate(hists):
        a[i] = None if h is None else np.asarray(h, dtype=np.int32)
    return a


def test_host_sampler_matches_reference_known_answers():
    z = np.load(os.path.join(GOLDEN, "g1_sampler.npz"))
    # random.seed(2022); AddNegSamples(5, 60000, 4, user2history={1:[5,6,7,8], 2:[9,10]})
    b = HostRowBuilder(5, 60000, 4, history=HistoryCSR(_u2h(None, [5, 6, 7, 8], [9, 10])), seed=2022)
    rows = np.concatenate([b.build([1], [7], with_seq=False)["item_id"] for _ in range(3)])
    assert rows.dtype == np.int64 and np.array_equal(rows, z["kat_seed2022_n60000_k4"])
    assert rows[0].tolist() == [7, 34841, 18935, 29007, 35767]
    # the same three rows in ONE native call (stream order is row-major)
    b = HostRowBuilder(5, 60000, 4, history=HistoryCSR(_u2h(None, [5, 6, 7, 8], [9, 10])), seed=2022)
    assert np.array_equal(b.build([1, 1, 1], [7, 7, 7], with_seq=False)["item_id"], z["kat_seed2022_n60000_k4"])
    # rejections in a small catalogue
    b = HostRowBuilder(3, 24, 6, history=HistoryCSR(_u2h(None, np.arange(1, 12), np.arange(1, 12))), seed=7)
    out = b.build(z["small_rows_user"], z["small_rows_pos"], with_seq=False)["item_id"]
    assert np.array_equal(out, z["small_out"])
    # exhaustion: every id is in the history -> 100 tries each, then id 0; stream position afterwards identical
    b = HostRowBuilder(2, 8, 3, history=HistoryCSR(_u2h(None, np.arange(1, 8))), seed=11)
    out = b.build([1], [3], with_seq=False)["item_id"][0]
    assert np.array_equal(out, z["exhaust_out"]) and out[1:].tolist() == [0, 0, 0]
    from unirec_amd._lib import lib
    assert lib.ur_host_sampler_getrandbits(b._h, 32) == int(z["after_exhaust_getrandbits32"][0])
    # popularity (alias) sampler
    b = HostRowBuilder(3, 30, 8, seed=5, item_popularity=z["pop"], neg_by_pop_alpha=0.5)
    rows = np.concatenate([b.build([1], [4], with_seq=False)["item_id"] for _ in range(4)])
    assert np.array_equal(rows, z["pop_out"])


def test_add_neg_samples_class_is_drop_in():
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    z = np.load(os.path.join(GOLDEN, "g1_sampler.npz"))
    t = AddNegSamples(5, 60000, 4, user2history=_u2h(None, [5, 6, 7, 8], [9, 10]), seed=2022)
    rows = [t(np.array([1, 7], dtype=object))[1] for _ in range(3)]
    assert np.array_equal(np.stack(rows), z["kat_seed2022_n60000_k4"])


def test_history_cut_and_padding_match_reference():
    z = np.load(os.path.join(GOLDEN, "g2_history.npz"))
    h = _u2h(None, z["h1"], z["h2"], z["h3"])
    calls = [(1, 7), (2, 9), (2, 9), (2, 9), (3, 15), (7, 3)]
    L = 40
    for mode in ("autoregressive", "unorder", "autoagressive"):
        for sl in (0, 1):
            # n_neg = 0: the id group is just the positive, and no negative draws perturb the stream (golden: seed 3)
            b = HostRowBuilder(8, 1000, 0, L, HistoryCSR(h, 8), reject_history=True, mask_mode=mode, seq_last=sl, seed=3)
            lens, cat = z[f"{mode}_sl{sl}.lens"], z[f"{mode}_sl{sl}.cat"]
            off = 0
            for (u, it), n in zip(calls, lens):
                ref_len, ref_hist = int(cat[off]), cat[off + 1:off + n]
                off += n
                r = b.build([u], [it])
                assert int(r["item_seq_len"][0]) == min(ref_len, L), (mode, sl, u)
                assert np.array_equal(r["item_seq"][0], data_ref.left_pad(ref_hist, L)), (mode, sl, u)
    b = HostRowBuilder(4, 1000, 0, 6, HistoryCSR(_u2h(None, [4, 5], [1, 2, 3, 4, 5, 6], list(range(1, 10)))), mask_mode="x")
    r = b.build([1, 2, 3, 9], [999, 999, 999, 999])
    assert np.array_equal(r["item_seq"][0], z["pad_short"]) and np.array_equal(r["item_seq"][1], z["pad_exact"])
    assert np.array_equal(r["item_seq"][2], z["pad_long"]) and np.array_equal(r["item_seq"][3], z["pad_one"])
    assert r["item_seq_len"].tolist() == [2, 6, 6, 1]


def test_whole_rows_match_oracle_stream():
    """negatives + autoregressive cut + padding on ONE stream, row after row (SeqRecDataset.__getitem__ order)."""
    rng = np.random.default_rng(0)
    n_users, n_items, K, L = 30, 400, 5, 12
    u2h = _u2h(None, *[rng.integers(1, n_items, rng.integers(1, 25)) for _ in range(n_users - 1)])
    sets = [None if h is None else set(int(x) for x in h) for h in u2h]
    users = rng.integers(1, n_users, 200)
    items = np.array([int(rng.choice(u2h[u])) for u in users])
    for mode, sl in (("autoregressive", 0), ("autoregressive", 1), ("unorder", 0)):
        ref = data_ref.MT19937(99)
        b = HostRowBuilder(n_users, n_items, K, L, HistoryCSR(u2h), mask_mode=mode, seq_last=sl, seed=99)
        got = b.build(users, items)
        for r in range(len(users)):
            _, ids, lab, seq, ln = data_ref.make_row(ref, int(users[r]), int(items[r]), n_items=n_items, n_neg=K, max_seq_len=L,
                                                     user2history=u2h, history_sets=sets, mask_mode=mode, seq_last=sl)
            assert np.array_equal(got["item_id"][r], ids) and np.array_equal(got["label"][r], lab)
            assert np.array_equal(got["item_seq"][r], seq) and got["item_seq_len"][r] == ln


def test_datasets_and_transforms_compose_like_the_reference():
    from unirec_amd.data.dataset.seqrecdataset import SeqRecDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.data.transform.adduserhistory import AddUserHistory
    u2h = _u2h(None, [5, 6, 7, 8], [9, 10, 11])
    cfg = {"max_seq_len": 5, "seed": 1}
    ds = SeqRecDataset(cfg, transform=AddNegSamples(3, 50, 2, user2history=u2h, seed=1), data=np.array([[1, 7], [2, 11], [1, 8]]))
    ds.add_user_history_transform(AddUserHistory(u2h, "autoregressive", seq_last=1))
    assert list(ds.return_key_2_index) == ["user_id", "item_id", "label", "item_seq", "item_seq_len"]
    u, ids, lab, seq, ln = ds[0]
    assert u == 1 and ids[0] == 7 and lab.tolist() == [1, 0, 0] and seq.tolist() == [0, 0, 0, 5, 6] and ln == 2
    bt = ds.get_batch(np.array([1, 2]))
    assert bt["item_seq"].tolist() == [[0, 0, 0, 9, 10], [0, 0, 5, 6, 7]] and bt["item_id"].shape == (2, 3)
    assert not (set(bt["item_id"][1, 1:].tolist()) & {5, 6, 7, 8})


@pytest.mark.gpu
def test_device_sampler_bit_exact_vs_philox_oracle_and_distribution():
    from unirec_amd.data.rows import sample_negatives_device
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    n_users, n_items, K, B = 20, 97, 6, 64
    u2h = _u2h(None, *[rng.integers(1, n_items, rng.integers(1, 40)) for _ in range(n_users - 1)])
    csr = HistoryCSR(u2h)
    users = rng.integers(0, n_users + 3, B)           # includes unknown users (id >= n_users) and user 0 (no history)
    pos = rng.integers(1, n_items, B)
    for step in (0, 5):
        ids, lab = sample_negatives_device(torch.from_numpy(pos).to(dev), K, n_items, torch.from_numpy(users).to(dev), csr, seed=2022, step=step)
        ref = philox_ref.sample_negatives(users, pos, K, n_items, csr.ptr, csr.sorted, seed=2022, step=step)
        assert torch.equal(ids.cpu(), torch.from_numpy(ref))            # bit-exact
        assert lab.cpu().tolist() == [[1] + [0] * K] * B
    ids = ids.cpu().numpy()
    for b in range(B):
        h = set() if not (0 <= users[b] < n_users) or u2h[users[b]] is None else set(int(x) for x in u2h[users[b]])
        for c in ids[b, 1:]:
            assert c == 0 or (1 <= c < n_items and c != pos[b] and c not in h)
    # uniformity over [1, N-1] without a history: chi-square at 5 sigma
    N, Bn, Kn = 1000, 2048, 64
    big, _ = sample_negatives_device(torch.full((Bn,), N + 5, dtype=torch.int64, device=dev).clamp_(max=N - 1), Kn, N, seed=7, step=1)
    x = big[:, 1:].reshape(-1).cpu().numpy()
    cnt = np.bincount(x, minlength=N)[1:N]
    cnt = np.delete(cnt, N - 2)                                      # the positive (id N-1) is excluded by construction
    e = cnt.sum() / len(cnt)
    chi2 = ((cnt - e) ** 2 / e).sum()
    dof = len(cnt) - 1
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof), (chi2, dof)
    assert x.min() >= 1 and x.max() <= N - 2
    # exhaustion -> 0
    csr2 = HistoryCSR(_u2h(None, np.arange(1, 8)))
    ex, _ = sample_negatives_device(torch.tensor([3], device=dev), 3, 8, torch.tensor([1], device=dev), csr2, seed=1)
    assert ex.cpu().tolist() == [[3, 0, 0, 0]]


def test_native_alias_table_equals_the_reference_algorithm():
    """ur_alias_table_build == oracle/data_ref.alias_table (itself pinned to the reference's sampler by g1's pop_out)."""
    from oracle import data_ref
    from unirec_amd.data.rows import alias_table, pop_sample_ratio
    rng = np.random.default_rng(4)
    for n, alpha in ((30, 0.5), (257, 1.0), (1000, 0.25)):
        pop = rng.integers(0, 50, n).astype(np.float64)
        pop[rng.integers(1, n, 5)] = 0          # unseen items
        w = pop_sample_ratio(pop, alpha)
        assert np.array_equal(w, data_ref.pop_sample_ratio(pop, alpha))
        odds, idx = alias_table(w)
        ro, ri = data_ref.alias_table(list(w))
        assert np.array_equal(odds, np.asarray(ro, dtype=np.float64)) and np.array_equal(idx, np.asarray(ri, dtype=np.int64))


@pytest.mark.gpu
def test_device_popularity_sampler_bit_exact_and_distribution():
    """neg_by_pop_alpha > 0 on the device (SURVEY.md 8 a1 + f2): alias draw from Philox words, bit-exact vs oracle/philox_ref.py,
    empirical frequencies = pop^alpha / sum, history / positive rejection as in the uniform sampler."""
    from unirec_amd.data.rows import DeviceRowBuilder, alias_table, pop_sample_ratio, sample_negatives_device
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2)
    n_users, n_items, K, B = 20, 97, 6, 64
    pop = rng.integers(0, 30, n_items).astype(np.float64)
    pop[[5, 17]] = 0
    w = pop_sample_ratio(pop, 0.75)
    odds, idx = alias_table(w)
    alias_dev = (torch.from_numpy(odds).to(dev), torch.from_numpy(idx).to(dev))
    u2h = _u2h(None, *[rng.integers(1, n_items, rng.integers(1, 40)) for _ in range(n_users - 1)])
    csr = HistoryCSR(u2h)
    users = rng.integers(0, n_users + 3, B)
    pos = rng.integers(1, n_items, B)
    for step in (0, 9):
        ids, _ = sample_negatives_device(torch.from_numpy(pos).to(dev), K, n_items, torch.from_numpy(users).to(dev), csr, seed=31, step=step,
                                         alias=alias_dev)
        ref = philox_ref.sample_negatives(users, pos, K, n_items, csr.ptr, csr.sorted, seed=31, step=step, alias=(odds, idx))
        assert torch.equal(ids.cpu(), torch.from_numpy(ref))
    got = ids.cpu().numpy()
    for b in range(B):
        h = set() if not (0 <= users[b] < n_users) or u2h[users[b]] is None else set(int(x) for x in u2h[users[b]])
        for c in got[b, 1:]:
            assert c == 0 or (1 <= c < n_items and c != pos[b] and c not in h and w[c] > 0)
    # frequencies: no history, positive = an item of weight 0 -> draws follow w exactly
    Bn, Kn = 4096, 64
    big, _ = sample_negatives_device(torch.full((Bn,), 5, dtype=torch.int64, device=dev), Kn, n_items, seed=3, step=2, alias=alias_dev)
    x = big[:, 1:].reshape(-1).cpu().numpy()
    cnt = np.bincount(x, minlength=n_items).astype(np.float64)
    assert cnt[0] == 0 and cnt[5] == 0 and cnt[17] == 0
    e = w / w.sum() * cnt.sum()
    nz = e > 0
    chi2 = ((cnt[nz] - e[nz]) ** 2 / e[nz]).sum()
    dof = int(nz.sum()) - 1
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof), (chi2, dof)
    # through the row builder
    bld = DeviceRowBuilder(n_users, n_items, K, history=csr, seed=31, device="cuda:0", item_popularity=pop, neg_by_pop_alpha=0.75)
    out = bld.build(torch.from_numpy(users).to(dev), torch.from_numpy(pos).to(dev), with_seq=False, step=9)
    assert torch.equal(out["item_id"].cpu(), torch.from_numpy(ref))


# ------------------------------------------------------------------------------------------ device row builder (8 f2)
def _random_history(rng, n_users, n_items, max_len):
    u2h = np.empty(n_users, dtype=object)
    for u in range(n_users):
        n = int(rng.integers(0, max_len))
        u2h[u] = None if (n == 0 or u % 9 == 0) else rng.integers(1, n_items, n).astype(np.int32)
    return u2h


@pytest.mark.gpu
@pytest.mark.parametrize("mask_mode,seq_last,reject", [("autoregressive", 0, True), ("autoregressive", 1, True), ("unorder", 0, True),
                                                       ("autoregressive", 0, False), ("unorder", 0, False)])
def test_device_row_builder_bit_exact_vs_philox_oracle(mask_mode, seq_last, reject):
    import torch
    from oracle import data_ref, philox_ref
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR
    rng = np.random.default_rng(17)
    n_users, n_items, K, L, B = 70, 60, 5, 12, 300      # small catalogue: repeated items, positives occurring several times
    u2h = _random_history(rng, n_users, n_items, 150)
    csr = HistoryCSR(u2h)
    user = rng.integers(0, n_users + 4, B).astype(np.int64)          # some users beyond the table
    pos = rng.integers(1, n_items, B).astype(np.int64)
    for b in range(0, B, 2):                                          # make the positive an item of the user's history
        if user[b] < n_users and u2h[user[b]] is not None:
            pos[b] = int(rng.choice(u2h[user[b]]))
    bld = DeviceRowBuilder(n_users, n_items, K, L, csr, reject_history=reject, mask_mode=mask_mode, seq_last=seq_last, seed=99)
    out = bld.build(torch.from_numpy(user).cuda(), torch.from_numpy(pos).cuda(), step=7)
    item_id = out["item_id"].cpu().numpy()
    ref_ids = philox_ref.sample_negatives(user, pos, K, n_items, csr.ptr if reject else None, csr.sorted if reject else None, seed=99, step=7)
    assert np.array_equal(item_id, ref_ids)
    seq, slen = philox_ref.build_seq(user, item_id, csr.ptr, csr.items, L, mask_mode, seq_last, match_all=not reject, seed=99, step=7)
    assert np.array_equal(out["item_seq"].cpu().numpy(), seq)
    assert np.array_equal(out["item_seq_len"].cpu().numpy(), slen)
    assert np.array_equal(out["label"].cpu().numpy()[:, 0], np.ones(B, np.int32)) and out["label"].sum().item() == B
    if seq_last or mask_mode == "unorder":
        # no random choice involved: must equal the reference-pinned host restatement (oracle/data_ref.py) too
        for b in range(B):
            h, hl = data_ref.add_user_history(None, int(user[b]), item_id[b], u2h, mask_mode, seq_last)
            assert np.array_equal(data_ref.left_pad(h, L), seq[b]) and min(hl, L) == slen[b], b


def test_batch_loader_reproduces_the_references_dataloader_batches():
    """the batches of the reference's own DataLoader (unirec/main/main.py:121-204 get_data_loader ->
    SeqRecDataset.__getitem__ -> default collate) over tests/golden/g12_dataset, two epochs of one sampler stream, captured by
    tools/capture_goldens.py; BatchLoader over this package's SeqRecDataset must produce the same tensors, bit for bit."""
    import torch
    from conftest import GOLDEN
    from unirec_amd.data.dataset.seqrecdataset import SeqRecDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.data.transform.adduserhistory import AddUserHistory
    from unirec_amd.facility.trainer import BatchLoader
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.file_io import load_data_info
    from unirec_amd.utils.general import load_user_history
    g = np.load(os.path.join(GOLDEN, "g3_dataloader_batches.npz"))
    ddir = os.path.join(GOLDEN, "g12_dataset")
    info = load_data_info(ddir)
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format=info["user_history_file_format"])
    cfg = parse_arguments(dict(model="SASRec", n_users=info["n_users"], n_items=info["n_items"], device="cpu", max_seq_len=8,
                               batch_size=int(g["batch_size"]), n_sample_neg_train=4, history_mask_mode="autoregressive", seed=int(g["seed"])))
    ds = SeqRecDataset(cfg, path=ddir, filename="train",
                       transform=AddNegSamples(info["n_users"], info["n_items"], 4, user2history=u2h, seed=int(g["seed"])))
    ds.add_user_history_transform(AddUserHistory(u2h, "autoregressive", seq_last=0))
    ld = BatchLoader(ds, int(g["batch_size"]), device="cpu")
    assert len(ld) == int(g["n_batches"])
    for e in range(2):
        batches = list(ld)
        assert len(batches) == int(g["n_batches"])
        for i, b in enumerate(batches):
            for k in ("user_id", "item_id", "label", "item_seq", "item_seq_len"):
                want = g[f"e{e}.b{i}.{k}"]
                got = b[k].numpy()
                assert got.shape == want.shape, (e, i, k, got.shape, want.shape)
                assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), (e, i, k)
            assert b["item_seq"].dtype == torch.int32 and b["item_id"].dtype == torch.int64     # the dtypes the reference collates to


def test_loss_check_draw_keeps_the_sampler_on_the_references_stream():
    """The reference's BPR / CCL loss draws ``random.random()`` once per training forward (the 10 % label check,
    unirec/model/base/reco_abc.py:238-246) from the process-global stream its negative sampler uses -- between the rows of batch i and
    those of batch i + 1.  BatchLoader.loss_check_draws = 1 (set by Trainer.fit for those losses) does the same on the native stream:
    the batches then equal CPython's own ``random`` driven that way."""
    import random
    from unirec_amd.data.dataset.basedataset import BaseDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.facility.trainer import BatchLoader
    n_users, n_items, K, B = 30, 400, 4, 16
    rng = np.random.default_rng(9)
    data = np.stack([rng.integers(1, n_users, 80), rng.integers(1, n_items, 80)], 1)
    ds = BaseDataset({}, transform=AddNegSamples(n_users, n_items, K, seed=77), data=data)
    ld = BatchLoader(ds, B, device="cpu")
    ld.loss_check_draws = 1
    got = [b["item_id"].numpy() for b in ld]
    random.seed(77)

    def rows_of(i):
        want = []
        for p in data[i * B:(i + 1) * B, 1]:
            row = [int(p)]
            while len(row) < K + 1:                      # addnegsamples.py:90-108 without a history: redraw what equals the positive
                x = random.randint(1, n_items - 1)
                if x != int(p):
                    row.append(x)
            want.append(row)
        return np.array(want)
    # Accelerate's DataLoaderShard (the reference's prepared loader) builds batch i + 1 before it hands out batch i: rows(0), rows(1),
    # step 0's draw, rows(2), step 1's draw, ...
    want = [rows_of(0)]
    for i in range(1, len(got)):
        want.append(rows_of(i))
        random.random()
    for i, g in enumerate(got):
        assert np.array_equal(g, want[i]), i


# ------------------------------------------------------------------------------------- the reference's MT19937 stream ON THE DEVICE
@pytest.mark.gpu
def test_mt_device_builder_reproduces_the_references_dataloader_batches():
    """VERDICT r5 "missing 2": the device-resident row builder on the reference's own random stream.  DeviceRowBuilder(rng="mt19937")
    must hand out the tensors the REFERENCE's DataLoader produced (golden G3: unirec/main/main.py get_data_loader -> SeqRecDataset ->
    AddNegSamples + AddUserHistory on one `random` stream, two epochs) -- negatives, history cuts, padding, bit for bit.  The 90-item
    catalogue makes history / positive rejections frequent: the exact-replay path of csrc/mt_sampler.hip carries most rows here."""
    import pandas as pd
    import torch
    from conftest import GOLDEN
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR
    from unirec_amd.utils.file_io import load_data_info
    from unirec_amd.utils.general import load_user_history
    g = np.load(os.path.join(GOLDEN, "g3_dataloader_batches.npz"))
    ddir = os.path.join(GOLDEN, "g12_dataset")
    info = load_data_info(ddir)
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format=info["user_history_file_format"])
    pairs = pd.read_pickle(os.path.join(ddir, "train.pkl"))[["user_id", "item_id"]].values.astype(np.int64)
    B, nb = int(g["batch_size"]), int(g["n_batches"])
    bld = DeviceRowBuilder(info["n_users"], info["n_items"], 4, 8, HistoryCSR(u2h, info["n_users"]), reject_history=True,
                           mask_mode="autoregressive", seq_last=0, seed=int(g["seed"]), rng="mt19937")
    dev = torch.device("cuda:0")
    for e in range(2):
        for i in range(nb):
            rows = pairs[i * B:(i + 1) * B]
            out = bld.build(torch.from_numpy(rows[:, 0].copy()).to(dev), torch.from_numpy(rows[:, 1].copy()).to(dev))
            for k in ("user_id", "item_id", "label", "item_seq", "item_seq_len"):
                want = g[f"e{e}.b{i}.{k}"]
                got = out[k].cpu().numpy()
                assert got.shape == want.shape, (e, i, k, got.shape, want.shape)
                assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), (e, i, k)
    bld.check()


@pytest.mark.gpu
@pytest.mark.parametrize("mask_mode,seq_last,reject", [("autoregressive", 0, True), ("autoregressive", 1, True), ("unorder", 0, True),
                                                       ("autoregressive", 0, False), ("unorder", 0, False)])
def test_mt_device_builder_equals_the_host_builder_on_a_tiny_catalogue(mask_mode, seq_last, reject):
    """60 items, users beyond the table, positives that occur several times in a history, negatives allowed inside it (reject = False:
    the cut then depends on the sampled negatives): every rule of the host builder -- itself pinned to the reference (G1-G3) -- on the
    device stream, several batches in a row (the stream's state carries over)."""
    import torch
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR, HostRowBuilder
    rng = np.random.default_rng(23)
    n_users, n_items, K, L, B = 70, 60, 5, 12, 200
    u2h = _random_history(rng, n_users, n_items, 150)
    csr = HistoryCSR(u2h)
    host = HostRowBuilder(n_users, n_items, K, L, csr, reject_history=reject, mask_mode=mask_mode, seq_last=seq_last, seed=4242)
    devb = DeviceRowBuilder(n_users, n_items, K, L, csr, reject_history=reject, mask_mode=mask_mode, seq_last=seq_last, seed=4242, rng="mt19937")
    for it in range(4):
        user = rng.integers(0, n_users + 4, B).astype(np.int64)
        pos = rng.integers(1, n_items, B).astype(np.int64)
        for b in range(0, B, 2):
            if user[b] < n_users and u2h[user[b]] is not None:
                pos[b] = int(rng.choice(u2h[user[b]]))
        want = host.build(user, pos)
        got = devb.build(torch.from_numpy(user).cuda(), torch.from_numpy(pos).cuda())
        for k in ("item_id", "label", "item_seq", "item_seq_len"):
            assert np.array_equal(got[k].cpu().numpy().astype(np.int64), np.asarray(want[k]).astype(np.int64)), (it, k)
    devb.check()


@pytest.mark.gpu
@pytest.mark.parametrize("K,B,n_items", [(4, 512, 100_000_000), (4, 512, 60_000), (1000, 128, 2_000_000)])
def test_mt_device_builder_equals_the_host_builder_on_1e5_rows(K, B, n_items):
    """10^5 rows at the shapes of BASELINE's configs (K = 4 at 100 M / 60 K items, K = 1000 at 2 M items): item_id, item_seq and
    item_seq_len of every batch torch.equal to HostRowBuilder's on one continued stream; rows/s of both printed."""
    import time
    import torch
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR, HostRowBuilder
    rng = np.random.default_rng(K + B)
    n_users, L, n_rows = 5000, 50, 100_000 if K <= 4 else 12_800
    u2h = np.empty(n_users, dtype=object)
    u2h[0] = None
    for u in range(1, n_users):
        u2h[u] = rng.integers(1, n_items, rng.integers(2, 120)).astype(np.int32)
    csr = HistoryCSR(u2h)
    users = rng.integers(1, n_users, n_rows).astype(np.int64)
    pos = np.array([int(u2h[u][rng.integers(0, len(u2h[u]))]) for u in users], dtype=np.int64)
    host = HostRowBuilder(n_users, n_items, K, L, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=7)
    devb = DeviceRowBuilder(n_users, n_items, K, L, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=7, rng="mt19937")
    du, dp = torch.from_numpy(users).cuda(), torch.from_numpy(pos).cuda()
    t0 = time.perf_counter()
    want = [host.build(users[i:i + B], pos[i:i + B]) for i in range(0, n_rows, B)]
    t_host = time.perf_counter() - t0
    devb.build(du[:B], dp[:B])                      # warm-up (allocations) on a throw-away stream position ...
    devb._mt_state = None                           # ... then start the stream over
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = [devb.build(du[i:i + B], dp[i:i + B]) for i in range(0, n_rows, B)]
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    devb.check()
    for i, (w, o) in enumerate(zip(want, got)):
        for k in ("item_id", "item_seq", "item_seq_len"):
            assert torch.equal(o[k].cpu().to(torch.int64), torch.from_numpy(np.asarray(w[k]).astype(np.int64))), (i, k)
    print(f"MT19937 row builders, K={K} B={B} N={n_items}: device {n_rows / t_dev:,.0f} rows/s, host {n_rows / t_host:,.0f} rows/s")
