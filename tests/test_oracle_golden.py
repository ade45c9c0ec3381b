"""The CPU oracle (oracle/) must reproduce every golden vector captured from the imported
reference (tools/capture_goldens.py).  Runs without a GPU."""
import glob
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import data_ref, model_ref

RTOL, ATOL = 1e-4, 1e-6   # north_star: logits / loss within 1e-4 relative fp32; grads 1e-4 rel + 1e-6 abs

MODEL_FIXTURES = sorted(os.path.basename(p)[:-4] for pat in ("g[578]_*.npz", "g14_*.npz", "g15_*.npz", "g16_*.npz") for p in glob.glob(os.path.join(GOLDEN, pat)))


def _t(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_forward_backward_matches_reference(name):
    cfg, g = load_golden(name)
    P, batch = _t(g["sd"]), _t(g["in"])
    loss, scores, user_emb, G = model_ref.grads_of(P, batch, cfg)
    np.testing.assert_allclose(user_emb.numpy(), g["out"]["user_emb"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(scores.numpy(), g["out"]["scores"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(loss.numpy(), g["out"]["loss"], rtol=RTOL, atol=ATOL)
    for k, ref in g["grad"].items():
        np.testing.assert_allclose(G[k].numpy(), ref, rtol=RTOL, atol=ATOL, err_msg=k)


@pytest.mark.parametrize("name", ["g17_sasrec_dropout_bpr", "g17_sasrec_dropout_softmax_nopos", "g18_convformer_dropout",
                                  "g18_fastconvformer_dropout", "g19_gru_dropout", "g19_atthist_dropout"])
def test_dropout_placement_and_scaling_match_reference(name):
    """The reference SASRec in training mode with its nn.Dropout modules replaced by recorded Bernoulli/(1-p) multipliers
    (tools/capture_goldens.py G17); the oracle replays the same multipliers at the sites it claims they sit at."""
    cfg, g = load_golden(name)
    P, batch = _t(g["sd"]), _t(g["in"])
    assert any((v == 0).any() for v in g["mask"].values())       # dropout was really on
    batch["drop_masks"] = _t(g["mask"])
    loss, scores, user_emb, G = model_ref.grads_of(P, batch, cfg)
    np.testing.assert_allclose(user_emb.numpy(), g["out"]["user_emb"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(loss.numpy(), g["out"]["loss"], rtol=RTOL, atol=ATOL)
    for k, ref in g["grad"].items():
        np.testing.assert_allclose(G[k].numpy(), ref, rtol=RTOL, atol=ATOL, err_msg=k)
    batch.pop("drop_masks")
    _, _, ue0, _ = model_ref.grads_of(P, batch, cfg)
    assert not np.allclose(ue0.numpy(), g["out"]["user_emb"], rtol=1e-2)   # ... and it matters


def test_device_dropout_hash_restatement():
    """oracle/dropout_ref.py: known answers of mix32 (computed by hand from its definition in C unsigned arithmetic),
    keep rate, independence of the sites / steps, p = 0."""
    from oracle import dropout_ref as dr
    assert int(dr.mix32(0)) == 0
    x = 1                                            # the definition, step by step, in Python integers
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF; x ^= x >> 16
    assert int(dr.mix32(1)) == x
    assert dr.threshold(0.5) == 2 ** 31 and dr.threshold(0.0) == 0
    m = dr.mask(512, 256, 0.3, seed=2022, step=7, site=6)
    assert set(np.unique(m)) == {np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(0.3))}
    keep = (m > 0).mean()
    assert abs(keep - 0.7) < 4 * np.sqrt(0.21 / m.size)
    assert abs((m > 0).mean(0) - 0.7).max() < 0.12 and abs((m > 0).mean(1) - 0.7).max() < 0.15     # no dead rows / columns
    m2 = dr.mask(512, 256, 0.3, seed=2022, step=8, site=6)
    m3 = dr.mask(512, 256, 0.3, seed=2022, step=7, site=7)
    for other in (m2, m3):
        agree = ((m > 0) == (other > 0)).mean()
        assert abs(agree - (0.49 + 0.09)) < 0.01       # independent masks agree with probability p^2 + (1-p)^2
    assert (dr.mask(4, 8, 0.0, 1, 1, 1) == 1).all()
    assert np.array_equal(m, dr.mask(512, 256, 0.3, seed=2022, step=7, site=6))


def test_mask_matches_reference():
    cfg, g = load_golden("g5_sasrec_h2_swish_bpr")
    m = model_ref.sasrec_attention_mask(torch.from_numpy(g["in"]["item_seq"]), True)
    assert np.array_equal(m.numpy(), g["mid"]["mask"])  # bit-exact {0,-10000}


@pytest.mark.parametrize("name", ["g9_adam_wd0", "g9_adam_wd1e-6_clip"])
def test_three_dense_adam_steps(name):
    cfg, g = load_golden(name)
    P = {k: v.clone() for k, v in _t(g["sd0"]).items()}
    state = {}
    wd, clip = float(g["hp"]["wd"]), float(g["hp"]["clip"])
    for step in range(3):
        batch = _t(g[f"in{step}"])
        loss = model_ref.train_step(P, state, batch, cfg, lr=1e-3, wd=wd, grad_clip=clip if clip > 0 else None)
        np.testing.assert_allclose(loss, float(g[f"loss{step}"][""] if "" in g[f"loss{step}"] else 0), rtol=RTOL)
        for k, ref in g[f"sd{step + 1}"].items():
            if k.endswith("key.bias"):
                # d loss / d key.bias == 0 analytically (softmax is shift-invariant along keys); the fp32
                # gradient is rounding noise (~1e-12) that Adam's g/(sqrt(v)+eps) turns into +-1e-7 steps.
                assert np.abs(P[k].numpy()).max() < 1e-5 and np.abs(ref).max() < 1e-5
                continue
            np.testing.assert_allclose(P[k].numpy(), ref, rtol=RTOL, atol=2e-7, err_msg=f"step{step} {k}")


# ------------------------------------------------------------------ integer path: bit-exact
def test_mt19937_matches_cpython():
    for seed in (0, 1, 2022, 2 ** 40 + 7, 12345678901234567890123):
        a, b = data_ref.MT19937(seed), random.Random(seed)
        for _ in range(1300):
            assert a.getrandbits(17) == b.getrandbits(17)
        assert a.random() == b.random()
        assert a.randint(1, 59999) == b.randint(1, 59999)
        assert a.choice([3, 4, 5, 9, 11]) == b.choice([3, 4, 5, 9, 11])
        assert a.getrandbits(77) == b.getrandbits(77)


def test_sampler_known_answers():
    _, g = load_golden("g1_sampler")
    g = {k: v for d in g.values() for k, v in d.items()} if "kat_seed2022_n60000_k4" not in g else g
    z = np.load(os.path.join(GOLDEN, "g1_sampler.npz"))
    # SURVEY.md Appendix C values, also stored in the fixture
    kat = z["kat_seed2022_n60000_k4"]
    assert kat.tolist()[0] == [7, 34841, 18935, 29007, 35767]
    rng = data_ref.MT19937(2022)
    hist = [None, {5, 6, 7, 8}, {9, 10}]
    rows = [data_ref.add_neg_samples(rng, 1, 7, 60000, 4, hist) for _ in range(3)]
    assert np.array_equal(np.stack(rows), kat) and rows[0].dtype == np.int64

    rng = data_ref.MT19937(7)
    hist = [None, set(range(1, 12)), set(range(1, 12))]
    out = [data_ref.add_neg_samples(rng, int(u), int(p), 24, 6, hist)
           for u, p in zip(z["small_rows_user"], z["small_rows_pos"])]
    assert np.array_equal(np.stack(out), z["small_out"])

    rng = data_ref.MT19937(11)
    out = data_ref.add_neg_samples(rng, 1, 3, 8, 3, [None, set(range(1, 8))])
    assert np.array_equal(out, z["exhaust_out"]) and out[1:].tolist() == [0, 0, 0]
    assert rng.getrandbits(32) == int(z["after_exhaust_getrandbits32"][0])  # 300 draws consumed, same stream position

    rng = data_ref.MT19937(5)
    ratio = data_ref.pop_sample_ratio(z["pop"], 0.5)
    odds, alias = data_ref.alias_table(list(ratio))
    rows = [data_ref.add_neg_samples(rng, 1, 4, 30, 8, None, sampler=lambda: data_ref.alias_draw(rng, odds, alias))
            for _ in range(4)]
    assert np.array_equal(np.stack(rows), z["pop_out"])


def test_history_and_padding():
    z = np.load(os.path.join(GOLDEN, "g2_history.npz"))
    h = [None, z["h1"], z["h2"], z["h3"]]
    calls = [(1, 7), (2, 9), (2, 9), (2, 9), (3, 15), (7, 3), (1, np.array([7, 99, 6]))]
    for mode in ("autoregressive", "unorder", "autoagressive"):
        for sl in (0, 1):
            rng = data_ref.MT19937(3)
            lens, cat = z[f"{mode}_sl{sl}.lens"], z[f"{mode}_sl{sl}.cat"]
            off = 0
            for (u, it), n in zip(calls, lens):
                ref = cat[off:off + n]
                off += n
                hist, ln = data_ref.add_user_history(rng, u, it, h, mode, sl)
                assert ln == ref[0] and np.array_equal(np.asarray(hist, dtype=np.int64), ref[1:]), (mode, sl, u)
    assert np.array_equal(data_ref.left_pad([4, 5], 6), z["pad_short"])
    assert np.array_equal(data_ref.left_pad([1, 2, 3, 4, 5, 6], 6), z["pad_exact"])
    assert np.array_equal(data_ref.left_pad(range(1, 10), 6), z["pad_long"])
    assert np.array_equal(data_ref.left_pad([0], 6), z["pad_one"])
