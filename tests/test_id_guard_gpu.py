"""Id guard (SURVEY.md 8b "index-range checks"; the reference: nn.Embedding(n_items, d) at unirec/model/base/reco_abc.py:168-170 raises
IndexError for an id < 0 or >= n_items).  Here the range check rides in the first pass of the row plan every training batch goes
through: the step that looked a bad id up and every step after it are skipped on the device (like a NaN step: no table byte, no moment,
no dense parameter moves), the host raises IndexError one or two steps later with the offending id; under row-sharding the rank's step
flags say "skip" to every rank.  The bounds-checked build (UR_DEBUG_BOUNDS=1) traps at the gather itself."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ITEMS, N_USERS, L, G, B = 1000, 60, 12, 5, 16


def _cfg(kind):
    return dict(model=kind, n_users=N_USERS, n_items=N_ITEMS, device="cuda:0", loss_type="bpr", embedding_size=32, hidden_size=32,
                dropout_prob=0.0, init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=kind == "MF", has_user_bias=False,
                has_item_bias=False, distance_type="dot", tau=1.0, train_file_format="user-item", exp_name="guard", n_layers=1,
                n_heads=4, inner_size=64, hidden_dropout_prob=0.0, attn_dropout_prob=0.0, hidden_act="swish", layer_norm_eps=1e-10,
                max_seq_len=L, use_position_emb=True, seed=3)


def _model(kind):
    from unirec_amd.model.cf.mf import MF
    from unirec_amd.model.sequential.sasrec import SASRec
    torch.manual_seed(0)
    return (MF if kind == "MF" else SASRec)(_cfg(kind))


def _batch(seed, dev):
    g = torch.Generator().manual_seed(seed)
    seq = torch.randint(1, N_ITEMS, (B, L), generator=g, dtype=torch.int32)
    for b in range(B):
        seq[b, : b % L] = 0
    lab = torch.zeros(B, G, dtype=torch.int32)
    lab[:, 0] = 1
    return {k: v.to(dev) for k, v in dict(item_seq=seq, item_id=torch.randint(1, N_ITEMS, (B, G), generator=g), label=lab,
                                          user_id=torch.randint(1, N_USERS, (B,), generator=g)).items()}


def _state(m, opt):
    torch.cuda.synchronize()
    out = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for name, st in opt.tables.items():
        for k in ("m", "v", "last"):
            if st.get(k) is not None:
                out[f"{name}.{k}"] = st[k].clone()
    out["dense_m"] = opt.dense_m.clone()
    return out


def _plain_step(m, opt, b, kind):
    ids = dict(item_seq=b["item_seq"] if kind != "MF" else None, item_id=b["item_id"], user_id=b["user_id"] if kind == "MF" else None)
    opt.plan_batch(**ids)
    loss, _, _, _ = m(user_id=b["user_id"], item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    loss.backward()
    opt.step()


@pytest.mark.parametrize("mode", ["rowwise", "lazy_dense"])
@pytest.mark.parametrize("kind,field,bad", [("SASRec", "item_seq", N_ITEMS), ("SASRec", "item_id", -1), ("SASRec", "item_id", N_ITEMS + 7),
                                            ("MF", "user_id", N_USERS), ("MF", "item_id", -1)])
def test_out_of_range_id_skips_the_step_and_raises_index_error(kind, field, bad, mode):
    from unirec_amd import ops
    from unirec_amd.facility.optimizer import SparseDenseAdam
    dev = torch.device("cuda:0")
    ops.id_guard_reset()
    m = _model(kind)
    opt = SparseDenseAdam(m, lr=1e-2, table_mode=mode)
    m.train()
    try:
        for s in range(2):
            _plain_step(m, opt, _batch(s, dev), kind)
        before = _state(m, opt)
        poisoned = _batch(7, dev)
        poisoned[field].view(-1)[3] = bad
        raised = None
        for s in range(3):              # the bad batch, then up to two clean ones: IndexError one or two plans later
            try:
                _plain_step(m, opt, poisoned if s == 0 else _batch(10 + s, dev), kind)
            except IndexError as e:
                raised = e
                break
        assert raised is not None and str(bad) in str(raised), raised
        after = _state(m, opt)
        if mode == "rowwise":           # nothing moved since the bad batch arrived: weights, moments, dense parameters -- bit for bit
            for k, v in before.items():
                assert torch.equal(v, after[k]), k
        else:
            # lazy_dense: a skipped step IS a zero-gradient step of dense Adam, so looked-up rows may have taken their
            # pending replays.  The dense half must not have moved at all; the tables must equal a twin that trained the two clean
            # batches and then let the same number of zero-gradient steps pass
            for k in [k for k in before if "embedding" not in k]:
                assert torch.equal(before[k], after[k]), k
            ops.id_guard_reset()
            opt.flush()
            twin_m = _model(kind)
            twin = SparseDenseAdam(twin_m, lr=1e-2, table_mode=mode)
            twin_m.train()
            for s in range(2):
                _plain_step(twin_m, twin, _batch(s, dev), kind)
            twin.t = opt.t
            twin.flush()
            torch.cuda.synchronize()
            for name in opt.tables:
                torch.testing.assert_close(opt.tables[name]["w"], twin.tables[name]["w"], rtol=1e-5, atol=1e-7)
    finally:
        torch.cuda.synchronize()
        ops.id_guard_reset()
    # the guard cleared, training goes on
    before = _state(m, opt)
    _plain_step(m, opt, _batch(20, dev), kind)
    _plain_step(m, opt, _batch(21, dev), kind)
    opt.flush()
    torch.cuda.synchronize()
    assert not torch.equal(before["item_embedding.weight"], m.state_dict()["item_embedding.weight"])


@pytest.mark.parametrize("kind", ["SASRec", "MF"])
def test_out_of_range_id_under_the_sharded_step(kind):
    """the multi-GPU step at world 1 (same kernels, same flags as W ranks): the rank's flag row says "skip", flush() raises"""
    from unirec_amd import ops
    from unirec_amd.facility.distributed import ShardedSparseDenseAdam
    dev = torch.device("cuda:0")
    ops.id_guard_reset()
    m = _model(kind)
    opt = ShardedSparseDenseAdam(m, 0, 1, lr=1e-2, table_mode="rowwise")   # (rowwise: a skipped step moves nothing, bit for bit)
    m.train()
    keys = ("user_id", "item_id", "label") if kind == "MF" else ("item_id", "label", "item_seq", "user_id")
    try:
        for s in range(2):
            opt.train_step({k: _batch(s, dev)[k] for k in keys})
        opt.flush()
        before = _state(m, opt)
        poisoned = {k: _batch(7, dev)[k] for k in keys}
        poisoned["item_id"].view(-1)[5] = N_ITEMS
        losses = []
        with pytest.raises(IndexError, match=str(N_ITEMS)):     # (two tables: the second table's plan of the SAME step may already see it)
            losses.append(opt.train_step(poisoned))
            for s in range(3):
                losses.append(opt.train_step({k: _batch(30 + s, dev)[k] for k in keys}))
            opt.flush()
        torch.cuda.synchronize()
        assert all(torch.isnan(x) for x in losses)      # what every rank reads in the step flags: skipped like a NaN step
        after = _state(m, opt)
        for k, v in before.items():
            assert torch.equal(v, after[k]), k
    finally:
        torch.cuda.synchronize()
        ops.id_guard_reset()


def test_bounds_checked_build_traps_at_the_gather():
    """UR_DEBUG_BOUNDS=1 loads libunirec_amd_dbg.so: an out-of-range index in a forward-only gather (no plan, so the release build does
    not see it) kills the launch (and prints the site when the device's printf buffer makes it out before the abort)."""
    code = ("import torch\n"
            "from unirec_amd import ops, _lib\n"
            "assert _lib.LIB_PATH.endswith('libunirec_amd_dbg.so')\n"
            "t = torch.randn(100, 32, device='cuda:0')\n"
            "ok = ops.embedding_gather(t, torch.tensor([1, 99, 0], device='cuda:0'))\n"
            "torch.cuda.synchronize(); print('IN_RANGE_OK', flush=True)\n"
            "ops.embedding_gather(t, torch.tensor([1, 100, 0], device='cuda:0'))\n"
            "torch.cuda.synchronize(); print('NOT_REACHED', flush=True)\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UR_DEBUG_BOUNDS="1"), cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    out = r.stdout + r.stderr
    assert "IN_RANGE_OK" in out and "NOT_REACHED" not in out and r.returncode != 0, out[-2000:]
    # (the device printf names the site when its buffer is flushed before the abort; the hardware exception is there either way)
    assert "bounds check: row index 100 outside [0, 100)" in out or "HSA_STATUS_ERROR_EXCEPTION" in out, out[-2000:]
