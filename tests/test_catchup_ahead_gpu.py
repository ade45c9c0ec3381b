"""Lazy-Adam catch-up of the NEXT batch's rows issued ahead of time, at the tail of the step in flight.

The reference's dense torch.optim.Adam moves every embedding row every step (unirec/facility/trainer.py:349).  The lazy table replays a
row's missed zero-gradient steps when the row is next looked up; with a plan prefetched for the next batch `SparseDenseAdam.step()` does
that replay right after its own row update (main stream, under the tail of the dense-gradient stream).  A zero-gradient step depends on
the step index only, so the trajectory must be BIT-identical to catching up at the head of the next step -- that is what is asserted here,
on a small table (every row comes back every few steps), for Adam and AdamW-with-decay, with a prefetched batch that is then not the one
trained on, and with steps whose loss guard skips the update.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from unirec_amd.utils.argument_parser import parse_arguments
    return parse_arguments(dict(dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, model="SASRec", n_users=50, n_items=300, device="cuda:0",
                                     loss_type="bpr", embedding_size=32, hidden_size=32, inner_size=64, n_heads=4, max_seq_len=10, epochs=1,
                                     batch_size=16, seed=4, n_sample_neg_train=4), **kw))


def _batches(n, B=16, L=10, N=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in range(n):
        seq = torch.randint(1, N, (B, L), generator=g, dtype=torch.int32)
        seq[::3, : 2 + s % 4] = 0
        out.append(dict(item_seq=seq.cuda(), item_id=torch.randint(1, N, (B, 5), generator=g).cuda()))
    return out


def _train(ahead, algo="adam", wd=0.0, swap_at=None, n_steps=14, nan_at=None, table_mode="lazy_dense"):
    """ahead: False = catch-up at the head of the next step (no plan lookahead); "tail" / "late" / "early" = with a prefetched plan: on
    the main stream between the row update and the join of the dense-gradient stream ("early": only the rows in both batches there, the
    others on the plan stream under the step in flight)"""
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.utils.general import get_class_instance, init_seed
    init_seed(4)
    model = get_class_instance("SASRec", "unirec_amd/model")(_cfg())
    opt = SparseDenseAdam(model, lr=5e-3, weight_decay=wd, algo=algo, table_mode=table_mode)
    # "early": rows of the next batch that the step in flight does not touch are replayed on the PLAN stream under that step (what a
    # plan of >= 65 536 ids gets by itself); "tail" / "late": the whole replay at the tail of step(), main stream
    opt._early_catchup = ahead == "early"
    model.train()
    bs = _batches(n_steps + 2)
    other = _batches(3, seed=99)
    lab = torch.zeros(16, 5, dtype=torch.int32, device="cuda:0")
    lab[:, 0] = 1
    losses = []
    for s in range(n_steps):
        b = bs[s]
        opt.zero_grad()
        opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
        nxt = other[0] if swap_at == s else bs[s + 1]       # swap_at: the batch planned ahead is NOT the one trained on next
        if ahead:
            opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
        loss = model.forward_backward(item_id=b["item_id"], label=lab, item_seq=b["item_seq"])
        if nan_at == s:                       # a NaN loss: the update kernels read the guard and skip the step
            model.loss_guard.fill_(-1.0)
        # "late": the loop says another step follows (as Trainer.fit / bench.py do): the next forward pass joins the dense half
        opt.step(late_join=(ahead in ("late", "early") and s + 1 < n_steps))
        losses.append(float(loss))
    opt.flush()
    torch.cuda.synchronize()
    st = opt.tables["item_embedding"]
    return losses, st["w"].clone(), st["m"].clone(), st["v"].clone(), model.dense_flat.data.clone()


@pytest.mark.parametrize("algo,wd", [("adam", 0.0), ("adamw", 0.01), ("adam", 0.001), ("rmsprop", 0.0)])
def test_catchup_ahead_is_bit_identical(algo, wd):
    b = _train(False, algo, wd)
    for mode in ("tail", "late", "early"):
        a = _train(mode, algo, wd)
        assert a[0] == b[0]
        for x, y, what in zip(a[1:], b[1:], ("w", "m", "v", "dense")):
            assert torch.equal(x, y), (mode, what, float((x - y).abs().max()))


def test_a_prefetched_batch_that_is_not_trained_on_changes_nothing():
    """rows caught up for a batch that is then not trained on are simply up to date earlier: the same zero-gradient steps, summed in
    two pieces instead of one (fp32 re-association of the replay sum: a few ulp of an lr-sized term, not bit-equal)"""
    c = _train(False)
    for mode in ("tail", "late", "early"):
        a = _train(mode, swap_at=5)
        for x, z, what in zip(a[1:4], c[1:4], ("w", "m", "v")):
            assert torch.allclose(x, z, rtol=1e-4, atol=1e-6), (mode, what, float((x - z).abs().max()))


@pytest.mark.gpu
def test_dense_half_on_the_side_stream_is_bit_identical():
    """SparseDenseAdam(dense_side=...): the dense half of the optimizer step runs on the encoder's side stream behind the dense-gradient reductions
    (ur_sasrec_side_stream / ur_sasrec_side_publish) and the NEXT forward pass joins it after its first launch ("late", default), or step()
    joins it ("join"), or the main stream waits and runs it itself ("0", round 2a).  Same kernels, same inputs: bit-identical parameters,
    optimizer state and losses after every step; a state_dict() taken right after step() must already see the update (the model joins)."""
    import copy
    import torch
    from unirec_amd import ops
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.sasrec import SASRec

    def run(mode):
        # "per-call": no override -- step(late_join=True) while another step follows, step() for the last one
        torch.manual_seed(7)
        model = SASRec(_cfg(n_items=5000, embedding_size=128, hidden_size=128, inner_size=512, n_heads=16, max_seq_len=50, batch_size=64))
        opt = SparseDenseAdam(model, lr=1e-3, table_mode="lazy_dense", dense_side=None if mode == "per-call" else mode)
        model.train()
        g = torch.Generator(device="cpu").manual_seed(11)
        batches = []
        for _ in range(13):
            seq = torch.randint(1, 5000, (64, 50), generator=g, dtype=torch.int32)
            pad = torch.randint(0, 40, (64,), generator=g)
            for b in range(64):
                seq[b, :pad[b]] = 0
            batches.append(dict(item_seq=seq.cuda(), item_id=torch.randint(1, 5000, (64, 5), generator=g).cuda(),
                                label=torch.tensor([1, 0, 0, 0, 0], dtype=torch.int32).repeat(64, 1).cuda()))
        losses, snaps = [], []
        for k in range(12):
            b, nxt = batches[k], batches[k + 1]
            opt.zero_grad()
            opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
            opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
            losses.append(model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"]))
            opt.step(late_join=k < 11)
            if k in (0, 5):   # read through the public surface right after step(): must be the UPDATED parameters
                snaps.append(copy.deepcopy({n: t.clone() for n, t in model.state_dict().items()}))
        torch.cuda.synchronize()
        assert not ops._side.hold or mode == "late"       # ("per-call": the last step() joined itself)
        return ([float(x) for x in losses], model.dense_flat.data.clone(), opt.dense_m.clone(), opt.dense_v.clone(),
                model.item_embedding.weight.data.clone(), snaps)

    ref = run("0")
    for mode in ("join", "late", "per-call"):
        got = run(mode)
        assert got[0] == ref[0], mode
        for a, b in zip(got[1:5], ref[1:5]):
            assert torch.equal(a, b), mode
        for sa, sb in zip(got[5], ref[5]):
            for n in sb:
                assert torch.equal(sa[n], sb[n]), (mode, n)


@pytest.mark.parametrize("table_mode", ["lazy_dense", "rowwise"])
def test_a_skipped_step_in_the_middle_is_bit_identical_with_the_lookahead(table_mode):
    """a step whose NaN guard skips the update, in the middle of a run with the plan lookahead and the late join: the rows the next batch
    reads take the skipped step as a zero-gradient step -- the trajectory of catching up at the head of the next step, bit for bit"""
    b = _train(False, nan_at=6, table_mode=table_mode)
    for mode in ("late", "early"):
        a = _train(mode, nan_at=6, table_mode=table_mode)
        assert a[0] == b[0]
        for x, y, what in zip(a[1:], b[1:], ("w", "m", "v", "dense")):
            assert torch.equal(x, y), (mode, what, float((x - y).abs().max()))


def test_flush_with_a_plan_ahead_pending_sees_its_replay():
    """flush() (evaluation, checkpoints) right after a step whose lookahead replayed rows on the plan stream: the current stream is ordered
    behind that replay first, and the flushed table is the one of a run without lookahead, bit for bit."""
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.utils.general import get_class_instance, init_seed

    def run(ahead):
        init_seed(4)
        model = get_class_instance("SASRec", "unirec_amd/model")(_cfg())
        opt = SparseDenseAdam(model, lr=5e-3, table_mode="lazy_dense")
        opt._early_catchup = True
        model.train()
        bs = _batches(8)
        lab = torch.zeros(16, 5, dtype=torch.int32, device="cuda:0")
        lab[:, 0] = 1
        for s in range(6):
            b = bs[s]
            opt.zero_grad()
            opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
            if ahead:
                opt.prefetch_plan(item_seq=bs[s + 1]["item_seq"], item_id=bs[s + 1]["item_id"])
            model.forward_backward(item_id=b["item_id"], label=lab, item_seq=b["item_seq"])
            opt.step()
        opt.flush()                      # (a plan for batch 6 is pending in the `ahead` run)
        st = opt.tables["item_embedding"]
        out = (st["w"].clone(), st["m"].clone(), st["v"].clone(), st["last"].clone())
        torch.cuda.synchronize()
        return out

    for x, y in zip(run(True), run(False)):
        assert torch.equal(x, y)
