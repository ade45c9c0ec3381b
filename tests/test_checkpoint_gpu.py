"""Checkpoint round trip and API-misuse guards (round-1 advisor findings).

* ``fit(save_model=True)`` then ``evaluate(load_best_model=True)`` (the reference's standard flow, unirec/facility/trainer.py:
  303-305, 389-412) must score exactly the checkpointed weights: the lazily-updated table may not replay pending momentum on top of
  a freshly loaded checkpoint.
* a second training forward before ``backward()`` overwrites the saved activations: it must fail loudly, while an evaluation
  forward in between is legal (its own workspace) and leaves the gradients untouched.
* a sequence tensor whose length is not ``max_seq_len`` is rejected before any kernel indexes it.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(model="SASRec", **kw):
    from unirec_amd.utils.argument_parser import parse_arguments
    return parse_arguments(dict(dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, model=model, n_users=50, n_items=400, device="cuda:0",
                                     loss_type="bpr", embedding_size=32, hidden_size=32, inner_size=64, n_heads=4, max_seq_len=10, epochs=2,
                                     batch_size=32, seed=9, n_sample_neg_train=4), **kw))


def _batches(n, B=32, L=10, N=400, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in range(n):
        seq = torch.randint(1, N, (B, L), generator=g, dtype=torch.int32)
        seq[::4, : 3 + s % 3] = 0
        lab = torch.zeros(B, 5, dtype=torch.int32)
        lab[:, 0] = 1
        out.append(dict(item_seq=seq.cuda(), item_id=torch.randint(1, N, (B, 5), generator=g).cuda(), label=lab.cuda(),
                        user_id=torch.randint(1, 50, (B,), generator=g).cuda()))
    return out


@pytest.mark.parametrize("table_mode", ["lazy_dense", "rowwise"])
def test_evaluate_with_load_best_model_scores_the_checkpoint(tmp_path, table_mode):
    from unirec_amd.facility.trainer import Trainer
    from unirec_amd.utils.general import get_class_instance, init_seed
    cfg = _cfg(output_path=str(tmp_path), embedding_optimizer=table_mode, early_stop=10)
    init_seed(9)
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    tr = Trainer(cfg, model)
    train, valid = _batches(6), _batches(2, seed=5)
    tr.fit(train, valid_data=valid, save_model=True)
    assert os.path.exists(tr.saved_model_file)
    ck = torch.load(tr.saved_model_file, map_location="cpu", weights_only=False)["state_dict"]
    # the in-memory weights have moved on since the checkpoint (one more epoch of steps) ...
    assert not torch.equal(model.item_embedding.weight.detach().cpu(), ck["item_embedding.weight"])
    res = tr.evaluate(valid, load_best_model=True)
    # ... and after the reference's standard `evaluate(load_best_model=True)` they ARE the checkpoint, bit for bit
    for k, v in model.state_dict().items():
        assert torch.equal(v.detach().cpu(), ck[k]), k
    # the metrics are those of the checkpointed weights: a fresh model that only loads the file scores the same
    init_seed(9)
    m2 = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    tr2 = Trainer(cfg, m2)
    tr2.load_model(tr.saved_model_file)
    res2 = tr2.evaluate(valid, load_best_model=False)
    assert res == res2
    # and the rows stay put when training continues for a step and flushes again: nothing stale is pending
    before = model.item_embedding.weight.detach().clone()
    tr.optimizer.flush()
    assert torch.equal(before, model.item_embedding.weight.detach())


def test_second_training_forward_before_backward_is_rejected():
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.utils.general import get_class_instance, init_seed
    for name in ("SASRec", "GRU"):
        cfg = _cfg(name)
        init_seed(3)
        model = get_class_instance(name, "unirec_amd/model")(cfg)
        opt = SparseDenseAdam(model, lr=1e-3)
        model.train()
        b1, b2 = _batches(2)
        kw = lambda b: {k: b[k] for k in ("item_id", "label", "item_seq")}   # noqa: E731
        opt.zero_grad()
        loss1, _, _, _ = model(**kw(b1))
        loss2, _, _, _ = model(**kw(b2))          # overwrites the activations loss1's backward needs
        with pytest.raises(RuntimeError, match="workspace"):
            loss1.backward()
        # an evaluation forward between a training forward and its backward is fine and changes nothing
        opt.zero_grad()
        la, _, _, _ = model(**kw(b1))
        la.backward()
        ga = model.dense_flat.grad.clone()
        opt.zero_grad()
        lb, _, _, _ = model(**kw(b1))
        model.eval()
        with torch.no_grad():
            model.forward_user_emb(item_seq=b2["item_seq"])
        model.train()
        lb.backward()
        assert torch.equal(ga, model.dense_flat.grad)


def test_wrong_sequence_length_is_rejected():
    from unirec_amd import _lib, ops
    from unirec_amd.utils.general import get_class_instance, init_seed
    for name in ("SASRec", "GRU"):
        cfg = _cfg(name)
        init_seed(3)
        model = get_class_instance(name, "unirec_amd/model")(cfg)
        model.eval()
        short = torch.randint(1, 400, (8, 7), dtype=torch.int32).cuda()
        with pytest.raises(ValueError):
            model.forward_user_emb(item_seq=short)
        model.train()
        with pytest.raises(ValueError):
            model.forward_backward(item_id=torch.ones(8, 5, dtype=torch.int64).cuda(), label=None, item_seq=short)
    gcfg = ops.gru_cfg(8, 10, 32, 32)
    g = get_class_instance("GRU", "unirec_amd/model")(_cfg("GRU"))
    with pytest.raises(_lib.UnirecAmdError):
        ops.gru_fwd(gcfg, g.item_embedding.weight.data, g.dense_flat.data, short, ops.gru_workspace(gcfg, "cuda:0"))


def test_rank_metrics_never_see_a_negative_rank():
    from unirec_amd.facility.trainer import Trainer
    m = Trainer._metrics_from_rank(np.array([-1, 0, 3]), 100)
    assert np.isfinite(list(m.values())).all() and m["mrr"] == pytest.approx((1 + 1 + 0.25) / 3)
