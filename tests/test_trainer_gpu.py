"""End-to-end: datasets + native row builder + Trainer.fit on the GPU vs the oracle stepping the SAME batches
with the reference's semantics (dense embedding gradient, dense Adam).  SURVEY.md 8c G10."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(model_name, loss):
    from unirec_amd.data.dataset.basedataset import BaseDataset
    from unirec_amd.data.dataset.seqrecdataset import SeqRecDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.data.transform.adduserhistory import AddUserHistory
    from unirec_amd.utils.argument_parser import parse_arguments
    rng = np.random.default_rng(3)
    n_users, n_items = 120, 1017     # ML-100K-shaped catalogue (config #1)
    u2h = np.empty(n_users, dtype=object)
    u2h[0] = None
    for u in range(1, n_users):
        u2h[u] = rng.integers(1, n_items, rng.integers(3, 40)).astype(np.int32)
    users = rng.integers(1, n_users, 640)
    data = np.stack([users, [int(rng.choice(u2h[u])) for u in users]], 1)
    cfg = parse_arguments(dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, model=model_name, n_users=n_users, n_items=n_items, device="cuda:0", loss_type=loss, embedding_size=32,
                               hidden_size=32, inner_size=64, n_heads=4, max_seq_len=12, epochs=1, batch_size=64, seed=5,
                               n_sample_neg_train=4, history_mask_mode="autoregressive", user_sequence_alpha=0.5, asymmetric=True,
                               **({"has_user_emb": True} if model_name == "SVDPlusPlus" else {}),
                               **(dict(conv_size=5, padding_mode="reflect", n_layers=2, layer_norm_eps=1e-9, seq_merge=model_name == "ConvFormer",
                                       seq_decay=-0.3, init_ratio=0.05, hidden_act="gelu") if "ConvFormer" in model_name else {})))
    neg = AddNegSamples(n_users, n_items, 4, user2history=u2h, seed=5)
    if model_name == "MF":
        ds = BaseDataset(cfg, transform=neg, data=data)
    else:
        ds = SeqRecDataset(cfg, transform=neg, data=data)
        ds.add_user_history_transform(AddUserHistory(u2h, "autoregressive", seq_last=0))
    return cfg, ds


@pytest.mark.parametrize("model_name,loss", [("SASRec", "bpr"), ("SASRec", "softmax"), ("MF", "bpr"), ("GRU", "softmax"),
                                             ("AvgHist", "bpr"), ("SVDPlusPlus", "softmax"), ("AttHist", "softmax"),
                                             ("ConvFormer", "softmax"), ("FASTConvFormer", "bpr")])
def test_fit_losses_follow_the_oracle(model_name, loss):
    from oracle import model_ref
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.general import get_class_instance, init_seed
    cfg, ds = _setup(model_name, loss)
    init_seed(cfg["seed"])
    model = get_class_instance(model_name, "unirec_amd/model")(cfg)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = Trainer(cfg, model)
    loader = BatchLoader(ds, cfg["batch_size"], device="cuda:0")
    loader.loss_check_draws = int(loss in ("bpr", "ccl"))              # (what fit() sets: the reference's per-step random.random() of those losses)
    batches = [{k: v.cpu() for k, v in b.items()} for b in loader]     # the builder's stream is consumed here ...
    cfg2, ds2 = _setup(model_name, loss)                                # ... so rebuild an identical one for training
    tr.fit(BatchLoader(ds2, cfg["batch_size"], device="cuda:0"))
    assert len(tr.step_losses) == len(batches) == 10
    state = {}
    ref = [model_ref.train_step(P, state, b, cfg, lr=cfg["learning_rate"]) for b in batches]
    np.testing.assert_allclose(tr.step_losses, ref, rtol=2e-4)
    tr.optimizer.flush()
    for k, v in model.state_dict().items():
        if k.endswith("key.bias"):
            continue
        if k == "item_src_embedding.weight":      # an alias of item_embedding in the model; an independent stale copy in P
            continue
        np.testing.assert_allclose(v.cpu().numpy(), P[k].numpy(), rtol=1e-3, atol=1e-4, err_msg=k)
    res = tr.evaluate(BatchLoader(ds2, 64, device="cuda:0"), load_best_model=False)
    assert 0.0 <= res["hit@5"] <= 1.0 and 0.0 < res["mrr"] <= 1.0


def test_prepared_dataset_directory_trains_and_evaluates():
    """SURVEY.md 8 f3 end to end: the reference's on-disk files (tests/golden/g12_dataset) -> datasets -> Trainer.fit ->
    one_vs_all evaluation with the history loaded from user_history.pkl."""
    import os
    from conftest import GOLDEN
    from unirec_amd.data.dataset.seqrecdataset import SeqRecDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.data.transform.adduserhistory import AddUserHistory
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.file_io import load_data_info
    from unirec_amd.utils.general import get_class_instance, init_seed, load_user_history
    ddir = os.path.join(GOLDEN, "g12_dataset")
    info = load_data_info(ddir)
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format=info["user_history_file_format"])
    cfg = parse_arguments(dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, model="SASRec", n_users=info["n_users"], n_items=info["n_items"], device="cuda:0", loss_type="softmax",
                               embedding_size=32, hidden_size=32, inner_size=64, n_heads=4, max_seq_len=8, epochs=2, batch_size=64,
                               seed=3, n_sample_neg_train=4, history_mask_mode="autoregressive"))
    init_seed(3)

    def dataset(name):
        ds = SeqRecDataset(cfg, path=ddir, filename=name, transform=AddNegSamples(info["n_users"], info["n_items"], 4, user2history=u2h, seed=3))
        ds.add_user_history_transform(AddUserHistory(u2h, "autoregressive", seq_last=0))
        return ds
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    tr = Trainer(cfg, model)
    tr.fit(BatchLoader(dataset("train"), 64, device="cuda:0"), save_model=False)
    assert len(tr.step_losses) == 2 * 5 and np.isfinite(tr.step_losses).all()
    assert np.mean(tr.step_losses[5:]) < np.mean(tr.step_losses[:5])           # it learns something
    tr.set_user_history(u2h)
    tr.reset_evaluator("user-item", "one_vs_all")
    res = tr.evaluate(BatchLoader(dataset("valid"), 64, device="cuda:0"), load_best_model=False)
    assert 0.0 < res["mrr"] <= 1.0 and 0.0 <= res["hit@10"] <= 1.0 and 0.0 < res["group_auc"] <= 1.0


def test_fit_with_the_device_resident_input_pipeline():
    """interaction pairs + CSR history in HBM, batches built on the device; the model must train on them
    exactly as on host-built rows (same kernels downstream), here checked by the loss going down and by the batches
    being well formed (positive in column 0, no negative from the user's history, left-padded sequences)."""
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR
    from unirec_amd.facility.trainer import DeviceBatchLoader, Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.general import get_class_instance, init_seed
    rng = np.random.default_rng(4)
    n_users, n_items, L, K = 200, 500, 10, 4
    u2h = np.empty(n_users, dtype=object)
    for u in range(n_users):
        u2h[u] = rng.integers(1, n_items, rng.integers(2, 30)).astype(np.int32) if u else None
    users = rng.integers(1, n_users, 1500)
    pairs = np.stack([users, [int(rng.choice(u2h[u])) for u in users]], 1)
    csr = HistoryCSR(u2h)
    cfg = parse_arguments(dict(hidden_dropout_prob=0.2, attn_dropout_prob=0.2, model="SASRec", n_users=n_users, n_items=n_items, device="cuda:0", loss_type="softmax", embedding_size=32,
                               hidden_size=32, inner_size=64, n_heads=4, max_seq_len=L, epochs=3, batch_size=128, seed=6))
    init_seed(6)
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    bld = DeviceRowBuilder(n_users, n_items, K, L, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=6)
    loader = DeviceBatchLoader(pairs, bld, 128, shuffle=True, seed=6)
    first = next(iter(loader))
    assert first["item_id"].shape == (128, K + 1) and first["item_seq"].shape == (128, L) and first["item_seq"].dtype == torch.int32
    ids, seq, uid = first["item_id"].cpu().numpy(), first["item_seq"].cpu().numpy(), first["user_id"].cpu().numpy()
    for b in range(128):
        assert not (set(ids[b, 1:].tolist()) - {0}) & set(u2h[uid[b]].tolist())      # negatives avoid the history
        nz = np.flatnonzero(seq[b])
        assert len(nz) == 0 or (nz == np.arange(L - len(nz), L)).all()               # left padded
    # the builds run two batches ahead on the loader's own stream: every batch == what an in-line build of the same (epoch, index) gives
    loader.epoch = 0
    ahead = [{k: v.clone() for k, v in b.items()} for b in loader]
    torch.cuda.synchronize()
    g = torch.Generator(device="cuda:0").manual_seed(6 + 0)
    order = torch.randperm(len(pairs), generator=g, device="cuda:0")
    assert len(ahead) == len(loader) == (len(pairs) + 127) // 128
    for k, b in enumerate(ahead):
        ref = loader._build(order, k, 0)
        assert all(torch.equal(b[key], ref[key]) for key in ref), k
    loader.epoch = 0
    tr = Trainer(cfg, model)
    tr.fit(loader, save_model=False)
    per_epoch = np.array(tr.step_losses).reshape(3, -1).mean(1)
    assert np.isfinite(per_epoch).all() and per_epoch[2] < per_epoch[0]


def test_fit_over_the_lookahead_loader_equals_fit_over_prebuilt_batches():
    """Trainer.fit puts the loader on the optimizer's plan stream with the `joined` promise (no wait / event packets of the hand-out in the
    main queue, batches released by stream order instead of Tensor.record_stream): the trajectory must be the one of the same batches
    built in line beforehand -- bit for bit, over enough steps that a batch recycled early or read before its build would show."""
    from unirec_amd.data.rows import DeviceRowBuilder, HistoryCSR
    from unirec_amd.facility.trainer import DeviceBatchLoader, Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.general import get_class_instance, init_seed
    rng = np.random.default_rng(5)
    n_users, n_items, L, K, B = 3000, 20000, 20, 4, 256
    u2h = np.empty(n_users, dtype=object)
    for u in range(n_users):
        u2h[u] = rng.integers(1, n_items, rng.integers(2, 60)).astype(np.int32) if u else None
    users = rng.integers(1, n_users, B * 40)
    pairs = np.stack([users, [int(u2h[u][-1]) for u in users]], 1)
    csr = HistoryCSR(u2h)
    cfg = parse_arguments(dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, model="SASRec", n_users=n_users, n_items=n_items, device="cuda:0", loss_type="softmax",
                               embedding_size=64, hidden_size=64, inner_size=128, n_heads=4, max_seq_len=L, epochs=2, batch_size=B, seed=6))

    def run(prebuilt):
        init_seed(6)
        model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
        bld = DeviceRowBuilder(n_users, n_items, K, L, csr, reject_history=True, mask_mode="autoregressive", seq_last=0, seed=6)
        loader = DeviceBatchLoader(pairs, bld, B, shuffle=True, seed=6)
        tr = Trainer(cfg, model)
        if prebuilt:
            class Epochs:       # the same (epoch, index) builds, made in line on the main stream and synchronised
                def __init__(self):
                    self.e = 0

                def __iter__(self):
                    g = torch.Generator(device="cuda:0").manual_seed(6 + self.e)
                    order = torch.randperm(len(pairs), generator=g, device="cuda:0")
                    nb = len(loader)
                    out = [loader._build(order, k, self.e * nb) for k in range(nb)]
                    torch.cuda.synchronize()
                    self.e += 1
                    return iter(out)
            tr.fit(Epochs(), save_model=False)
        else:
            own, seen, step = loader.stream, [], tr.train_step

            def watched(cur, nxt=None):
                seen.append(loader.joined and loader.stream is tr.optimizer.plan_stream())
                return step(cur, nxt)
            tr.train_step = watched
            tr.fit(loader, save_model=False)
            assert all(seen) and len(seen) == 80
            # ... and only there: after fit() the loader is back on its own stream with the waiting hand-out, so a consumer
            # that makes no promise (evaluate(), a user loop) reads finished batches -- epoch 2's, equal to the in-line builds
            assert loader.joined is False and loader.stream is own
            after = [{k: v.clone() for k, v in b.items()} for b in loader]
            torch.cuda.synchronize()
            order = torch.randperm(len(pairs), generator=torch.Generator(device="cuda:0").manual_seed(6 + 2), device="cuda:0")
            assert len(after) == len(loader)
            for k, b in enumerate(after):
                ref = loader._build(order, k, 2 * len(loader))
                assert all(torch.equal(b[key], ref[key]) for key in ref), k
        torch.cuda.synchronize()
        return list(tr.step_losses), {k: v.detach().clone() for k, v in model.state_dict().items()}

    la, sa = run(False)
    lb, sb = run(True)
    assert len(la) == 80 and la == lb
    for k in sb:
        assert torch.equal(sa[k], sb[k]), k


def test_fit_fullsoftmax_follows_the_oracle():
    """fullsoftmax end to end: dense table gradient + dense Adam on the table, encoder rows folded in, vs the oracle."""
    from oracle import model_ref
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.general import get_class_instance, init_seed
    cfg, ds = _setup("SASRec", "fullsoftmax")
    init_seed(cfg["seed"])
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = Trainer(cfg, model)
    batches = [{k: v.cpu() for k, v in b.items()} for b in BatchLoader(ds, cfg["batch_size"], device="cuda:0")]
    cfg2, ds2 = _setup("SASRec", "fullsoftmax")
    tr.fit(BatchLoader(ds2, cfg["batch_size"], device="cuda:0"))
    state = {}
    ref = [model_ref.train_step(P, state, dict(b, item_id=b["item_id"][:, 0]), cfg, lr=cfg["learning_rate"]) for b in batches]
    np.testing.assert_allclose(tr.step_losses, ref, rtol=2e-4)
    for k, v in model.state_dict().items():
        if k.endswith("key.bias"):
            continue
        np.testing.assert_allclose(v.cpu().numpy(), P[k].numpy(), rtol=1e-3, atol=1e-4, err_msg=k)


def test_a_step_with_nan_loss_is_skipped_without_a_host_sync():
    """Trainer.fit's NaN check (unirec/facility/trainer.py:164-168,343-350): the update of a step whose loss is NaN is not applied.
    Here the check is a device flag written by the loss kernel and read by the update kernels."""
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.sasrec import SASRec
    dev = torch.device("cuda:0")
    for clip in (None, 1.0):
        cfg = dict(model="SASRec", n_users=10, n_items=500, device="cuda:0", loss_type="softmax", embedding_size=32, hidden_size=32,
                   dropout_prob=0.0, init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, distance_type="dot", tau=1.0,
                   train_file_format="user-item", exp_name="t", n_layers=2, n_heads=4, inner_size=64, hidden_dropout_prob=0.0,
                   attn_dropout_prob=0.0, hidden_act="gelu", layer_norm_eps=1e-10, max_seq_len=8, use_position_emb=True)
        torch.manual_seed(0)
        m = SASRec(cfg)
        opt = SparseDenseAdam(m, lr=1e-2, grad_clip=clip)
        g = torch.Generator().manual_seed(1)
        seq = torch.randint(1, 500, (16, 8), generator=g, dtype=torch.int32).to(dev)
        ids = torch.randint(1, 500, (16, 5), generator=g).to(dev)
        lab = torch.zeros(16, 5, dtype=torch.int32, device=dev)
        lab[:, 0] = 1
        m.train()

        def step():
            opt.zero_grad()
            opt.plan_batch(item_seq=seq, item_id=ids)
            loss = m.forward_backward(item_id=ids, label=lab, item_seq=seq)
            opt.step()
            return loss
        step()                                                          # a normal step moves the weights
        w1, t1 = m.dense_flat.data.clone(), m.item_embedding.weight.data.clone()
        saved = m.item_embedding.weight.data[int(ids[0, 0])].clone()
        m.item_embedding.weight.data[int(ids[0, 0])] = float("nan")     # poison one candidate row -> NaN loss
        loss = step()
        assert torch.isnan(loss)
        m.item_embedding.weight.data[int(ids[0, 0])] = saved
        assert torch.equal(m.dense_flat.data, w1) and torch.equal(m.item_embedding.weight.data, t1)   # nothing moved
        loss = step()                                                   # and training goes on
        assert torch.isfinite(loss) and not torch.equal(m.dense_flat.data, w1)


def test_deferred_dense_join_gives_the_same_steps_as_the_eager_join():
    """SparseDenseAdam (no clipping) lets the SASRec backward return while the dense-gradient reductions still run on the side
    stream (ur_sasrec_bwd_deferred / ur_sasrec_bwd_join): same kernels, same order of every sum -> bit-identical weights; and the
    dense gradient is not visible (None, not stale) before the join."""
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.facility.trainer import BatchLoader
    from unirec_amd.utils.general import get_class_instance, init_seed
    finals = []
    for overlap in (True, False):
        cfg, ds = _setup("SASRec", "bpr")
        init_seed(cfg["seed"])
        model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
        model.train()
        opt = SparseDenseAdam(model, lr=1e-2, table_mode="rowwise", overlap_dense_join=overlap)
        assert model.defer_dense_join == overlap
        for i, b in enumerate(BatchLoader(ds, cfg["batch_size"], device="cuda:0")):
            model.forward_backward(**b)
            if overlap:
                assert model.dense_flat.grad is None and model._deferred_dense_grad is not None
                if i == 1:   # a reader that is not the optimizer joins explicitly
                    model.finish_backward()
                    assert model.dense_flat.grad is not None and bool(torch.isfinite(model.dense_flat.grad).all())
            else:
                assert model.dense_flat.grad is not None
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        finals.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
    # clipping needs the global norm first: the optimizer then asks for complete gradients
    cfg, ds = _setup("SASRec", "bpr")
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    SparseDenseAdam(model, lr=1e-2, table_mode="rowwise", grad_clip=1.0)
    assert model.defer_dense_join is False


def test_fit_reproduces_the_references_own_trainer_run():
    """per-step losses and final parameters of the reference's own Trainer.fit (unirec/facility/trainer.py:
    234-357; 2 epochs over tests/golden/g12_dataset, Adam, grad_clip_value 0.5, negatives drawn by the reference's sampler stream),
    captured by tools/capture_goldens.py.  Here: the same files -> SeqRecDataset -> BatchLoader -> Trainer.fit on the GPU, from the
    reference's initial state_dict."""
    import os
    from conftest import GOLDEN, load_golden
    from unirec_amd.data.dataset.seqrecdataset import SeqRecDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.data.transform.adduserhistory import AddUserHistory
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.file_io import load_data_info
    from unirec_amd.utils.general import get_class_instance, load_user_history
    gcfg, groups = load_golden("g10_trainer_fit")
    ddir = os.path.join(GOLDEN, "g12_dataset")
    info = load_data_info(ddir)
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format=info["user_history_file_format"])
    keys = ("n_heads", "n_layers", "inner_size", "embedding_size", "hidden_size", "max_seq_len", "hidden_act", "loss_type", "layer_norm_eps",
            "use_position_emb", "init_std", "tau", "learning_rate", "grad_clip_value", "epochs", "batch_size", "weight_decay", "optimizer")
    cfg = parse_arguments(dict({k: (gcfg[k].item() if hasattr(gcfg[k], "item") else gcfg[k]) for k in keys}, model="SASRec",
                               n_users=info["n_users"], n_items=info["n_items"], device="cuda:0", hidden_dropout_prob=0.0, attn_dropout_prob=0.0,
                               n_sample_neg_train=4, history_mask_mode="autoregressive", seed=21, early_stop=0))
    model = get_class_instance("SASRec", "unirec_amd/model")(cfg)
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in groups["sd0"].items()}, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.check_views()
    ds = SeqRecDataset(cfg, path=ddir, filename="train", transform=AddNegSamples(info["n_users"], info["n_items"], 4, user2history=u2h, seed=21))
    ds.add_user_history_transform(AddUserHistory(u2h, "autoregressive", seq_last=0))
    tr = Trainer(cfg, model)
    tr.fit(BatchLoader(ds, int(cfg["batch_size"]), device="cuda:0"), save_model=False)
    want = groups["step_losses"][""] if "" in groups.get("step_losses", {}) else np.load(os.path.join(GOLDEN, "g10_trainer_fit.npz"))["step_losses"]
    assert len(tr.step_losses) == len(want) == 10
    np.testing.assert_allclose(tr.step_losses, want, rtol=1e-4)
    tr.optimizer.flush()
    for k, v in model.state_dict().items():
        if k.endswith("key.bias"):
            continue
        np.testing.assert_allclose(v.cpu().numpy(), groups["sd1"][k], rtol=1e-3, atol=5e-4, err_msg=k)   # atol = 1/4 of ONE lr-sized Adam step, after 10 steps that move a weight by up to 2e-2 (Adam turns the rounding noise of a near-zero gradient into a sign-sized move: 1 element in 1024 differs by 3e-4)


def test_fit_reproduces_the_references_own_gru_run():
    """G10 for GRU: the reference's Trainer.fit over tests/golden/g12_dataset with the GRU encoder (torch nn.GRU,
    unirec/model/sequential/gru.py:13-35), BPR, clipping 0.5, 2 epochs -- per-step losses and final parameters."""
    import os
    from conftest import GOLDEN, load_golden
    from unirec_amd.data.dataset.seqrecdataset import SeqRecDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.data.transform.adduserhistory import AddUserHistory
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.file_io import load_data_info
    from unirec_amd.utils.general import get_class_instance, load_user_history
    gcfg, groups = load_golden("g10_trainer_fit_gru")
    ddir = os.path.join(GOLDEN, "g12_dataset")
    info = load_data_info(ddir)
    u2h, _ = load_user_history(ddir, "user_history", n_users=info["n_users"], format=info["user_history_file_format"])
    keys = ("n_layers", "embedding_size", "hidden_size", "max_seq_len", "loss_type", "init_std", "tau", "learning_rate", "grad_clip_value",
            "epochs", "batch_size", "weight_decay", "optimizer", "dropout_prob")
    cfg = parse_arguments(dict({k: (gcfg[k].item() if hasattr(gcfg[k], "item") else gcfg[k]) for k in keys}, model="GRU",
                               n_users=info["n_users"], n_items=info["n_items"], device="cuda:0", n_sample_neg_train=4,
                               history_mask_mode="autoregressive", seed=23, early_stop=0))
    model = get_class_instance("GRU", "unirec_amd/model")(cfg)
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in groups["sd0"].items()}, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys
    ds = SeqRecDataset(cfg, path=ddir, filename="train", transform=AddNegSamples(info["n_users"], info["n_items"], 4, user2history=u2h, seed=23))
    ds.add_user_history_transform(AddUserHistory(u2h, "autoregressive", seq_last=0))
    tr = Trainer(cfg, model)
    tr.fit(BatchLoader(ds, int(cfg["batch_size"]), device="cuda:0"), save_model=False)
    want = np.load(os.path.join(GOLDEN, "g10_trainer_fit_gru.npz"))["step_losses"]
    assert len(tr.step_losses) == len(want) == 10
    np.testing.assert_allclose(tr.step_losses, want, rtol=1e-4)
    tr.optimizer.flush()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), groups["sd1"][k], rtol=1e-3, atol=5e-4, err_msg=k)


def test_fit_reproduces_the_references_mf_run_at_the_c1_shape(tmp_path):
    """BASELINE configs[0] (C1) at its NAMED shape against the reference itself: MF + BPR, 943 users x 1 682 items, d = 64
    (unirec/config/model/MF.yaml:2), batch 400, 100 000 ML-100K-shaped synthetic interactions = 250 steps of the reference's own
    Trainer.fit (golden g10_trainer_fit_mf_c1; the data set is regenerated here by the generator the capture used)."""
    import os
    import ml100k_shaped
    from conftest import GOLDEN, load_golden
    from unirec_amd.data.dataset.basedataset import BaseDataset
    from unirec_amd.data.transform.addnegsamples import AddNegSamples
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.general import get_class_instance, load_user_history
    gcfg, groups = load_golden("g10_trainer_fit_mf_c1")
    ddir = ml100k_shaped.write(str(tmp_path / "ml100k_shaped"))
    n_users, n_items = ml100k_shaped.N_USERS, ml100k_shaped.N_ITEMS
    assert (int(gcfg["n_users"]), int(gcfg["n_items"]), int(gcfg["embedding_size"]), int(gcfg["batch_size"])) == (n_users, n_items, 64, 400)
    u2h, _ = load_user_history(ddir, "user_history", n_users=n_users, format="user-item")
    keys = ("embedding_size", "hidden_size", "loss_type", "tau", "learning_rate", "grad_clip_value", "epochs", "batch_size", "weight_decay", "optimizer",
            "has_user_emb")
    cfg = parse_arguments(dict({k: (gcfg[k].item() if hasattr(gcfg[k], "item") else gcfg[k]) for k in keys}, model="MF", n_users=n_users,
                               n_items=n_items, device="cuda:0", n_sample_neg_train=4, seed=25, early_stop=0))
    model = get_class_instance("MF", "unirec_amd/model")(cfg)
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in ml100k_shaped.initial_state().items()}, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys
    ds = BaseDataset(cfg, path=ddir, filename="train", transform=AddNegSamples(n_users, n_items, 4, user2history=u2h, seed=25))
    tr = Trainer(cfg, model)
    tr.fit(BatchLoader(ds, 400, device="cuda:0"), save_model=False)
    want = np.load(os.path.join(GOLDEN, "g10_trainer_fit_mf_c1.npz"))["step_losses"]
    assert len(tr.step_losses) == len(want) == 250
    np.testing.assert_allclose(tr.step_losses, want, rtol=1e-4)
    tr.optimizer.flush()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy()[::8], groups["sd1_every8"][k], rtol=1e-3, atol=5e-4, err_msg=k)

