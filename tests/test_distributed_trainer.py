"""a14: `Trainer(config, model)` under torch.distributed trains ONE model.

W ranks x batch B must equal 1 rank x the concatenated batch of W*B rows (DDP's mean of equal-sized rank means == the mean over
the concatenated batch): per-step losses, and every parameter after several steps with gradient clipping on -- for SASRec, GRU
(BASELINE config C4's encoder) and MF (user + item table) -- plus the checkpoint round trip across world sizes: a checkpoint
written by 1 rank loads into W ranks (`scatter_state_dict`), one written by W ranks (`gather_state_dict`, shards streamed to
rank 0) loads into 1, and evaluation over the sharded tables (sampled and full-item protocol) gives the single-GPU metrics.
The ranks share cuda:0 and talk over gloo (rows staged through the host): the routing, not RCCL, is under test here; the
RCCL path has its own self-check (bench.py --gpus N).  BatchLoader's rank sharding is checked on the CPU."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_batch_loader_gives_every_rank_the_same_number_of_batches():
    from unirec_amd.facility.trainer import BatchLoader

    class DS:
        def __len__(self):
            return 23

        def get_batch(self, idx):
            return {"i": np.asarray(idx, dtype=np.int64)}

    for W in (1, 2, 3, 4):
        seen = []
        for r in range(W):
            ld = BatchLoader(DS(), 4, rank=r, world=W, device="cpu")
            got = [b["i"].numpy() for b in ld]
            assert len(got) == len(ld) == -(-6 // W)
            seen.append(got)
        order = [seen[r][k] for k in range(len(seen[0])) for r in range(W)]
        flat = np.concatenate(order)
        assert np.array_equal(flat[:23], np.arange(23))          # dataset order = (step, rank) order, nothing missing
        assert set(flat[23:].tolist()) <= set(range(23))         # the wrap-around repeats the first batches


N_ITEMS, N_USERS, L, G = 2003, 41, 12, 5
# "identical loss (1e-6 rel)".  The first step's loss IS that close (same parameters; the mean of W equal-sized rank means
# against one mean over W * B rows).  From the second step on the parameters differ by what Adam makes of fp32 summation order -- the
# first update of an element is lr * sign(g) whatever |g|, and a dense gradient summed as W partial sums + an all-reduce differs from
# the one-GEMM sum in the last bits, which decides the sign of the elements whose gradient is ~0 -- so later losses agree to ~1e-5.
LOSS_RTOL_FIRST, LOSS_RTOL = 1e-6, 2e-5


def _cfg(kind, **kw):
    from unirec_amd.utils.argument_parser import parse_arguments
    base = dict(hidden_dropout_prob=0.0, attn_dropout_prob=0.0, dropout_prob=0.0, model=kind, n_users=N_USERS, n_items=N_ITEMS, device="cuda:0",
                loss_type="softmax", embedding_size=32, hidden_size=32, inner_size=64, n_heads=4, n_layers=2, max_seq_len=L, epochs=1,
                batch_size=16, seed=11, n_sample_neg_train=G - 1, grad_clip_value=0.05, learning_rate=2e-3, early_stop=0)
    if kind == "MF":
        base.update(has_user_emb=True, has_user_bias=True, has_item_bias=True, loss_type="bpr")
    base.update(kw)
    return parse_arguments(base)


def _batches(n_steps, B, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in range(n_steps):
        seq = torch.randint(1, N_ITEMS, (B, L), generator=g, dtype=torch.int32)
        seq[::3, : 2 + s] = 0
        seq[1] = 0                                   # an all-padding history
        lab = torch.zeros(B, G, dtype=torch.int32)
        lab[:, 0] = 1
        ids = torch.randint(1, N_ITEMS, (B, G), generator=g)
        ids[0, 1] = ids[0, 0]                        # duplicates inside a row and across ranks
        ids[B - 1, 0] = ids[0, 0]
        out.append(dict(item_seq=seq, item_id=ids, label=lab, user_id=torch.randint(1, N_USERS, (B,), generator=g)))
    return out


def _to(b, dev, lo=None, hi=None):
    return {k: v[lo:hi].to(dev).contiguous() for k, v in b.items()}


class _TdComm:
    """how a parity body talks to its peers: torch.distributed's default group here, a pgroup.LoopbackGroup in tests/test_loopback_gpu.py"""
    accelerator = None

    @staticmethod
    def barrier():
        dist.barrier()

    @staticmethod
    def all_gather_object(box, obj):
        dist.all_gather_object(box, obj)

    @staticmethod
    def exclusive():
        import contextlib
        return contextlib.nullcontext()      # (one process per rank: nothing is shared)


def _worker(rank, world, port, q, kind, tmp, clip=0.05, loss=None):
    if loss == "serial-rows":       # the row exchange inside the step (rounds 1-3) instead of a step ahead + fix-up
        os.environ["UR_PREFETCH_ROWS"] = "0"
        loss = None
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _parity_body(rank, world, kind, tmp, clip, loss, _TdComm)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


def _parity_body(rank, world, kind, tmp, clip, loss, comm, tweak=None):
    """W ranks x batch B == 1 rank x the concatenated batch, through Trainer.fit / evaluate / checkpoints (module docstring).
    comm: the peers (barrier, all_gather_object, the `accelerator` Trainer gets); tweak(optimizer): test hooks on the W-rank optimizer"""
    from unirec_amd.facility.trainer import Trainer
    from unirec_amd.utils.general import get_class_instance, init_seed
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = _cfg(kind, output_path=tmp, grad_clip_value=clip, **({"loss_type": loss} if loss else {}))
    B = 16
    full = _batches(4, B * world)
    ev = _batches(2, B * world, seed=9)
    single = types.SimpleNamespace(process_index=0, num_processes=1)      # forces the 1-GPU path although a group exists
    ref = {}
    if rank == 0:      # ---- the reference run: ONE rank on the concatenated batches
        init_seed(cfg["seed"])
        m1 = get_class_instance(kind, "unirec_amd/model")(cfg)
        t1 = Trainer(cfg, m1, single)
        assert t1.world == 1
        t1.fit([_to(b, dev) for b in full[:2]], save_model=False)
        t1.save_model(os.path.join(tmp, "w1.pth"))
        ref["eval_k"] = t1.evaluate([_to(b, dev) for b in ev], load_best_model=False)
        if kind != "MF":
            t1.reset_evaluator("user-item", "one_vs_all")
            ref["eval_all"] = t1.evaluate([_to(b, dev) for b in ev], load_best_model=False)
            t1.reset_evaluator("user-item", None)
        t1.fit([_to(b, dev) for b in full[2:]], save_model=False)
        t1.optimizer.flush()
        ref["losses"] = list(t1.step_losses)
        ref["state"] = {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()}
    comm.barrier()
    # ---- W ranks, each on its slice
    with comm.exclusive():      # (rank THREADS share the device's random generator: seed + parameter initialisation one rank at a time)
        init_seed(cfg["seed"])
        m = get_class_instance(kind, "unirec_amd/model")(cfg)
    tr = Trainer(cfg, m, comm.accelerator)
    if tweak is not None:
        tweak(tr.optimizer)
    assert tr.world == world and type(tr.optimizer).__name__ == "ShardedSparseDenseAdam"
    assert m.item_embedding.weight.shape[0] < N_ITEMS                       # the model holds a shard, not the table
    mine = lambda bs: [_to(b, dev, rank * B, (rank + 1) * B) for b in bs]   # noqa: E731
    tr.fit(mine(full[:2]), save_model=False)
    tr.save_model(os.path.join(tmp, "ww.pth"))                              # collective: shards streamed to rank 0
    got_k = tr.evaluate(mine(ev), load_best_model=False)
    got_all = None
    if kind != "MF":
        tr.reset_evaluator("user-item", "one_vs_all")
        got_all = tr.evaluate(mine(ev), load_best_model=False)
        tr.reset_evaluator("user-item", None)
    tr.load_model(os.path.join(tmp, "w1.pth"))                              # a 1-rank checkpoint dealt out to W ranks
    tr.fit(mine(full[2:]), save_model=False)
    losses = [None] * world
    comm.all_gather_object(losses, list(tr.step_losses))
    sd = tr.optimizer.gather_state_dict()
    if rank == 0:
        # every rank returns the SAME value: the mean over the ranks (trainer.py:353 gather_for_metrics(loss).mean())
        assert all(np.array_equal(np.asarray(losses[0]), np.asarray(x)) for x in losses[1:])
        rel = np.abs(np.asarray(losses[0]) - np.asarray(ref["losses"])) / np.abs(np.asarray(ref["losses"]))
        assert rel[0] <= LOSS_RTOL_FIRST and rel.max() <= LOSS_RTOL, (rel.tolist(), losses[0], ref["losses"])
        for k, v in ref["state"].items():
            if k.endswith("key.bias"):
                continue
            np.testing.assert_allclose(sd[k].numpy(), v.numpy(), rtol=1e-4, atol=2e-5, err_msg=k)   # atol = 1% of an lr-sized step
        # the checkpoint written by W ranks holds FULL tables under the reference's names, equal to the 1-rank one
        c1 = torch.load(os.path.join(tmp, "w1.pth"), map_location="cpu", weights_only=False)["state_dict"]
        cw = torch.load(os.path.join(tmp, "ww.pth"), map_location="cpu", weights_only=False)["state_dict"]
        assert set(c1) == set(cw)
        for k in c1:
            assert tuple(c1[k].shape) == tuple(cw[k].shape), k
            if not k.endswith("key.bias"):
                np.testing.assert_allclose(cw[k].numpy(), c1[k].numpy(), rtol=1e-4, atol=2e-5, err_msg=k)
        # ... and loads into ONE rank
        init_seed(cfg["seed"] + 1)
        m2 = get_class_instance(kind, "unirec_amd/model")(cfg)
        t2 = Trainer(cfg, m2, single)
        t2.load_model(os.path.join(tmp, "ww.pth"))
        for k, v in m2.state_dict().items():
            assert torch.equal(v.detach().cpu(), cw[k]), k
        # evaluation over the sharded tables == over the full ones
        for key in ("hit@1", "hit@5", "ndcg@10", "mrr", "group_auc"):
            assert abs(got_k[key] - ref["eval_k"][key]) < 2e-3, (key, got_k[key], ref["eval_k"][key])
            if got_all is not None:
                assert abs(got_all[key] - ref["eval_all"][key]) < 2e-3, (key, got_all[key], ref["eval_all"][key])
    comm.barrier()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,clip,loss", [
    ("SASRec", 2, 0.05, None), ("GRU", 2, 0.05, None), ("MF", 2, 0.05, None), ("SASRec", 3, 0.05, None),
    ("SASRec", 2, 0.0, None), ("GRU", 2, 0.0, None),     # clip 0: the dense half on the encoder's side stream
    # the reference's own DDP test trains exactly this loss (tests/test_model/run_ddp_test.sh:28): every item a candidate, the catalogue
    # row-sharded -- per-shard logsumexp partials, the shard's table gradient final on its owner
    ("SASRec", 2, 0.05, "fullsoftmax"), ("SASRec", 3, 0.0, "fullsoftmax"), ("MF", 2, 0.05, "fullsoftmax"),
    ("SASRec", 2, 0.0, "serial-rows")])
def test_trainer_fit_on_w_ranks_equals_one_rank_on_the_concatenated_batches(kind, world, clip, loss, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind, str(tmp_path), clip, loss)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _overflow_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unirec_amd.facility.distributed import ShardedSparseDenseAdam
        from unirec_amd.utils.general import get_class_instance, init_seed
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = _cfg("SASRec", grad_clip_value=0.0)
        B = 16
        full = _batches(8, B * world)
        mine = [_to(b, dev, rank * B, (rank + 1) * B) for b in full]

        def run(slack):
            init_seed(cfg["seed"])
            m = get_class_instance("SASRec", "unirec_amd/model")(cfg)
            opt = ShardedSparseDenseAdam(m, rank, world, lr=2e-3, cap_slack=slack)
            m.train()
            losses = [opt.train_step(b, mine[i + 1] if i + 1 < len(mine) else None) for i, b in enumerate(mine)]
            m.join_side_updates()
            torch.cuda.synchronize()
            return opt, m, [float(x) for x in losses]

        opt, m, losses = run(0.25)         # a quarter of the expected rows per owner: the first steps overflow on every rank
        assert opt.n_overflow >= 1 and opt._cap_scale >= 2, (opt.n_overflow, opt._cap_scale)
        scales = [None] * world
        dist.all_gather_object(scales, (opt._cap_scale, opt.n_overflow, opt.t))
        assert len(set(scales)) == 1, scales                                   # lockstep: every rank doubled at the same steps
        assert opt.t == len(mine) + opt.n_overflow                             # every overflowed batch was trained again
        assert all(np.isfinite(losses[-3:])), losses
        dense = [None] * world
        dist.all_gather_object(dense, m.dense_flat.data.cpu())
        assert all(torch.equal(dense[0], x) for x in dense[1:])                # the replicas stayed identical through the skips
        ref_opt, _, ref_losses = run(1.5)
        assert ref_opt.n_overflow == 0
        assert abs(losses[-1] - ref_losses[-1]) < 0.05 * abs(ref_losses[-1])   # same batches, slightly different order: same ballpark
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


def _many_steps_worker(rank, world, port, q, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unirec_amd.facility.trainer import Trainer
        from unirec_amd.utils.general import get_class_instance, init_seed
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = _cfg("SASRec", output_path=tmp, grad_clip_value=0.0)
        B, n = 16, 9
        full = _batches(n, B * world)
        ref = None
        if rank == 0:
            init_seed(cfg["seed"])
            t1 = Trainer(cfg, get_class_instance("SASRec", "unirec_amd/model")(cfg), types.SimpleNamespace(process_index=0, num_processes=1))
            t1.fit([_to(b, dev) for b in full], save_model=False)
            ref = list(t1.step_losses)
        dist.barrier()
        init_seed(cfg["seed"])
        tr = Trainer(cfg, get_class_instance("SASRec", "unirec_amd/model")(cfg))
        tr.fit([_to(b, dev, rank * B, (rank + 1) * B) for b in full], save_model=False)
        got = list(tr.step_losses)       # read in drain() at the END of the epoch: every entry must still be its own step's value
        every = [None] * world
        dist.all_gather_object(every, got)
        if rank == 0:
            assert len(got) == n and all(np.array_equal(np.asarray(every[0]), np.asarray(x)) for x in every[1:])
            rel = np.abs(np.asarray(got) - np.asarray(ref)) / np.abs(np.asarray(ref))
            assert rel[0] <= LOSS_RTOL_FIRST and rel.max() <= 5 * LOSS_RTOL, (rel.tolist(), got, ref)
            assert len(set(np.round(got, 6))) == n          # (nine different values, not four repeated)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_step_losses_of_a_long_epoch_are_each_steps_own(tmp_path):
    """Advisor r3: train_step returned a view into a four-entry ring that fit() only read at the end of the epoch.  Nine steps per epoch on
    W = 2: the per-step losses (mean over the ranks) against the 1-rank run on the concatenated batches, every one of the nine."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_steps_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(2)], res


def _second_table_overflow_worker(rank, world, port, q, both=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unirec_amd.facility.distributed import ShardedSparseDenseAdam
        from unirec_amd.utils.general import get_class_instance, init_seed
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        n_users = 4001
        cfg = _cfg("MF", grad_clip_value=0.0, n_users=n_users)
        B = 128 if both else 256
        g = torch.Generator().manual_seed(17)
        mine = []
        for s_ in range(6):
            lab = torch.zeros(B, G, dtype=torch.int32)
            lab[:, 0] = 1
            uid = torch.randperm(n_users // 2 - 1, generator=g)[:B] * 2 + 2      # EVEN user ids only: every one of them lives on rank 0
            iid = torch.randint(1, N_ITEMS, (B, G), generator=g)
            if both:      # EVEN item ids only as well: both tables overflow on rank 0 in the SAME step (advisor r4: 1 + 1 = 2 read as "no overflow")
                # 500 distinct even ids per batch (> the 447 slots planned at slack 1.25 for 640 lookups), from alternating halves of the
                # id range: consecutive batches share no row, so the fix-up exchange of the reference run (slack 4) stays empty
                perm = (torch.randperm(500, generator=g) + 1 + 500 * (s_ % 2)) * 2
                iid = torch.cat([perm, perm[: B * G - perm.numel()]]).reshape(B, G)
            b = dict(item_id=iid, label=lab, user_id=uid)
            mine.append(_to(b, dev))

        def run(slack):
            init_seed(cfg["seed"])
            m = get_class_instance("MF", "unirec_amd/model")(cfg)
            opt = ShardedSparseDenseAdam(m, rank, world, lr=2e-3, cap_slack=slack)
            m.train()
            losses = [opt.train_step(b, mine[i + 1] if i + 1 < len(mine) else None) for i, b in enumerate(mine)]
            opt.flush()
            torch.cuda.synchronize()
            return opt, m, [float(x) for x in losses]

        # the ITEM table (first in lookup_tables) fits at slack 1.25; the USER table needs 256 slots on rank 0 where 1.25 x 256 / 2 are planned
        opt, m, losses = run(1.25)
        assert opt.n_overflow >= 1 and opt._cap_scale >= 2, (opt.n_overflow, opt._cap_scale)
        state = [None] * world
        dist.all_gather_object(state, (opt._cap_scale, opt.n_overflow, opt.t))
        assert len(set(state)) == 1, state
        assert opt.t == len(mine) + opt.n_overflow and all(np.isfinite(losses)), losses
        bias = [None] * world
        dist.all_gather_object(bias, m.item_bias.data.cpu())
        assert all(torch.equal(bias[0], x) for x in bias[1:])                  # replicated parameters stayed identical through the skips
        ref_opt, _, ref_losses = run(4.0)
        assert ref_opt.n_overflow == 0
        assert abs(losses[-1] - ref_losses[-1]) < 0.05 * abs(ref_losses[-1]), (losses, ref_losses)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("both", [False, True])
def test_overflow_of_the_second_table_skips_the_step_too(both):
    """Advisor r3: only the first table's flags reached the step flags.  MF with every user id on one owner: the user table (second in
    lookup_tables) overflows while the item table fits -- the step must be skipped everywhere, the capacity doubled, the batch re-trained.
    both (advisor r4): the item ids sit on that owner too -- two tables raise their flag in the same step, and the merged flag word must
    still have bit 0 set (a SUM of the two flag words read as 2 = "no overflow")."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_second_table_overflow_worker, args=(r, 2, port, q, both)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(2)], res


def _fixup_overflow_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unirec_amd.facility.distributed import ShardedSparseDenseAdam
        from unirec_amd.utils.general import get_class_instance, init_seed
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = _cfg("SASRec", grad_clip_value=0.0)
        B = 16
        full = _batches(5, B * world)
        full = [b for b in full for _ in range(2)]          # every batch twice in a row: the second time ALL of its rows are hot
        mine = [_to(b, dev, rank * B, (rank + 1) * B) for b in full]

        def run(prefetch_rows):
            init_seed(cfg["seed"])
            m = get_class_instance("SASRec", "unirec_amd/model")(cfg)
            opt = ShardedSparseDenseAdam(m, rank, world, lr=2e-3, cap_slack=1.5, fix_cap_min=4)
            opt.prefetch_rows = prefetch_rows
            m.train()
            losses = [opt.train_step(b, mine[i + 1] if i + 1 < len(mine) else None) for i, b in enumerate(mine)]
            opt.flush()
            m.join_side_updates()
            torch.cuda.synchronize()
            return opt, m, [float(x) for x in losses]

        opt, m, losses = run(True)
        assert opt.n_overflow >= 1 and opt._cap_scale >= 2, (opt.n_overflow, opt._cap_scale)      # the fix-up lists overflowed (the main capacity never does at slack 1.5)
        scales = [None] * world
        dist.all_gather_object(scales, (opt._cap_scale, opt.n_overflow, opt.t))
        assert len(set(scales)) == 1, scales
        assert opt.t == len(mine) + opt.n_overflow
        dense = [None] * world
        dist.all_gather_object(dense, m.dense_flat.data.cpu())
        assert all(torch.equal(dense[0], x) for x in dense[1:])
        ref_opt, ref_m, ref_losses = run(False)              # rows inside the step: no fix-up exchange, no overflow
        assert ref_opt.n_overflow == 0
        # once the capacity has settled the two runs train the same batches on almost the same parameters
        assert abs(losses[-1] - ref_losses[-1]) < 0.05 * abs(ref_losses[-1]), (losses, ref_losses)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_fixup_exchange_overflow_is_an_overflow_like_any_other():
    """Row prefetch: the rows of the next batch that the step in flight updates come in a small second exchange (cap2 slots per pair).
    Batches repeated back to back make EVERY row hot: more than cap2 per pair raises the batch's overflow flag on the owner, the flag
    travels with the gradients, every rank skips, the capacities double, the batch is trained again -- and the replicas stay identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fixup_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(2)], res


@pytest.mark.gpu
def test_capacity_overflow_skips_the_step_everywhere_doubles_and_retrains():
    """A (source, owner) pair that needs more than cap - 1 slots: the flag travels with the row gradients, EVERY rank skips that step's
    update (like a NaN loss), the host sees it two steps later, doubles the capacity and trains the batch again."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(2)], res
