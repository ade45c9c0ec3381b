"""Multi-GPU path.

* world_size-2 gloo test on the CPU: the routing of the row-sharded table (owner mapping, all-to-all split
  sizes, ordering of the three exchanges) reproduces the single-process row gradients.  The HIP kernels that
  sit between the exchanges are stood in for by torch CPU indexing INSIDE THIS TEST (the product code has no
  CPU path); what is under test is unirec_amd.sharded.RowExchange / owner_and_local / shard_rows.
* GPU test: the sharded step with world == 1 is bit-identical in structure to the plain optimizer path and must
  produce the same parameters after a few steps.
"""
import os
import socket

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


def _worker(rank, world, port, N, d, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unirec_amd.sharded import RowExchange, owner_and_local, shard_rows
    try:
        n_local = shard_rows(N, world)
        g = torch.Generator().manual_seed(7)
        full = torch.randn(N, d, generator=g)          # the logical table, known to every rank for checking
        full[0] = 0
        ids_all = torch.arange(N)
        owner, local = owner_and_local(ids_all, world)
        shard = torch.zeros(n_local, d)
        mine = owner == rank
        shard[local[mine]] = full[mine]
        # this rank's lookups (with duplicates and padding) and the gradient row of each lookup
        gb = torch.Generator().manual_seed(100 + rank)
        ids = torch.randint(0, N, (57 + 10 * rank,), generator=gb)
        ids[:3] = 0
        ids[5] = ids[6]
        grads = torch.randn(len(ids), d, generator=gb)
        # ---- what rows_plan_sharded produces: unique keys sorted by (owner, local row) + per-owner counts
        o, l = owner_and_local(ids, world)
        keys = o * n_local + l
        uniq, inv = torch.unique(keys, sorted=True, return_inverse=True)
        send_counts = [int(((uniq // n_local) == r).sum()) for r in range(world)]
        x = RowExchange(world, rank)
        recv_counts = x.exchange_counts(send_counts)
        assert x.exchange_counts_dev(torch.tensor(send_counts, dtype=torch.int32)) == (send_counts, recv_counts)
        assert x.exchange_counts_host(send_counts) == (send_counts, recv_counts)      # lookahead path: counts already on the host
        x.cpu_group = None                                                             # no gloo group: falls back to the main group
        assert x.exchange_counts_host(send_counts) == (send_counts, recv_counts)
        req = x.all_to_all_rows((uniq % n_local).to(torch.int32), send_counts, recv_counts)
        assert int(req.max()) < n_local and len(req) == sum(recv_counts)
        compact = x.all_to_all_rows(shard[req.long()], recv_counts, send_counts)      # rows come back in key order
        assert torch.equal(compact[inv], full[ids])                                  # lookups see the right rows, bit-exact
        ug = torch.zeros(len(uniq), d).index_add_(0, inv, grads)
        ug[uniq == 0] = 0                                                            # padding row
        grads_in = x.all_to_all_rows(ug, send_counts, recv_counts)
        shard_grad = torch.zeros(n_local, d).index_add_(0, req.long(), grads_in)
        # ---- reference: dense gradient of the whole table summed over both ranks' lookups
        all_ids = [None] * world
        all_gr = [None] * world
        dist.all_gather_object(all_ids, ids)
        dist.all_gather_object(all_gr, grads)
        dense = torch.zeros(N, d)
        for i_, g_ in zip(all_ids, all_gr):
            dense.index_add_(0, i_, g_)
        dense[0] = 0
        expect = torch.zeros(n_local, d)
        expect[local[mine]] = dense[mine]
        np.testing.assert_allclose(shard_grad.numpy(), expect.numpy(), rtol=1e-5, atol=1e-6)
        assert torch.equal(shard_grad[0], torch.zeros(d))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [101, 64])
def test_row_exchange_world2_gloo(N):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_owner_mapping_matches_kernel_contract():
    from unirec_amd.sharded import owner_and_local, shard_rows
    ids = torch.arange(0, 23)
    for W in (1, 2, 4, 8):
        o, l = owner_and_local(ids, W)
        assert int(o[0]) == 0 and int(l[0]) == 0
        if W > 1:
            assert torch.equal(o[1:], ids[1:] % W) and torch.equal(l[1:], ids[1:] // W + 1)
            assert int(l.max()) < shard_rows(23, W)
            pairs = set(zip(o.tolist(), l.tolist()))
            assert len(pairs) == len(ids)                      # injective
        else:
            assert torch.equal(l, ids)


_CFG = dict(model="SASRec", n_users=10, n_items=3001, device="cuda:0", loss_type="bpr", embedding_size=64, hidden_size=64,
            dropout_prob=0.0, init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, has_user_bias=False,
            has_item_bias=False, distance_type="dot", tau=1.0, train_file_format="user-item", exp_name="t", n_layers=2,
            n_heads=16, inner_size=128, hidden_dropout_prob=0.0, attn_dropout_prob=0.0, hidden_act="swish",
            layer_norm_eps=1e-10, max_seq_len=20, use_position_emb=True)


def _batches(n_steps, B):
    g = torch.Generator().manual_seed(11)
    out = []
    for s in range(n_steps):
        seq = torch.randint(1, 3001, (B, 20), generator=g, dtype=torch.int32)
        seq[::3, : 4 + s] = 0
        out.append(dict(item_seq=seq, item_id=torch.randint(1, 3001, (B, 5), generator=g),
                        label=torch.zeros(B, 5, dtype=torch.int32)))
    return out


def _gpu_worker(rank, world, port, q, kind="SASRec"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share cuda:0; rows are staged via host
    try:
        from unirec_amd.sharded import ShardedSasrecStep, owner_and_local
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        T0 = torch.randn(3001, 64, generator=torch.Generator().manual_seed(5)) * 0.05
        T0[0] = 0
        cfg_k = dict(_CFG, model=kind)
        st = ShardedSasrecStep(cfg_k, dev, rank, world, table_mode="lazy_dense")
        owner, local = owner_and_local(torch.arange(3001), world)
        st.table.zero_()
        st.table[local[owner == rank].to(dev)] = T0[owner == rank].to(dev)
        B = 16
        losses = []
        mine = [{k: v[rank * B:(rank + 1) * B].to(dev).contiguous() for k, v in b.items()} for b in _batches(3, B * world)]
        for i, b in enumerate(mine):   # with the plan lookahead (next batch's id sort on a side stream); the 1-rank run has none
            losses.append(float(st.step(b, mine[i + 1] if i + 1 < len(mine) else None)))
        st.flush()
        full = st.gather_table().cpu()
        dense = st.model.dense_flat.data.cpu()
        # ---- full-item ranking over the sharded table == ur_full_rank over the gathered table (exact integers)
        from unirec_amd import ops
        gh = torch.Generator().manual_seed(100 + rank)
        uid = torch.randint(0, 12, (B,), generator=gh).to(dev)                # users 10, 11: outside the history table
        tgt = mine[0]["item_id"][:, 0].contiguous()
        lens = torch.randint(0, 40, (10,), generator=torch.Generator().manual_seed(7))
        hp = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(dev)
        hs_g = torch.randint(1, 3001, (int(lens.sum()),), generator=torch.Generator().manual_seed(8), dtype=torch.int32)
        hs = torch.cat([hs_g[int(hp[u]):int(hp[u + 1])].sort().values for u in range(10)]).to(dev)
        got = st.full_item_ranks(mine[0]["item_seq"], tgt, user_id=uid, hist_ptr=hp, hist_sorted=hs)
        want, _ = ops.full_rank(st.encode(mine[0]["item_seq"]), full.to(dev), tgt, user_id=uid, hist_ptr=hp, hist_sorted=hs)
        assert torch.equal(got.cpu(), want.cpu()), (got.cpu(), want.cpu())
        got2 = st.full_item_ranks(mine[0]["item_seq"], tgt)                    # no history
        want2, _ = ops.full_rank(st.encode(mine[0]["item_seq"]), full.to(dev), tgt)
        assert torch.equal(got2.cpu(), want2.cpu()) and int(want2.max()) > 0
        all_losses = [None] * world
        dist.all_gather_object(all_losses, losses)
        if rank == 0:
            one = ShardedSasrecStep(cfg_k, dev, 0, 1, table_mode="lazy_dense")
            one.table.copy_(T0.to(dev))
            ref_losses = [float(one.step({k: v.to(dev) for k, v in b.items()})) for b in _batches(3, B * world)]
            one.flush()
            # global-mean loss == mean of the equal-sized rank means; parameters after 3 steps agree
            np.testing.assert_allclose(np.mean(all_losses, axis=0), ref_losses, rtol=1e-5)
            np.testing.assert_allclose(dense.numpy(), one.model.dense_flat.data.cpu().numpy(), rtol=1e-4, atol=2e-5)  # Adam amplifies rounding noise on ~zero gradients (key.bias): atol = 2% of an lr-sized step
            np.testing.assert_allclose(full.numpy(), one.table.cpu().numpy(), rtol=1e-4, atol=2e-5)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", [("SASRec", 2), ("GRU", 2), ("SASRec", 4)])
def test_two_ranks_equal_one_rank_with_the_concatenated_batch(kind, world):
    """SURVEY.md 8e parity test: W ranks x batch B == 1 rank x batch W*B (losses and updated parameters; full-item ranks over the
    sharded table == over the gathered table); GRU = config C4; W = 4: more runs in the owner-side merge, unequal shard sizes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["SASRec", "GRU"])
def test_sharded_step_world1_equals_plain_path(kind):
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.model.sequential.gru import GRU
    from unirec_amd.model.sequential.sasrec import SASRec as _S
    from unirec_amd.sharded import ShardedSasrecStep
    SASRec = GRU if kind == "GRU" else _S
    dev = torch.device("cuda:0")
    cfg = dict(model=kind, n_users=10, n_items=5000, device="cuda:0", loss_type="bpr", embedding_size=64, hidden_size=64,
               dropout_prob=0.0, init_method="normal", init_mean=0.0, init_std=0.05, has_user_emb=False, has_user_bias=False,
               has_item_bias=False, distance_type="dot", tau=1.0, train_file_format="user-item", exp_name="t", n_layers=2,
               n_heads=16, inner_size=128, hidden_dropout_prob=0.0, attn_dropout_prob=0.0, hidden_act="swish",
               layer_norm_eps=1e-10, max_seq_len=20, use_position_emb=True)
    for mode in ("lazy_dense", "rowwise"):
        st = ShardedSasrecStep(cfg, dev, rank=0, world=1, table_mode=mode)
        m = SASRec(cfg)
        with torch.no_grad():
            m.dense_flat.data.copy_(st.model.dense_flat.data)
            m.item_embedding.weight.copy_(st.table)
        opt = SparseDenseAdam(m, lr=1e-3, table_mode=mode)
        m.train()
        g = torch.Generator().manual_seed(3)
        for step in range(3):
            seq = torch.randint(1, 5000, (32, 20), generator=g, dtype=torch.int32)
            seq[:, : step * 3] = 0
            batch = dict(item_seq=seq.to(dev), item_id=torch.randint(1, 5000, (32, 5), generator=g).to(dev),
                         label=torch.zeros(32, 5, dtype=torch.int32, device=dev))
            l1 = st.step(batch)
            opt.zero_grad()
            opt.plan_batch(item_seq=batch["item_seq"], item_id=batch["item_id"])
            l2, _, _, _ = m(item_id=batch["item_id"], label=batch["label"], item_seq=batch["item_seq"])
            l2.backward()
            opt.step()
            np.testing.assert_allclose(float(l1), float(l2.detach()), rtol=1e-6)
        st.flush()
        opt.flush()
        np.testing.assert_allclose(st.model.dense_flat.data.cpu().numpy(), m.dense_flat.data.cpu().numpy(), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(st.gather_table().cpu().numpy(), m.item_embedding.weight.detach().cpu().numpy(), rtol=1e-6, atol=1e-8)
