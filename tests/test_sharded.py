"""Multi-GPU path: the fixed-capacity row exchange.

* world_size-2 gloo test on the CPU: the routing of the row-sharded table (owner mapping, the fixed-capacity block layout, the
  ordering of the three exchanges, the flags riding in slot 0) reproduces the single-process row gradients.  The HIP kernels that sit
  between the exchanges are stood in for by torch CPU indexing INSIDE THIS TEST (the product code has no CPU path); what is under
  test is unirec_amd.sharded.RowExchange / owner_and_local / shard_rows / pack_layout.
* GPU tests: the pack / scatter / flag kernels against that host statement (exact integers), the RCCL transport at world 1 (the
  library's own communicator, send / recv to itself), and the sharded optimizer with world == 1 against the plain optimizer path.
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


def _pack_host(uniq, n_local, world, cap):
    """torch restatement of ur_shard_exchange_ids' packing for sorted unique sharded keys `uniq` -> (send_ids, slot_of_uniq, u_of_slot)"""
    from unirec_amd.sharded import pack_layout
    counts = [int(((uniq // n_local) == r).sum()) for r in range(world)]
    has0 = len(uniq) > 0 and int(uniq[0]) == 0
    lay = pack_layout(counts, cap, key0_first=has0)
    send = torch.zeros(world * cap, dtype=torch.int32)
    slot = torch.zeros(len(uniq), dtype=torch.int64)
    uos = torch.full((world * cap,), -1, dtype=torch.int64)
    u = 0
    for o, (first, n) in enumerate(lay):
        if o == 0 and has0:
            slot[0] = 0
            u += 1
        for j in range(n):
            send[first + j] = int(uniq[u] % n_local) if world > 1 else int(uniq[u])
            slot[u] = first + j
            uos[first + j] = u
            u += 1
    assert u == len(uniq)
    return send, slot, uos, counts


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
        # ---- what rows_plan_sharded produces: unique keys sorted by (owner, local row); then the fixed-capacity block
        o, l = owner_and_local(ids, world)
        keys = o * n_local + l
        uniq, inv = torch.unique(keys, sorted=True, return_inverse=True)
        cap = 48
        send, slot, uos, counts = _pack_host(uniq, n_local, world, cap)
        x = RowExchange(world, rank)
        req = x.all_to_all_equal(send)                                               # (1) ids -> owners
        assert len(req) == world * cap and int(req.max()) < n_local
        for s_ in range(world):                                                      # every block ascending (the owner's merge plan needs that)
            blk = req[s_ * cap:(s_ + 1) * cap]
            assert bool((blk[1:] >= blk[:-1]).all()) and int(blk[0]) == 0           # slot 0: reserved padding
        compact = x.all_to_all_equal(shard[req.long()])                              # (2) rows -> requesters, slot layout
        assert torch.equal(compact[slot[inv]], full[ids])                            # lookups see the right rows, bit-exact
        assert torch.equal(compact[0], torch.zeros(d))                               # compact row 0 = the padding row
        ug = torch.zeros(len(uniq), d).index_add_(0, inv, grads)
        ug[uniq == 0] = 0                                                            # padding row
        sendg = torch.zeros(world * cap, d)
        sendg[uos >= 0] = ug[uos[uos >= 0]]
        for s_ in range(world):                                                      # slot 0 of every block: this rank's flags
            sendg[s_ * cap, :4] = torch.tensor([0.0, 0.0, 1.5 + rank, 1.0])
        grads_in = x.all_to_all_equal(sendg)                                         # (3) row gradients -> owners
        flags = torch.stack([grads_in[s_ * cap, :4] for s_ in range(world)])
        assert torch.equal(flags[:, 2], torch.tensor([1.5 + r for r in range(world)]))   # every rank sees every rank's loss
        grads_in[torch.arange(world) * cap] = 0                                      # (the owner's segment sum ignores local row 0)
        shard_grad = torch.zeros(n_local, d).index_add_(0, req.long(), grads_in)
        shard_grad[0] = 0
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
        assert torch.equal(x.all_reduce_sum(torch.ones(3) * (rank + 1)), torch.ones(3) * sum(range(1, world + 1)))
        assert torch.equal(x.all_gather_cat(torch.tensor([rank])), torch.arange(world))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
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


def test_pack_layout_reserves_slot_zero_and_detects_overflow():
    from unirec_amd.sharded import pack_layout
    assert pack_layout([3, 0, 5], 8) == [(5, 3), (16, 0), (19, 5)]
    assert pack_layout([3, 2], 4, key0_first=True) == [(2, 2), (6, 2)]          # key 0 takes no slot
    assert pack_layout([7], 8) == [(1, 7)]
    with pytest.raises(OverflowError):
        pack_layout([8], 8)


# ------------------------------------------------------------------------------------------------ GPU: kernels of the exchange
@pytest.mark.gpu
@pytest.mark.parametrize("world,with_zero", [(1, True), (2, True), (4, False), (8, True)])
def test_pack_kernel_matches_host_layout(world, with_zero):
    """ur_shard_exchange_ids (pack only) == the host statement: send ids, slot of every unique key, key of every slot (exact integers);
    ur_compact_index with the slot map; the overflow flag."""
    from unirec_amd import ops
    from unirec_amd.sharded import shard_rows
    dev = torch.device("cuda:0")
    N = 5003
    g = torch.Generator().manual_seed(world)
    ids_a = torch.randint(1, N, (700,), generator=g, dtype=torch.int32)
    if with_zero:
        ids_a[::7] = 0
    ids_b = torch.randint(1, N, (90,), generator=g)
    n_local = shard_rows(N, world)
    pl, counts = ops.rows_plan_sharded(ids_a.to(dev), ids_b.to(dev), N, world)
    n_uniq = int(pl.n_uniq)
    uniq = pl.uniq_idx[:n_uniq].cpu().to(torch.int64)
    cap = 790 + 1 if world == 1 else 64 * ((int(counts.max()) + 64) // 64)
    i32 = dict(dtype=torch.int32, device=dev)
    send, slot, uos, flags = torch.empty(world * cap, **i32), torch.zeros(pl.n, **i32), torch.empty(world * cap, **i32), torch.zeros(4, **i32)
    ops.shard_exchange_ids(pl, counts, n_local, world, cap, send, slot, uos, flags)
    want_send, want_slot, want_uos, want_counts = _pack_host(uniq, n_local, world, cap)
    assert counts.cpu().tolist() == want_counts and int(flags[0]) == 0
    assert torch.equal(send.cpu(), want_send)
    assert torch.equal(slot[:n_uniq].cpu().to(torch.int64), want_slot)
    got_uos = uos.cpu().to(torch.int64)
    if with_zero:      # the padding id reads slot 0; its own key has no slot
        want_uos = want_uos.clone()
    assert torch.equal(got_uos, want_uos)
    idx_a, idx_b = ops.compact_index(pl, slot)
    # every lookup's slot asks its owner for exactly the lookup's row
    req_local = send.cpu().to(torch.int64)
    for ids, idx in ((ids_a.to(torch.int64), idx_a.cpu().to(torch.int64)), (ids_b, idx_b.cpu())):
        owner = idx // cap
        glob = torch.where(req_local[idx] > 0, (req_local[idx] - 1) * world + owner, torch.zeros_like(idx)) if world > 1 else req_local[idx]
        assert torch.equal(glob, ids)
        assert bool((idx[ids == 0] == 0).all())
    # overflow: a capacity one below what the fullest owner needs raises the flag
    need = int(counts.max()) - (1 if with_zero and int(counts[0]) == int(counts.max()) else 0)
    small = max(2, need)          # holds need - 1 keys
    flags.zero_()
    ops.shard_exchange_ids(pl, counts, n_local, world, small, torch.empty(world * small, **i32), slot, torch.empty(world * small, **i32), flags)
    assert int(flags[0]) == 1


@pytest.mark.gpu
def test_grad_scatter_carries_the_flags_and_step_flags_reads_them():
    from unirec_amd import ops
    dev = torch.device("cuda:0")
    W, cap, d = 4, 8, 16
    uos = torch.full((W * cap,), -1, dtype=torch.int32, device=dev)
    uos[5], uos[9], uos[31] = 0, 2, 1
    ug = torch.arange(3 * d, dtype=torch.float32, device=dev).view(3, d) + 1
    loss_out = torch.tensor([0.75, 16.0, 1.0, 0.0], device=dev)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    ws = torch.empty(W * cap, d, device=dev)
    out = ops.shard_exchange_grads(ug, uos, W, cap, ws, loss_out=loss_out, flags=flags)
    assert torch.equal(out[5], ug[0]) and torch.equal(out[9], ug[2]) and torch.equal(out[31], ug[1])
    for s in range(W):
        assert out[s * cap, :4].tolist() == [0.0, 0.0, 0.75, 1.0] and float(out[s * cap, 4:].abs().sum()) == 0
    mask = torch.ones(W * cap, dtype=torch.bool, device=dev)
    mask[[5, 9, 31] + [s * cap for s in range(W)]] = False
    assert float(out[mask].abs().sum()) == 0
    out4 = torch.zeros(4, device=dev)
    # as received by an owner: one block per source rank; rank 2 reports an overflow, rank 1 a NaN loss
    recv = out.clone()
    for s, (nan, ovf, loss) in enumerate([(0, 0, 0.5), (0, 0, 1.5), (0, 0, 1.0), (0, 0, 2.0)]):
        recv[s * cap, :4] = torch.tensor([nan, ovf, loss, 1.0])
    ops.shard_step_flags(recv, W, cap, out4)
    assert out4.tolist() == [0.25, 1.25, 0.0, 0.0]
    recv[2 * cap, 1] = 1.0
    ops.shard_step_flags(recv, W, cap, out4)
    assert out4[0] == -1.0 and out4[3] == 1.0 and out4[1] == 1.25
    recv[1 * cap, 0] = 1.0
    ops.shard_step_flags(recv, W, cap, out4)
    assert out4[0] == -1.0 and out4[2] == 1.0 and bool(torch.isnan(out4[1]))
    # a NaN loss on THIS rank: the flag row says so (and carries no NaN into the sums)
    nan_loss = torch.tensor([float("nan"), 16.0, -1.0, 0.0], device=dev)
    out = ops.shard_exchange_grads(ug, uos, W, cap, ws, loss_out=nan_loss, flags=flags)
    assert out[0, :4].tolist() == [1.0, 0.0, 0.0, 1.0]


@pytest.mark.gpu
def test_rccl_transport_at_world_one():
    """The library's own RCCL communicators (ur_comm_init): at world 1 every exchange is a send / recv to itself and the all-reduce the
    identity -- the transport code path of the multi-GPU step, executed through RCCL on the one GPU there is."""
    from unirec_amd import ops
    if ops.comm_world() < 0:
        pytest.skip("no RCCL library in this process")
    dev = torch.device("cuda:0")
    ops.comm_init(0, 1)
    assert ops.comm_world() == 1
    assert ops.comm_count() == 1       # RCCL's own answer (ncclCommCount / ncclCommUserRank of both communicators)
    try:
        N, d = 3001, 32
        g = torch.Generator().manual_seed(3)
        table = torch.randn(N, d, generator=g).to(dev)
        table[0] = 0
        ids = torch.randint(0, N, (500,), generator=g, dtype=torch.int32).to(dev)
        pl, counts = ops.rows_plan_sharded(ids, None, N, 1)
        cap = 501
        i32 = dict(dtype=torch.int32, device=dev)
        bufs = lambda: (torch.empty(cap, **i32), torch.zeros(500, **i32), torch.empty(cap, **i32), torch.zeros(4, **i32))   # noqa: E731
        s0, slot0, uos0, f0 = bufs()
        ops.shard_exchange_ids(pl, counts, N, 1, cap, s0, slot0, uos0, f0)
        s1, slot1, uos1, f1 = bufs()
        recv = torch.empty(cap, **i32)
        ops.shard_exchange_ids(pl, counts, N, 1, cap, s1, slot1, uos1, f1, recv_ids=recv, transport=True)
        assert torch.equal(recv, s0) and torch.equal(slot0, slot1)
        ws, compact = torch.empty(cap, d, device=dev), torch.empty(cap, d, device=dev)
        ops.shard_exchange_rows(table, recv, 1, cap, ws, compact=compact, transport=True)
        idx_a, _ = ops.compact_index(pl, slot1)
        assert torch.equal(compact[idx_a.long()], table[ids.long()])
        ug = torch.randn(int(pl.n_uniq), d, generator=torch.Generator().manual_seed(1)).to(dev)
        gin = torch.empty(cap, d, device=dev)
        ops.shard_exchange_grads(ug, uos1, 1, cap, ws, grads_in=gin, transport=True)
        assert torch.equal(gin, ops.shard_exchange_grads(ug, uos1, 1, cap, torch.empty(cap, d, device=dev)))
        t = torch.arange(1000, dtype=torch.float32, device=dev)
        assert torch.equal(ops.comm_all_reduce_sum(t.clone()), t)
        # ---- the row prefetch's use of the two communicators (facility/distributed.py _prefetch_rows): the next batch's slot list and
        # rows go through the SECOND communicator on a side stream while the step's own exchange (first communicator) runs on the main
        # stream; then the fix-up: plan against an owner-side plan, gather, exchange on the first communicator, scatter into the table
        own = ops.rows_plan_merge(recv, [cap])
        cap2 = 64
        req2, slot2 = torch.empty(cap2, **i32), torch.empty(cap2, **i32)
        f2 = torch.zeros(4, **i32)
        prev = ops.RowsPlan()                       # "the step in flight updates these rows": 40 of the requested ones
        hot_rows = torch.unique(recv[recv > 0])[:40].contiguous()
        prev.n, prev.n_a, prev.uniq_idx, prev.n_uniq = 64, 0, torch.cat([hot_rows, torch.zeros(24, **i32)]), torch.tensor([40], **i32)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.shard_fixup_plan(recv, 1, cap, prev, cap2, req2, slot2, f2, torch.zeros(1, **i32))
            slot2_r = ops.comm_all_to_all(slot2, torch.empty_like(slot2), 1, ahead=True, kind="ids")
            ws_n, compact_n = torch.empty(cap, d, device=dev), torch.empty(cap, d, device=dev)
            ops.shard_exchange_rows(table, recv, 1, cap, ws_n, compact=compact_n, transport=False)
            ops.comm_all_to_all(ws_n, compact_n, 1, ahead=True)
        ops.shard_exchange_grads(ug, uos1, 1, cap, ws, grads_in=gin, transport=True)      # (main stream, first communicator, meanwhile)
        torch.cuda.current_stream().wait_stream(side)
        assert int(f2[0]) == 0 and int((slot2_r >= 0).sum()) == 40
        table2 = table.clone()
        table2[hot_rows.long()] += 1.0              # "the update": the prefetched copies of the hot rows are stale now
        rows2, rows2_r = torch.empty(cap2, d, device=dev), torch.empty(cap2, d, device=dev)
        ops.shard_exchange_rows(table2, req2, 1, cap2, rows2, compact=rows2_r, transport=True)
        ops.shard_fixup_apply(compact_n, rows2_r, slot2_r, 1, cap, cap2)
        assert torch.equal(compact_n[idx_a.long()], table2[ids.long()])
        assert own.n == cap
        torch.cuda.synchronize()
    finally:
        ops.comm_destroy()
    assert ops.comm_world() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["SASRec", "GRU"])
def test_sharded_optimizer_world1_equals_plain_path(kind):
    """ShardedSparseDenseAdam at world 1 (the whole fixed-capacity step, collectives degenerate) == SparseDenseAdam: same losses, same
    parameters, with and without the plan lookahead, lazy_dense and rowwise, with and without gradient clipping."""
    from unirec_amd.facility.distributed import ShardedSparseDenseAdam
    from unirec_amd.facility.optimizer import SparseDenseAdam
    from unirec_amd.utils.argument_parser import parse_arguments
    from unirec_amd.utils.general import get_class_instance, init_seed
    dev = torch.device("cuda:0")
    cfg = parse_arguments(dict(model=kind, n_users=10, n_items=5000, device="cuda:0", loss_type="bpr", embedding_size=64, hidden_size=64,
                               n_layers=2, n_heads=16, inner_size=128, hidden_dropout_prob=0.0, attn_dropout_prob=0.0, dropout_prob=0.0,
                               hidden_act="swish", max_seq_len=20, batch_size=32, seed=5, n_sample_neg_train=4))
    g = torch.Generator().manual_seed(3)
    batches = []
    for step in range(5):
        seq = torch.randint(1, 5000, (32, 20), generator=g, dtype=torch.int32)
        seq[:, : step * 3] = 0
        lab = torch.zeros(32, 5, dtype=torch.int32)
        lab[:, 0] = 1
        batches.append(dict(item_seq=seq.to(dev), item_id=torch.randint(1, 5000, (32, 5), generator=g).to(dev), label=lab.to(dev)))
    for mode, clip, look in (("lazy_dense", None, True), ("lazy_dense", None, False), ("rowwise", None, True), ("lazy_dense", 0.05, True)):
        init_seed(5)
        m1 = get_class_instance(kind, "unirec_amd/model")(cfg)
        init_seed(5)
        m2 = get_class_instance(kind, "unirec_amd/model")(cfg)
        o1 = ShardedSparseDenseAdam(m1, 0, 1, lr=2e-3, table_mode=mode, grad_clip=clip)
        o2 = SparseDenseAdam(m2, lr=2e-3, table_mode=mode, grad_clip=clip)
        m1.train(), m2.train()
        for i, b in enumerate(batches):
            l1 = o1.train_step(b, batches[i + 1] if look and i + 1 < len(batches) else None)
            o2.zero_grad()
            o2.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
            l2 = m2.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
            o2.step()
            np.testing.assert_allclose(float(l1), float(l2), rtol=1e-6)
        m1.join_side_updates()
        o1.flush(), o2.flush()
        assert o1.n_overflow == 0
        # (bit-identical without clipping; WITH it the two paths sum the squared gradient norm in different orders -- owners' rows + one
        # flat all-reduce against one pass -- so the clip coefficient, and with it every update, differs by an ulp: 1e-7 relative, and
        # on an element that is ~0, 1e-8 absolute)
        np.testing.assert_allclose(m1.dense_flat.data.cpu().numpy(), m2.dense_flat.data.cpu().numpy(), rtol=1e-6, atol=3e-8)
        # (the lazy replay of a row's missed steps is summed in two pieces when the tail catch-up ran: a few ulp of an lr-sized term)
        np.testing.assert_allclose(m1.item_embedding.weight.detach().cpu().numpy(), m2.item_embedding.weight.detach().cpu().numpy(),
                                   rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_rows_reduce_into_caller_rows():
    """ur_rows_reduce(out_rows): the per-id sums written to rows of the caller's choice == the default output scattered by the same map
    (bit-exact), rows outside the map untouched"""
    import torch
    from unirec_amd import ops
    torch.manual_seed(5)
    dev, d, n_a, n_b, G = "cuda", 64, 900, 30, 5
    ids_a = torch.randint(0, 200, (n_a,), device=dev, dtype=torch.int32)
    ids_b = torch.randint(0, 200, (n_b * G,), device=dev, dtype=torch.int64)
    rows = torch.randn(n_a, d, device=dev)
    coef, vec = torch.randn(n_b * G, device=dev), torch.randn(n_b, d, device=dev)
    pl = ops.rows_plan(ids_a, ids_b, 200)
    want = ops.rows_reduce(pl, rows, coef, vec, G, d)
    n_uniq = int(pl.n_uniq.item())
    perm = torch.randperm(pl.n + 17, device=dev)[: pl.n].to(torch.int32)
    out = torch.full((pl.n + 17, d), 7.0, device=dev)
    got = ops.rows_reduce(pl, rows, coef, vec, G, d, out=out, out_rows=perm)
    assert got is out
    ref = torch.full_like(out, 7.0)
    ref[perm[:n_uniq].long()] = want[:n_uniq]
    assert torch.equal(out, ref)


@pytest.mark.gpu
def test_native_transport_selftest_world_one():
    """What ShardedSparseDenseAdam runs once before it trusts the library's communicators (world > 1 on the nccl backend), executed at
    world 1 on a one-rank nccl group: library all-to-all / all-reduce == torch.distributed's."""
    from unirec_amd import ops
    from unirec_amd.facility.distributed import native_transport_selftest
    if ops.comm_world() < 0:
        pytest.skip("no RCCL library in this process")
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        assert ops.comm_init(0, 1) is True
        try:
            assert native_transport_selftest(0, 1, None, dev) is True
        finally:
            if ops.comm_world() >= 1:
                ops.comm_destroy()
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,cap,cap2,n_prev", [(1, 300, 64, 120), (4, 192, 64, 400), (8, 128, 64, 0), (3, 257, 64, 2000)])
def test_fixup_plan_split_and_apply_match_a_host_statement(world, cap, cap2, n_prev):
    """Row prefetch (facility/distributed.py _prefetch_rows): the ids-only kernels of the fix-up exchange against numpy -- which slots of
    the next batch's request list ask for rows the step in flight updates (in slot order, padding behind, overflow flag), the hot / cold
    split of the next owner-side plan, and the scatter of the received rows into the compact table."""
    from unirec_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(world * 1000 + cap)
    n_local = 5000
    prev_rows = np.sort(rng.choice(np.arange(1, n_local), size=n_prev, replace=False)).astype(np.int32) if n_prev else np.zeros(0, np.int32)
    recv = np.zeros((world, cap), np.int32)
    for s in range(world):                       # every block: slot 0 reserved, keys right-aligned and ascending behind padding
        cnt = int(rng.integers(0, cap))
        keys = np.sort(rng.choice(np.arange(1, n_local), size=cnt, replace=False))
        if n_prev and cnt:                       # make sure some are hot
            k = min(cnt, n_prev, 40 if s != 1 else cap - 1)
            keys[:k] = rng.choice(prev_rows, size=k, replace=False)
            keys = np.sort(np.unique(keys))
            cnt = len(keys)
        recv[s, cap - cnt:] = keys
    prev = None
    if n_prev:
        prev = ops.RowsPlan()
        prev.n, prev.n_a = max(n_prev, 1) + 7, 0
        buf = np.zeros(prev.n, np.int32)
        buf[:n_prev] = prev_rows
        prev.uniq_idx = torch.from_numpy(buf).to(dev)
        prev.n_uniq = torch.tensor([n_prev], dtype=torch.int32, device=dev)
    req2 = torch.full((world * cap2,), 77, dtype=torch.int32, device=dev)
    slot2 = torch.full((world * cap2,), 77, dtype=torch.int32, device=dev)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    ops.shard_fixup_plan(torch.from_numpy(recv.reshape(-1)).to(dev), world, cap, prev, cap2, req2, slot2, flags,
                         torch.zeros(world, dtype=torch.int32, device=dev))
    req2, slot2 = req2.cpu().numpy().reshape(world, cap2), slot2.cpu().numpy().reshape(world, cap2)
    hotset = set(prev_rows.tolist())
    overflow = False
    for s in range(world):
        want = [(int(r), p) for p, r in enumerate(recv[s]) if p > 0 and int(r) in hotset]
        got = [(int(r), int(p)) for r, p in zip(req2[s], slot2[s]) if p >= 0]      # (order inside a list is arbitrary)
        if len(want) > cap2:
            overflow = True
            assert len(got) == cap2 and set(got) <= set(want)
        else:
            assert sorted(got) == sorted(want)
        k = len(got)
        assert all(p >= 0 for p in slot2[s, :k]) and (req2[s, k:] == 0).all() and (slot2[s, k:] == -1).all()
    assert bool(int(flags[0]) & 1) == overflow
    # ---- hot / cold split of an owner-side plan against the previous one
    own_rows = np.unique(recv.reshape(-1))
    own = ops.RowsPlan()
    own.n, own.n_a = len(own_rows) + 5, 0
    buf = np.zeros(own.n, np.int32)
    buf[:len(own_rows)] = own_rows
    own.uniq_idx = torch.from_numpy(buf).to(dev)
    own.n_uniq = torch.tensor([len(own_rows)], dtype=torch.int32, device=dev)
    last = torch.zeros(n_local, dtype=torch.int32, device=dev)
    touched = rng.random(n_local) < 0.5
    last[torch.from_numpy(np.nonzero(touched)[0]).to(dev)] = 3
    for use_last in (True, False):
        cold, hot = ops.rows_split_hot(own, last if use_last else None, prev)
        got_c = sorted(cold.uniq_idx[: int(cold.n_uniq)].cpu().tolist())
        got_h = sorted(hot.uniq_idx[: int(hot.n_uniq)].cpu().tolist())
        assert got_h == sorted(int(r) for r in own_rows if r != 0 and int(r) in hotset)
        assert got_c == sorted(int(r) for r in own_rows if r != 0 and int(r) not in hotset and (touched[r] or not use_last))
    # ---- apply
    d = 8
    compact = torch.zeros(world * cap, d, device=dev)
    rows2 = torch.arange(world * cap2 * d, device=dev, dtype=torch.float32).reshape(world * cap2, d) + 1
    ops.shard_fixup_apply(compact, rows2, torch.from_numpy(slot2.reshape(-1)).to(dev), world, cap, cap2)
    exp = np.zeros((world * cap, d), np.float32)
    r2 = rows2.cpu().numpy()
    for q in range(world * cap2):
        if slot2.reshape(-1)[q] >= 0:
            exp[(q // cap2) * cap + slot2.reshape(-1)[q]] = r2[q]
    assert np.array_equal(compact.cpu().numpy(), exp)
