"""Multi-GPU training step behind the Trainer: ONE model trained by W processes.

Reference behaviour being replaced: ``accelerator.prepare(model, optimizer, loaders)`` wraps the model in DDP
(unirec/facility/trainer.py:67, 261) and ``accelerator.backward(loss)`` (:346) all-reduces EVERY gradient -- including the
dense [N, d] embedding gradient -- before ``optimizer.step()`` (:349) updates W identical replicas.  Here, with one process
per GPU over ``torch.distributed`` (RCCL on the GPUs):

  * dense parameters (the flat encoder buffer, user / item bias vectors) are replicated; their gradients are summed by ONE
    flat all-reduce per step and the 1/W of DDP's mean is folded into the optimizer kernels' gradient scale;
  * every embedding table is ROW-SHARDED: row ``i`` lives on rank ``i % W`` at local row ``i // W + 1`` (local row 0 is the
    padding row of every shard).  Per step and table: sort/unique the batch's ids by (owner, row) -> all-to-all #1 (row ids) ->
    owners bring the rows up to date (lazy Adam) and gather them -> all-to-all #2 (rows) -> the model runs its NORMAL
    ``forward_backward`` on the compact table of fetched rows (ids re-indexed; same HIP kernels, same model code) ->
    segment-reduce the row gradients -> all-to-all #3 (row gradients) -> owners sum the contributions in source-rank order
    (deterministic) and apply the optimizer rule to their rows.  No collective ever touches a full table; every exchange has a
    FIXED capacity per (source, owner) pair (no count exchange, no host synchronisation: ShardedSparseDenseAdam's docstring);
  * ``grad_clip_value`` (trainer.py:347-348): the global norm is that of the AVERAGED gradient = sqrt(sum of squares of the
    all-reduced dense gradients + of the owner-side unique row gradients, all-reduced) / W; every optimizer rule of
    ``Trainer._build_optimizer`` and ``weight_decay`` go through the same kernels as on one GPU (SparseDenseAdam);
  * a NaN loss on ANY rank skips the step on every rank (the flag rides in the dense all-reduce; the reference checks
    per process, trainer.py:343-350, and would desynchronise its replicas);
  * checkpoints hold the FULL tables under the reference's state_dict names: ``gather_state_dict`` streams the shards to
    rank 0 chunk by chunk (never more than one chunk of foreign rows on any device), ``scatter_state_dict`` deals a full
    state_dict out to whatever world size is running -- a checkpoint written by W ranks loads into 1 or 2W.

Parity: W ranks x batch B == 1 rank x the concatenated batch (tests/test_distributed_trainer.py; gloo, CPU-staged on one GPU).
"""
import collections
import os
import warnings

import torch

from .. import pgroup as dist     # torch.distributed's surface, or the in-process loopback group's (pgroup.py)

from .. import ops
from ..sharded import RowExchange, shard_rows   # noqa: F401  (shard_rows: re-exported for callers)
from .optimizer import SparseDenseAdam

TABLE_NAMES = ("item_embedding", "user_embedding", "item_dst_embedding")


def dist_info(accelerator=None):
    """(rank, world) of this process: an Accelerate-style object if one was passed (trainer.py:21-40 gets one), else the
    default torch.distributed group, else (0, 1)."""
    if accelerator is not None and hasattr(accelerator, "num_processes"):
        return int(accelerator.process_index), int(accelerator.num_processes)
    if dist.is_loopback(accelerator):
        return accelerator.rank, accelerator.world
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def dist_group(accelerator=None):
    """the process group the ranks of `accelerator` talk through: its ``process_group`` attribute when it has one (a
    pgroup.LoopbackGroup: W rank threads in this process), else None = torch.distributed's default group"""
    if dist.is_loopback(accelerator):
        return accelerator
    return getattr(accelerator, "process_group", None)


def owned_ids(n_rows, rank, world):
    """(first id, count) of the global ids > 0 that live on `rank`: first, first + W, ...; their local rows are consecutive,
    starting at first // W + 1."""
    first = rank if rank > 0 else world
    cnt = (n_rows - 1 - first) // world + 1 if n_rows - 1 >= first else 0
    return first, cnt


def extract_shard(full, rank, world):
    """[N, d] -> this rank's [shard_rows(N, W), d] (row 0 = padding row, zeros)."""
    N = full.shape[0]
    shard = torch.zeros((shard_rows(N, world),) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    first, cnt = owned_ids(N, rank, world)
    if cnt:
        shard[first // world + 1: first // world + 1 + cnt] = full[first::world][:cnt]
    return shard


class _Look:
    """ids-only state of one batch (plans, packed ids, owner-side plan, re-indexed lookups): made a step ahead on the plan stream"""
    __slots__ = ("key", "tabs", "event", "keep", "waited")


def native_transport_selftest(rank, world, group, dev):
    """One tiny all-to-all and all-reduce through the library's communicators (ops.comm_init) against torch.distributed's: the first RCCL
    traffic of the process goes through a check, not through the first training step.  Every rank gets the same verdict (a MIN
    all-reduce).  A mismatch RAISES on every rank: the torch.distributed route is several times slower and nothing but
    ``config.parallelism`` would say so.  UR_ALLOW_TD_FALLBACK=1: warn, destroy the library's communicators and go on over
    torch.distributed instead."""
    W, cap, d = world, 4, 8
    send = (torch.arange(W * cap * d, device=dev, dtype=torch.float32) + 1000.0 * rank).reshape(W * cap, d)
    got = torch.empty_like(send)
    ops.shard_exchange_grads(send.clone(), torch.arange(W * cap, device=dev, dtype=torch.int32), W, cap, send.clone(), grads_in=got, transport=True)
    want = torch.empty_like(send)
    dist.all_to_all_single(want, send.clone(), group=group)
    got[::cap] = want[::cap]          # (slot 0 of every block carries the flag row: not part of the comparison)
    s = torch.full((16,), float(rank + 1), device=dev)
    ops.comm_all_reduce_sum(s)
    ids = (torch.arange(W * cap, device=dev, dtype=torch.int32) + 100 * rank).contiguous()      # the second communicator's all-to-all (work issued a step ahead)
    got_i = ops.comm_all_to_all(ids, torch.empty_like(ids), W, ahead=True, kind="ids")
    want_i = torch.empty_like(ids)
    dist.all_to_all_single(want_i, ids.clone(), group=group)
    ok = torch.tensor([1.0 if torch.equal(got, want) and torch.equal(got_i, want_i) and float(s[0]) == W * (W + 1) / 2 else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if float(ok) != 1.0:
        ops.comm_destroy()
        if os.environ.get("UR_ALLOW_TD_FALLBACK", "0") not in ("", "0"):
            warnings.warn("unirec_amd: the library's RCCL transport failed its self-test; using torch.distributed for the row exchange "
                          "(UR_ALLOW_TD_FALLBACK is set)")
            return False
        raise RuntimeError("unirec_amd: the library's RCCL transport failed its self-test (all-to-all / all-reduce through ur_comm_* != "
                           "torch.distributed's).  Set UR_ALLOW_TD_FALLBACK=1 to train over torch.distributed collectives instead "
                           "(slower), or UR_NATIVE_TRANSPORT=0 to skip the library's communicators altogether.")
    return True


class ShardedSparseDenseAdam(SparseDenseAdam):
    """SparseDenseAdam for world > 1 (see the module docstring).  The model keeps its SHARD under each table's ``weight``;
    ``train_step`` swaps the compact table of the batch's rows in for the duration of the model's forward_backward.

    The step (round 3): every exchange has FIXED capacity (``cap`` slots per (source, owner) pair, include/unirec_amd.h
    ur_shard_exchange_*), so there is no per-step count exchange, no host synchronisation, and every buffer is allocated once.
      plan stream, one step AHEAD: id sort by (owner, row) -> pack -> all-to-all #1 (ids) -> owner-side merge plan -> re-indexed lookups
                   -> (round 4, _prefetch_rows) fix-up plan, zero-gradient steps of the requested rows the step in flight does not
                   touch, gather + all-to-all #2 (rows) of the NEXT batch
      main stream: fix-up exchange (the few rows the previous step's update changed after they were fetched; without a lookahead: owner
                   rows caught up, then the full all-to-all #2) -> the model's own forward_backward on the compact table -> row-gradient
                   reduce into the exchange slots -> all-to-all #3 (slot 0 of every block carries the rank's NaN / overflow flags and
                   loss: an all-gather riding along) -> owner-side reduce in source-rank order (+ the step flags) -> row update
      encoder's side stream (no gradient clipping): dense-gradient reductions -> all-reduce (second communicator) -> dense update, joined
                   by the next forward pass after its first launch (SparseDenseAdam's late join)
    With ``grad_clip`` the global norm needs every gradient first: the dense half runs on the main stream behind ONE flat all-reduce
    that also carries the owners' row-gradient norms.  A capacity overflow (a rank asks one owner for more than cap - 1 rows: skewed
    ids) skips the step on every rank, like a NaN loss; the host sees the flag two steps later, doubles the capacity and trains that
    batch again.  Transport: the library's RCCL communicators (ops.comm_init) when the process group is nccl, else torch.distributed
    on the packed blocks (gloo: CPU-staged -- the routing tests)."""

    def __init__(self, model, rank, world, group=None, sync_init=True, full_rows=None, cap_slack=1.25, fix_cap_min=64, **kw):
        """full_rows (optional): {table: N} for tables the model ALREADY holds as this rank's shard (shard_rows(N, W) rows,
        initialised per rank): nothing is broadcast or cut for them -- how a 100 M-row table is brought up without ever
        existing in one piece (bench.py); by default the model's full tables are broadcast from rank 0 and cut here.
        cap_slack: capacity per (source, owner) pair = cap_slack x lookups / world (uniform ids need ~1.1)."""
        self.rank, self.world = rank, world
        self.xchg = RowExchange(world, rank, group)
        self.full_rows = {}
        full_rows = dict(full_rows or {})
        dev = model.device
        if world > 64:
            raise NotImplementedError("row exchange: at most 64 ranks (the owner-side plan is a 64-way merge)")
        if sync_init and world > 1:      # what DDP's wrap-time parameter broadcast does (trainer.py:67)
            self._broadcast(model.dense_flat.data)
            for n, p in model.named_parameters():
                if n in ("user_bias", "item_bias"):
                    self._broadcast(p.data)
        for name in TABLE_NAMES:
            if not hasattr(model, name) or (name == "item_dst_embedding" and model.item_dst_embedding is model.item_embedding):
                continue
            w = getattr(model, name).weight
            if name in full_rows:
                if w.shape[0] != shard_rows(full_rows[name], world):
                    raise ValueError(f"{name}: a pre-sharded table of {full_rows[name]} rows has {shard_rows(full_rows[name], world)} "
                                     f"rows per rank, the model holds {w.shape[0]}")
                self.full_rows[name] = int(full_rows[name])
                w.data[0].zero_()
                continue
            self.full_rows[name] = w.shape[0]
            if sync_init and world > 1:
                self._broadcast(w.data)
            w.data = extract_shard(w.data, rank, world) if world > 1 else w.data
        super().__init__(model, **kw)
        self.inv_w = torch.full((1,), 1.0 / world, dtype=torch.float32, device=dev)
        self.cap_slack = float(cap_slack)
        self.fix_cap_min = int(fix_cap_min)
        self._cap_scale = 1               # doubled after a capacity overflow
        # native transport: RCCL through the library's own communicators, on whatever stream the step is on
        self._native = False
        self._loop = self.xchg.loop      # the in-process loopback group (pgroup.LoopbackGroup), or None
        if (world > 1 and dev.type == "cuda" and dist.get_backend(group) == "nccl" and ops.comm_world() >= 0
                and os.environ.get("UR_NATIVE_TRANSPORT", "1") not in ("", "0")):
            self._native = bool(ops.comm_init(rank, world, group)) and self._native_selftest(dev)
        elif (world == 1 and dev.type == "cuda" and self._loop is None and os.environ.get("UR_NATIVE_W1", "0") not in ("", "0")
              and os.environ.get("UR_NATIVE_TRANSPORT", "1") not in ("", "0") and ops.comm_world() >= 0):
            # one rank through the library's RCCL communicators (every exchange a send / recv to itself, the all-reduce the identity):
            # what a one-GPU box can execute of the multi-GPU transport (bench.py --gpus 1 --supervised; there is no second route to
            # compare with at world 1, so no self-test)
            self._native = bool(ops.comm_init(0, 1))
        self._bufs = {}                   # (table, n, n_a, parity) -> preallocated exchange buffers
        self._look = None                 # _Look of the next batch (plan stream)
        self._parity = 0
        self._out4 = [torch.zeros(4, dtype=torch.float32, device=dev) for _ in range(4)]   # per-step flags (ring: the host reads them late)
        self._flag_pool = []              # pinned host copies of a step's flags, recycled
        self._pending = collections.deque()   # (step, event, host flags, batch, capacity scale of the step): read by _check_overflow
        self._replaying = False
        self.n_overflow = 0
        # fullsoftmax (the loss of the reference's own DDP test, tests/test_model/run_ddp_test.sh:28): every rank scores ALL ranks' users
        # against the rows it owns; the table gradient of a shard is complete on its owner (dense over the shard, no exchange)
        # rows of the next batch fetched a step ahead (_prefetch_rows); UR_PREFETCH_ROWS=0: in the step itself (rounds 1-3).  Not with
        # fullsoftmax: its dense update moves every row of the item shard in every step
        # UR_DENSE_SIDE=0 (rung "native-1comm-1stream" of bench.py's fallback ladder): the dense half -- all-reduce + update -- on the main
        # stream behind the encoder's reductions, as with gradient clipping; with UR_COMM_SINGLE=1 every collective of a step then goes
        # through ONE communicator on ONE stream, in program order: the shape of the reference's own DDP step
        self._dense_on_side = os.environ.get("UR_DENSE_SIDE", "1") not in ("", "0")
        self.plan_delay_us = 0            # test aid (see prefetch)
        self.prefetch_rows = os.environ.get("UR_PREFETCH_ROWS", "1") not in ("", "0") and model.loss_type != "fullsoftmax"
        self._fs_dgrad = None
        if model.loss_type == "fullsoftmax":
            if "item_embedding" not in self.tables or self.full_rows["item_embedding"] < world:
                raise NotImplementedError("fullsoftmax under row-sharding needs an item table of at least `world` rows")
            self._fs_dgrad = torch.zeros_like(self.tables["item_embedding"]["w"])
            self._fs_w = torch.full((1,), float(world), dtype=torch.float32, device=dev)
            object.__setattr__(model, "_fs_shard_step", self._fs_shard_step)

    def _native_selftest(self, dev):
        return native_transport_selftest(self.rank, self.world, self.xchg.group, dev)

    # ------------------------------------------------------------------ collectives on top of RowExchange
    def _broadcast(self, t, chunk=1 << 26):
        flat = t.view(-1)
        staged = self.xchg._staged(flat)
        for o in range(0, flat.numel(), chunk):
            piece = flat[o:o + chunk]
            if staged:
                h = piece.cpu()
                dist.broadcast(h, src=0, group=self.xchg.group)
                piece.copy_(h)
            else:
                dist.broadcast(piece, src=0, group=self.xchg.group)

    _LOOP_KIND = {"a2a_ids": ("ids", True), "a2a_fix_slots": ("ids", True), "a2a_rows": ("rows", None), "a2a_fix_rows": ("rows", False),
                  "a2a_row_grads": ("grads", False)}

    def _a2a(self, send, recv, label, ahead=None):
        """equal-split all-to-all of a packed block when the library's RCCL communicators are not in use: the in-process loopback
        transport (stream-ordered copies, communicator index as the RCCL route would pick it: `ahead` = issued a step ahead), else
        torch.distributed"""
        if self.world == 1:
            assert recv.data_ptr() == send.data_ptr()     # (aliased at world 1: see _buffers)
            return recv
        if self._loop is not None:
            kind, dflt = self._LOOP_KIND[label]
            return self._loop.all_to_all(send, recv, ahead=dflt if ahead is None else ahead, kind=kind)
        recv.copy_(self.xchg.all_to_all_equal(send, label=label))
        return recv

    def _all_reduce(self, t):
        if self.world == 1:
            return t
        if self._native:
            return ops.comm_all_reduce_sum(t)
        if self._loop is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
            return self._loop.all_reduce_sum(t)
        r = self.xchg.all_reduce_sum(t)
        if r is not t:
            t.copy_(r)
        return t

    # ------------------------------------------------------------------ id plans
    def _table_inputs(self, batch):
        """-> {table: (field_a, field_b, ids_a int32 | None, ids_b int64 | None)}"""
        spec = self.model.lookup_tables()
        out, seen = {}, {}
        for name, (ka, kb) in spec.items():
            if name not in self.tables:
                continue
            ta = batch.get(ka) if ka else None
            tb = batch.get(kb) if kb else None
            if ta is None and tb is None:
                continue
            for k in (ka if ta is not None else None, kb if tb is not None else None):
                if k is not None and seen.setdefault(k, name) != name:
                    raise NotImplementedError(f"batch field {k!r} indexes two sharded tables ({seen[k]}, {name})")
            a = ta.reshape(-1).to(torch.int32).contiguous() if ta is not None else None
            b = tb.reshape(-1).to(torch.int64).contiguous() if tb is not None else None
            out[name] = (ka if ta is not None else None, kb if tb is not None else None, a, b)
        return out

    def _capacity(self, n):
        """slots per (source, owner) pair for a batch of n lookups of one table; slot 0 of every block is reserved"""
        W = self.world
        if W == 1:
            return n + 1
        cap = int(self.cap_slack * self._cap_scale * n / W) + 2
        cap = (cap + 63) // 64 * 64
        return min(cap, (n + 1 + 63) // 64 * 64)

    def _capacity2(self, n, cap):
        """slots per (owner, requester) pair of the FIX-UP exchange (rows of the next batch that the step in flight updates): an eighth
        of the unclamped main capacity -- uniform ids over a large table need a handful -- growing with it after an overflow, up to cap;
        a multiple of `fix_cap_min` (64; the tests shrink it to make the fix-up overflow on its own)."""
        W, q = self.world, self.fix_cap_min
        want = int(self.cap_slack * self._cap_scale * n / W / 8) if W > 1 else n // 8 * self._cap_scale
        return min(cap, max(q, (want + q - 1) // q * q))

    def _buffers(self, name, n, n_a, d, parity):
        key = (name, n, n_a, parity, self._cap_scale)
        bf = self._bufs.get(key)
        if bf is None:
            dev, W = self.model.device, self.world
            cap = self._capacity(n)
            cap2 = self._capacity2(n, cap)
            i32 = dict(dtype=torch.int32, device=dev)
            f32 = dict(dtype=torch.float32, device=dev)
            send_ids = torch.empty(W * cap, **i32)
            rows_ws = torch.empty(W * cap, d, **f32)
            slot2, rows2 = torch.empty(W * cap2, **i32), torch.empty(W * cap2, d, **f32)
            bf = dict(cap=cap, plan=ops.rows_plan_alloc(n, n_a, dev), counts=torch.empty(W, **i32), send_ids=send_ids,
                      # (world 1: every exchange is the identity -- the receive buffers ARE the send buffers, nothing is copied)
                      recv_ids=torch.empty(W * cap, **i32) if W > 1 else send_ids, slot=torch.zeros(n, **i32), uos=torch.empty(W * cap, **i32),
                      flags=torch.zeros(4, **i32), own=ops.rows_plan_alloc(W * cap, W * cap, dev),
                      idx_a=torch.empty(n_a, **i32) if n_a else None,
                      idx_b=torch.empty(n - n_a, dtype=torch.int64, device=dev) if n > n_a else None,
                      # the rows themselves, per parity: batch t + 1's are fetched on the plan stream while step t computes on batch t's
                      rows_ws=rows_ws, compact=torch.empty(W * cap, d, **f32) if W > 1 else rows_ws,
                      # fix-up exchange (rows of this batch that the previous step's update changed after they were fetched)
                      cap2=cap2, req2=torch.zeros(W * cap2, **i32), slot2=slot2, slot2_recv=torch.empty(W * cap2, **i32) if W > 1 else slot2,
                      rows2=rows2, rows2_recv=torch.empty(W * cap2, d, **f32) if W > 1 else rows2, split=None,
                      fix_cnt=torch.zeros(W, **i32))
            self._bufs[key] = bf
        skey = (name, n, "step", self._cap_scale)
        sb = self._bufs.get(skey)
        if sb is None:
            dev, W, cap = self.model.device, self.world, bf["cap"]
            f32 = dict(dtype=torch.float32, device=dev)
            send_grads = torch.empty(W * cap, d, **f32)
            sb = dict(send_grads=send_grads, grads_in=torch.empty(W * cap, d, **f32) if W > 1 else send_grads)
            self._bufs[skey] = sb
        return bf, sb

    def _prepare(self, batch, parity, prev=None):
        """everything of a step that depends on the ids only, on the CURRENT stream -> {table: state}.  prev (the tabs of the step in
        flight; lookahead only): the ROWS are fetched here as well, a step ahead (_prefetch_rows)."""
        W, tabs = self.world, {}
        ahead = prev is not None
        for name, (ka, kb, a, b) in self._table_inputs(batch).items():
            st = self.tables[name]
            n_a = a.numel() if a is not None else 0
            n = n_a + (b.numel() if b is not None else 0)
            bf, sb = self._buffers(name, n, n_a, st["w"].shape[1], parity)
            cap = bf["cap"]
            pl, counts = ops.rows_plan_sharded(a, b, self.full_rows[name], W, out=(bf["plan"], bf["counts"]), want_counts=False)
            bf["flags"].zero_()
            ops.shard_exchange_ids(pl, counts, st["w"].shape[0], W, cap, bf["send_ids"], bf["slot"], bf["uos"], bf["flags"],
                                   recv_ids=bf["recv_ids"], transport=self._native)
            if not self._native:
                self._a2a(bf["send_ids"], bf["recv_ids"], "a2a_ids")
            own = ops.rows_plan_merge(bf["recv_ids"], [cap] * W, out=bf["own"])
            ops.compact_index(pl, bf["slot"], out=(bf["idx_a"], bf["idx_b"]))
            filt = None
            if st["last"] is not None and self.wd == 0.0 and not ahead:
                filt = ops.rows_filter_touched(own, st["last"])     # (rows first touched by the step in flight need no catch-up)
            tabs[name] = dict(ka=ka, kb=kb, pl=pl, own=own, filt=filt, bf=bf, sb=sb, cap=cap, ids=(a, b), caught_up=None,
                              rows_ready=False, hot=None)
        if ahead:
            self._prefetch_rows(tabs, prev)
        return tabs

    def _prefetch_rows(self, tabs, prev):
        """The row exchange (all-to-all #2) of the NEXT batch, on the plan stream under the step in flight (step self.t, owner-side plans
        `prev`).  What that step still changes are the rows ITS update writes -- prev's owner-side unique rows; everything else the next
        batch asks for can travel now:
          * (lazy_dense) the requested rows the step in flight does NOT touch take their zero-gradient steps up to and including it
            here (it gives them a zero gradient by construction): `cold`; the others, `hot`, are current once that step's update ran;
          * all requested rows are gathered and exchanged (second communicator: work issued a step ahead never queues in front of the
            step's own exchanges); the copies of hot rows are stale and are replaced at the head of the next step by a FIX-UP exchange
            of cap2 << cap slots per pair, planned here from the ids alone (ur_shard_fixup_plan; its slot list travels now too).
        A pair with more than cap2 hot rows raises the batch's overflow flag: skipped everywhere, capacities doubled, re-trained."""
        W = self.world
        lazy = self.table_mode == "lazy_dense"
        cfg = self._cfg(self.t + 1)
        for name, c in tabs.items():
            st, bf, cap, cap2 = self.tables[name], c["bf"], c["cap"], c["bf"]["cap2"]
            pown = prev[name]["own"] if name in prev else None
            ops.shard_fixup_plan(bf["recv_ids"], W, cap, pown, cap2, bf["req2"], bf["slot2"], bf["flags"], bf["fix_cnt"])
            if W > 1:
                if self._native:
                    ops.comm_all_to_all(bf["slot2"], bf["slot2_recv"], W, ahead=True, kind="ids")
                else:
                    self._a2a(bf["slot2"], bf["slot2_recv"], "a2a_fix_slots")
            if lazy and st["last"] is not None:
                bf["split"] = ops.rows_split_hot(c["own"], st["last"] if self.wd == 0.0 else None, pown, out=bf["split"])
                cold, c["hot"] = bf["split"][0], bf["split"][1]
                ops.lazy_adam_catchup(cfg, st["w"], st["m"], st["v"], st["last"], cold)
                c["caught_up"] = self.t
            ops.shard_exchange_rows(st["w"], bf["recv_ids"], W, cap, bf["rows_ws"], compact=bf["compact"], transport=False)
            if W > 1:
                if self._native:
                    ops.comm_all_to_all(bf["rows_ws"], bf["compact"], W, ahead=True)
                else:
                    self._a2a(bf["rows_ws"], bf["compact"], "a2a_rows", ahead=True)
            c["rows_ready"] = True

    def prefetch(self, batch, cur_tabs=None):
        """ids-only half of the NEXT batch's step on the plan stream, all-to-all #1 included: nothing of it is left for the step itself.
        cur_tabs (the step in flight's state): its ROWS travel now as well (_prefetch_rows), unless switched off."""
        if batch is None or self.model.device.type != "cuda":
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.model.device)
        ops.stream_wait_stream(self._side, main)      # the ids may still be in flight on the main stream; `last` is being updated there
        if self.plan_delay_us:      # test aid: this rank's plan stream runs late (tests/test_loopback_gpu.py)
            ops.debug_delay(self.plan_delay_us, self._side)
        look = _Look()
        self._parity ^= 1
        with torch.cuda.stream(self._side):
            look.tabs = self._prepare(batch, self._parity, prev=cur_tabs if self.prefetch_rows else None)
            look.event = torch.cuda.Event()
            look.event.record(self._side)
        look.key, look.keep, look.waited = self._batch_key(batch), batch, False
        self._look = look

    @staticmethod
    def _batch_key(batch):
        return tuple((k, v.data_ptr(), v.numel()) for k, v in sorted(batch.items()) if torch.is_tensor(v))

    def _catchup(self, tabs, target_t):
        """bring the owner rows of `tabs` to the state after step target_t (lazy_dense): the rows with history only when the filter exists"""
        cfg = self._cfg(target_t + 1)
        for name, c in tabs.items():
            st = self.tables[name]
            if st["last"] is not None and c["caught_up"] != target_t:
                ops.lazy_adam_catchup(cfg, st["w"], st["m"], st["v"], st["last"], c["filt"] if c["filt"] is not None else c["own"])
                c["caught_up"] = target_t

    # ------------------------------------------------------------------ fullsoftmax over the sharded catalogue
    def _fs_rows(self):
        """(first local row, count) of the rows this rank scores: every global id it owns INCLUDING id 0 (rank 0's local row 1: the
        reference's logsumexp runs over all n in [0, n_items), recommender.py:47-50), never the shard's own padding row 0.  Row j of
        that range is global id j * W + rank."""
        N, W, r = self.full_rows["item_embedding"], self.world, self.rank
        if W == 1:
            return 0, N
        return 1, ((N - 1 - r) // W + 1 if N - 1 >= r else 0)

    def _fs_shard_step(self, user_emb, target, user_id):
        """model.forward_backward's fullsoftmax half (reco_abc.py:266-270 under DDP): all-gather the user vectors, per-shard
        (max, sum-exp, target score) partials, ONE small all-gather of those, the global logsumexp, then the backward over this rank's
        rows with it: the shard's table gradient is final here (summed over every rank's users), d_user is summed over the ranks.
        -> (loss_out [loss of THIS rank's users, B, guard, -], d_user [B, d]).  Gradients carry 1 / (B tau): the optimizer's 1 / W
        (DDP's mean) applies to them like to every other gradient of the step."""
        model, W, r, x = self.model, self.world, self.rank, self.xchg
        st = self.tables["item_embedding"]
        B = user_emb.shape[0]
        lo, cnt = self._fs_rows()
        rows = st["w"][lo:lo + cnt]
        gather = (lambda t: t) if W == 1 else x.all_gather_cat
        ue_all = gather(user_emb.contiguous())
        tgt_all = gather(target.to(torch.int64).contiguous())
        ltgt = tgt_all if W == 1 else torch.where(tgt_all % W == r, tgt_all // W, torch.full_like(tgt_all, -1))
        uid_all = ub_all = ib_rows = None
        if model.has_user_bias:      # the value per column travels (user_id may be a compact index on this rank)
            ub_all = gather(model.user_bias.data.reshape(-1)[user_id.to(torch.int64)].contiguous())
            uid_all = torch.arange(ub_all.numel(), dtype=torch.int64, device=ub_all.device)
        if model.has_item_bias:
            ib = model.item_bias.data.reshape(-1)
            ib_rows = ib if W == 1 else ib[r::W][:cnt].contiguous()
        part3, ws = ops.full_softmax_fwd_shard(ue_all, rows, ltgt, uid_all, ub_all, ib_rows, model.tau, model.SCORE_CLIP)
        parts = part3.view(1, 3, -1) if W == 1 else x.all_gather_cat(part3.view(1, 3, -1))
        lse, loss_out = ops.full_softmax_combine_shards(parts.contiguous(), r * B, B)
        d_user_all, d_ib = ops.full_softmax_bwd_shard(ue_all, rows, ltgt, lse, ws, self._fs_dgrad[lo:lo + cnt], uid_all, ub_all, ib_rows,
                                                      model.tau, model.SCORE_CLIP, d_loss=self._fs_w, zero_row0=(r == 0))
        d_user_all = self._all_reduce(d_user_all)
        if model.has_item_bias:
            g = torch.zeros_like(model.item_bias.data)
            if W == 1:
                g.view(-1).copy_(d_ib)
            else:
                g.view(-1)[r::W][:cnt] = d_ib
            model.item_bias.grad = g
        if model.has_user_bias:
            model.user_bias.grad = torch.zeros_like(model.user_bias.data)      # softmax is shift invariant along n: exactly 0
        return loss_out, d_user_all[r * B:(r + 1) * B].contiguous()

    def _update_rows(self, cfg, tabs, owner_grads, dense_shard, scale):
        for name, c in tabs.items():
            if name not in dense_shard and name in owner_grads:     # (not in owner_grads: updated by the fused owner-side launch already)
                st = self.tables[name]
                ops.sparse_adam_rows(cfg, st["w"], st["m"], st["v"], c["own"], owner_grads[name], st["last"], scale)
        for name, dg in dense_shard.items():      # fullsoftmax: every row of the shard moves -> plain dense rule on the shard
            st = self.tables[name]
            ops.dense_adam(cfg, st["w"], dg, st["m"], st["v"], scale)

    def _catchup_next(self):
        """tail of a step: the NEXT batch's owner rows take this step too (lazy_dense) -- unless the plan stream has done that already for
        the rows this step does not touch (_prefetch_rows): then there is nothing to do here, and nothing to wait for yet"""
        nxt = self._look
        if nxt is None or self.table_mode != "lazy_dense" or all(c["rows_ready"] for c in nxt.tabs.values()):
            return
        torch.cuda.current_stream().wait_event(nxt.event)
        nxt.waited = True
        self._catchup(nxt.tabs, self.t)

    # ------------------------------------------------------------------ the step
    def _check_overflow(self, drain=False):
        """The flags of the steps that finished at least two enqueues ago (long done: no stall), in step order; drain=True: of every
        step enqueued so far (flush / end of an epoch).  An overflow -> every rank skipped that step: double the capacity and train the
        batch again (every rank reads the same all-gathered flags at the same point of its loop: lockstep).  The step RIGHT AFTER an
        overflowed one ran under the same, too small capacity: its entry is next in the queue and is looked at in the same call -- the
        replay has advanced the step count past it."""
        if self._replaying:
            return
        while self._pending and (drain or self._pending[0][0] <= self.t - 1):
            step, ev, host, batch, scale_then = self._pending.popleft()
            if ev is not None:
                ev.synchronize()
            overflow = float(host[3]) > 0
            self._flag_pool.append(host)
            if not overflow:
                continue
            self.n_overflow += 1
            if scale_then == self._cap_scale:   # (not doubled yet by an earlier entry of the same capacity)
                self._cap_scale *= 2
            self._drop_look()             # made with the old capacity
            warnings.warn(f"row exchange: capacity overflow at step {step}; capacity x{self._cap_scale}, batch re-trained")
            self._replaying = True
            try:
                self.train_step(batch, None)
            finally:
                self._replaying = False

    def _join_look(self):
        """The current stream waits for the plan stream's work on the lookahead batch.  Since round 4 that work WRITES optimizer state
        (_prefetch_rows: the zero-gradient steps of the cold rows -- w, m, v, last), so whatever touches the tables on the current
        stream without adopting the lookahead (a replay, flush, a checkpoint) has to be ordered behind it first."""
        look = self._look
        if look is not None and not look.waited and look.event is not None:
            torch.cuda.current_stream().wait_event(look.event)
            look.waited = True

    def _drop_look(self):
        self._join_look()
        self._look = None

    def flush(self):
        self._check_overflow(drain=True)
        self._join_look()
        return super().flush()

    def state_dict(self):
        self._join_look()
        return super().state_dict()

    def mark_tables_current(self):
        self._join_look()
        return super().mark_tables_current()

    def train_step(self, batch, next_batch=None):
        """One optimisation step of the ONE model on this rank's batch (a dict of device tensors as the Trainer builds it).
        Returns the mean loss over the ranks (device scalar; trainer.py:353 gather_for_metrics(loss).mean())."""
        model, W = self.model, self.world
        if not model.training:
            model.train()
        cuda = model.device.type == "cuda"
        self._check_overflow()
        self.zero_grad()
        self.t += 1
        cfg = self._cfg(self.t)
        # ---- 1. ids-only state: adopt the lookahead (an event wait, unless the previous step's tail already waited) or make it now
        look, self._look = self._look, None
        tabs = None
        if look is not None:
            if not look.waited:
                torch.cuda.current_stream().wait_event(look.event)
            if look.key == self._batch_key(batch):
                tabs = look.tabs
        if tabs is None:
            self._parity ^= 1
            tabs = self._prepare(batch, self._parity)
        if next_batch is not None:
            self.prefetch(next_batch, tabs)
        # ---- 2. the rows.  Fetched a step ahead (lookahead): only the rows the previous step's update changed after that are re-sent
        # (fix-up exchange, cap2 slots per pair).  Otherwise: owner rows up to date, then the full exchange
        if self.t > 1:
            self._catchup({n: c for n, c in tabs.items() if not c["rows_ready"]}, self.t - 1)
        cbatch = dict(batch)
        for name, c in tabs.items():
            st, bf, sb = self.tables[name], c["bf"], c["sb"]
            if c["rows_ready"]:
                if c["hot"] is not None:   # current after the previous step's update -- or, had that step been skipped everywhere, after this
                    ops.lazy_adam_catchup(self._cfg(self.t), st["w"], st["m"], st["v"], st["last"], c["hot"])
                ops.shard_exchange_rows(st["w"], bf["req2"], W, bf["cap2"], bf["rows2"], compact=bf["rows2_recv"], transport=self._native)
                if not self._native:
                    self._a2a(bf["rows2"], bf["rows2_recv"], "a2a_fix_rows")
                ops.shard_fixup_apply(bf["compact"], bf["rows2_recv"], bf["slot2_recv"], W, c["cap"], bf["cap2"])
            else:
                ops.shard_exchange_rows(st["w"], bf["recv_ids"], W, c["cap"], bf["rows_ws"], compact=bf["compact"], transport=self._native)
                if not self._native:
                    self._a2a(bf["rows_ws"], bf["compact"], "a2a_rows", ahead=False)
            if c["ka"] is not None:
                cbatch[c["ka"]] = bf["idx_a"].view(batch[c["ka"]].shape)
            if c["kb"] is not None:
                cbatch[c["kb"]] = bf["idx_b"].view(batch[c["kb"]].shape)
        # ---- 3. the model's own forward / backward on the compact tables (bias vectors compacted the same way)
        swapped = []
        try:
            for name, c in tabs.items():
                p = getattr(model, name).weight
                swapped.append((p, p.data))
                p.data = c["bf"]["compact"]
            bias_ctx = self._compact_biases(tabs, batch, swapped)
            if "user_id" in cbatch and cbatch["user_id"].dtype != torch.int64:
                cbatch["user_id"] = cbatch["user_id"].to(torch.int64)
            kw = {k: cbatch[k] for k in ("user_id", "item_id", "label", "item_seq", "item_seq_len") if k in cbatch}
            model.forward_backward(**kw)
        finally:
            for p, data in reversed(swapped):
                p.data = data
        guard = getattr(model, "loss_guard", None)
        loss_buf = guard._base if guard is not None and guard._base is not None else None   # the loss kernels' [loss, n, guard, .]
        # ---- 4. row gradients: reduce per unique key, scatter to slots (+ this rank's flags in slot 0), owners sum in source-rank order
        owner_grads, out4 = {}, self._out4[self.t % 4]
        first = True
        fuse_first = self._fused_update and self.grad_clip is None
        if len(tabs) > 1:     # ONE flag row per step rides in the first table's exchange: it must say "overflow" for every table
            # (bitwise OR, not a sum: every consumer tests single bits of flags[0] -- two tables overflowing in one step must not read as "none")
            f0 = next(iter(tabs.values()))["bf"]["flags"]
            for c in list(tabs.values())[1:]:
                f0.bitwise_or_(c["bf"]["flags"])
        for name, c in tabs.items():
            ids_a, rows, ids_b, coef, vec, G = self._collect(name)
            sb, bf = c["sb"], c["bf"]
            d = bf["compact"].shape[1]
            # (the sums land in their exchange slots: no scatter pass; padding slots keep stale bytes that no owner reads)
            # (this rank's flag row rides in the reduce launch -- slot 0 of every block; the step flags of all ranks ride in the owner-side
            # reduce of the first table: two launches less per step than ur_shard_exchange_grads(NULL) + ur_shard_step_flags, which
            # remain in the ABI for callers that reduce elsewhere)
            ops.rows_reduce_riders(c["pl"], rows, coef.reshape(-1) if coef is not None else None, vec, G, d, W, c["cap"], out=sb["send_grads"],
                                   out_rows=bf["slot"], flag_rows=(loss_buf, bf["flags"]))
            if self._native:
                ops.comm_all_to_all(sb["send_grads"], sb["grads_in"], W, ahead=False, kind="grads")
            else:
                self._a2a(sb["send_grads"], sb["grads_in"], "a2a_row_grads")
            if first and fuse_first and not (name == "item_embedding" and self._fs_dgrad is not None):
                # nothing needs the owner-side row gradients between their reduction and the update (no clipping, no dense fold): the
                # update is the reduce launch's epilogue, and the scale it needs -- the step flags in the block it reduces -- is read there
                st = self.tables[name]
                ops.rows_reduce_update_owner(cfg, st["w"], st["m"], st["v"], c["own"], sb["grads_in"], W, c["cap"], out4, st["last"])
            elif not first and fuse_first and not (name == "item_embedding" and self._fs_dgrad is not None):
                st = self.tables[name]      # (a later table of the step: the flags are on the device already)
                ops.rows_reduce_update(cfg, st["w"], st["m"], st["v"], c["own"], sb["grads_in"], None, None, 1, st["last"], out4[0:1])
            else:
                owner_grads[name] = ops.rows_reduce_riders(c["own"], sb["grads_in"], None, None, 1, d, W, c["cap"],
                                                           zero_tail=self.grad_clip is not None, step_flags_out4=out4 if first else None)
            first = False
        dense_shard = {}
        if self._fs_dgrad is not None:      # fullsoftmax: the encoder's row-sparse part of the item table's gradient folds into the dense one
            dense_shard["item_embedding"] = self._fs_dgrad
            if "item_embedding" in owner_grads:
                ops.rows_scatter_add(tabs["item_embedding"]["own"], owner_grads["item_embedding"], self._fs_dgrad)
                self._fs_dgrad[0].zero_()   # (the shard's padding row: requests for id 0 land there)
        scale = out4[0:1]
        flags_copied = False

        def copy_flags():   # the step's flags -> pinned host memory (read two steps later); on the side stream when there is one:
            # a device-to-host copy in the main queue is 4 us of copy and 8 us of idle queue behind it
            host = self._flag_pool.pop() if self._flag_pool else (torch.zeros(4, dtype=torch.float32).pin_memory() if cuda else torch.zeros(4))
            host.copy_(out4, non_blocking=True)
            ev = None
            if cuda:
                ev = torch.cuda.Event()
                ev.record()
            self._pending.append((self.t, ev, host, batch, self._cap_scale))
        # ---- 5./6. updates
        side = None
        if self.grad_clip is None:
            # rows first (nothing to wait for), then the next batch's owner rows take this step too -- under the dense-gradient stream
            self._update_rows(cfg, tabs, owner_grads, dense_shard, scale)
            self._catchup_next()
            g = getattr(model, "_deferred_dense_grad", None)
            if g is not None and g.numel() and not self.extra and not bias_ctx and self._dense_on_side:
                side = ops.sasrec_side_stream()
            if side is not None:
                # the dense half behind the encoder's own reductions on ITS stream: all-reduce (second communicator) + update; the next
                # forward pass joins after its first launch.  `scale` was made on the main stream: the side stream waits for it
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    if cuda:
                        copy_flags()
                        flags_copied = True
                    self._all_reduce(g)
                    ops.dense_adam(cfg, model.dense_flat.data, g, self.dense_m, self.dense_v, scale)
                ops.sasrec_side_publish(late=next_batch is not None, hold=(g, out4) + tuple(getattr(model, "_deferred_reads", ())))
                model.dense_flat.grad = g
                object.__setattr__(model, "_deferred_dense_grad", None)
        if not flags_copied:
            copy_flags()
        if side is None:
            self._dense_main(cfg, bias_ctx, owner_grads, tabs, scale, dense_shard)
        object.__setattr__(model, "loss_guard", None)
        model.sparse_grads.clear()
        model.dense_flat.grad = None
        for p in self.extra:
            p.grad = None
        return out4[1].clone()        # (out4 is a slot of a four-step ring: callers keep the loss for a whole epoch)

    def _dense_main(self, cfg, bias_ctx, owner_grads, tabs, scale, dense_shard=None):
        dense_shard = dense_shard or {}
        """the dense half on the main stream: ONE flat all-reduce (dense gradients, bias gradients and -- with clipping -- the owners'
        row-gradient norms), the clip coefficient, the updates (and, with clipping, the row updates that waited for the norm)"""
        model = self.model
        model.finish_backward()
        dgrad = model.dense_flat.grad if model.dense_flat.grad is not None else torch.zeros_like(model.dense_flat.data)   # (MF: no dense parameters)
        pieces = [dgrad.reshape(-1)]
        for p, gid in bias_ctx:       # compact bias gradient -> the bias vector's own index space
            g = torch.zeros_like(p.data)
            if p.grad is not None:
                g.index_add_(0, gid, p.grad.reshape(-1)[: gid.numel()])
            pieces.append(g)
        for p in self.extra:
            if not any(p is q for q, _ in bias_ctx):
                pieces.append(p.grad.reshape(-1) if p.grad is not None else torch.zeros_like(p.data))
        tail = []
        if self.grad_clip is not None:   # sum of squares of THIS rank's owned row gradients rides along
            ss = self._scalars[0:1]
            first = True
            for og in [g for n, g in owner_grads.items() if n not in dense_shard] + [dense_shard[n] for n in dense_shard]:
                ops.sumsq(og, ss, accumulate=not first, ws=self._sumsq_ws)
                first = False
            if first:
                ss.zero_()
            tail = [ss]
        flat = torch.cat(pieces + tail) if len(pieces) + len(tail) > 1 else pieces[0]
        flat = self._all_reduce(flat)
        if self.grad_clip is not None:
            body = flat[: flat.numel() - 1]
            ss_all = self._scalars[2:3]
            ops.sumsq(body, ss_all, accumulate=False, ws=self._sumsq_ws)
            total = (ss_all + flat[flat.numel() - 1:]) * (self.inv_w * self.inv_w)       # squared norm of the AVERAGED gradient
            coef = self._scalars[1:2]
            ops.clip_coef(total, self.grad_clip, coef)
            scale = torch.where(scale < 0, scale, scale * coef)
            self._update_rows(cfg, tabs, owner_grads, dense_shard, scale)
            self._catchup_next()
        o = model.dense_flat.numel()
        if o:
            ops.dense_adam(cfg, model.dense_flat.data, flat[:o].contiguous(), self.dense_m, self.dense_v, scale)
        for p, (m, v) in zip(self._bias_order(bias_ctx), self._bias_state(bias_ctx)):
            n = p.numel()
            ops.dense_adam(cfg, p.data, flat[o:o + n].contiguous(), m, v, scale)
            o += n

    # ---- bias vectors under compacted ids -------------------------------------------------------------------------------
    def _global_ids(self, c):
        """global id of every compact row (= slot) of a table's fixed-capacity exchange (padding slots: id 0)."""
        local = c["bf"]["send_ids"].to(torch.int64)
        if self.world == 1:
            return local
        owner = torch.arange(local.numel(), device=local.device) // c["cap"]
        return torch.where(local > 0, (local - 1) * self.world + owner, torch.zeros_like(local))

    def _compact_biases(self, tabs, batch, swapped):
        """item_bias is indexed by item_id, user_bias by user_id: when that field was re-indexed to compact rows, the model must
        see the bias entries of those rows.  -> [(parameter, global ids of its compact entries)]"""
        model, out = self.model, []
        for pname, field in (("item_bias", "item_id"), ("user_bias", "user_id")):
            if not getattr(model, "has_" + pname, False):
                continue
            p = getattr(model, pname)
            for name, c in tabs.items():
                if field in (c["ka"], c["kb"]):
                    gid = self._global_ids(c)
                    swapped.append((p, p.data))
                    p.data = p.data[gid].contiguous()
                    out.append((p, gid))
                    break
        return out

    def _bias_order(self, bias_ctx):
        return [p for p, _ in bias_ctx] + [p for p in self.extra if not any(p is q for q, _ in bias_ctx)]

    def _bias_state(self, bias_ctx):
        state = {id(p): s for p, s in zip(self.extra, self.extra_state)}
        return [state[id(p)] for p in self._bias_order(bias_ctx)]

    # ------------------------------------------------------------------ evaluation-time access to rows
    @torch.no_grad()
    def compact_batch(self, batch):
        """For evaluation forwards: fetch the rows `batch` looks up (the same fixed-capacity exchange as a training step, on the current
        stream).  -> (re-indexed batch, restore()); between the call and restore() the model's tables (and bias vectors) are the compact
        ones.  Call flush() first in lazy_dense mode.  Evaluation is not the hot path: the overflow flag is checked on the spot (one
        host synchronisation) and the capacity doubled until the batch fits."""
        model, W = self.model, self.world
        if self._look is not None and not self._look.waited:   # (a prefetched training batch: its plan-stream work reads `last` and the tables)
            torch.cuda.current_stream().wait_event(self._look.event)
            self._look.waited = True
        while True:
            tabs = self._prepare(batch, 2)          # evaluation has buffers of its own (key 2): a pending lookahead keeps parity 0 / 1
            ovf = torch.stack([c["bf"]["flags"][0:1].to(torch.float32) for c in tabs.values()]).sum().reshape(1)
            if float(self._all_reduce(ovf)) == 0:
                break
            self._cap_scale *= 2
            self.n_overflow += 1
        cbatch, swapped = dict(batch), []
        for name, c in tabs.items():
            st, bf, sb = self.tables[name], c["bf"], c["sb"]
            compact = torch.empty_like(bf["compact"])      # (the caller keeps it until restore(): not the step's buffer)
            ops.shard_exchange_rows(st["w"], bf["recv_ids"], W, c["cap"], bf["rows_ws"] if W > 1 else compact, compact=compact,
                                    transport=self._native)
            if not self._native and W > 1:
                self._a2a(bf["rows_ws"], compact, "a2a_rows", ahead=False)
            if c["ka"] is not None:
                cbatch[c["ka"]] = bf["idx_a"].clone().view(batch[c["ka"]].shape)
            if c["kb"] is not None:
                cbatch[c["kb"]] = bf["idx_b"].clone().view(batch[c["kb"]].shape)
            p = getattr(model, name).weight
            swapped.append((p, p.data))
            p.data = compact
        self._compact_biases(tabs, batch, swapped)
        if "user_id" in cbatch and cbatch["user_id"].dtype != torch.int64:
            cbatch["user_id"] = cbatch["user_id"].to(torch.int64)

        def restore():
            for p, data in reversed(swapped):
                p.data = data
        return cbatch, restore

    # ------------------------------------------------------------------ full-item ranking over the sharded catalogue
    def local_history(self, hist_ptr, hist_sorted):
        """CSR history (global ids, ascending per user) -> the same CSR restricted to this rank's items, in local row ids."""
        if hist_ptr is None or self.world == 1:
            return hist_ptr, hist_sorted
        W, r = self.world, self.rank
        own = (hist_sorted % W == r) & (hist_sorted > 0)
        csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=own.device), own.to(torch.int64).cumsum(0)])
        return csum[hist_ptr].contiguous(), (hist_sorted[own] // W + 1).to(torch.int32).contiguous()

    @torch.no_grad()
    def full_item_ranks(self, user_emb, target, user_id=None, local_hist=(None, None)):
        """one_vs_all rank (Evaluator.evaluate_with_full_items semantics, as ops.full_rank) of this rank's rows over the row-SHARDED
        catalogue: all-gather the user vectors, every rank counts on its own shard, two small all-reduces (thresholds, counts).
        No score leaves a GPU, the table is never gathered.  Collective; ranks may bring different row counts.  -> int32[B]."""
        W, r, xchg = self.world, self.rank, self.xchg
        st = self.tables["item_embedding"]
        N = self.full_rows["item_embedding"]
        B = user_emb.shape[0]
        Bmax = B
        if W > 1:      # common row count (the last batch of an epoch may be short on one rank)
            box = [None] * W
            dist.all_gather_object(box, B, group=xchg.cpu_group or xchg.group)
            Bmax = max(box)
        target = target.reshape(B, -1)[:, 0].to(torch.int64)
        if Bmax > B:
            user_emb = torch.cat([user_emb, torch.zeros(Bmax - B, user_emb.shape[1], device=user_emb.device)])
            target = torch.cat([target, torch.ones(Bmax - B, dtype=torch.int64, device=target.device)])
            if user_id is not None:
                user_id = torch.cat([user_id, torch.zeros(Bmax - B, dtype=user_id.dtype, device=user_id.device)])
        ue_all, tgt_all = xchg.all_gather_cat(user_emb.contiguous()), xchg.all_gather_cat(target.contiguous())
        uid_all = xchg.all_gather_cat(user_id.to(torch.int64).contiguous()) if user_id is not None else None
        ltgt = tgt_all if W == 1 else torch.where(tgt_all % W == r, tgt_all // W + 1, torch.full_like(tgt_all, -1))
        hp, hs = local_hist if uid_all is not None else (None, None)
        if W == 1:
            n_rows, excl = N, -1
        elif r == 0:      # rank 0 holds ids W, 2W, .. at local rows 2.. (row 1 is the slot id 0 would take: never an item)
            n_rows, excl = (N - 1) // W + 2, 1
        else:
            n_rows, excl = ((N - 1 - r) // W + 1 if N - 1 >= r else 0) + 1, -1
        bias_local = None
        if getattr(self.model, "has_item_bias", False):
            bias_local = extract_shard(self.model.item_bias.data.view(-1, 1), r, W).view(-1).contiguous() if W > 1 else self.model.item_bias.data
        thr = xchg.all_reduce_sum(ops.full_rank_shard(1, ue_all, st["w"], ltgt, item_bias_local=bias_local, n_rows=n_rows))
        part = ops.full_rank_shard(2, ue_all, st["w"], ltgt, thr=thr, user_id=uid_all, hist_ptr=hp, hist_sorted_local=hs,
                                   item_bias_local=bias_local, n_rows=n_rows, excl_row=excl)
        return xchg.all_reduce_sum(part)[r * Bmax: r * Bmax + B].clamp_(min=0)

    # ------------------------------------------------------------------ checkpoints: full tables <-> shards
    def _send(self, t, dst):
        dist.send(t.cpu() if self.xchg._staged(t) else t, dst=dst, group=self.xchg.group)

    def _recv(self, shape, dtype, src, device):
        staged = device.type == "cuda" and dist.get_backend(self.xchg.group) != "nccl"
        buf = torch.empty(shape, dtype=dtype, device="cpu" if staged else device)
        dist.recv(buf, src=src, group=self.xchg.group)
        return buf

    def gather_table(self, name, chunk_rows=1 << 18):
        """-> the FULL [N, d] table as a CPU tensor on rank 0 (None elsewhere).  The shards travel chunk by chunk through one
        staging buffer: no device ever holds more than `chunk_rows` foreign rows and nothing of size W x shard exists."""
        st, N, W = self.tables[name], self.full_rows[name], self.world
        w = st["w"]
        d = w.shape[1]
        if W == 1:
            return w.detach().cpu() if self.rank == 0 else None
        full = torch.zeros(N, d, dtype=w.dtype) if self.rank == 0 else None
        for r in range(W):
            first, cnt = owned_ids(N, r, W)
            lo = first // W + 1
            for c0 in range(0, cnt, chunk_rows):
                c1 = min(cnt, c0 + chunk_rows)
                piece = None
                if r == self.rank:
                    piece = w[lo + c0: lo + c1].contiguous()
                    if r != 0:
                        self._send(piece, 0)
                elif self.rank == 0:
                    piece = self._recv((c1 - c0, d), w.dtype, r, w.device)
                if self.rank == 0:
                    full[first + W * c0: first + W * (c1 - 1) + 1: W] = piece.cpu()
        return full

    def scatter_table(self, name, full, chunk_rows=1 << 18):
        """the inverse: rank 0 holds `full` [N, d] (CPU or device); every rank receives its rows."""
        st, N, W = self.tables[name], self.full_rows[name], self.world
        w = st["w"]
        d = w.shape[1]
        if W == 1:
            w.copy_(full.to(w.device))
            return
        if self.rank == 0 and tuple(full.shape) != (N, d):
            raise ValueError(f"{name}: checkpoint table has shape {tuple(full.shape)}, the model expects {(N, d)}")
        for r in range(W):
            first, cnt = owned_ids(N, r, W)
            lo = first // W + 1
            for c0 in range(0, cnt, chunk_rows):
                c1 = min(cnt, c0 + chunk_rows)
                if self.rank == 0:
                    piece = full[first + W * c0: first + W * (c1 - 1) + 1: W].to(w.device).contiguous()
                    if r == 0:
                        w[lo + c0: lo + c1] = piece
                    else:
                        self._send(piece, r)
                elif self.rank == r:
                    w[lo + c0: lo + c1] = self._recv((c1 - c0, d), w.dtype, 0, w.device).to(w.device)
        w[0].zero_()
        if self.rank == 0 and W > 1:
            w[1].zero_()      # the slot id 0 would take: never an item

    def gather_state_dict(self):
        """the model's state_dict with FULL tables, on rank 0 (CPU tensors; None elsewhere).  Collective."""
        self.flush()
        sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()} if self.rank == 0 else None
        for name in self.tables:
            key = name + ".weight"
            full = self.gather_table(name)
            if self.rank == 0:
                for k, v in list(sd.items()):      # aliases of the same table (symmetric AvgHist) share the storage
                    if k == key or (k.endswith("embedding.weight") and getattr(self.model, k[:-7]).weight is getattr(self.model, name).weight):
                        sd[k] = full
        return sd

    def scatter_state_dict(self, sd):
        """load a FULL state_dict (rank 0's `sd`; other ranks may pass None): tables are dealt out row by row, everything else is
        broadcast.  Collective."""
        model = self.model
        small = None
        if self.rank == 0:
            small = {k: v for k, v in sd.items() if not any(k == n + ".weight" for n in TABLE_NAMES)
                     and not (k.endswith("embedding.weight") and k[:-7] in ("item_src_embedding",))}
        box = [small]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=self.xchg.group)
        own = model.state_dict()
        for k, v in box[0].items():
            if k in own and tuple(own[k].shape) == tuple(v.shape):
                own[k].copy_(v.to(own[k].device))
        for name in self.tables:
            self.scatter_table(name, sd[name + ".weight"] if self.rank == 0 else None)
        model.check_views()
        self.mark_tables_current()
