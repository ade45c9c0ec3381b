"""Row-sharded embedding table + data-parallel SASRec step for the 1/2/4/8-GPU path.

Not in the reference: UniRec's only strategy is DDP over a replicated dense table (unirec/facility/trainer.py:67,
346) whose [N,d] gradient is all-reduced every step.  Here (one process per GPU, torch.distributed over RCCL/xGMI):

  * batch: each rank trains on its own B rows; dense-parameter gradients (~0.4 M floats) are summed with ONE flat
    all-reduce and scaled by 1/W inside the Adam kernel (== DDP's mean);
  * table: row ``i`` lives on rank ``i % W`` at local row ``i // W + 1`` (local row 0 is the padding row of every
    shard).  Per step and rank:  sort/unique the batch ids by (owner, row) -> all-to-all #1 (row ids, int32) ->
    owners catch up / gather the rows -> all-to-all #2 (rows) -> forward/backward on the COMPACT table of the
    fetched rows (lookups re-indexed, same HIP kernels) -> segment-reduce the row gradients -> all-to-all #3
    (row gradients) -> owners sum the contributions in rank order (deterministic) and apply row-wise Adam.
    No collective ever touches the full table.

``RowExchange`` is pure torch.distributed routing (any backend / device): the world_size-2 gloo tests drive it on
the CPU; everything else is the HIP library.
"""
from typing import List

import os
import warnings

import torch
import torch.distributed as dist

from . import ops


class RowExchange:
    """Variable-size all-to-all of per-owner contiguous row blocks."""

    def __init__(self, world: int, rank: int, group=None):
        self.world, self.rank, self.group = world, rank, group
        self.profile = None        # {label: [ms, bytes sent to peers, calls]} while profiling is on (bench.py: per-collective times)
        self._pending = []
        # host-side group for the tiny per-step counts exchange (2W integers that the HOST needs as split sizes): with
        # an RCCL main group this is a second, gloo group, so the exchange neither queues behind device collectives nor
        # forces the host to drain the compute stream (exchange_counts_host).  Created collectively by every rank.
        self.cpu_group = None
        if world > 1:
            if dist.get_backend(group) == "gloo":
                self.cpu_group = group
            elif os.environ.get("UR_COUNTS_VIA_DEVICE") != "1":   # (switch: counts through the main group, as before)
                try:
                    self.cpu_group = dist.new_group(backend="gloo")
                except Exception as e:   # no gloo transport on this host: every rank fails alike and takes the device path
                    warnings.warn(f"row exchange: no gloo group for the counts ({e}); using the device group")

    def exchange_counts_host(self, send_counts: List[int]):
        """send_counts already on the HOST (read from a plan computed ahead on a side stream): all-to-all over the gloo group.
        No device work, no stream synchronisation."""
        if self.world == 1:
            return list(send_counts), list(send_counts)
        if self.cpu_group is None:
            return [int(x) for x in send_counts], self.exchange_counts(send_counts)
        src = torch.tensor(send_counts, dtype=torch.int64)
        recv = torch.empty_like(src)
        dist.all_to_all_single(recv, src, group=self.cpu_group)
        return [int(x) for x in send_counts], [int(x) for x in recv.tolist()]

    def exchange_counts(self, send_counts: List[int]) -> List[int]:
        if self.world == 1:
            return list(send_counts)
        t = torch.tensor(send_counts, dtype=torch.int64)
        out = torch.empty(self.world, dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            t, out = t.to(dev), out.to(dev)
        dist.all_to_all_single(out, t, group=self.group)
        return [int(x) for x in out.tolist()]

    def exchange_counts_dev(self, counts_dev: torch.Tensor):
        """counts_dev: int32/int64 [world] ON THE DEVICE (rows this rank wants from each owner).  Exchanges them without
        a host round trip and reads send + recv counts back together: ONE host sync per step instead of two."""
        send = counts_dev.to(torch.int64)
        if self.world == 1:
            c = [int(x) for x in send.tolist()]
            return c, list(c)
        nccl = dist.get_backend(self.group) == "nccl"
        src = send if nccl else send.cpu()
        recv = torch.empty_like(src)
        dist.all_to_all_single(recv, src, group=self.group)
        both = torch.cat([src, recv]).tolist()
        return [int(x) for x in both[: self.world]], [int(x) for x in both[self.world:]]

    # ---- per-collective timing (off by default: two events per collective).  bytes = what this rank sends to its W - 1 peers.
    def profile_start(self):
        self.profile, self._pending = {}, []

    def profile_stop(self):
        """-> {label: {"ms": device time of the collective on the compute stream, "MB_to_peers": ..., "calls": ...}}"""
        out = {}
        for label, e0, e1, nbytes in self._pending:
            e1.synchronize()
            rec = out.setdefault(label, {"ms": 0.0, "MB_to_peers": 0.0, "calls": 0})
            rec["ms"] += e0.elapsed_time(e1)
            rec["MB_to_peers"] += nbytes / 1e6
            rec["calls"] += 1
        self.profile, self._pending = None, []
        return out

    def _timed(self, label, nbytes, fn, on_device):
        if self.profile is None or not on_device:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self._pending.append((label, e0, e1, nbytes))
        return r

    def all_to_all_rows(self, send: torch.Tensor, send_counts: List[int], recv_counts: List[int], label="all_to_all") -> torch.Tensor:
        """send: [sum(send_counts), ...] blocks ordered by destination rank -> [sum(recv_counts), ...] ordered by source."""
        if self.world == 1:
            return send
        stage = send.is_cuda and dist.get_backend(self.group) != "nccl"   # gloo has no device all-to-all: stage via host
        src = send.contiguous().cpu() if stage else send.contiguous()
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=src.device)
        row_bytes = send.element_size() * (send[0].numel() if send.shape[0] else 1)
        peers = (sum(send_counts) - send_counts[self.rank]) * row_bytes
        self._timed(label, peers, lambda: dist.all_to_all_single(out, src, output_split_sizes=recv_counts,
                                                                 input_split_sizes=send_counts, group=self.group), send.is_cuda)
        return out.to(send.device) if stage else out


    def _staged(self, t: torch.Tensor):
        return t.is_cuda and self.world > 1 and dist.get_backend(self.group) != "nccl"

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        if self._staged(t):
            h = t.cpu()
            self._timed("all_reduce", 2 * t.numel() * t.element_size() * (self.world - 1) // self.world,
                        lambda: dist.all_reduce(h, group=self.group), t.is_cuda)
            return h.to(t.device)
        self._timed("all_reduce", 2 * t.numel() * t.element_size() * (self.world - 1) // self.world,
                    lambda: dist.all_reduce(t, group=self.group), t.is_cuda)
        return t

    def all_gather_cat(self, t: torch.Tensor) -> torch.Tensor:
        """equal-shaped [n, ...] from every rank -> [world * n, ...] in rank order"""
        if self.world == 1:
            return t
        src = t.contiguous().cpu() if self._staged(t) else t.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self.group)
        return torch.cat(parts).to(t.device)


def shard_rows(n_items: int, world: int) -> int:
    """rows per shard INCLUDING its padding row 0 (must match rows.hip build_keys_kernel)."""
    return (n_items + world - 1) // world + 1 if world > 1 else n_items


def owner_and_local(ids: torch.Tensor, world: int):
    """global id -> (owner rank, local row); id 0 (padding) -> (0, 0).  Mirrors build_keys_kernel."""
    if world == 1:
        return torch.zeros_like(ids), ids
    return torch.where(ids > 0, ids % world, torch.zeros_like(ids)), torch.where(ids > 0, ids // world + 1, torch.zeros_like(ids))


class ShardedSasrecStep:
    """One training step of a sequence encoder (SASRec or GRU, ``model_cfg['model']``) + sampled loss over a row-sharded
    table.  step(batch) -> local loss (device scalar).  (The class keeps its first name; GRU = BASELINE config C4.)"""

    def __init__(self, model_cfg: dict, device, rank: int, world: int, lr=1e-3, weight_decay=0.0, table_mode="lazy_dense",
                 batch_size=None, seed=2022):
        from .model.sequential.gru import GRU
        from .model.sequential.sasrec import SASRec
        self.rank, self.world, self.device = rank, world, device
        self.N, self.d = model_cfg["n_items"], model_cfg["embedding_size"]
        self.n_local = shard_rows(self.N, world)
        self.xchg = RowExchange(world, rank)
        cfg = dict(model_cfg)
        cfg["n_items"] = 8            # the model object only carries the dense parameters here
        cfg["device"] = str(device)
        cfg["dropout_seed"] = int(cfg.get("seed", seed) or 0) * 65536 + rank   # independent dropout masks on every rank
        torch.manual_seed(seed)       # identical dense init on every rank
        self.kind = model_cfg.get("model", "SASRec")
        if self.kind not in ("SASRec", "GRU"):
            raise NotImplementedError(f"row-sharded training is built for SASRec and GRU, not {self.kind}")
        self.model = (GRU if self.kind == "GRU" else SASRec)(cfg)
        self.model.train()
        if world > 1:
            dist.broadcast(self.model.dense_flat.data, src=0)
        g = torch.Generator(device=device).manual_seed(seed + 1000 * rank + 1)
        self.table = torch.empty(self.n_local, self.d, dtype=torch.float32, device=device)
        self.table.normal_(0.0, model_cfg.get("init_std", 0.02), generator=g)
        self.table[0].zero_()
        self.m = torch.zeros_like(self.table)
        self.v = torch.zeros_like(self.table)
        self.last = torch.zeros(self.n_local, dtype=torch.int32, device=device) if table_mode == "lazy_dense" else None
        self.dense_m = torch.zeros_like(self.model.dense_flat.data)
        self.dense_v = torch.zeros_like(self.model.dense_flat.data)
        self.lr, self.wd, self.t = lr, weight_decay, 0
        self.inv_w = torch.full((1,), 1.0 / world, dtype=torch.float32, device=device)
        self.zero_id = torch.zeros(1, dtype=torch.int64, device=device)
        self.zero_coef = torch.zeros(1, dtype=torch.float32, device=device)
        self.loss_type = model_cfg["loss_type"]
        self.tau = model_cfg.get("tau", 1.0)
        self._side, self._prefetched, self._counts_pinned = None, None, None

    # ---- the step ------------------------------------------------------------------------------------------
    def step(self, batch, next_batch=None):
        m, W, d = self.model, self.world, self.d
        item_seq, item_id, label = batch["item_seq"], batch["item_id"], batch.get("label")
        B, L = item_seq.shape
        G = item_id.shape[1]
        self.t += 1
        acfg = ops.adam_cfg(self.lr, self.t, self.wd)
        # 1. plan: unique (owner, row) keys of this batch; a trailing lookup of id 0 pins compact row 0 = padding row.
        #    Local work that depends on the ids only: the plan of the NEXT batch is started on a side stream here (no
        #    collective runs there), so from the second step on this is just an event wait.
        pl, counts_dev, counts_host = self._take_plan(batch)
        if next_batch is not None:
            self._prefetch_plan(next_batch)
        if counts_host is not None:
            # the plan came from the lookahead: its counts were copied to pinned host memory on the side stream, so the host
            # only waited for THAT event -- the compute stream is never drained and the host keeps running ahead of the GPU
            send_counts, recv_counts = self.xchg.exchange_counts_host(counts_host)
        else:
            send_counts, recv_counts = self.xchg.exchange_counts_dev(counts_dev)   # first step: one host sync (2W ints)
        n_uniq = sum(send_counts)
        keys = pl.uniq_idx[:n_uniq]
        req_send = (keys % self.n_local).to(torch.int32) if W > 1 else keys
        # 2. all-to-all #1: row ids -> owners
        req = self.xchg.all_to_all_rows(req_send, send_counts, recv_counts)
        # every sender's block is already ascending and unique (its plan sorted it): the owner-side plan is a W-way merge
        own = (ops.rows_plan_merge(req.contiguous(), recv_counts) if 1 < W <= 64 and req.numel() > 0
               else ops.rows_plan(req.contiguous(), None, self.n_local))
        if self.last is not None and self.t > 1:
            ops.lazy_adam_catchup(acfg, self.table, self.m, self.v, self.last, own)
        # 3. all-to-all #2: rows back -> compact table [n_uniq, d]
        rows_out = ops.embedding_gather(self.table, req)
        compact = self.xchg.all_to_all_rows(rows_out, recv_counts, send_counts)
        idx_a, idx_b = ops.compact_index(pl)
        seq_c = idx_a.view(B, L)
        item_c = idx_b[: B * G].view(B, G).contiguous()
        # 4. forward / backward on the compact table (same kernels as the single-GPU path)
        cfg = m._cfg(B) if self.kind == "GRU" else m._cfg(B, train=True)   # SASRec: training-time dropout as configured
        ws = m._workspace(cfg, train=True)
        enc_fwd, enc_bwd = (ops.gru_fwd, ops.gru_bwd) if self.kind == "GRU" else (ops.sasrec_fwd, ops.sasrec_bwd)
        user_emb = enc_fwd(cfg, compact, m.dense_flat.data, seq_c, ws)
        lcfg = ops.loss_cfg(B, G, d, self.loss_type, self.tau)
        lab = label.to(torch.int32).contiguous() if label is not None else None
        scores, _, loss_out = ops.gather_dot_loss_fwd(lcfg, user_emb, compact, item_c, lab)
        coef, d_user, _ = ops.gather_dot_loss_bwd(lcfg, user_emb, compact, item_c, lab, scores, loss_out)
        # SASRec: the dense-gradient reductions stay on the library's side stream (ur_sasrec_bwd_deferred) under the row-gradient
        # reduce and all-to-all #3 below; joined right before the first reader of dense_grad, the all-reduce
        deferred = self.kind != "GRU"
        if deferred:
            dense_grad, d_rows = ops.sasrec_bwd(cfg, compact, m.dense_flat.data, seq_c, d_user, ws, defer_join=True)
        else:
            dense_grad, d_rows = enc_bwd(cfg, compact, m.dense_flat.data, seq_c, d_user, ws)
        # 5. row gradients of the unique keys, then all-to-all #3 to the owners
        coef_b = torch.cat([coef.reshape(-1), self.zero_coef])
        vec_b = torch.cat([user_emb, torch.zeros(1, d, dtype=user_emb.dtype, device=user_emb.device)])   # the trailing id-0 lookup reads row B
        ug = ops.rows_reduce(pl, d_rows, coef_b, vec_b, G, d)[:n_uniq]
        grads_in = self.xchg.all_to_all_rows(ug, send_counts, recv_counts)
        # 6. dense parameters: one flat all-reduce (sum), mean applied inside the Adam kernel.  Issued behind the last
        #    all-to-all and waited for only before the dense update: it runs under the owner-side reduce + sparse update
        if deferred:
            ops.sasrec_bwd_join()
        work = dist.all_reduce(dense_grad, async_op=True) if W > 1 else None
        own_ug = ops.rows_reduce(own, grads_in.contiguous(), None, None, 1, d)   # sums ranks in source-rank order
        ops.sparse_adam_rows(acfg, self.table, self.m, self.v, own, own_ug, self.last, self.inv_w)
        if work is not None:
            work.wait()
        ops.dense_adam(acfg, m.dense_flat.data, dense_grad, self.dense_m, self.dense_v, self.inv_w)
        return loss_out[0]

    # ---- plan lookahead (ids only; mirrors SparseDenseAdam.prefetch_plan) ------------------------------------
    def _plan_ids(self, batch):
        return batch["item_seq"].reshape(-1), torch.cat([batch["item_id"].reshape(-1), self.zero_id])

    def _prefetch_plan(self, batch):
        if not batch["item_seq"].is_cuda:
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        ids_a, ids_b = self._plan_ids(batch)                 # built (and later released) under the main stream
        n, n_a = ids_a.numel() + ids_b.numel(), ids_a.numel()
        bufs = (ops.rows_plan_alloc(n, n_a, self.device), torch.empty(self.world, dtype=torch.int32, device=self.device))
        self._side.wait_stream(main)
        if self._counts_pinned is None:   # two slots: the host reads slot t while the side stream may already fill slot t+1
            self._counts_pinned = [torch.empty(self.world, dtype=torch.int32).pin_memory() for _ in range(2)]
        host = self._counts_pinned[self.t & 1]
        with torch.cuda.stream(self._side):
            pl, counts = ops.rows_plan_sharded(ids_a, ids_b, self.N, self.world, out=bufs)
            host.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._prefetched = ((batch["item_seq"].data_ptr(), batch["item_id"].data_ptr()), pl, counts, ev, (ids_a, ids_b, bufs), host)

    def _take_plan(self, batch):
        """-> (plan, counts on the device, counts on the host or None)"""
        pre, self._prefetched = self._prefetched, None
        if pre is not None:
            torch.cuda.current_stream().wait_event(pre[3])
            if pre[0] == (batch["item_seq"].data_ptr(), batch["item_id"].data_ptr()):
                pre[3].synchronize()           # the host waits for the side stream's plan only
                return pre[1], pre[2], [int(x) for x in pre[5].tolist()]
        ids_a, ids_b = self._plan_ids(batch)
        pl, counts = ops.rows_plan_sharded(ids_a, ids_b, self.N, self.world)
        return pl, counts, None

    # ---- evaluation over the sharded table -------------------------------------------------
    @torch.no_grad()
    def encode(self, item_seq):
        """user_emb [B,d] of this rank's sequences, evaluation mode: the rows are fetched through the same two all-to-alls as
        in step() (call flush() first when the table is lazily updated)."""
        m, W = self.model, self.world
        item_seq = item_seq.to(torch.int32).contiguous()
        B, L = item_seq.shape
        pl, counts_dev = ops.rows_plan_sharded(item_seq.reshape(-1), self.zero_id, self.N, W)
        send_counts, recv_counts = self.xchg.exchange_counts_dev(counts_dev)
        keys = pl.uniq_idx[: sum(send_counts)]
        req_send = (keys % self.n_local).to(torch.int32) if W > 1 else keys
        req = self.xchg.all_to_all_rows(req_send, send_counts, recv_counts)
        compact = self.xchg.all_to_all_rows(ops.embedding_gather(self.table, req), recv_counts, send_counts)
        idx_a, _ = ops.compact_index(pl)
        was_training = m.training
        m.eval()
        cfg = m._cfg(B)
        enc_fwd = ops.gru_fwd if self.kind == "GRU" else ops.sasrec_fwd
        out = enc_fwd(cfg, compact, m.dense_flat.data, idx_a.view(B, L), m._workspace(cfg)).clone()
        m.train(was_training)
        return out

    def local_history(self, hist_ptr, hist_sorted):
        """CSR history (global ids, ascending per user) -> the same CSR restricted to this rank's items, in local row ids."""
        if hist_ptr is None or self.world == 1:
            return hist_ptr, hist_sorted
        W, r = self.world, self.rank
        own = (hist_sorted % W == r) & (hist_sorted > 0)
        csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=own.device), own.to(torch.int64).cumsum(0)])
        return csum[hist_ptr].contiguous(), (hist_sorted[own] // W + 1).to(torch.int32).contiguous()

    @torch.no_grad()
    def full_item_ranks(self, item_seq, target, user_id=None, hist_ptr=None, hist_sorted=None, local_hist=None):
        """one_vs_all rank (Evaluator.evaluate_with_full_items semantics, as ops.full_rank) of this rank's B rows over the
        row-SHARDED catalogue: all-gather the user vectors, every rank counts on its own shard, two small all-reduces
        (thresholds, counts).  Every rank must call it with the same B.  -> int32[B]."""
        W, r = self.world, self.rank
        ue = self.encode(item_seq)
        B = ue.shape[0]
        target = target.reshape(B, -1)[:, 0].to(torch.int64).contiguous()
        ue_all, tgt_all = self.xchg.all_gather_cat(ue), self.xchg.all_gather_cat(target)
        uid_all = self.xchg.all_gather_cat(user_id.to(torch.int64).contiguous()) if user_id is not None else None
        ltgt = tgt_all if W == 1 else torch.where(tgt_all % W == r, tgt_all // W + 1, torch.full_like(tgt_all, -1))
        hp, hs = local_hist if local_hist is not None else self.local_history(hist_ptr, hist_sorted)
        if uid_all is None:
            hp = hs = None
        # valid local rows: rank 0 holds ids W, 2W, .. at local rows 2.. (row 1 is the slot id 0 would take: never an item);
        # rank r > 0 holds ids r, r + W, .. at local rows 1..
        if W == 1:
            n_rows, excl = self.N, -1
        elif r == 0:
            n_rows, excl = (self.N - 1) // W + 2, 1
        else:
            n_rows, excl = ((self.N - 1 - r) // W + 1 if self.N - 1 >= r else 0) + 1, -1
        thr = self.xchg.all_reduce_sum(ops.full_rank_shard(1, ue_all, self.table, ltgt, n_rows=n_rows))
        part = ops.full_rank_shard(2, ue_all, self.table, ltgt, thr=thr, user_id=uid_all, hist_ptr=hp, hist_sorted_local=hs,
                                   n_rows=n_rows, excl_row=excl)
        return self.xchg.all_reduce_sum(part)[r * B:(r + 1) * B].clamp_(min=0)   # (see ops.full_rank: a count is never < 0)

    def flush(self):
        if self.last is not None and self.t > 0:
            ops.lazy_adam_flush(ops.adam_cfg(self.lr, self.t, self.wd), self.table, self.m, self.v, self.last)

    def gather_table(self) -> torch.Tensor:
        """Full [N, d] table on every rank (checkpoint / tests): row i <- shard[i % W][i // W + 1]."""
        if self.world == 1:
            return self.table.clone()
        parts = [torch.empty_like(self.table) for _ in range(self.world)]
        dist.all_gather(parts, self.table)
        ids = torch.arange(self.N, device=self.table.device)
        owner, local = owner_and_local(ids, self.world)
        return torch.stack(parts)[owner, local]


def build_sharded_trainer(args, model_cfg, device, rank, world):
    """bench.py hook: returns (step_fn, model, info)."""
    st = ShardedSasrecStep(model_cfg, device, rank, world, lr=1e-3, table_mode=args.table_mode)
    return st.step, st.model, {"parallelism": f"dp{world} + embedding rows sharded {world}-way (3 all-to-alls + 1 all-reduce per step)"}
