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
    (deterministic) and apply the optimizer rule to their rows.  No collective ever touches a full table;
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
import torch
import torch.distributed as dist

from .. import ops
from ..sharded import RowExchange, shard_rows
from .optimizer import SparseDenseAdam

TABLE_NAMES = ("item_embedding", "user_embedding", "item_dst_embedding")


def dist_info(accelerator=None):
    """(rank, world) of this process: an Accelerate-style object if one was passed (trainer.py:21-40 gets one), else the
    default torch.distributed group, else (0, 1)."""
    if accelerator is not None and hasattr(accelerator, "num_processes"):
        return int(accelerator.process_index), int(accelerator.num_processes)
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


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


class ShardedSparseDenseAdam(SparseDenseAdam):
    """SparseDenseAdam for world > 1 (see the module docstring).  The model keeps its SHARD under each table's ``weight``;
    ``train_step`` swaps the compact table of the batch's rows in for the duration of the model's forward_backward."""

    def __init__(self, model, rank, world, group=None, sync_init=True, full_rows=None, **kw):
        """full_rows (optional): {table: N} for tables the model ALREADY holds as this rank's shard (shard_rows(N, W) rows,
        initialised per rank): nothing is broadcast or cut for them -- how a 100 M-row table is brought up without ever
        existing in one piece (bench.py); by default the model's full tables are broadcast from rank 0 and cut here."""
        self.rank, self.world = rank, world
        self.xchg = RowExchange(world, rank, group)
        self.full_rows = {}
        full_rows = dict(full_rows or {})
        dev = model.device
        if model.loss_type == "fullsoftmax":
            raise NotImplementedError("fullsoftmax scores every item against every user: not available over a row-sharded table")
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
        self._zero_id = torch.zeros(1, dtype=torch.int64, device=dev)
        self._zero_coef = torch.zeros(1, dtype=torch.float32, device=dev)
        self._comm = None
        self._look = None          # plans of the next batch, made on a side stream (ids only)

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

    # ------------------------------------------------------------------ id plans
    def _table_inputs(self, batch):
        """-> {table: (field_a, field_b, ids_a int32 | None, ids_b int64 with a trailing 0)}.  The trailing lookup of id 0 pins
        compact row 0 = the padding row, whatever the batch holds."""
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
            b = torch.cat([tb.reshape(-1).to(torch.int64), self._zero_id]) if tb is not None else self._zero_id
            out[name] = (ka if ta is not None else None, kb if tb is not None else None, a, b)
        return out

    def _sharded_plans(self, batch):
        return {name: (ka, kb, ops.rows_plan_sharded(a, b, self.full_rows[name], self.world))
                for name, (ka, kb, a, b) in self._table_inputs(batch).items()}

    def prefetch(self, batch):
        """plans (id sorts + per-owner counts) of the NEXT batch on a side stream, counts copied to pinned host memory there:
        the step that adopts them only waits for that event, never drains the compute stream."""
        if batch is None or not self.model.device.type == "cuda":
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.model.device)
        inputs = self._table_inputs(batch)        # built (and later released) under the main stream
        bufs = {name: (ops.rows_plan_alloc((a.numel() if a is not None else 0) + b.numel(), a.numel() if a is not None else 0,
                                           self.model.device), torch.empty(self.world, dtype=torch.int32, device=self.model.device))
                for name, (_, _, a, b) in inputs.items()}
        hosts = {name: torch.empty(self.world, dtype=torch.int32).pin_memory() for name in inputs}
        self._side.wait_stream(main)
        plans = {}
        with torch.cuda.stream(self._side):
            for name, (ka, kb, a, b) in inputs.items():
                pl, counts = ops.rows_plan_sharded(a, b, self.full_rows[name], self.world, out=bufs[name])
                hosts[name].copy_(counts, non_blocking=True)
                plans[name] = (ka, kb, (pl, counts))
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._look = (self._batch_key(batch), plans, hosts, ev, (inputs, bufs))

    @staticmethod
    def _batch_key(batch):
        return tuple((k, v.data_ptr(), v.numel()) for k, v in sorted(batch.items()) if torch.is_tensor(v))

    # ------------------------------------------------------------------ the step
    def train_step(self, batch, next_batch=None):
        """One optimisation step of the ONE model on this rank's batch (a dict of device tensors as the Trainer builds it).
        Returns this rank's loss (device scalar, detached)."""
        model, W, xchg = self.model, self.world, self.xchg
        if not model.training:
            model.train()
        self.zero_grad()
        self.t += 1
        cfg = self._cfg(self.t)
        # ---- 1. plans: adopt the lookahead (host-side counts: no stream sync) or sort now (one host sync for the counts)
        look, self._look = self._look, None
        host_counts = None
        if look is not None:
            torch.cuda.current_stream().wait_event(look[3])
            if look[0] == self._batch_key(batch):
                look[3].synchronize()
                plans, host_counts = look[1], {n: [int(x) for x in h.tolist()] for n, h in look[2].items()}
            else:
                plans = self._sharded_plans(batch)
        else:
            plans = self._sharded_plans(batch)
        if next_batch is not None:
            self.prefetch(next_batch)
        # ---- 2. per table: ids -> owners, rows back
        ctx = {}
        cbatch = dict(batch)
        for name, (ka, kb, (pl, counts)) in plans.items():
            st = self.tables[name]
            n_local = st["w"].shape[0]
            if host_counts is not None:
                send, recv = xchg.exchange_counts_host(host_counts[name])
            else:
                send, recv = xchg.exchange_counts_dev(counts)
            n_uniq = sum(send)
            keys = pl.uniq_idx[:n_uniq]
            req_send = (keys % n_local).to(torch.int32) if W > 1 else keys
            req = xchg.all_to_all_rows(req_send, send, recv, label="a2a_ids").contiguous()
            # every sender's block is ascending and unique (its plan sorted it): the owner-side plan is a W-way merge
            own = ops.rows_plan_merge(req, recv) if 1 < W <= 64 and req.numel() > 0 else ops.rows_plan(req, None, n_local)
            if st["last"] is not None and self.t > 1:
                ops.lazy_adam_catchup(cfg, st["w"], st["m"], st["v"], st["last"], own)
            compact = xchg.all_to_all_rows(ops.embedding_gather(st["w"], req), recv, send, label="a2a_rows")
            idx_a, idx_b = ops.compact_index(pl)
            if ka is not None:
                cbatch[ka] = idx_a.view(batch[ka].shape)
            if kb is not None:
                cbatch[kb] = idx_b[:-1].view(batch[kb].shape)
            ctx[name] = dict(pl=pl, own=own, send=send, recv=recv, n_uniq=n_uniq, keys=keys, compact=compact, n_local=n_local)
        # ---- 3. the model's own forward / backward on the compact tables (bias vectors compacted the same way)
        swapped = []
        try:
            for name, c in ctx.items():
                p = getattr(model, name).weight
                swapped.append((p, p.data))
                p.data = c["compact"]
            bias_ctx = self._compact_biases(ctx, plans, batch, swapped)
            if "user_id" in cbatch and cbatch["user_id"].dtype != torch.int64:
                cbatch["user_id"] = cbatch["user_id"].to(torch.int64)
            kw = {k: cbatch[k] for k in ("user_id", "item_id", "label", "item_seq", "item_seq_len") if k in cbatch}
            loss = model.forward_backward(**kw)
        finally:
            for p, data in reversed(swapped):
                p.data = data
        # ---- 4. row gradients: reduce per unique key, send to the owners, owners sum in source-rank order
        owner_grads = {}
        for name, c in ctx.items():
            ids_a, rows, ids_b, coef, vec, G = self._collect(name)
            d = c["compact"].shape[1]
            if ids_b is not None:    # the trailing id-0 lookup carries a zero coefficient and a zero vector row
                coef = torch.cat([coef.reshape(-1), self._zero_coef])
                vec = torch.cat([vec.reshape(-1, d), torch.zeros(1, d, dtype=vec.dtype, device=vec.device)])
            else:
                coef, vec, G = self._zero_coef, torch.zeros(1, d, dtype=torch.float32, device=c["compact"].device), 1
            ug = ops.rows_reduce(c["pl"], rows, coef, vec, G, d)[: c["n_uniq"]]
            grads_in = xchg.all_to_all_rows(ug, c["send"], c["recv"], label="a2a_row_grads").contiguous()
            owner_grads[name] = ops.rows_reduce(c["own"], grads_in, None, None, 1, d, zero_tail=self.grad_clip is not None)
        # ---- 5. dense gradients + bias gradients + flags: ONE flat all-reduce (sum)
        model.finish_backward()
        dgrad = model.dense_flat.grad if model.dense_flat.grad is not None else torch.zeros_like(model.dense_flat.data)   # (MF: no dense parameters)
        pieces = [dgrad.reshape(-1)]
        bias_full = []
        for p, gid in bias_ctx:       # compact bias gradient -> the bias vector's own index space
            g = torch.zeros_like(p.data)
            if p.grad is not None:
                g.index_add_(0, gid, p.grad.reshape(-1)[: gid.numel()])
            bias_full.append(g)
            pieces.append(g)
        for p in self.extra:
            if not any(p is q for q, _ in bias_ctx):
                bias_full.append(p.grad.reshape(-1) if p.grad is not None else torch.zeros_like(p.data))
                pieces.append(bias_full[-1])
        guard = getattr(model, "loss_guard", None)
        nan_flag = (guard < 0).to(torch.float32) if guard is not None else self._zero_coef
        extra = [nan_flag]
        if self.grad_clip is not None:   # sum of squares of THIS rank's owned row gradients rides along
            ss = self._scalars[0:1]
            first = True
            for name, og in owner_grads.items():
                ops.sumsq(og, ss, accumulate=not first, ws=self._sumsq_ws)
                first = False
            if first:
                ss.zero_()
            extra.append(ss)
        flat = torch.cat(pieces + extra)
        flat = xchg.all_reduce_sum(flat)
        n_tail = len(extra)
        tail = flat[flat.numel() - n_tail:]
        # gradient scale of every update kernel: 1/W (DDP's mean), times the clip coefficient, or -1 = skip (a NaN loss somewhere)
        scale = torch.where(tail[0:1] > 0, torch.full_like(self.inv_w, -1.0), self.inv_w)
        if self.grad_clip is not None:
            body = flat[: flat.numel() - n_tail]
            ss_all = self._scalars[2:3]
            ops.sumsq(body, ss_all, accumulate=False, ws=self._sumsq_ws)
            total = (ss_all + tail[1:2]) * (self.inv_w * self.inv_w)       # squared norm of the AVERAGED gradient
            coef = self._scalars[1:2]
            ops.clip_coef(total, self.grad_clip, coef)
            scale = torch.where(scale < 0, scale, scale * coef)
        # ---- 6. updates: owners' rows, then the replicated dense parameters
        for name, c in ctx.items():
            st = self.tables[name]
            ops.sparse_adam_rows(cfg, st["w"], st["m"], st["v"], c["own"], owner_grads[name], st["last"], scale)
        o = model.dense_flat.numel()
        ops.dense_adam(cfg, model.dense_flat.data, flat[:o].contiguous(), self.dense_m, self.dense_v, scale)
        for p, (m, v), g in zip(self._bias_order(bias_ctx), self._bias_state(bias_ctx), bias_full):
            n = p.numel()
            ops.dense_adam(cfg, p.data, flat[o:o + n].contiguous(), m, v, scale)
            o += n
        object.__setattr__(model, "loss_guard", None)
        model.sparse_grads.clear()
        model.dense_flat.grad = None
        for p in self.extra:
            p.grad = None
        return loss.detach()

    # ---- bias vectors under compacted ids -------------------------------------------------------------------------------
    def _global_ids(self, name, c):
        """global id of every compact row of table `name` (row 0 = id 0)."""
        keys = c["keys"].to(torch.int64)
        if self.world == 1:
            return keys
        owner, local = keys // c["n_local"], keys % c["n_local"]
        return torch.where(keys > 0, (local - 1) * self.world + owner, torch.zeros_like(keys))

    def _compact_biases(self, ctx, plans, batch, swapped):
        """item_bias is indexed by item_id, user_bias by user_id: when that field was re-indexed to compact rows, the model must
        see the bias entries of those rows.  -> [(parameter, global ids of its compact entries)]"""
        model, out = self.model, []
        for pname, field in (("item_bias", "item_id"), ("user_bias", "user_id")):
            if not getattr(model, "has_" + pname, False):
                continue
            p = getattr(model, pname)
            for name, (ka, kb, _) in plans.items():
                if field in (ka, kb):
                    gid = self._global_ids(name, ctx[name])
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
        """For evaluation forwards: fetch the rows `batch` looks up.  -> (re-indexed batch, restore()); between the call and
        restore() the model's tables (and bias vectors) are the compact ones.  Call flush() first in lazy_dense mode."""
        model, xchg, W = self.model, self.xchg, self.world
        plans = self._sharded_plans(batch)
        ctx, cbatch, swapped = {}, dict(batch), []
        for name, (ka, kb, (pl, counts)) in plans.items():
            st = self.tables[name]
            n_local = st["w"].shape[0]
            send, recv = xchg.exchange_counts_dev(counts)
            keys = pl.uniq_idx[: sum(send)]
            req_send = (keys % n_local).to(torch.int32) if W > 1 else keys
            req = xchg.all_to_all_rows(req_send, send, recv).contiguous()
            compact = xchg.all_to_all_rows(ops.embedding_gather(st["w"], req), recv, send)
            idx_a, idx_b = ops.compact_index(pl)
            if ka is not None:
                cbatch[ka] = idx_a.view(batch[ka].shape)
            if kb is not None:
                cbatch[kb] = idx_b[:-1].view(batch[kb].shape)
            ctx[name] = dict(keys=keys, n_local=n_local)
            p = getattr(model, name).weight
            swapped.append((p, p.data))
            p.data = compact
        self._compact_biases(ctx, plans, batch, swapped)
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
