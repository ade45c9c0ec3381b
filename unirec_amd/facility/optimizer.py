"""Optimizer for unirec_amd models: dense Adam over the flat buffer + row-wise Adam over the tables.

Reference behaviour being replaced: ``torch.optim.Adam(model.parameters())`` over every parameter,
including the dense [N,d] embedding gradient (unirec/facility/trainer.py:134-136,349), preceded by an
optional global-norm clip (:347-348).

table_mode
  "lazy_dense" (default) -- reproduces the reference's *dense* Adam exactly (for weight_decay == 0):
        rows that a batch does not touch still move in the reference (their momentum keeps pushing
        them); here those zero-gradient steps are replayed lazily, the next time the row is looked up
        (``plan_batch`` -> ur_lazy_adam_catchup) or when ``flush()`` is called (evaluation, checkpoint).
  "rowwise"  -- production semantics: only touched rows are updated (SparseAdam-like); half the traffic.
"""
import collections
import os

import torch

from .. import ops


class SparseDenseAdam:
    def __init__(self, model, lr=1e-3, weight_decay=0.0, betas=None, eps=None, grad_clip=None,
                 table_mode="lazy_dense", algo="adam", overlap_dense_join=True, dense_side=None):
        """algo: the torch.optim rule the reference's Trainer._build_optimizer would construct (trainer.py:134-152):
        adam (default) / adamw / sgd / adagrad / rmsprop; betas / eps None = torch's defaults for that rule.

        Stream contract (SASRec encoder, no gradient clipping): the dense half of the update runs on the encoder's side stream.
        ``step()`` makes the current stream wait for it before returning; ``step(late_join=True)`` -- what the training loop passes when
        another step follows at once -- leaves that to the model's next forward pass.  In that window everything that reads the dense
        parameters or this optimizer's dense state through the public surface joins first (``model.state_dict() / load_state_dict() /
        train() / eval()``, ``optimizer.state_dict() / flush()``); code that reads ``model.dense_flat.data`` or ``dense_m / dense_v``
        directly calls ``model.join_side_updates()`` first.  ``dense_side="late" / "join" / "0"`` overrides the per-call choice."""
        assert table_mode in ("lazy_dense", "rowwise")
        if algo not in ops.OPT_ALGOS:
            raise ValueError(f"unknown optimizer rule {algo!r}")
        d1, d2, de = ops.OPT_DEFAULTS[algo]
        self.algo = algo
        betas = (d1, d2) if betas is None else betas
        eps = de if eps is None else eps
        self.model, self.lr, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        self.grad_clip = grad_clip if grad_clip and grad_clip > 0 else None
        self.table_mode = table_mode
        # step() runs the row-sparse half first and joins the encoder's dense-gradient reductions itself: the model may leave
        # them running on the side stream when its backward returns (dense_flat.grad is None until model.finish_backward())
        model.defer_dense_join = bool(overlap_dense_join) and self.grad_clip is None
        self.t = 0
        dev = model.device
        self.dense_m = torch.zeros_like(model.dense_flat.data)
        self.dense_v = torch.zeros_like(model.dense_flat.data)
        self.extra = [p for n, p in model.named_parameters() if n in ("user_bias", "item_bias")]
        self.extra_state = [(torch.zeros_like(p.data), torch.zeros_like(p.data)) for p in self.extra]
        self.tables = {}
        for name in ("item_embedding", "user_embedding", "item_dst_embedding"):
            if hasattr(model, name) and not (name == "item_dst_embedding" and model.item_dst_embedding is model.item_embedding):
                w = getattr(model, name).weight.data
                st = dict(w=w, m=torch.zeros_like(w), v=torch.zeros_like(w))
                st["last"] = torch.zeros(w.shape[0], dtype=torch.int32, device=dev) if table_mode == "lazy_dense" else None
                if name == "item_embedding" and getattr(model, "loss_type", None) == "fullsoftmax":
                    st["last"] = None      # fullsoftmax: a dense [N,d] gradient every step -> plain dense Adam on this table
                self.tables[name] = st
        self._plans = {}
        # dense_side: "" = per call (step(late_join=...) -> "late" / "join"); "late" / "join" / "0" force one mode ("0": the dense half on
        # the main stream after it has waited for the reductions).  The next batch's rows take their missed zero-gradient steps
        # (lazy_dense) on the main stream right after this step's row update -- under the tail of the dense-gradient stream, and only the
        # rows WITH optimizer history (filtered next to the plan, on its stream) -- or, without a prefetched plan, at the head of the
        # next step (`plan_batch`)
        self._dense_side = dense_side or ""
        self._prefetched, self._side = None, None
        # None: by size (prefetch_plan); True / False: forced (tests; UR_EARLY_CATCHUP=0/1 for A/B runs of whole benchmarks)
        import os
        self._early_catchup = {"0": False, "1": True}.get(os.environ.get("UR_EARLY_CATCHUP", ""))
        self._fused_update = os.environ.get("UR_FUSED_UPDATE", "1") != "0"     # row reduce + row update in one launch (no clipping)
        self._scalars = torch.zeros(4, dtype=torch.float32, device=dev)   # [0] sumsq, [1] clip coef
        self._sumsq_ws = torch.empty(2048, dtype=torch.float32, device=dev)
        self.param_groups = [dict(lr=lr)]  # enough of torch's surface for schedulers / logging

    # ------------------------------------------------------------------ torch.optim surface
    def zero_grad(self, set_to_none=True):
        self.model.finish_backward()
        self.model.dense_flat.grad = None
        for p in self.extra:
            p.grad = None
        self.model.sparse_grads.clear()
        self.model.dense_table_grads.clear()

    def state_dict(self):
        self.model.join_side_updates()
        self._join_plan_stream()
        return dict(t=self.t, dense_m=self.dense_m, dense_v=self.dense_v, param_groups=self.param_groups,
                    tables={k: {kk: vv for kk, vv in v.items() if kk != "w"} for k, v in self.tables.items()})

    def _cfg(self, step):
        return ops.adam_cfg(self.param_groups[0]["lr"], step, self.wd, self.betas[0], self.betas[1], self.eps, algo=self.algo)

    # ------------------------------------------------------------------ per-batch plan (before forward)
    def _plan_inputs(self, item_seq, item_id, user_id):
        """-> {table: (ids_a int32 | None, ids_b int64 | None)} for the tables this batch looks up.  Which batch field feeds
        which table comes from ``model.lookup_tables()`` (default: item table <- item_seq rows + item_id candidates,
        user table <- user_id)."""
        src_a = {"item_seq": item_seq, "user_id": user_id, "item_id": item_id}
        spec = self.model.lookup_tables() if hasattr(self.model, "lookup_tables") else {
            "item_embedding": ("item_seq", "item_id"), "user_embedding": ("user_id", None)}
        req = {}
        for name, (ka, kb) in spec.items():
            if name not in self.tables:
                continue
            ta = src_a.get(ka) if ka else None
            tb = src_a.get(kb) if kb else None
            a = ta.reshape(-1).to(torch.int32).contiguous() if ta is not None else None
            b = tb.reshape(-1).to(torch.int64).contiguous() if tb is not None else None
            if a is not None or b is not None:
                req[name] = (a, b)
        return req

    def _make_plans(self, item_seq, item_id, user_id):
        return {name: ops.rows_plan(a, b, self.tables[name]["w"].shape[0])
                for name, (a, b) in self._plan_inputs(item_seq, item_id, user_id).items()}

    @staticmethod
    def _ids_key(item_seq, item_id, user_id):
        return tuple((t.data_ptr(), t.numel()) if t is not None else None for t in (item_seq, item_id, user_id))

    def plan_stream(self):
        """the stream the NEXT batch's id plan runs on (created on first use); a device batch loader builds its batches there too"""
        if self._side is None and self.model.device.type == "cuda":
            self._side = torch.cuda.Stream(device=self.model.device)
        return self._side

    def prefetch_plan(self, item_seq=None, item_id=None, user_id=None):
        """Sort/unique the ids of the NEXT batch on a side stream, so that the (latency-bound, ~0.15 ms) plan overlaps
        with the current step's forward/backward.  The plan depends on the ids only, never on the model state."""
        main = torch.cuda.current_stream()
        self._pre_waited = None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.model.device)
        # every buffer is allocated (and later released) under the MAIN stream; the side stream only fills them.  No
        # record_stream bookkeeping: that would put one event packet per tensor into the main queue when they are freed
        req = self._plan_inputs(item_seq, item_id, user_id)
        bufs = {name: ops.rows_plan_alloc((a.numel() if a is not None else 0) + (b.numel() if b is not None else 0),
                                          a.numel() if a is not None else 0, self.model.device) for name, (a, b) in req.items()}
        # (through an event without the system-scope fence: ~2 us cheaper on the recording stream than Stream.wait_stream)
        ops.stream_wait_stream(self._side, main)   # the ids may still be in flight (H2D copy) on the main stream; `last` is being updated there
        with torch.cuda.stream(self._side):
            plans = {name: ops.rows_plan(a, b, self.tables[name]["w"].shape[0], out=bufs[name]) for name, (a, b) in req.items()}
            filtered = splits = None
            lazy = [name for name in plans if self.tables[name]["last"] is not None] if self.table_mode == "lazy_dense" else []
            early = self._early_catchup
            if early is None:
                # The tail of step() hides a replay as long as row reduce + row update + replay fit under the side stream's last
                # weight-gradient launch + reductions + dense update (~135 us at 25 600 tokens): ~50 us of replay = 65 K rows x 6 row
                # arrays at the random-row rate.  Measured on one box (tools/ab_switch.sh): C5 (28 K ids a step) tail 0.550 / early 0.553 ms,
                # aged table 0.574 / 0.586; C3 (154 K ids a step) tail 0.853 / early 0.814 ms.
                early = sum(plans[name].n for name in lazy) >= 65536
            if lazy and early and all(name in self._plans for name in lazy):
                # Called between plan_batch(t) and step(t) (self.t == t - 1): the next batch's rows split against the plan of the step in
                # flight.  COLD rows (not in batch t, some optimizer history) get a zero gradient at step t by construction, so their replay
                # up to and INCLUDING step t depends on step indices only: it runs right here, on the plan stream, under step t's forward /
                # backward (disjoint from every row step t reads or writes).  HOT rows (in both batches) are brought to step t by step t's
                # own update; the tail of step() walks just that list (a no-op unless the update was skipped: NaN loss, id guard).
                # Same single replay per row as at the tail, bit for bit -- at C3 (150 K rows a step) it was 148 us of the main stream.
                filtered = {}
                cfg2 = self._cfg(self.t + 2)
                for name in lazy:
                    st = self.tables[name]
                    cold, hot = ops.rows_split_hot(plans[name], st["last"] if self.wd == 0.0 else None, self._plans[name])
                    ops.lazy_adam_catchup(cfg2, st["w"], st["m"], st["v"], st["last"], cold, background=True)
                    filtered[name] = hot
            elif lazy and self._fused_update and self.grad_clip is None and all(name in self._plans for name in lazy):
                # small plans: the replay stays at the tail of the step, but RIDES in the fused reduce + update launch (extra workgroups walk
                # the cold list there; ur_rows_reduce_update) instead of being a launch of its own behind it: the split is made here
                splits = {name: ops.rows_split_hot(plans[name], self.tables[name]["last"] if self.wd == 0.0 else None, self._plans[name])
                          for name in lazy}
            elif lazy and self.wd == 0.0:
                # rows of the next batch that have any optimizer history at all: the catch-up at the tail of this step only walks those
                # (a row first touched by the step in flight is missed here and needs no catch-up: its update leaves it current)
                filtered = {name: ops.rows_filter_touched(plans[name], self.tables[name]["last"]) for name in lazy}
            ev = torch.cuda.Event()
            ev.record(self._side)
        # keep ids + workspaces alive until the plan is adopted (the side stream reads / writes them asynchronously)
        # (and the in-flight step's plans: the catch-up reads their row lists after step() has dropped them)
        self._prefetched = (self._ids_key(item_seq, item_id, user_id), plans, ev, (req, bufs, dict(self._plans)), None, filtered, splits)

    def plan_batch(self, item_seq=None, item_id=None, user_id=None):
        """Sort/unique the ids this batch will look up (or adopt the plan `prefetch_plan` made for the same tensors);
        in lazy_dense mode bring those rows up to date."""
        pre, self._prefetched = self._prefetched, None
        if pre is not None:   # always order the main stream after the side stream's use of the prefetch buffers
            cur = torch.cuda.current_stream()
            # (the tail catch-up of the previous step() already made THIS stream wait for THIS event: a second wait is one more barrier
            # packet in front of the forward pass, ~5 us of idle main stream at every step boundary)
            pw = getattr(self, "_pre_waited", None)
            if pw is None or pw[0] is not pre[2] or pw[1] != cur.cuda_stream:
                cur.wait_event(pre[2])
            self._pre_waited = None
        caught_up = False
        if pre is not None and pre[0] == self._ids_key(item_seq, item_id, user_id):
            self._plans = pre[1]
            caught_up = pre[4] == self.t      # the tail of the previous step() already brought these rows to the state after step t
        else:
            self._plans = self._make_plans(item_seq, item_id, user_id)
        if self.table_mode == "lazy_dense" and self.t > 0 and not caught_up:
            cfg = self._cfg(self.t + 1)
            for name, pl in self._plans.items():
                st = self.tables[name]
                if st["last"] is not None:
                    ops.lazy_adam_catchup(cfg, st["w"], st["m"], st["v"], st["last"], pl)

    def _catchup_prefetched(self, rode=()):
        """step(), after the row update: the NEXT batch's rows (plan already made by `prefetch_plan`) take their zero-gradient steps
        up to and including this one now, on the main stream, while the dense-gradient reductions are still running on the side
        stream -- the same launch `plan_batch` would make at the head of the next step, moved into a slot where the main stream
        has nothing else to do."""
        pre = self._prefetched
        if pre is None or self.table_mode != "lazy_dense" or pre[4] is not None:
            return
        cur = torch.cuda.current_stream()
        pw = getattr(self, "_pre_waited", None)
        if pw is None or pw[0] is not pre[2] or pw[1] != cur.cuda_stream:
            cur.wait_event(pre[2])
        self._pre_waited = (pre[2], cur.cuda_stream)   # (the event object itself: an id() could be re-used by the next plan's event)
        cfg = self._cfg(self.t + 1)
        for name, pl in pre[1].items():
            st = self.tables[name]
            if st["last"] is not None and name not in rode:     # (rode: replayed inside this step's fused reduce + update launch)
                if pre[5] is not None and name in pre[5]:
                    pl = pre[5][name]
                ops.lazy_adam_catchup(cfg, st["w"], st["m"], st["v"], st["last"], pl)
        self._prefetched = pre[:4] + (self.t,) + pre[5:]

    def flush(self):
        """lazy_dense: apply all pending zero-gradient steps to every row (before eval / checkpoint)."""
        ops.id_guard_check()      # (an out-of-range id of the last steps: IndexError before anything is evaluated or saved)
        self._join_plan_stream()
        if self.table_mode != "lazy_dense" or self.t == 0:
            return
        self.model.join_side_updates()
        cfg = self._cfg(self.t)
        for st in self.tables.values():
            if st["last"] is not None:
                ops.lazy_adam_flush(cfg, st["w"], st["m"], st["v"], st["last"])

    def _join_plan_stream(self):
        """a plan made ahead replays rows of its batch on the plan stream (prefetch_plan): whoever reads or rewrites the tables outside a
        step (flush, checkpoints, evaluation) orders the current stream behind it first"""
        if self._prefetched is not None:
            torch.cuda.current_stream().wait_event(self._prefetched[2])

    def mark_tables_current(self):
        """The table rows were just replaced from outside (checkpoint load): they are up to date as of step ``t``, so no
        zero-gradient replay is pending for any of them."""
        self._join_plan_stream()
        for st in self.tables.values():
            if st["last"] is not None:
                st["last"].fill_(self.t)

    # ------------------------------------------------------------------ step
    def _collect(self, name):
        """-> (ids_a, rows_a, ids_b, coef, vec, G) for one table from model.sparse_grads."""
        a = [g for g in self.model.sparse_grads if g["table"] == name and "ids_a" in g]
        b = [g for g in self.model.sparse_grads if g["table"] == name and "ids_b" in g]
        if len(b) > 1:
            raise NotImplementedError("more than one scorer call per step")
        ids_a = rows = None
        if a:
            ids_a = a[0]["ids_a"] if len(a) == 1 else torch.cat([g["ids_a"] for g in a])
            rows = a[0]["rows"] if len(a) == 1 else torch.cat([g["rows"] for g in a])
        if b:
            return ids_a, rows, b[0]["ids_b"].reshape(-1), b[0]["coef"], b[0]["vec"], b[0]["G"]
        return ids_a, rows, None, None, None, 1

    def step(self, late_join=False):
        """late_join=True: the caller enqueues the model's next training forward pass right after this call (the training loop does): the
        dense half of the update, running on the encoder's side stream, is then joined by that forward pass instead of here.  With the
        default the current stream has joined when step() returns, and anything enqueued on it afterwards sees the updated parameters."""
        model = self.model
        dense_side = self._dense_side or ("late" if late_join else "join")
        self.t += 1
        cfg = self._cfg(self.t)
        reduced, rode = {}, set()
        guard = getattr(model, "loss_guard", None)
        for name, st in self.tables.items():
            ids_a, rows, ids_b, coef, vec, G = self._collect(name)
            if ids_a is None and ids_b is None:
                continue
            pl = self._plans.get(name)
            if pl is None:
                if self.table_mode == "lazy_dense" and self.t > 1:
                    raise RuntimeError("lazy_dense mode: call optimizer.plan_batch(...) before the forward pass")
                pl = ops.rows_plan(ids_a.contiguous() if ids_a is not None else None, ids_b, st["w"].shape[0])
            d = st["w"].shape[1]
            if self._fused_update and self.grad_clip is None and name not in model.dense_table_grads:
                # nothing needs this table's row gradients between the reduction and the update: one launch, the sums stay on chip
                split = None
                pre = self._prefetched
                if pre is not None and pre[4] is None and pre[6] and name in pre[6] and st["last"] is not None:
                    cur = torch.cuda.current_stream()    # the next batch's replay lists were made on the plan stream
                    pw = getattr(self, "_pre_waited", None)
                    if pw is None or pw[0] is not pre[2] or pw[1] != cur.cuda_stream:
                        cur.wait_event(pre[2])
                        self._pre_waited = (pre[2], cur.cuda_stream)
                    split = pre[6][name]
                    rode.add(name)
                ops.rows_reduce_update(cfg, st["w"], st["m"], st["v"], pl, rows, coef, vec, G, st["last"], guard, next_split=split)
                continue
            reduced[name] = (pl, ops.rows_reduce(pl, rows, coef, vec, G, d, zero_tail=self.grad_clip is not None))
        # fullsoftmax: the table's gradient is dense; the encoder's row-sparse part is folded into it
        dense_tables = dict(model.dense_table_grads)
        for name, dg in dense_tables.items():
            if name in reduced:
                pl, ug = reduced.pop(name)
                ops.rows_scatter_add(pl, ug, dg)
        # NaN guard (trainer.py:343-350: a step whose loss is NaN is not applied): the loss kernels left a device flag
        # (1, or -1 for NaN) that the update kernels read as their gradient scale -- < 0 = return untouched; no host round trip
        scale = guard
        sparse_done = False
        if self.grad_clip is None:
            # no global norm to wait for: the row-sparse half goes first, under the encoder's dense-gradient reductions that
            # may still be running on the side stream (model.defer_dense_join), then the join, then the dense half
            for name, (pl, ug) in reduced.items():
                st = self.tables[name]
                ops.sparse_adam_rows(cfg, st["w"], st["m"], st["v"], pl, ug, st["last"], scale)
            sparse_done = True
            self._catchup_prefetched(rode)
        side = None
        if (self.grad_clip is None and dense_side in ("late", "join") and getattr(model, "_deferred_dense_grad", None) is not None
                and model._deferred_dense_grad.numel()):
            side = ops.sasrec_side_stream()
        if side is not None:
            g = model._deferred_dense_grad
            with torch.cuda.stream(side):
                ops.dense_adam(cfg, model.dense_flat.data, g, self.dense_m, self.dense_v, scale)
            # (held until the main stream joins: the gradient buffer, and the row gradients the side stream's reductions read -- zero_grad()
            # drops both before the next forward pass, and the plan stream's buffers could land on their memory)
            # the guard / gradient scale is a view of the step's loss buffer: the side stream reads it too, so it is held as well (a
            # caller that drops the returned loss would otherwise hand that block back to the main stream's allocator under the read)
            ops.sasrec_side_publish(late=dense_side == "late",
                                    hold=(g,) + ((scale,) if scale is not None else ()) + tuple(getattr(model, "_deferred_reads", ())))
            model.dense_flat.grad = g
            object.__setattr__(model, "_deferred_dense_grad", None)
        model.finish_backward()
        if self.grad_clip is not None:
            ss = self._scalars[0:1]
            ss.zero_()
            if model.dense_flat.grad is not None and model.dense_flat.grad.numel():    # (MF: no dense encoder parameters)
                ops.sumsq(model.dense_flat.grad, ss, accumulate=True, ws=self._sumsq_ws)
            for p in self.extra:
                if p.grad is not None:
                    ops.sumsq(p.grad, ss, accumulate=True, ws=self._sumsq_ws)
            for pl, ug in reduced.values():
                ops.sumsq(ug, ss, accumulate=True, ws=self._sumsq_ws)
            for dg in dense_tables.values():
                ops.sumsq(dg, ss, accumulate=True, ws=self._sumsq_ws)
            scale = self._scalars[1:2]
            ops.clip_coef(ss, self.grad_clip, scale, guard=guard)
        if model.dense_flat.grad is not None and side is None:
            ops.dense_adam(cfg, model.dense_flat.data, model.dense_flat.grad, self.dense_m, self.dense_v, scale)
        for p, (m, v) in zip(self.extra, self.extra_state):
            if p.grad is not None:
                ops.dense_adam(cfg, p.data, p.grad.contiguous(), m, v, scale)
        if not sparse_done:
            for name, (pl, ug) in reduced.items():
                st = self.tables[name]
                ops.sparse_adam_rows(cfg, st["w"], st["m"], st["v"], pl, ug, st["last"], scale)
        for name, dg in dense_tables.items():
            st = self.tables[name]
            ops.dense_adam(cfg, st["w"], dg, st["m"], st["v"], scale)
        self._plans = {}
        if guard is not None:
            object.__setattr__(model, "loss_guard", None)
        model.sparse_grads.clear()
        model.dense_table_grads.clear()
