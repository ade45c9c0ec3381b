"""BaseRecommender -- mirror of unirec/model/base/recommender.py:14-197 on the HIP ops.

``forward`` keeps the reference's positional signature and return convention
(training: ``(loss, None, None, None)`` or all four; eval: ``(None, scores, user_emb, items_emb)``).
The scorer + loss is ONE fused HIP op (gather + dot + loss); the candidate tensor ``items_emb``
[B,G,d] is only materialised when the caller asks for it (eval / ``return_loss_only=False``).
"""
import inspect

import numpy as np
import torch

from ... import ops
from .reco_abc import AbstractRecommender


class _ScoreLossFn(torch.autograd.Function):
    """loss = _cal_loss(_predict_layer(user_emb, E[item_id]))  -- recommender.py:76-96, reco_abc.py:220-272."""

    @staticmethod
    def forward(ctx, user_emb, model, item_id, label, user_id):
        B, G = item_id.shape
        cfg = ops.loss_cfg(B, G, model.embedding_size, model.loss_type, model.tau, model.SCORE_CLIP,
                           model.config.get("ccl_w", 0.0), model.config.get("ccl_m", 0.0), model._group_rows(B, G))
        ub = model.user_bias.data if model.has_user_bias else None
        ib = model.item_bias.data if model.has_item_bias else None
        table = model.item_embedding.weight.data
        user_emb = user_emb.contiguous()
        scores, loss_rows, loss_out = ops.gather_dot_loss_fwd(cfg, user_emb, table, item_id, label, ub, ib,
                                                              user_id if ub is not None else None)
        ctx.model, ctx.cfg = model, cfg
        ctx.save_for_backward(user_emb, item_id, label if label is not None else item_id.new_empty(0), scores, loss_out,
                              user_id if user_id is not None else item_id.new_empty(0))
        ctx.mark_non_differentiable(scores, loss_rows)
        object.__setattr__(model, "loss_guard", loss_out[2:3])   # device flag: -1 = this step's loss is NaN (optimizer skips the update)
        return loss_out[0], scores, loss_rows

    @staticmethod
    def backward(ctx, d_loss, _ds, _dr):
        user_emb, item_id, label, scores, loss_out, user_id = ctx.saved_tensors
        model, cfg = ctx.model, ctx.cfg
        label = label if label.numel() else None
        d_loss = d_loss.contiguous().view(1).float()
        coef, d_user, d_ub = ops.gather_dot_loss_bwd(cfg, user_emb, model.item_embedding.weight.data, item_id, label, scores,
                                                     loss_out, d_loss, want_user_bias=model.has_user_bias)
        # implicit row gradient of the candidates: dE[item_id[b,g]] += coef[b,g] * user_emb[b]
        model.sparse_grads.append(dict(table="item_embedding", ids_b=item_id, coef=coef, vec=user_emb, G=cfg.G))
        if model.has_item_bias:   # bias vectors are tiny 1-d parameters: plain device index_add (not on the hot path)
            g = torch.zeros_like(model.item_bias)
            g.index_add_(0, item_id.reshape(-1), coef.reshape(-1))
            model.item_bias.grad = g if model.item_bias.grad is None else model.item_bias.grad + g
        if model.has_user_bias:
            g = torch.zeros_like(model.user_bias)
            g.index_add_(0, user_id, d_ub)
            model.user_bias.grad = g if model.user_bias.grad is None else model.user_bias.grad + g
        return d_user, None, None, None, None


class _FullSoftmaxFn(torch.autograd.Function):
    """loss = mean_b(logsumexp_n s(b,n) - s(b,target_b)) over ALL items -- recommender.py:46-55, reco_abc.py:266-270."""

    @staticmethod
    def forward(ctx, user_emb, model, target, user_id):
        user_emb = user_emb.contiguous()
        ub = model.user_bias.data if model.has_user_bias else None
        ib = model.item_bias.data if model.has_item_bias else None
        loss_out, lse, ws = ops.full_softmax_fwd(user_emb, model.item_embedding.weight.data, target, user_id if ub is not None else None,
                                                 ub, ib, model.tau, model.SCORE_CLIP)
        ctx.model, ctx.ws = model, ws
        ctx.save_for_backward(user_emb, target, lse, user_id if user_id is not None else target.new_empty(0))
        object.__setattr__(model, "loss_guard", loss_out[2:3])
        return loss_out[0]

    @staticmethod
    def backward(ctx, d_loss):
        user_emb, target, lse, user_id = ctx.saved_tensors
        model = ctx.model
        model._full_softmax_backward(user_emb, target, lse, ctx.ws, user_id if user_id.numel() else None,
                                     d_loss.contiguous().view(1).float())
        return model._fs_d_user, None, None, None


class _TableLookupFn(torch.autograd.Function):
    """user_emb = U[user_id] with a row-sparse gradient (MF: recommender.py:42-44)."""

    @staticmethod
    def forward(ctx, anchor, model, table_name, idx):
        ctx.model, ctx.table_name = model, table_name
        ctx.save_for_backward(idx)
        return getattr(model, table_name)(idx)

    @staticmethod
    def backward(ctx, d_out):
        (idx,) = ctx.saved_tensors
        ctx.model.sparse_grads.append(dict(table=ctx.table_name, ids_a=idx.to(torch.int32).contiguous(),
                                           rows=d_out.contiguous().view(-1, d_out.shape[-1])))
        return None, None, None, None


class BaseRecommender(AbstractRecommender):
    def _init_attributes(self):
        super()._init_attributes()
        self.dnn_inner_size = self.embedding_size
        self.time_seq = 0

    def _init_modules(self):
        scorer_type = self.config["distance_type"]
        if scorer_type != "dot":
            raise NotImplementedError(f"distance_type={scorer_type!r}: only the dot-product scorer is on the accelerated path")
        # autograd needs one differentiable input to hang the table-lookup node on
        object.__setattr__(self, "_anchor", torch.zeros(1, device=self.device, requires_grad=True))
        super()._init_modules()

    def _define_model_layers(self):
        pass

    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("BaseRecommender")

    # ---- encoders ---------------------------------------------------------------------------
    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        if torch.is_grad_enabled() and self.training:
            return _TableLookupFn.apply(self._anchor, self, "user_embedding", user_id.contiguous())
        return self.user_embedding(user_id)

    def forward_item_emb(self, items, item_features=None):
        return self.item_embedding(items)

    def item_embedding_for_user(self, item_seq, item_seq_features=None, time_seq=None):
        return self.item_embedding(item_seq)

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, user_id=None, item_id=None, label=None, item_features=None, item_seq=None, item_seq_len=None,
                item_seq_features=None, time_seq=None, session_id=None, reduction=True, return_loss_only=True, max_len=None):
        if self.loss_type == "fullsoftmax" and self.training:
            # label = item_id, candidates = every item (recommender.py:47-50)
            if getattr(self, "_fs_shard_step", None) is not None:
                raise NotImplementedError("fullsoftmax over a row-sharded table runs through forward_backward (the Trainer's fused step)")
            if not (reduction and return_loss_only):
                raise NotImplementedError("fullsoftmax: the [B, n_items] score matrix is never materialised (reduction / return_loss_only)")
            user_emb = self.forward_user_emb(user_id, item_seq, item_seq_len, item_seq_features, time_seq)
            target = item_id.reshape(item_id.shape[0], -1)[:, 0].contiguous()
            return _FullSoftmaxFn.apply(user_emb, self, target, user_id), None, None, None
        squeeze = item_id.dim() == 1     # one candidate per row (user-item-label rows): the reference's scores are [B] then (modules.py:53-55)
        if squeeze:
            item_id = item_id.unsqueeze(1)
            label = label.unsqueeze(1) if label is not None and label.dim() == 1 else label
        item_id = item_id.contiguous()
        user_emb = self.forward_user_emb(user_id, item_seq, item_seq_len, item_seq_features, time_seq)
        lab = label.to(torch.int32).contiguous() if label is not None else None
        if self.training:
            loss, scores, loss_rows = _ScoreLossFn.apply(user_emb, self, item_id, lab, user_id)
            if not reduction:
                if self.loss_type in ("bpr", "ccl"):
                    gs = self._group_rows(*item_id.shape)
                    loss = loss_rows[: item_id.shape[0] // gs if gs else item_id.shape[0]]
                else:
                    raise NotImplementedError("reduction=False is implemented for bpr/ccl only")
            if return_loss_only:
                return loss, None, None, None
            if squeeze:
                return loss, scores.squeeze(1), user_emb, self.forward_item_emb(item_id.squeeze(1))
            return loss, scores, user_emb, self.forward_item_emb(item_id)
        scores = self._predict_layer(user_emb, None, user_id, item_id)
        if squeeze:
            return None, scores.squeeze(1), user_emb, self.forward_item_emb(item_id.squeeze(1))
        return None, scores, user_emb, self.forward_item_emb(item_id)

    def _full_softmax_backward(self, user_emb, target, lse, ws, user_id, d_loss):
        ub = self.user_bias.data if self.has_user_bias else None
        ib = self.item_bias.data if self.has_item_bias else None
        d_user, d_table, d_ib = ops.full_softmax_bwd(user_emb, self.item_embedding.weight.data, target, lse, ws,
                                                     user_id if ub is not None else None, ub, ib, self.tau, self.SCORE_CLIP, d_loss)
        self.dense_table_grads["item_embedding"] = d_table      # dense: every row of the table moves (the reference's cost too)
        if self.has_item_bias:
            self.item_bias.grad = d_ib
        if self.has_user_bias:
            self.user_bias.grad = torch.zeros_like(self.user_bias)   # softmax is shift invariant along n: exactly 0
        object.__setattr__(self, "_fs_d_user", d_user)

    def lookup_tables(self):
        """Which batch field indexes which row-sparse table: {table: (explicit-row ids, scorer candidate ids)}; the optimizer
        plans the batch's id sort from it.  Default: MF-style (user table <- user_id, item table <- scorer candidates);
        sequence models add item_seq to the item table."""
        # fullsoftmax TRAINING: every item is a candidate, their gradient is dense; evaluation scores item_id like every other loss
        cand = None if (self.loss_type == "fullsoftmax" and self.training) else "item_id"
        spec = {"item_embedding": ("item_seq" if "SeqRecBase" in self.annotations else None, cand)}
        if hasattr(self, "user_embedding"):
            spec["user_embedding"] = ("user_id", None)
        return spec

    # ---- fused training step (no autograd graph) ------------------------------------------------
    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        """-> (user_emb, state for _encode_backward).  MF: the user table lookup."""
        return self.user_embedding(user_id), user_id

    def _encode_backward(self, state, d_user):
        self.sparse_grads.append(dict(table="user_embedding", ids_a=state.to(torch.int32).contiguous(),
                                      rows=d_user.view(-1, d_user.shape[-1])))

    # ---- activation workspaces.  One per mode: evaluation forwards never touch the buffer a pending backward will read; every
    # TRAINING forward bumps a generation counter, and an autograd backward whose saved activations were overwritten by a
    # later training forward of the same model fails loudly instead of producing silently wrong gradients (the reference's
    # autograd keeps activations per call; here they live in ONE caller-owned buffer per mode).
    def _ws_slot(self, key, train, alloc):
        slots = self.__dict__.get("_ws_slots")
        if slots is None:
            slots = {}
            object.__setattr__(self, "_ws_slots", slots)
        s = slots.get(bool(train))
        if s is None or s[0] != key:
            s = (key, alloc())
            slots[bool(train)] = s
        if train:
            object.__setattr__(self, "_ws_gen", getattr(self, "_ws_gen", 0) + 1)
        return s[1]

    def _ws_check(self, gen):
        if gen != getattr(self, "_ws_gen", 0):
            raise RuntimeError("the activation workspace of this forward pass was overwritten by a later training forward of the same "
                               "model before backward() ran: call backward() before the next training forward (evaluation forwards "
                               "use their own workspace and are fine)")

    # An optimizer that runs the row-sparse half of its step first sets ``defer_dense_join``: the encoder backward may then
    # return while its dense-gradient reductions are still running on a side stream (ur_sasrec_bwd_deferred);
    # ``dense_flat.grad`` stays None until ``finish_backward()`` joins them.  Default off: backward leaves complete gradients.
    defer_dense_join = False
    _deferred_dense_grad = None
    _deferred_reads = ()

    def join_side_updates(self):
        """The optimizer may leave the dense half of its step running on the encoder's side stream (joined by the next encoder forward
        pass): anything else that reads parameters or their optimizer state on the current stream calls this first."""
        ops.sasrec_side_join()

    def state_dict(self, *args, **kwargs):
        self.join_side_updates()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.join_side_updates()
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        # (train() reads no parameter: a call that leaves the mode as it is -- the training loop's, at the head of every step -- must not
        # take the late join away from the forward pass that hides it)
        if bool(mode) != self.training:
            self.join_side_updates()
        return super().train(mode)

    def finish_backward(self):
        g = self._deferred_dense_grad
        if g is not None:
            ops.sasrec_bwd_join()
            self.dense_flat.grad = g
            object.__setattr__(self, "_deferred_dense_grad", None)

    def forward_backward(self, user_id=None, item_id=None, label=None, item_seq=None, item_seq_len=None, **_unused):
        """``loss = model(...); loss.backward()`` (unirec/facility/trainer.py:340-346) as one straight-line sequence of
        the same HIP launches, without building/walking an autograd graph: leaves ``dense_flat.grad``, the bias
        gradients and ``sparse_grads`` exactly as ``backward()`` would, and returns the (detached) loss.
        The autograd engine hands backward to its device thread, which leaves the GPU idle for ~0.1 ms per step at
        this step size; the trainer therefore uses this entry point (config ``fused_step``, default on)."""
        if not self.training:
            raise RuntimeError("forward_backward is a training-mode entry point")
        if self.loss_type == "fullsoftmax":
            user_emb, state = self._encode_train(user_id, item_seq, item_seq_len)
            user_emb = user_emb.contiguous()
            target = item_id.reshape(item_id.shape[0], -1)[:, 0].contiguous()
            ub = self.user_bias.data if self.has_user_bias else None
            ib = self.item_bias.data if self.has_item_bias else None
            uid = user_id if ub is not None else None
            shard_step = getattr(self, "_fs_shard_step", None)
            if shard_step is not None:    # row-sharded catalogue (facility/distributed.py): every rank scores all users against ITS rows
                loss_out, d_user = shard_step(user_emb, target, user_id)
                object.__setattr__(self, "_fs_d_user", d_user)
            else:
                loss_out, lse, ws = ops.full_softmax_fwd(user_emb, self.item_embedding.weight.data, target, uid, ub, ib, self.tau, self.SCORE_CLIP)
                self._full_softmax_backward(user_emb, target, lse, ws, user_id, None)
            self._encode_backward(state, self._fs_d_user)
            object.__setattr__(self, "loss_guard", loss_out[2:3])
            return loss_out[0]
        if item_id.dim() == 1:
            item_id = item_id.unsqueeze(1)
            label = label.unsqueeze(1) if label is not None and label.dim() == 1 else label
        item_id = item_id.contiguous()
        lab = label.to(torch.int32).contiguous() if label is not None else None
        user_emb, state = self._encode_train(user_id, item_seq, item_seq_len)
        user_emb = user_emb.contiguous()
        B, G = item_id.shape
        cfg = ops.loss_cfg(B, G, self.embedding_size, self.loss_type, self.tau, self.SCORE_CLIP,
                           self.config.get("ccl_w", 0.0), self.config.get("ccl_m", 0.0), self._group_rows(B, G))
        ub = self.user_bias.data if self.has_user_bias else None
        ib = self.item_bias.data if self.has_item_bias else None
        table = self.item_embedding.weight.data
        scores, loss_out, coef, d_user, d_ub = ops.gather_dot_loss_fwd_bwd(cfg, user_emb, table, item_id, lab, ub, ib,
                                                                           user_id if ub is not None else None,
                                                                           want_user_bias=self.has_user_bias)
        self.sparse_grads.append(dict(table="item_embedding", ids_b=item_id, coef=coef, vec=user_emb, G=G))
        if self.has_item_bias:
            g = torch.zeros_like(self.item_bias)
            g.index_add_(0, item_id.reshape(-1), coef.reshape(-1))
            self.item_bias.grad = g
        if self.has_user_bias:
            g = torch.zeros_like(self.user_bias)
            g.index_add_(0, user_id, d_ub)
            self.user_bias.grad = g
        self._encode_backward(state, d_user)
        object.__setattr__(self, "loss_guard", loss_out[2:3])   # device flag: -1 = this step's loss is NaN (optimizer skips the update)
        return loss_out[0]

    def _group_rows(self, B, G):
        """``group_size`` of the loss's reshape (reco_abc.py:233-236: ``scores.view(-1, group_size)``) for a [B, G] candidate matrix.
        G == group_size: the rows already are the groups (the view is the identity).  G == 1 -- the user-item-label format, one
        (user, item, label) triple per row -- regroups every group_size consecutive rows into one score row (UrLossCfg.group_size)."""
        gs = int(self.group_size) if self.group_size else 0
        if gs <= 0 or G == gs:
            return 0
        if G != 1 or B % gs:
            raise ValueError(f"group_size={gs}: scores of shape [{B}, {G}] cannot be viewed as [-1, {gs}] row groups")
        return gs

    def _predict_layer(self, user_emb, items_emb, user_id, item_id):
        """scores only (no loss), through the same fused gather-dot kernel; items_emb is ignored."""
        if item_id.dim() == 1:
            item_id = item_id.unsqueeze(1)
        B, G = item_id.shape
        cfg = ops.loss_cfg(B, G, self.embedding_size, "bpr", self.tau, self.SCORE_CLIP)
        cfg.loss_type = -1  # UR_LOSS_NONE: scores only
        ub = self.user_bias.data if self.has_user_bias else None
        ib = self.item_bias.data if self.has_item_bias else None
        scores, _, _ = ops.gather_dot_loss_fwd(cfg, user_emb.detach().contiguous(), self.item_embedding.weight.data,
                                               item_id.contiguous(), None, ub, ib, user_id if ub is not None else None)
        return scores

    def predict(self, interaction):
        inputs = {k: v for k, v in interaction.items() if k in inspect.signature(self.forward_user_emb).parameters}
        with torch.no_grad():
            user_emb = self.forward_user_emb(**inputs)
            scores = self._predict_layer(user_emb, None, interaction.get("user_id"), interaction["item_id"])
        return scores.detach().cpu().numpy()

    def forward_all_item_emb(self, batch_size=None, numpy=True):
        w = self.item_embedding.weight.detach()
        return w.cpu().numpy() if numpy else w.clone()

    def get_all_item_bias(self):
        return self.item_bias.detach().cpu().numpy()

    def get_user_bias(self, interaction):
        return self.user_bias[interaction["user_id"]].detach().cpu().numpy()

    def topk(self, interaction, k, user_hist=None, candidates=None):
        """Top-k items for a batch of users -- recommender.py:149-197 with candidates=None (all items).
        user_hist: the reference's padded [B, H] tensor of history items (0 = padding), or a HistoryCSR (then
        interaction['user_id'] selects the rows), or None.  Returns (scores [B,k], ids [B,k]) on the device, best first.
        Item 0 (the padding row) is never returned."""
        if candidates is not None:
            raise NotImplementedError("topk over an explicit candidate list: use predict() on the candidates")
        inputs = {kk: v for kk, v in interaction.items() if kk in inspect.signature(self.forward_user_emb).parameters}
        with torch.no_grad():
            user_emb = self.forward_user_emb(**inputs).contiguous()
        B = user_emb.shape[0]
        dev = user_emb.device
        uid = interaction.get("user_id")
        hp = hs = None
        hist_uid = uid
        if user_hist is not None:
            if torch.is_tensor(user_hist):   # padded per-row histories -> a batch-local CSR (rows sorted, zeros first)
                hs = torch.sort(user_hist.to(dev).to(torch.int32), dim=1).values.reshape(-1).contiguous()
                hp = (torch.arange(B + 1, device=dev, dtype=torch.int64) * user_hist.shape[1]).contiguous()
                hist_uid = torch.arange(B, device=dev, dtype=torch.int64)
            else:
                hp, hs = user_hist.to_device(dev)
        if self.has_user_bias and hist_uid is not uid:
            # two different row keys (history rows vs user ids): fold the user bias in afterwards
            scores, ids = ops.full_topk(user_emb, self.item_embedding.weight.data, k, hist_uid, hp, hs, None,
                                        self.item_bias.data if self.has_item_bias else None, 1.0)
            scores = torch.where(torch.isinf(scores), scores, (scores + self.user_bias.data[uid].unsqueeze(1)) / self.tau)
            return scores, ids
        return ops.full_topk(user_emb, self.item_embedding.weight.data, k, hist_uid if (hp is not None or self.has_user_bias) else None,
                             hp, hs, self.user_bias.data if self.has_user_bias else None,
                             self.item_bias.data if self.has_item_bias else None, self.tau)
