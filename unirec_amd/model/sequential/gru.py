"""GRU -- mirror of unirec/model/sequential/gru.py:9-35 on the HIP encoder (ur_gru_fwd / ur_gru_bwd).
state_dict names as the reference: gru_layers.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0}, dense.{weight,bias}."""
import math

import torch
import torch.nn as nn

from ... import ops
from ..base.recommender import BaseRecommender
from ..base.reco_abc import ParamHolder


class _GruEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, model, item_seq):
        cfg = model._cfg(item_seq.shape[0], train=True)
        ws = model._workspace(cfg, train=True)
        out = ops.gru_fwd(cfg, model.item_embedding.weight.data, dense.data, item_seq, ws)
        ctx.model, ctx.cfg, ctx.ws, ctx.gen = model, cfg, ws, model._ws_gen
        ctx.save_for_backward(item_seq)
        return out

    @staticmethod
    def backward(ctx, d_user):
        (item_seq,) = ctx.saved_tensors
        model = ctx.model
        model._ws_check(ctx.gen)
        dense_grad, d_rows = ops.gru_bwd(ctx.cfg, model.item_embedding.weight.data, model.dense_flat.data, item_seq, d_user.contiguous(), ctx.ws)
        model.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))
        return dense_grad, None, None


class GRU(BaseRecommender):
    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("SeqRecBase")

    def _cfg(self, B, train=False):
        """train=True: embedding dropout (gru.py:17,29, config dropout_prob) when the module is in training mode."""
        p = float(self.dropout_prob or 0.0)
        drop = train and self.training and p > 0
        if drop:
            object.__setattr__(self, "_drop_step", getattr(self, "_drop_step", 0) + 1)
        return ops.gru_cfg(B, self.config["max_seq_len"], self.embedding_size, self.hidden_size, p_drop=p if drop else 0.0,
                           drop_seed=int(self.config.get("dropout_seed", self.config.get("seed", 0)) or 0),
                           drop_step=getattr(self, "_drop_step", 0), mfma_arith=self.config.get("mfma_arith"))

    def _workspace(self, cfg, train=False):
        return self._ws_slot(cfg.B, train, lambda: ops.gru_workspace(cfg, self.device))

    def _check_seq(self, item_seq):
        if item_seq.dim() != 2 or item_seq.shape[1] != self.config["max_seq_len"]:
            raise ValueError(f"item_seq has shape {tuple(item_seq.shape)}, expected [B, max_seq_len={self.config['max_seq_len']}]")

    def _define_model_layers(self):
        d, H = self.embedding_size, self.hidden_size
        offs, total = ops.gru_param_layout(self._cfg(1))
        self._alloc_dense(total)
        v = self._view
        self.gru_layers = ParamHolder(weight_ih_l0=v(offs[0], (3 * H, d)), weight_hh_l0=v(offs[1], (3 * H, H)),
                                      bias_ih_l0=v(offs[2], (3 * H,)), bias_hh_l0=v(offs[3], (3 * H,)))
        self.dense = ParamHolder(weight=v(offs[4], (d, H)), bias=v(offs[5], (d,)))
        k = 1.0 / math.sqrt(H)   # nn.GRU keeps torch's default U(-1/sqrt(H), 1/sqrt(H)) (reco_abc.py init skips it)
        with torch.no_grad():
            for p in self.gru_layers.parameters():
                p.uniform_(-k, k)

    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        item_seq = item_seq.to(torch.int32).contiguous()
        self._check_seq(item_seq)
        cfg = self._cfg(item_seq.shape[0], train=True)
        ws = self._workspace(cfg, train=True)
        return ops.gru_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, ws), (cfg, ws, item_seq)

    def _encode_backward(self, state, d_user):
        cfg, ws, item_seq = state
        dense_grad, d_rows = ops.gru_bwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, d_user, ws)
        self.dense_flat.grad = dense_grad
        self.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))

    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        item_seq = item_seq.to(torch.int32).contiguous()
        self._check_seq(item_seq)
        if torch.is_grad_enabled() and self.training:
            return _GruEncoderFn.apply(self.dense_flat, self, item_seq)
        cfg = self._cfg(item_seq.shape[0])
        return ops.gru_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, self._workspace(cfg))
