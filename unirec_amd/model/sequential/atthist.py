"""AttHist -- mirror of unirec/model/sequential/atthist.py:9-23 on the HIP encoder (ur_atthist_fwd / ur_atthist_bwd).
state_dict names as the reference: attention.dense.{weight,bias}, attention.h ([d,1])."""
import torch
import torch.nn as nn

from ... import ops
from ..base.recommender import BaseRecommender
from ..base.reco_abc import ParamHolder


class _AttHistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, model, item_seq):
        cfg = model._cfg(item_seq.shape[0], item_seq.shape[1], train=True)
        ws = model._workspace(cfg, train=True)
        out = ops.atthist_fwd(cfg, model.item_embedding.weight.data, dense.data, item_seq, ws)
        ctx.model, ctx.cfg, ctx.ws, ctx.gen = model, cfg, ws, model._ws_gen
        ctx.save_for_backward(item_seq)
        return out

    @staticmethod
    def backward(ctx, d_user):
        (item_seq,) = ctx.saved_tensors
        model = ctx.model
        model._ws_check(ctx.gen)
        dense_grad, d_rows = ops.atthist_bwd(ctx.cfg, model.item_embedding.weight.data, model.dense_flat.data, item_seq, d_user.contiguous(), ctx.ws)
        model.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))
        return dense_grad, None, None


class AttHist(BaseRecommender):
    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("SeqRecBase")

    def _cfg(self, B, L, train=False):
        """train=True: dropout on the pooled output (modules.py:231,242, config dropout_prob) when the module is in training mode."""
        p = float(self.dropout_prob or 0.0)
        drop = train and self.training and p > 0
        if drop:
            object.__setattr__(self, "_drop_step", getattr(self, "_drop_step", 0) + 1)
        return ops.atthist_cfg(B, L, self.embedding_size, p_drop=p if drop else 0.0,
                               drop_seed=int(self.config.get("dropout_seed", self.config.get("seed", 0)) or 0),
                               drop_step=getattr(self, "_drop_step", 0))

    def _workspace(self, cfg, train=False):
        return self._ws_slot((cfg.B, cfg.L), train, lambda: ops.atthist_workspace(cfg, self.device))

    def _define_model_layers(self):
        d = self.embedding_size
        offs, total = ops.atthist_param_layout(self._cfg(1, 1))
        self._alloc_dense(total)
        v = self._view
        self.attention = nn.Module()
        self.attention.dense = ParamHolder(weight=v(offs[0], (d, d)), bias=v(offs[1], (d,)))
        self.attention.register_parameter("h", v(offs[2], (d, 1)))

    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        item_seq = item_seq.to(torch.int32).contiguous()
        cfg = self._cfg(*item_seq.shape, train=True)
        ws = self._workspace(cfg, train=True)
        return ops.atthist_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, ws), (cfg, ws, item_seq)

    def _encode_backward(self, state, d_user):
        cfg, ws, item_seq = state
        dense_grad, d_rows = ops.atthist_bwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, d_user, ws)
        self.dense_flat.grad = dense_grad
        self.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))

    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        item_seq = item_seq.to(torch.int32).contiguous()
        if torch.is_grad_enabled() and self.training:
            return _AttHistFn.apply(self.dense_flat, self, item_seq)
        cfg = self._cfg(*item_seq.shape)
        return ops.atthist_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, self._workspace(cfg))
