"""AvgHist -- mirror of unirec/model/sequential/avghist.py:9-42: the user embedding is the history's pooled item
embedding, ``(item_seq_len + 1)^(-alpha) * sum_l E_dst[item_seq[:, l]]``.  ``asymmetric`` (config, reference default
True) gives the history its own table ``item_dst_embedding`` (a copy of ``item_embedding`` at construction); otherwise
``item_src_embedding`` / ``item_dst_embedding`` are aliases of ``item_embedding``, as in the reference (so the
state_dict carries the same three key names)."""
import torch

from ... import ops
from ..base.recommender import BaseRecommender
from ..base.reco_abc import SparseTable


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, item_seq, item_seq_len, base):
        ctx.model = model
        ctx.save_for_backward(item_seq, item_seq_len)
        ctx.has_base = base is not None
        return ops.pool_rows_fwd(model.item_dst_embedding.weight.data, item_seq, item_seq_len, model.alpha,
                                 base.detach().contiguous() if base is not None else None)

    @staticmethod
    def backward(ctx, d_user):
        item_seq, item_seq_len = ctx.saved_tensors
        ctx.model._pool_backward(item_seq, item_seq_len, d_user.contiguous())
        return None, None, None, None, (d_user if ctx.has_base else None)


class AvgHist(BaseRecommender):
    def __init__(self, config):
        self.asymmetric = bool(config.get("asymmetric", True))
        self.alpha = float(config.get("user_sequence_alpha", 0.5))
        super().__init__(config)

    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("SeqRecBase")

    def _define_model_layers(self):
        self.item_src_embedding = self.item_embedding
        self.item_dst_embedding = SparseTable(self.n_items, self.embedding_size, self.device) if self.asymmetric else self.item_embedding
        self._alloc_dense(4)   # no dense parameters; a tiny buffer keeps the optimizer path uniform

    def _init_params(self):
        super()._init_params()
        if self.asymmetric:     # copy.deepcopy(self.item_embedding) in the reference: both tables start equal
            with torch.no_grad():
                self.item_dst_embedding.weight.copy_(self.item_embedding.weight)

    def lookup_tables(self):
        if self.asymmetric:
            return {"item_embedding": (None, "item_id"), "item_dst_embedding": ("item_seq", None)}
        return {"item_embedding": ("item_seq", "item_id")}

    def _dst_table_name(self):
        return "item_dst_embedding" if self.asymmetric else "item_embedding"

    def _pool_backward(self, item_seq, item_seq_len, d_user):
        rows = ops.pool_rows_bwd(d_user, item_seq_len, self.alpha, item_seq.shape[1])
        self.sparse_grads.append(dict(table=self._dst_table_name(), ids_a=item_seq.reshape(-1), rows=rows))

    def _prep(self, item_seq, item_seq_len):
        item_seq = item_seq.to(torch.int32).contiguous()
        if item_seq_len is None:
            raise ValueError(f"{type(self).__name__} needs item_seq_len (the (len+1)^-alpha normaliser)")
        return item_seq, item_seq_len.to(torch.int64).contiguous()

    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        item_seq, item_seq_len = self._prep(item_seq, item_seq_len)
        if torch.is_grad_enabled() and self.training:
            return _PoolFn.apply(self._anchor, self, item_seq, item_seq_len, None)
        return ops.pool_rows_fwd(self.item_dst_embedding.weight.data, item_seq, item_seq_len, self.alpha)

    def item_embedding_for_user(self, item_seq, item_seq_features=None, time_seq=None):
        return self.item_dst_embedding(item_seq)

    # fused (autograd-free) step hooks
    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        item_seq, item_seq_len = self._prep(item_seq, item_seq_len)
        return ops.pool_rows_fwd(self.item_dst_embedding.weight.data, item_seq, item_seq_len, self.alpha), (item_seq, item_seq_len)

    def _encode_backward(self, state, d_user):
        self._pool_backward(state[0], state[1], d_user)
