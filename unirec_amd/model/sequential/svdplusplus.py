"""SVDPlusPlus -- mirror of unirec/model/sequential/svdplusplus.py:10-40:
``user_emb = user_embedding[user_id] + (item_seq_len + 1)^(-alpha) * sum_l item_dst_embedding[item_seq[:, l]]`` with
``item_dst_embedding`` a separate copy of the item table (scoring uses ``item_src_embedding`` = ``item_embedding``)."""
import torch

from ... import ops
from .avghist import AvgHist, _PoolFn


class SVDPlusPlus(AvgHist):
    def __init__(self, config):
        config = dict(config)
        config["asymmetric"] = True      # svdplusplus.py:18-20: always two tables
        if not config.get("has_user_emb", False):
            raise ValueError("SVDPlusPlus needs has_user_emb=True (unirec/config/model/SVDPlusPlus.yaml)")
        super().__init__(config)

    def lookup_tables(self):
        return {"item_embedding": (None, "item_id"), "item_dst_embedding": ("item_seq", None), "user_embedding": ("user_id", None)}

    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        item_seq, item_seq_len = self._prep(item_seq, item_seq_len)
        if torch.is_grad_enabled() and self.training:
            from ..base.recommender import _TableLookupFn
            base = _TableLookupFn.apply(self._anchor, self, "user_embedding", user_id.contiguous())
            return _PoolFn.apply(self._anchor, self, item_seq, item_seq_len, base)
        return ops.pool_rows_fwd(self.item_dst_embedding.weight.data, item_seq, item_seq_len, self.alpha,
                                 self.user_embedding(user_id).contiguous())

    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        item_seq, item_seq_len = self._prep(item_seq, item_seq_len)
        base = self.user_embedding(user_id).contiguous()
        out = ops.pool_rows_fwd(self.item_dst_embedding.weight.data, item_seq, item_seq_len, self.alpha, base)
        return out, (item_seq, item_seq_len, user_id)

    def _encode_backward(self, state, d_user):
        item_seq, item_seq_len, user_id = state
        self._pool_backward(item_seq, item_seq_len, d_user)
        self.sparse_grads.append(dict(table="user_embedding", ids_a=user_id.to(torch.int32).contiguous(), rows=d_user.view(-1, d_user.shape[-1])))
