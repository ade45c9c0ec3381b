"""AbstractRecommender -- host-side mirror of unirec/model/base/reco_abc.py:61-272.

Same constructor contract (``Model(config: dict)``), attribute names and state_dict key names as the
reference, but no arithmetic happens here: every tensor op is a call into the HIP library
(unirec_amd.ops).  Differences that matter to a caller, all by design:

* embedding tables are ``SparseTable`` modules: ``weight`` has the reference's name and shape but is
  updated by row-sparse gradients; no dense ``[N, d]`` gradient is ever created.  After
  ``loss.backward()`` the row gradients are queued in ``model.sparse_grads`` for the optimizer.
* all other ("dense") parameters are views into ONE flat fp32 buffer ``model.dense_flat`` so the HIP
  encoder, the optimizer and the gradient all-reduce each see a single array.
"""
import logging

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...constants import LOSS_TYPES


class SparseTable(nn.Module):
    """Stand-in for nn.Embedding(n, d, padding_idx=0) (reco_abc.py:168,170): same ``weight`` name/shape."""

    def __init__(self, n_rows, dim, device, padding_idx=0):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = n_rows, dim, padding_idx
        self.weight = nn.Parameter(torch.zeros(n_rows, dim, dtype=torch.float32, device=device), requires_grad=False)

    def forward(self, idx):
        return ops.embedding_gather(self.weight.data, idx.contiguous())

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}, row-sparse gradients"


class ParamHolder(nn.Module):
    """Names a group of parameters (e.g. ``query.weight``/``query.bias``) that live in the flat buffer."""

    def __init__(self, **params):
        super().__init__()
        for k, v in params.items():
            self.register_parameter(k, v)


class AbstractRecommender(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.logger = logging.getLogger(config.get("exp_name", "unirec_amd"))
        self.__optimized_by_SGD__ = True
        self.config = config
        self.sparse_grads = []     # filled by backward: dicts consumed by facility.optimizer
        self.dense_table_grads = {}  # fullsoftmax: {table name: dense [N,d] gradient} (every row moves)
        self._init_attributes()
        self._init_modules()
        self.annotations = []
        self.add_annotation()
        self._parameter_validity_check()

    # ---- reference hooks -----------------------------------------------------------------
    def _parameter_validity_check(self):
        if self.loss_type == "softmax" and self.config.get("train_file_format") in ("user-item-label", "user-item-label-session") \
                and self.group_size <= 0:
            raise ValueError("softmax loss on user-item-label data needs a positive group_size")  # reco_abc.py:83-90
        if self.loss_type not in LOSS_TYPES:
            raise ValueError(f"unknown loss_type {self.loss_type}")

    def _define_model_layers(self):
        raise NotImplementedError

    def add_annotation(self):
        self.annotations.append("AbstractRecommender")

    def _init_attributes(self):  # reco_abc.py:124-157
        config = self.config
        self.n_users = config["n_users"]
        self.n_items = config["n_items"]
        self.device = torch.device(config["device"])
        if self.device.type != "cuda":
            raise RuntimeError(f"unirec_amd models run on an MI355X only (config['device']={config['device']!r}); "
                               "there is no CPU fallback")
        self.loss_type = config.get("loss_type", "bce")
        self.embedding_size = config.get("embedding_size", 0)
        self.hidden_size = config.get("hidden_size", self.embedding_size)
        self.dropout_prob = config.get("dropout_prob", 0.0)
        self.init_method = config.get("init_method", "normal")
        for unsupported in ("use_features", "use_text_emb", "use_pre_item_emb", "time_seq"):
            if config.get(unsupported, 0):
                raise NotImplementedError(f"config['{unsupported}'] is outside the accelerated hot path (DESIGN.md section 7)")
        self.use_features = self.use_text_emb = False
        self.group_size = config.get("group_size", -1)
        self.SCORE_CLIP = config.get("score_clip_value", -1) or -1
        self.has_user_bias = bool(config.get("has_user_bias", False))
        self.has_item_bias = bool(config.get("has_item_bias", False))
        if self.loss_type == "fullsoftmax" and self.has_user_bias and self.SCORE_CLIP > 0:
            # the fullsoftmax kernels take d loss / d user_bias as exactly 0 (a per-user shift cancels in the softmax over n); a score clamp
            # (recommender.py:94-95) breaks that invariance wherever it saturates, and that term is not computed: refuse instead of drifting
            raise NotImplementedError("loss_type='fullsoftmax' with has_user_bias and score_clip_value > 0: the user-bias gradient under an "
                                      "active score clamp is not implemented (drop the clamp or the user bias)")
        self.tau = config.get("tau", 1.0)

    def _init_modules(self):  # reco_abc.py:159-208
        dev = self.device
        if self.has_user_bias:
            self.user_bias = nn.Parameter(torch.normal(0, 0.1, size=(self.n_users,), device=dev))
        if self.has_item_bias:
            self.item_bias = nn.Parameter(torch.normal(0, 0.1, size=(self.n_items,), device=dev))
        if self.config["has_user_emb"]:
            self.user_embedding = SparseTable(self.n_users, self.embedding_size, dev)
        self.item_embedding = SparseTable(self.n_items, self.embedding_size, dev)
        self._define_model_layers()
        self._init_params()

    # ---- flat dense buffer helpers ---------------------------------------------------------
    def _alloc_dense(self, total):
        """Allocate the flat buffer of all dense parameters; kept OUT of the module registry."""
        flat = nn.Parameter(torch.zeros(total, dtype=torch.float32, device=self.device))
        object.__setattr__(self, "dense_flat", flat)
        return flat

    def _view(self, off, shape):
        n = int(np.prod(shape))
        return nn.Parameter(self.dense_flat.data[off:off + n].view(*shape))

    def check_views(self):
        """Parameters must still alias the flat buffer (``.to()`` / manual reassignment would break it)."""
        lo = self.dense_flat.data_ptr()
        hi = lo + self.dense_flat.numel() * 4
        for name, p in self.named_parameters():
            if name in self._sparse_param_names():
                continue
            if p.dim() and p.requires_grad and not (lo <= p.data_ptr() < hi) and name not in ("user_bias", "item_bias"):
                raise RuntimeError(f"parameter {name} no longer aliases model.dense_flat; do not move/replace parameters")

    def _sparse_param_names(self):
        return {"item_embedding.weight", "user_embedding.weight", "item_dst_embedding.weight", "item_src_embedding.weight"}

    def _init_params(self):  # reco_abc.py:210-218 + :19-58
        method = self.init_method
        mean, std = self.config.get("init_mean", 0.0), self.config.get("init_std", 0.02)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name in ("user_bias", "item_bias"):
                    continue
                if name.endswith("LayerNorm.weight"):
                    p.fill_(1.0)
                elif "depthwise_conv" in name:
                    continue   # nn.Conv1d of the ConvFormer layers: initialised at construction, no init hook touches it
                elif name.endswith(".bias") or name.endswith("LayerNorm.bias"):
                    p.zero_()
                elif name.startswith("gru_layers."):
                    continue  # nn.GRU keeps torch's default U(-1/sqrt(H), 1/sqrt(H)) (set by the GRU model)
                elif "depthwise_conv" in name or name.endswith("conv_weight"):
                    continue   # set at construction by the ConvFormer / FASTConvFormer layers (no init hook touches them)
                elif name == "attention.h":
                    p.normal_(0.0, 1.0)   # a bare nn.Parameter(torch.randn(...)): no init hook touches it (modules.py:233)
                elif p.dim() >= 2:
                    if method == "normal":
                        p.normal_(mean=mean, std=std)
                    elif method == "xavier_normal":
                        nn.init.xavier_normal_(p)
                    elif method == "xavier_uniform":
                        nn.init.xavier_uniform_(p)
                    else:
                        raise KeyError(method)
            for t in ("item_embedding", "user_embedding", "item_dst_embedding"):
                if hasattr(self, t):
                    getattr(self, t).weight[0].zero_()  # padding row

    def __str__(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters())
        return super().__str__() + f"\nparameter number (incl. row-sparse tables): {n}"
