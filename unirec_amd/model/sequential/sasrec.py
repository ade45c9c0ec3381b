"""SASRec -- mirror of unirec/model/sequential/sasrec.py:10-76 on the HIP encoder.

Same config keys (sasrec.py:12-20) and the same state_dict names as the reference
(``position_embedding.weight``, ``LayerNorm.*``, ``trm_encoder.layer.{i}.multi_head_attention.{query,key,value,dense}.*``,
``...multi_head_attention.LayerNorm.*``, ``trm_encoder.layer.{i}.feed_forward.{dense_1,dense_2,LayerNorm}.*``);
every one of them is a view into ``model.dense_flat`` at the offset the C ABI reports
(``ur_sasrec_param_layout``), so ``ur_sasrec_fwd/_bwd`` read the weights without any packing step.
"""
import torch
import torch.nn as nn

from ... import ops
from ..base.recommender import BaseRecommender
from ..base.reco_abc import ParamHolder


class _SasrecEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, model, item_seq):
        cfg = model._cfg(item_seq.shape[0], train=True)
        ws = model._workspace(cfg, train=True)
        user_emb = ops.sasrec_fwd(cfg, model.item_embedding.weight.data, dense.data, item_seq, ws)
        ctx.model, ctx.cfg, ctx.ws, ctx.gen = model, cfg, ws, model._ws_gen
        ctx.save_for_backward(item_seq)
        return user_emb

    @staticmethod
    def backward(ctx, d_user):
        (item_seq,) = ctx.saved_tensors
        model = ctx.model
        model._ws_check(ctx.gen)
        dense_grad, d_rows = ops.sasrec_bwd(ctx.cfg, model.item_embedding.weight.data, model.dense_flat.data, item_seq,
                                            d_user.contiguous(), ctx.ws)
        model.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))
        return dense_grad, None, None


class SASRec(BaseRecommender):
    def __init__(self, config):
        self.n_layers = config["n_layers"]
        self.n_heads = config["n_heads"]
        self.inner_size = config["inner_size"]
        self.hidden_dropout_prob = float(config["hidden_dropout_prob"])
        self.attn_dropout_prob = float(config["attn_dropout_prob"])
        self.hidden_act = config["hidden_act"]
        self.layer_norm_eps = float(config["layer_norm_eps"])
        self.max_seq_len = config["max_seq_len"]
        self.use_pos_emb = config["use_position_emb"]
        super().__init__(config)

    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("SeqRecBase")

    def _cfg(self, B, train=False):
        """train=True: dropout on (nn.Dropout semantics: only in training mode), one fresh mask stream per call."""
        drop = train and self.training and (self.hidden_dropout_prob > 0 or self.attn_dropout_prob > 0)
        if drop:
            self._drop_step += 1
        return ops.sasrec_cfg(B, self.max_seq_len, self.hidden_size, self.n_heads, self.inner_size, self.n_layers,
                              self.hidden_act, self.use_pos_emb, self.layer_norm_eps,
                              last_only=int(self.config.get("last_row_only", 1)),
                              skip_padding=int(self.config.get("skip_padding", 1)),
                              p_hidden=self.hidden_dropout_prob if drop else 0.0, p_attn=self.attn_dropout_prob if drop else 0.0,
                              drop_seed=int(self.config.get("dropout_seed", self.config.get("seed", 0)) or 0), drop_step=self._drop_step,
                              mfma_arith=self.config.get("mfma_arith"))

    def _workspace(self, cfg, train=False):
        def alloc():
            big = ops.sasrec_cfg(cfg.B, cfg.L, cfg.d, cfg.n_heads, cfg.inner, cfg.n_layers, self.hidden_act, cfg.use_pos, cfg.eps,
                                 p_hidden=self.hidden_dropout_prob)     # the training layout (dropout scratch included) fits both
            return ops.sasrec_workspace(big, self.device)
        return self._ws_slot(cfg.B, train, alloc)   # one buffer per mode (batch size is fixed in training)

    def _define_model_layers(self):
        if self.hidden_size != self.embedding_size:
            raise ValueError("SASRec adds position embeddings of hidden_size to item embeddings of embedding_size: "
                             "they must be equal (sasrec.py:25,60-66)")
        object.__setattr__(self, "_drop_step", 0)
        d, I, L = self.hidden_size, self.inner_size, self.max_seq_len
        offs, total = ops.sasrec_param_layout(self._cfg(1))
        self._alloc_dense(total)
        v = self._view
        # sasrec.py:25 keeps the table even when unused only if use_pos_emb; we always keep the slot (zeros if unused)
        self.position_embedding = ParamHolder(weight=v(offs[0], (L + 1, d))) if self.use_pos_emb else None
        self.LayerNorm = ParamHolder(weight=v(offs[1], (d,)), bias=v(offs[2], (d,)))
        layers = []
        for i in range(self.n_layers):
            o = offs[3 + 16 * i: 3 + 16 * (i + 1)]
            mha = nn.Module()
            mha.query = ParamHolder(weight=v(o[0], (d, d)), bias=v(o[3], (d,)))
            mha.key = ParamHolder(weight=v(o[1], (d, d)), bias=v(o[4], (d,)))
            mha.value = ParamHolder(weight=v(o[2], (d, d)), bias=v(o[5], (d,)))
            mha.dense = ParamHolder(weight=v(o[6], (d, d)), bias=v(o[7], (d,)))
            mha.LayerNorm = ParamHolder(weight=v(o[8], (d,)), bias=v(o[9], (d,)))
            ff = nn.Module()
            ff.dense_1 = ParamHolder(weight=v(o[10], (I, d)), bias=v(o[11], (I,)))
            ff.dense_2 = ParamHolder(weight=v(o[12], (d, I)), bias=v(o[13], (d,)))
            ff.LayerNorm = ParamHolder(weight=v(o[14], (d,)), bias=v(o[15], (d,)))
            layer = nn.Module()
            layer.multi_head_attention, layer.feed_forward = mha, ff
            layers.append(layer)
        self.trm_encoder = nn.Module()
        self.trm_encoder.layer = nn.ModuleList(layers)

    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        item_seq = item_seq.to(torch.int32).contiguous()
        if item_seq.shape[1] != self.max_seq_len:
            raise ValueError(f"item_seq has length {item_seq.shape[1]}, expected max_seq_len={self.max_seq_len}")
        cfg = self._cfg(item_seq.shape[0], train=True)
        ws = self._workspace(cfg, train=True)
        return ops.sasrec_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, ws), (cfg, ws, item_seq)

    def _encode_backward(self, state, d_user):
        cfg, ws, item_seq = state
        defer = bool(getattr(self, "defer_dense_join", False))
        dense_grad, d_rows = ops.sasrec_bwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, d_user, ws,
                                            defer_join=defer)
        if defer:   # the reductions into dense_grad are still running on the side stream: finish_backward() publishes it
            self.dense_flat.grad = None
            object.__setattr__(self, "_deferred_dense_grad", dense_grad)
            object.__setattr__(self, "_deferred_reads", (d_rows,))   # what the side stream's reductions read besides the workspace (position-table gradient)
        else:
            self.dense_flat.grad = dense_grad
        self.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))

    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        item_seq = item_seq.to(torch.int32).contiguous()
        if item_seq.shape[1] != self.max_seq_len:
            raise ValueError(f"item_seq has length {item_seq.shape[1]}, expected max_seq_len={self.max_seq_len}")
        if torch.is_grad_enabled() and self.training:
            return _SasrecEncoderFn.apply(self.dense_flat, self, item_seq)
        cfg = self._cfg(item_seq.shape[0])
        return ops.sasrec_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, self._workspace(cfg))
