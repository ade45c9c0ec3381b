"""ConvFormer -- mirror of unirec/model/sequential/convformer.py:16-129 on the HIP encoder (ur_convformer_fwd / _bwd).
state_dict names as the reference: position_embedding.weight [L,d], LayerNorm.*, encoder.{i}.filterlayer.conv.depthwise_conv.{weight
[d,1,K], bias}, encoder.{i}.filterlayer.LayerNorm.*, encoder.{i}.intermediate.{dense_1,dense_2,LayerNorm}.*"""
import torch
import torch.nn as nn

from ... import ops
from ..base.recommender import BaseRecommender
from ..base.reco_abc import ParamHolder


class _ConvFormerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, model, item_seq, item_seq_len):
        cfg = model._cfg(item_seq.shape[0], train=True)
        ws = model._workspace(cfg, train=True)
        out = ops.convformer_fwd(cfg, model.item_embedding.weight.data, dense.data, item_seq, item_seq_len, ws)
        ctx.model, ctx.cfg, ctx.ws, ctx.gen = model, cfg, ws, model._ws_gen
        ctx.save_for_backward(item_seq, item_seq_len if item_seq_len is not None else item_seq.new_empty(0, dtype=torch.int64))
        return out

    @staticmethod
    def backward(ctx, d_user):
        item_seq, item_seq_len = ctx.saved_tensors
        model = ctx.model
        model._ws_check(ctx.gen)
        dense_grad, d_rows = ops.convformer_bwd(ctx.cfg, model.item_embedding.weight.data, model.dense_flat.data, item_seq,
                                                item_seq_len if item_seq_len.numel() else None, d_user.contiguous(), ctx.ws)
        model.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))
        return dense_grad, None, None, None


class ConvFormer(BaseRecommender):
    FAST = False

    def __init__(self, config):
        self.conv_size = config["conv_size"]
        self.padding_mode = config.get("padding_mode", "circular")
        self.n_layers = config["n_layers"]
        self.inner_size = config["inner_size"]
        self.hidden_act = config.get("hidden_act", "gelu")
        self.layer_norm_eps = float(config["layer_norm_eps"])
        self.hidden_dropout_prob = float(config.get("hidden_dropout_prob", 0.0))
        self.max_seq_len = config["max_seq_len"]
        self.seq_decay = float(config.get("seq_decay", -0.3))
        self.seq_merge = bool(config.get("seq_merge", False))
        self.init_ratio = float(config.get("init_ratio", 0.005))
        if self.conv_size > self.max_seq_len:
            raise ValueError(f"`conv_size` should be smaller than `max_seq_len`, while get `conv={self.conv_size}` and "
                             f"`max_seq_len={self.max_seq_len}`.")   # convformer.py:32-33
        super().__init__(config)

    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("SeqRecBase")

    def _act(self):
        return self.hidden_act

    def _padding_mode(self):
        return self.padding_mode

    def _cfg(self, B, train=False, p_hidden=None):
        """train=True: hidden dropout on when the module is in training mode (nn.Dropout semantics), a fresh mask stream per call."""
        drop = train and self.training and self.hidden_dropout_prob > 0
        if drop:
            self._drop_step += 1
        p = p_hidden if p_hidden is not None else (self.hidden_dropout_prob if drop else 0.0)
        return ops.convformer_cfg(B, self.max_seq_len, self.hidden_size, self.inner_size, self.n_layers, self._act(), self.conv_size,
                                  self._padding_mode(), self.FAST, self.seq_merge, self.layer_norm_eps, self.seq_decay, p_hidden=p,
                                  drop_seed=int(self.config.get("dropout_seed", self.config.get("seed", 0)) or 0), drop_step=self._drop_step)

    def _workspace(self, cfg, train=False):
        return self._ws_slot(cfg.B, train, lambda: ops.convformer_workspace(self._cfg(cfg.B, p_hidden=self.hidden_dropout_prob),
                                                                            self.device))   # the training layout fits both

    def _mixer_holder(self, v, o, d, K):
        """parameters of one layer's mixer, named as in the reference"""
        fl = nn.Module()
        fl.conv = nn.Module()
        fl.conv.depthwise_conv = ParamHolder(weight=v(o[0], (d, 1, K)), bias=v(o[1], (d,)))
        with torch.no_grad():   # nn.Conv1d created, then weight / bias ~ N(0, init_ratio) (convformer.py:82-85)
            fl.conv.depthwise_conv.weight.normal_(0.0, self.init_ratio)
            fl.conv.depthwise_conv.bias.normal_(0.0, self.init_ratio)
        return fl

    def _define_model_layers(self):
        if self.hidden_size != self.embedding_size:
            raise ValueError("ConvFormer adds position embeddings of hidden_size to item embeddings of embedding_size: they must be equal")
        object.__setattr__(self, "_drop_step", 0)
        d, I, L, K = self.hidden_size, self.inner_size, self.max_seq_len, self.conv_size
        offs, total = ops.convformer_param_layout(self._cfg(1))
        self._alloc_dense(total)
        v = self._view
        self.position_embedding = ParamHolder(weight=v(offs[0], (L, d)))
        self.LayerNorm = ParamHolder(weight=v(offs[1], (d,)), bias=v(offs[2], (d,)))
        layers = []
        for i in range(self.n_layers):
            o = offs[3 + 10 * i: 3 + 10 * (i + 1)]
            layer = nn.Module()
            layer.filterlayer = self._mixer_holder(v, o, d, K)
            layer.filterlayer.LayerNorm = ParamHolder(weight=v(o[2], (d,)), bias=v(o[3], (d,)))
            im = nn.Module()
            im.dense_1 = ParamHolder(weight=v(o[4], (I, d)), bias=v(o[5], (I,)))
            im.dense_2 = ParamHolder(weight=v(o[6], (d, I)), bias=v(o[7], (d,)))
            im.LayerNorm = ParamHolder(weight=v(o[8], (d,)), bias=v(o[9], (d,)))
            layer.intermediate = im
            layers.append(layer)
        self.encoder = nn.ModuleList(layers)

    def _prep(self, item_seq, item_seq_len):
        item_seq = item_seq.to(torch.int32).contiguous()
        if item_seq.shape[1] != self.max_seq_len:
            raise ValueError(f"item_seq has length {item_seq.shape[1]}, expected max_seq_len={self.max_seq_len}")
        if self.seq_merge and item_seq_len is None:
            raise ValueError("seq_merge needs item_seq_len")
        return item_seq, (item_seq_len.to(torch.int64).contiguous() if item_seq_len is not None else None)

    def _encode_train(self, user_id, item_seq, item_seq_len=None):
        item_seq, item_seq_len = self._prep(item_seq, item_seq_len)
        cfg = self._cfg(item_seq.shape[0], train=True)
        ws = self._workspace(cfg, train=True)
        out = ops.convformer_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, item_seq_len, ws)
        return out, (cfg, ws, item_seq, item_seq_len)

    def _encode_backward(self, state, d_user):
        cfg, ws, item_seq, item_seq_len = state
        dense_grad, d_rows = ops.convformer_bwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, item_seq_len, d_user, ws)
        self.dense_flat.grad = dense_grad
        self.sparse_grads.append(dict(table="item_embedding", ids_a=item_seq.reshape(-1), rows=d_rows))

    def forward_user_emb(self, user_id=None, item_seq=None, item_seq_len=None, item_seq_features=None, time_seq=None):
        item_seq, item_seq_len = self._prep(item_seq, item_seq_len)
        if torch.is_grad_enabled() and self.training:
            return _ConvFormerFn.apply(self.dense_flat, self, item_seq, item_seq_len)
        cfg = self._cfg(item_seq.shape[0])
        return ops.convformer_fwd(cfg, self.item_embedding.weight.data, self.dense_flat.data, item_seq, item_seq_len, self._workspace(cfg))
