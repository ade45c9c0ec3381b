"""``torch.ops.unirec_amd.*`` -- the hot-path entry points of the C ABI registered with ``torch.library``.

BASELINE.json's north_star asks for the HIP kernels "exposed as torch ops".  The C ABI (include/unirec_amd.h) stays the drop-in
boundary; this module puts a dispatcher-visible schema on top of it: every op below is a ``torch.library.custom_op`` whose
implementation is the ctypes call in ``unirec_amd.ops`` (no arithmetic here) and whose ``register_fake`` gives output shapes and
dtypes, so ``torch.compile`` / FakeTensor tracing / ``torch.export`` see through the calls and the ops appear as
``torch.ops.unirec_amd.<name>``.  Structs of the C ABI (UrSasrecCfg, UrLossCfg, UrAdamCfg) become flat scalar arguments.

``HEADER_TO_OP`` maps each compute symbol of the header to its op; ``NOT_OPS`` lists the symbols that are deliberately not
dispatcher ops with the reason (queries, host-side objects, process-wide switches, raw test hooks, model families outside the
north_star's named path).  tests/test_abi.py checks that the two tables partition the header's symbol list exactly and that every
op exists with a fake implementation.

Import this module to register the ops (``import unirec_amd.torch_ops``); the model classes keep calling ``unirec_amd.ops``
directly -- same kernels, no dispatcher round trip on the hot path.
"""
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import ops

NS = "unirec_amd"


def _plan(uniq_idx, seg_start, sorted_pos, n_uniq, n, n_a):
    pl = ops.RowsPlan()
    pl.n, pl.n_a, pl.uniq_idx, pl.seg_start, pl.sorted_pos, pl.n_uniq = n, n_a, uniq_idx, seg_start, sorted_pos, n_uniq
    return pl


# ------------------------------------------------------------------------------------------------ embedding lookup
@custom_op(f"{NS}::embedding_gather", mutates_args=())
def embedding_gather(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return ops.embedding_gather(table, idx.contiguous())


@embedding_gather.register_fake
def _(table, idx):
    return table.new_empty(tuple(idx.shape) + (table.shape[1],))


# ------------------------------------------------------------------------------------------------ SASRec encoder
def _sasrec_cfg(item_seq, d, n_heads, inner, n_layers, act, use_pos, eps, last_only, skip_padding, p_hidden, p_attn, drop_seed, drop_step):
    return ops.sasrec_cfg(item_seq.shape[0], item_seq.shape[1], d, n_heads, inner, n_layers, act, use_pos, eps, last_only, skip_padding,
                          p_hidden, p_attn, drop_seed, drop_step)


@custom_op(f"{NS}::sasrec_workspace", mutates_args=())
def sasrec_workspace(item_seq: torch.Tensor, d: int, n_heads: int, inner: int, n_layers: int, p_hidden: float) -> torch.Tensor:
    return ops.sasrec_workspace(_sasrec_cfg(item_seq, d, n_heads, inner, n_layers, "gelu", True, 1e-12, 1, 1, p_hidden, 0.0, 0, 0), item_seq.device)


@sasrec_workspace.register_fake
def _(item_seq, d, n_heads, inner, n_layers, p_hidden):
    return item_seq.new_empty((torch.library.get_ctx().new_dynamic_size(),), dtype=torch.uint8)


@custom_op(f"{NS}::sasrec_fwd", mutates_args=("ws",))
def sasrec_fwd(item_table: torch.Tensor, dense: torch.Tensor, item_seq: torch.Tensor, ws: torch.Tensor, n_heads: int, inner: int,
               n_layers: int, act: str, use_pos: bool, eps: float, last_only: int = 1, skip_padding: int = 1, p_hidden: float = 0.0,
               p_attn: float = 0.0, drop_seed: int = 0, drop_step: int = 0) -> torch.Tensor:
    cfg = _sasrec_cfg(item_seq, item_table.shape[1], n_heads, inner, n_layers, act, use_pos, eps, last_only, skip_padding, p_hidden,
                      p_attn, drop_seed, drop_step)
    return ops.sasrec_fwd(cfg, item_table, dense, item_seq, ws)


@sasrec_fwd.register_fake
def _(item_table, dense, item_seq, ws, n_heads, inner, n_layers, act, use_pos, eps, last_only=1, skip_padding=1, p_hidden=0.0, p_attn=0.0,
      drop_seed=0, drop_step=0):
    return item_table.new_empty((item_seq.shape[0], item_table.shape[1]))


@custom_op(f"{NS}::sasrec_bwd", mutates_args=("ws",))
def sasrec_bwd(item_table: torch.Tensor, dense: torch.Tensor, item_seq: torch.Tensor, d_user_emb: torch.Tensor, ws: torch.Tensor,
               n_heads: int, inner: int, n_layers: int, act: str, use_pos: bool, eps: float, last_only: int = 1, skip_padding: int = 1,
               p_hidden: float = 0.0, p_attn: float = 0.0, drop_seed: int = 0, drop_step: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    cfg = _sasrec_cfg(item_seq, item_table.shape[1], n_heads, inner, n_layers, act, use_pos, eps, last_only, skip_padding, p_hidden,
                      p_attn, drop_seed, drop_step)
    return ops.sasrec_bwd(cfg, item_table, dense, item_seq, d_user_emb.contiguous(), ws)


@sasrec_bwd.register_fake
def _(item_table, dense, item_seq, d_user_emb, ws, n_heads, inner, n_layers, act, use_pos, eps, last_only=1, skip_padding=1, p_hidden=0.0,
      p_attn=0.0, drop_seed=0, drop_step=0):
    return torch.empty_like(dense), item_table.new_empty((item_seq.shape[0] * item_seq.shape[1], item_table.shape[1]))


# ------------------------------------------------------------------------------------------------ GRU encoder
@custom_op(f"{NS}::gru_fwd", mutates_args=("ws",))
def gru_fwd(item_table: torch.Tensor, dense: torch.Tensor, item_seq: torch.Tensor, ws: torch.Tensor, hidden: int, p_drop: float = 0.0,
            drop_seed: int = 0, drop_step: int = 0) -> torch.Tensor:
    cfg = ops.gru_cfg(item_seq.shape[0], item_seq.shape[1], item_table.shape[1], hidden, p_drop, drop_seed, drop_step)
    return ops.gru_fwd(cfg, item_table, dense, item_seq, ws)


@gru_fwd.register_fake
def _(item_table, dense, item_seq, ws, hidden, p_drop=0.0, drop_seed=0, drop_step=0):
    return item_table.new_empty((item_seq.shape[0], item_table.shape[1]))


@custom_op(f"{NS}::gru_bwd", mutates_args=("ws",))
def gru_bwd(item_table: torch.Tensor, dense: torch.Tensor, item_seq: torch.Tensor, d_user_emb: torch.Tensor, ws: torch.Tensor, hidden: int,
            p_drop: float = 0.0, drop_seed: int = 0, drop_step: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    cfg = ops.gru_cfg(item_seq.shape[0], item_seq.shape[1], item_table.shape[1], hidden, p_drop, drop_seed, drop_step)
    return ops.gru_bwd(cfg, item_table, dense, item_seq, d_user_emb.contiguous(), ws)


@gru_bwd.register_fake
def _(item_table, dense, item_seq, d_user_emb, ws, hidden, p_drop=0.0, drop_seed=0, drop_step=0):
    return torch.empty_like(dense), item_table.new_empty((item_seq.shape[0] * item_seq.shape[1], item_table.shape[1]))


# ------------------------------------------------------------------------------------------------ fused scorer + loss
@custom_op(f"{NS}::gather_dot_loss_fwd", mutates_args=())
def gather_dot_loss_fwd(user_emb: torch.Tensor, item_table: torch.Tensor, item_id: torch.Tensor, label: Optional[torch.Tensor],
                        user_bias: Optional[torch.Tensor], item_bias: Optional[torch.Tensor], user_id: Optional[torch.Tensor],
                        loss_type: str, tau: float = 1.0, score_clip: float = -1.0, ccl_w: float = 0.0,
                        ccl_m: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    B, G = item_id.shape
    cfg = ops.loss_cfg(B, G, user_emb.shape[1], "bpr" if loss_type == "none" else loss_type, tau, score_clip, ccl_w, ccl_m)
    if loss_type == "none":
        cfg.loss_type = -1
    return ops.gather_dot_loss_fwd(cfg, user_emb, item_table, item_id, label, user_bias, item_bias, user_id)


@gather_dot_loss_fwd.register_fake
def _(user_emb, item_table, item_id, label, user_bias, item_bias, user_id, loss_type, tau=1.0, score_clip=-1.0, ccl_w=0.0, ccl_m=0.0):
    B, G = item_id.shape
    return user_emb.new_empty((B, G)), user_emb.new_empty((2 * B,)), user_emb.new_empty((4,))


@custom_op(f"{NS}::gather_dot_loss_bwd", mutates_args=())
def gather_dot_loss_bwd(user_emb: torch.Tensor, item_table: torch.Tensor, item_id: torch.Tensor, label: Optional[torch.Tensor],
                        scores: torch.Tensor, loss_out: torch.Tensor, d_loss: Optional[torch.Tensor], loss_type: str, tau: float = 1.0,
                        score_clip: float = -1.0, ccl_w: float = 0.0, ccl_m: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    B, G = item_id.shape
    cfg = ops.loss_cfg(B, G, user_emb.shape[1], loss_type, tau, score_clip, ccl_w, ccl_m)
    coef, d_user, d_ub = ops.gather_dot_loss_bwd(cfg, user_emb, item_table, item_id, label, scores, loss_out, d_loss, want_user_bias=True)
    return coef, d_user, d_ub


@gather_dot_loss_bwd.register_fake
def _(user_emb, item_table, item_id, label, scores, loss_out, d_loss, loss_type, tau=1.0, score_clip=-1.0, ccl_w=0.0, ccl_m=0.0):
    B, G = item_id.shape
    return user_emb.new_empty((B, G)), torch.empty_like(user_emb), user_emb.new_empty((B,))


# ------------------------------------------------------------------------------------------------ negative sampler (device, Philox)
@custom_op(f"{NS}::sample_negatives", mutates_args=())
def sample_negatives(user_id: torch.Tensor, pos_item: torch.Tensor, n_neg: int, n_items: int, hist_ptr: Optional[torch.Tensor],
                     hist_sorted: Optional[torch.Tensor], seed: int, step: int) -> Tuple[torch.Tensor, torch.Tensor]:
    import ctypes as C
    from ._lib import check, lib
    B, dev = user_id.numel(), user_id.device
    item_id = torch.empty(B, n_neg + 1, dtype=torch.int64, device=dev)
    label = torch.empty(B, n_neg + 1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
    n_users = hist_ptr.numel() - 1 if hist_ptr is not None else 0
    check(lib.ur_sample_negatives(p(user_id), p(pos_item.contiguous()), B, n_neg, n_items, n_users, p(hist_ptr), p(hist_sorted),
                                  int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFF, p(item_id), p(label),
                                  ops._stream()), "ur_sample_negatives")
    return item_id, label


@sample_negatives.register_fake
def _(user_id, pos_item, n_neg, n_items, hist_ptr, hist_sorted, seed, step):
    B = user_id.numel()
    return user_id.new_empty((B, n_neg + 1), dtype=torch.int64), user_id.new_empty((B, n_neg + 1), dtype=torch.int32)


# ------------------------------------------------------------------------------------------------ row-sparse gradient + optimizer
@custom_op(f"{NS}::rows_plan", mutates_args=())
def rows_plan(ids_a: Optional[torch.Tensor], ids_b: Optional[torch.Tensor], n_rows: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    pl = ops.rows_plan(ids_a, ids_b, n_rows)
    return pl.uniq_idx, pl.seg_start, pl.sorted_pos, pl.n_uniq


@rows_plan.register_fake
def _(ids_a, ids_b, n_rows):
    ref = ids_a if ids_a is not None else ids_b
    n = (ids_a.numel() if ids_a is not None else 0) + (ids_b.numel() if ids_b is not None else 0)
    i32 = dict(dtype=torch.int32)
    return ref.new_empty((n,), **i32), ref.new_empty((n + 1,), **i32), ref.new_empty((n,), **i32), ref.new_empty((1,), **i32)


# ------------------------------------------------------------------------------------------------ row exchange (row-sharded tables)
# a2a_embedding_exchange.  The three legs of a step's exchange over the library's RCCL communicator (ops.comm_init); the
# fixed capacity `cap` per (source, owner) pair makes every shape static.
@custom_op(f"{NS}::a2a_embedding_ids", mutates_args=())
def a2a_embedding_ids(uniq_key: torch.Tensor, n_uniq: torch.Tensor, counts: torch.Tensor, n_local: int, world: int,
                      cap: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (recv_ids [world*cap] on the owners, slot_of_uniq, u_of_slot, flags)"""
    dev, i32 = uniq_key.device, torch.int32
    pl = _plan(uniq_key, None, None, n_uniq, uniq_key.numel(), 0)
    send = torch.empty(world * cap, dtype=i32, device=dev)
    recv = torch.empty(world * cap, dtype=i32, device=dev)
    slot = torch.zeros(uniq_key.numel(), dtype=i32, device=dev)
    uos = torch.empty(world * cap, dtype=i32, device=dev)
    flags = torch.zeros(4, dtype=i32, device=dev)
    ops.shard_exchange_ids(pl, counts, n_local, world, cap, send, slot, uos, flags, recv_ids=recv, transport=True)
    return recv, slot, uos, flags


@a2a_embedding_ids.register_fake
def _(uniq_key, n_uniq, counts, n_local, world, cap):
    i32 = dict(dtype=torch.int32)
    return (uniq_key.new_empty((world * cap,), **i32), uniq_key.new_empty((uniq_key.numel(),), **i32),
            uniq_key.new_empty((world * cap,), **i32), uniq_key.new_empty((4,), **i32))


@custom_op(f"{NS}::a2a_embedding_exchange", mutates_args=())
def a2a_embedding_exchange(table_shard: torch.Tensor, req_ids: torch.Tensor, world: int, cap: int) -> torch.Tensor:
    """owner side gathers the requested rows of its shard, all-to-all -> the requesters' compact [world*cap, d] tables"""
    d = table_shard.shape[1]
    ws = torch.empty(world * cap, d, dtype=torch.float32, device=table_shard.device)
    out = torch.empty(world * cap, d, dtype=torch.float32, device=table_shard.device)
    ops.shard_exchange_rows(table_shard, req_ids, world, cap, ws, compact=out, transport=True)
    return out


@a2a_embedding_exchange.register_fake
def _(table_shard, req_ids, world, cap):
    return table_shard.new_empty((world * cap, table_shard.shape[1]))


@custom_op(f"{NS}::a2a_embedding_grads", mutates_args=())
def a2a_embedding_grads(uniq_grad: torch.Tensor, u_of_slot: torch.Tensor, world: int, cap: int) -> torch.Tensor:
    """row gradients in unique order -> slot layout, all-to-all -> [world*cap, d] on the owners (block s = from rank s)"""
    d = uniq_grad.shape[1]
    ws = torch.empty(world * cap, d, dtype=torch.float32, device=uniq_grad.device)
    out = torch.empty(world * cap, d, dtype=torch.float32, device=uniq_grad.device)
    ops.shard_exchange_grads(uniq_grad, u_of_slot, world, cap, ws, grads_in=out, transport=True)   # (no flags: slot 0 rows say "no NaN, no overflow")
    return out


@a2a_embedding_grads.register_fake
def _(uniq_grad, u_of_slot, world, cap):
    return uniq_grad.new_empty((world * cap, uniq_grad.shape[1]))


@custom_op(f"{NS}::rows_reduce", mutates_args=())
def rows_reduce(uniq_idx: torch.Tensor, seg_start: torch.Tensor, sorted_pos: torch.Tensor, n_uniq: torch.Tensor, rows_a: Optional[torch.Tensor],
                coef_b: Optional[torch.Tensor], vec_b: Optional[torch.Tensor], n_a: int, G: int, d: int) -> torch.Tensor:
    return ops.rows_reduce(_plan(uniq_idx, seg_start, sorted_pos, n_uniq, uniq_idx.numel(), n_a), rows_a, coef_b, vec_b, G, d)


@rows_reduce.register_fake
def _(uniq_idx, seg_start, sorted_pos, n_uniq, rows_a, coef_b, vec_b, n_a, G, d):
    return uniq_idx.new_empty((uniq_idx.numel(), d), dtype=torch.float32)


@custom_op(f"{NS}::sparse_adam_rows", mutates_args=("table", "m", "v", "last_step"))
def sparse_adam_rows(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, last_step: Optional[torch.Tensor], uniq_idx: torch.Tensor,
                     n_uniq: torch.Tensor, uniq_grad: torch.Tensor, grad_scale: Optional[torch.Tensor], lr: float, step: int,
                     weight_decay: float = 0.0, algo: str = "adam") -> None:
    pl = _plan(uniq_idx, None, None, n_uniq, uniq_idx.numel(), 0)
    ops.sparse_adam_rows(ops.adam_cfg(lr, step, weight_decay, algo=algo), table, m, v, pl, uniq_grad, last_step, grad_scale)


@custom_op(f"{NS}::rows_reduce_update", mutates_args=("table", "m", "v", "last_step"))
def rows_reduce_update(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, last_step: Optional[torch.Tensor], uniq_idx: torch.Tensor,
                       seg_start: torch.Tensor, sorted_pos: torch.Tensor, n_uniq: torch.Tensor, rows_a: Optional[torch.Tensor],
                       coef_b: Optional[torch.Tensor], vec_b: Optional[torch.Tensor], n_a: int, G: int, grad_scale: Optional[torch.Tensor],
                       lr: float, step: int, weight_decay: float = 0.0, algo: str = "adam") -> None:
    """rows_reduce + sparse_adam_rows in one launch: the row-gradient sums never reach HBM"""
    pl = _plan(uniq_idx, seg_start, sorted_pos, n_uniq, uniq_idx.numel(), n_a)
    ops.rows_reduce_update(ops.adam_cfg(lr, step, weight_decay, algo=algo), table, m, v, pl, rows_a, coef_b, vec_b, G, last_step, grad_scale)


@custom_op(f"{NS}::lazy_adam_catchup", mutates_args=("table", "m", "v", "last_step"))
def lazy_adam_catchup(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, last_step: torch.Tensor, uniq_idx: torch.Tensor,
                      n_uniq: torch.Tensor, lr: float, step: int, weight_decay: float = 0.0, algo: str = "adam") -> None:
    pl = _plan(uniq_idx, None, None, n_uniq, uniq_idx.numel(), 0)
    ops.lazy_adam_catchup(ops.adam_cfg(lr, step, weight_decay, algo=algo), table, m, v, last_step, pl)


@custom_op(f"{NS}::lazy_adam_flush", mutates_args=("table", "m", "v", "last_step"))
def lazy_adam_flush(table: torch.Tensor, m: torch.Tensor, v: torch.Tensor, last_step: torch.Tensor, lr: float, step: int,
                    weight_decay: float = 0.0, algo: str = "adam") -> None:
    ops.lazy_adam_flush(ops.adam_cfg(lr, step, weight_decay, algo=algo), table, m, v, last_step)


@custom_op(f"{NS}::dense_adam", mutates_args=("param", "m", "v"))
def dense_adam(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad_scale: Optional[torch.Tensor], lr: float,
               step: int, weight_decay: float = 0.0, algo: str = "adam") -> None:
    ops.dense_adam(ops.adam_cfg(lr, step, weight_decay, algo=algo), param, grad, m, v, grad_scale)


for _op in (sparse_adam_rows, rows_reduce_update, lazy_adam_catchup, lazy_adam_flush, dense_adam):
    _op.register_fake(lambda *a, **k: None)


# ------------------------------------------------------------------------------------------------ full-item ranking / top-k
@custom_op(f"{NS}::full_rank", mutates_args=())
def full_rank(user_emb: torch.Tensor, item_table: torch.Tensor, target: torch.Tensor, user_id: Optional[torch.Tensor],
              hist_ptr: Optional[torch.Tensor], hist_sorted: Optional[torch.Tensor], user_bias: Optional[torch.Tensor],
              item_bias: Optional[torch.Tensor], tau: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    return ops.full_rank(user_emb, item_table, target, user_id, hist_ptr, hist_sorted, user_bias, item_bias, tau)


@full_rank.register_fake
def _(user_emb, item_table, target, user_id, hist_ptr, hist_sorted, user_bias, item_bias, tau=1.0):
    B = user_emb.shape[0]
    return user_emb.new_empty((B,), dtype=torch.int32), user_emb.new_empty((B,))


@custom_op(f"{NS}::full_topk", mutates_args=())
def full_topk(user_emb: torch.Tensor, item_table: torch.Tensor, k: int, user_id: Optional[torch.Tensor], hist_ptr: Optional[torch.Tensor],
              hist_sorted: Optional[torch.Tensor], user_bias: Optional[torch.Tensor], item_bias: Optional[torch.Tensor],
              tau: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    return ops.full_topk(user_emb, item_table, k, user_id, hist_ptr, hist_sorted, user_bias, item_bias, tau)


@full_topk.register_fake
def _(user_emb, item_table, k, user_id, hist_ptr, hist_sorted, user_bias, item_bias, tau=1.0):
    B = user_emb.shape[0]
    return user_emb.new_empty((B, k)), user_emb.new_empty((B, k), dtype=torch.int64)


# ------------------------------------------------------------------------------------------------ header symbol <-> op
HEADER_TO_OP = {
    "ur_embedding_gather_f32": "embedding_gather",
    "ur_sasrec_workspace_bytes": "sasrec_workspace",
    "ur_sasrec_fwd": "sasrec_fwd",
    "ur_sasrec_bwd": "sasrec_bwd",
    "ur_gru_fwd": "gru_fwd",
    "ur_gru_bwd": "gru_bwd",
    "ur_gather_dot_loss_fwd": "gather_dot_loss_fwd",
    "ur_gather_dot_loss_bwd": "gather_dot_loss_bwd",
    "ur_sample_negatives": "sample_negatives",
    "ur_rows_plan": "rows_plan",
    "ur_shard_exchange_ids": "a2a_embedding_ids",
    "ur_shard_exchange_rows": "a2a_embedding_exchange",
    "ur_shard_exchange_grads": "a2a_embedding_grads",
    "ur_rows_reduce": "rows_reduce",
    "ur_sparse_adam_rows": "sparse_adam_rows",
    "ur_rows_reduce_update": "rows_reduce_update",
    "ur_lazy_adam_catchup": "lazy_adam_catchup",
    "ur_lazy_adam_catchup_background": "lazy_adam_catchup",
    "ur_lazy_adam_flush": "lazy_adam_flush",
    "ur_dense_adam": "dense_adam",
    "ur_full_rank": "full_rank",
    "ur_full_topk": "full_topk",
}

_QUERY = "layout / workspace-size query or error plumbing: plain host function, nothing to dispatch"
_HOST = "host-side object (CPython-compatible MT19937 sampler, row builder, alias table): CPU memory, no tensor dispatch"
_SWITCH = "process-wide switch / profiler control"
_HOOK = "raw kernel hook for unit tests and micro-benchmarks (the encoder ops call these kernels internally)"
_PLUMB = "stream plumbing of the deferred-join backward: ordering between HIP streams, not a tensor computation"
_SHARD = "row-sharded (multi-GPU) variant driven by facility/distributed.py around torch.distributed collectives"
_COMM = "the library's RCCL communicator (process-wide state behind the a2a_embedding_exchange ops): set-up / tear-down / the flat all-reduce, no tensor dispatch"
_FAMILY = "sibling model family / loss outside the north_star's named ops (SURVEY.md 8 f4): reached through unirec_amd.ops"
NOT_OPS = {
    "ur_rows_reduce_update_owner": _SHARD,
    "ur_last_error": _QUERY, "ur_version": _QUERY, "ur_id_guard_state": _QUERY, "ur_id_guard_reset": _SWITCH, "ur_trace_ranges_pushed": _QUERY, "ur_sasrec_param_layout": _QUERY, "ur_gru_param_layout": _QUERY,
    "ur_gru_workspace_bytes": _QUERY, "ur_rows_plan_workspace_bytes": _QUERY, "ur_gemm_tn_workspace_floats": _QUERY,
    "ur_full_topk_workspace_bytes": _QUERY, "ur_full_softmax_workspace_bytes": _QUERY, "ur_convformer_param_layout": _QUERY,
    "ur_convformer_workspace_bytes": _QUERY, "ur_atthist_param_layout": _QUERY, "ur_atthist_workspace_bytes": _QUERY,
    "ur_host_sampler_create": _HOST, "ur_host_sampler_destroy": _HOST, "ur_host_sampler_getrandbits": _HOST, "ur_host_sampler_random": _HOST,
    "ur_host_sampler_randint": _HOST, "ur_host_sampler_set_alias": _HOST, "ur_host_build_rows": _HOST, "ur_alias_table_build": _HOST,
    "ur_sasrec_set_side_stream": _SWITCH, "ur_sasrec_set_chain": _SWITCH, "ur_prof_enable": _SWITCH, "ur_prof_set_mask": _SWITCH,
    "ur_set_mfma_arith": _SWITCH, "ur_get_mfma_arith": _SWITCH, "ur_prof_reset": _SWITCH, "ur_prof_num_classes": _SWITCH, "ur_prof_class_name": _SWITCH, "ur_prof_read": _SWITCH,
    "ur_gemm_nt": _HOOK, "ur_gemm_tn": _HOOK, "ur_gemm_tn_group": _HOOK, "ur_debug_delay": _HOOK,
    "ur_sasrec_bwd_deferred": _PLUMB, "ur_sasrec_bwd_join": _PLUMB, "ur_stream_wait_stream": _PLUMB, "ur_sasrec_side_stream": _PLUMB, "ur_sasrec_side_publish": _PLUMB,
    "ur_rows_plan_merge": _SHARD, "ur_rows_plan_sharded": _SHARD, "ur_compact_index": _SHARD, "ur_full_rank_shard": _SHARD,
    "ur_shard_step_flags": _SHARD, "ur_comm_world": _COMM, "ur_comm_count": _COMM, "ur_loop_create": _COMM, "ur_loop_destroy": _COMM, "ur_loop_attach": _COMM, "ur_loop_detach": _COMM,
    "ur_loop_world": _COMM, "ur_loop_post": _COMM, "ur_loop_all_to_all_pull": _COMM, "ur_loop_all_reduce_pull": _COMM, "ur_loop_finish": _COMM, "ur_comm_unique_id": _COMM, "ur_comm_init": _COMM, "ur_comm_destroy": _COMM, "ur_comm_all_reduce_sum": _COMM,
    "ur_comm_all_to_all": _COMM, "ur_shard_fixup_plan": _SHARD, "ur_shard_fixup_apply": _SHARD, "ur_rows_split_hot": _SHARD,
    "ur_rows_reduce_riders": "ur_rows_reduce + the flag rows / step flags of the row-sharded step riding in the same launch: a scheduling variant of the registered ops rows_reduce and a2a_embedding_grads",
    "ur_sumsq": _FAMILY + " (gradient-clipping helpers of the optimizer)", "ur_clip_coef": _FAMILY + " (gradient-clipping helpers)",
    "ur_clip_coef_guarded": _FAMILY + " (gradient-clipping helpers)",
    "ur_sample_negatives_pop": _FAMILY + " (popularity-biased sampler)", "ur_host_sampler_state": _HOST, "ur_mt_workspace_bytes": _QUERY,
    "ur_mt_build_rows": _FAMILY + " (device row builder on the reference's MT19937 stream)", "ur_device_build_seq_choice": _FAMILY + " (device row builder)", "ur_device_build_seq": _FAMILY + " (device row builder)",
    "ur_convformer_fwd": _FAMILY, "ur_convformer_bwd": _FAMILY, "ur_atthist_fwd": _FAMILY, "ur_atthist_bwd": _FAMILY,
    "ur_pool_rows_fwd": _FAMILY, "ur_pool_rows_bwd": _FAMILY, "ur_full_softmax_fwd": _FAMILY, "ur_full_softmax_bwd": _FAMILY,
    "ur_full_softmax_fwd_shard": _SHARD, "ur_full_softmax_combine_shards": _SHARD, "ur_full_softmax_bwd_shard": _SHARD,
    "ur_rows_scatter_add": _FAMILY,
    "ur_gather_dot_loss_fused_supported": _QUERY,
    "ur_gather_dot_loss_fwd_bwd": "fusion of the two ops gather_dot_loss_fwd + gather_dot_loss_bwd (both registered) for the graph-free training step: a scheduling choice, not a new op",
    "ur_rows_filter_touched": "index bookkeeping of the lazy optimizer schedule (which rows of the next plan have any history): no arithmetic, a scheduling aid",
}
