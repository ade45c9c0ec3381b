"""Torch-tensor front end of the C ABI: validates tensors, passes raw device pointers and the current
HIP stream.  PyTorch is only the allocator / stream provider here; all arithmetic is in the HIP library.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import ACT_IDS, LOSS_IDS, UrAdamCfg, UrLossCfg, UrSasrecCfg, check, lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream as a raw pointer.  torch.cuda.current_stream() builds a Stream object through three layers of Python
    (~8 us); a step makes 20-40 ctypes calls, which made the multi-GPU step host-bound (0.65 ms of enqueue per 0.60 ms of device
    work).  The raw accessor is one C call."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, dtype, name, allow_none=False):
    if t is None:
        if allow_none:
            return
        raise _lib.UnirecAmdError(f"{name}: tensor required")
    if not t.is_cuda:
        raise _lib.UnirecAmdError(f"{name}: expected a GPU tensor (unirec_amd has no CPU path), got {t.device}")
    if t.dtype != dtype:
        raise _lib.UnirecAmdError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.UnirecAmdError(f"{name}: tensor must be contiguous")


# --------------------------------------------------------------------------------------------- gather
def embedding_gather(table: torch.Tensor, idx: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """out[..., :] = table[idx[...], :]  (bit-exact).  idx: int32 or int64, any shape.  out: optional preallocated result."""
    _chk(table, torch.float32, "table")
    if idx.dtype not in (torch.int32, torch.int64):
        raise _lib.UnirecAmdError(f"idx: expected int32/int64, got {idx.dtype}")
    _chk(idx, idx.dtype, "idx")
    n_rows, d = table.shape
    if out is None:
        out = torch.empty(*idx.shape, d, dtype=torch.float32, device=table.device)
    else:
        _chk(out, torch.float32, "out")
        if out.numel() != idx.numel() * d:
            raise _lib.UnirecAmdError("embedding_gather: out has the wrong size")
    check(lib.ur_embedding_gather_f32(_p(table), n_rows, d, _p(idx), idx.element_size(), idx.numel(), _p(out), _stream()),
          "ur_embedding_gather_f32")
    return out


# --------------------------------------------------------------------------------------------- SASRec
def default_mfma_arith() -> int:
    """Arithmetic of the encoder's weight-gradient products and row-chain kernels when a model's config does not say (``mfma_arith``):
    6 = the fp32 operands split exactly into three bf16 pieces, six piece products accumulated in fp32 on the bf16 matrix pipes
    (fp32-equivalent; include/unirec_amd.h: UrSasrecCfg.mfma_arith, ur_set_mfma_arith); ``UR_MFMA_ARITH=0`` (or 9) in the environment
    selects the exact fp32-input MFMA (all nine terms in the weight-gradient products)."""
    v = os.environ.get("UR_MFMA_ARITH", "")
    return int(v) if v in ("0", "6", "9") else 6


def _arith(v) -> int:
    v = default_mfma_arith() if v is None else int(v)
    if (v & 0xFF) not in (0, 3, 6, 9) or v & ~0x1FF:
        raise ValueError(f"mfma_arith={v}: 0 (exact fp32 MFMA), 6 or 9 (split-bf16 terms); + 0x100 = also for products narrower than the "
                         "split kernel's 128 x 128 tile (unit tests)")
    return v


def sasrec_cfg(B, L, d, n_heads, inner, n_layers, act, use_pos, eps, last_only=1, skip_padding=1, p_hidden=0.0, p_attn=0.0,
               drop_seed=0, drop_step=0, mfma_arith=None) -> UrSasrecCfg:
    """last_only=1: exact last-row specialisation of the final layer (only position L-1 reaches the loss).
    skip_padding=1: padded prefixes get no token rows (exact; applies when L <= 64 and head dim is 4/8/16).
    p_hidden / p_attn: training-time dropout; the mask is a function of (drop_seed, drop_step), the backward must get the
    same cfg as the forward."""
    return UrSasrecCfg(B, L, d, n_heads, inner, n_layers, ACT_IDS[act], int(bool(use_pos)), float(eps), int(last_only), int(skip_padding),
                       float(p_hidden), float(p_attn), int(drop_seed), int(drop_step), _arith(mfma_arith), 0)


def sasrec_param_layout(cfg: UrSasrecCfg):
    n = _lib.UR_SASREC_N_GLOBAL + cfg.n_layers * _lib.UR_SASREC_N_PER_LAYER
    offs = (C.c_int64 * n)()
    total = check(lib.ur_sasrec_param_layout(C.byref(cfg), offs), "ur_sasrec_param_layout")
    return list(offs), int(total)


def sasrec_workspace(cfg: UrSasrecCfg, device) -> torch.Tensor:
    nbytes = check(lib.ur_sasrec_workspace_bytes(C.byref(cfg)), "ur_sasrec_workspace_bytes")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def sasrec_fwd(cfg, item_table, dense, item_seq, ws):
    _chk(item_table, torch.float32, "item_table")
    _chk(dense, torch.float32, "dense")
    _chk(item_seq, torch.int32, "item_seq")
    if tuple(item_seq.shape) != (cfg.B, cfg.L):
        raise _lib.UnirecAmdError(f"sasrec_fwd: item_seq has shape {tuple(item_seq.shape)}, the configuration says [{cfg.B}, {cfg.L}]")
    user_emb = torch.empty(cfg.B, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_sasrec_fwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(user_emb),
                            _p(ws), _stream()), "ur_sasrec_fwd")
    _side.hold.clear()   # (a late join of side-stream work, if one was pending, is now enqueued on this stream)
    _side.late = False
    return user_emb


def sasrec_bwd(cfg, item_table, dense, item_seq, d_user_emb, ws, defer_join=False):
    """defer_join=True: dense_grad is NOT complete in stream order on return -- call sasrec_bwd_join() before reading it
    (d_emb_rows is complete); see ur_sasrec_bwd_deferred."""
    _chk(d_user_emb, torch.float32, "d_user_emb")
    dense_grad = torch.empty_like(dense)
    d_emb_rows = torch.empty(cfg.B * cfg.L, cfg.d, dtype=torch.float32, device=dense.device)
    fn = lib.ur_sasrec_bwd_deferred if defer_join else lib.ur_sasrec_bwd
    check(fn(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(d_user_emb),
             _p(ws), _p(dense_grad), _p(d_emb_rows), _stream()), "ur_sasrec_bwd")
    return dense_grad, d_emb_rows


def sasrec_bwd_join():
    check(lib.ur_sasrec_bwd_join(_stream()), "ur_sasrec_bwd_join")
    _side.hold.clear()
    _side.late = False


def stream_wait_stream(waiter, waited):
    """torch's ``waiter.wait_stream(waited)`` through an event without the system-scope fence (ur_stream_wait_stream)."""
    check(lib.ur_stream_wait_stream(C.c_void_p(waiter.cuda_stream), C.c_void_p(waited.cuda_stream)), "ur_stream_wait_stream")


def sasrec_side_stream():
    """The encoder's side stream as a torch stream while a deferred backward is pending (else None): see ur_sasrec_side_stream."""
    p = lib.ur_sasrec_side_stream()
    return torch.cuda.ExternalStream(p) if p else None


class _SideState(threading.local):
    """per THREAD, as the library's side-stream context is (csrc/common.h: g_ctx_id): the rank threads of the in-process loopback
    transport each have their own encoder side stream"""

    def __init__(self):
        self.late = False    # a late join is armed (sasrec_side_publish(late=True)) and no forward pass / explicit join has taken it yet
        self.hold = []       # tensors the side-stream work reads (allocated under the main stream): kept alive until the main stream has joined


_side = _SideState()


def sasrec_side_publish(late=True, hold=()):
    """End of the work the caller put on the encoder's side stream.  late=True: the next sasrec_fwd joins it after its first launch;
    `hold`: tensors that work reads -- kept referenced until that join is enqueued (their memory belongs to the main stream's allocator)."""
    check(lib.ur_sasrec_side_publish(1 if late else 0), "ur_sasrec_side_publish")
    if late:
        _side.hold.extend(hold)
        _side.late = True
    else:
        sasrec_bwd_join()


def sasrec_side_join():
    """Joins a late side-stream update now, on the current stream (no-op when none is pending): for readers of the dense parameters
    other than the encoder's forward pass."""
    if _side.late:
        sasrec_bwd_join()


# --------------------------------------------------------------------------------------------- GRU
def gru_cfg(B, L, d, H, p_drop=0.0, drop_seed=0, drop_step=0, mfma_arith=None):
    return _lib.UrGruCfg(B, L, d, H, float(p_drop), int(drop_seed), int(drop_step), _arith(mfma_arith), 0)


def gru_param_layout(cfg):
    offs = (C.c_int64 * 6)()
    total = check(lib.ur_gru_param_layout(C.byref(cfg), offs), "ur_gru_param_layout")
    return list(offs), int(total)


def gru_workspace(cfg, device):
    return torch.empty(check(lib.ur_gru_workspace_bytes(C.byref(cfg)), "ur_gru_workspace_bytes"), dtype=torch.uint8, device=device)


def gru_fwd(cfg, item_table, dense, item_seq, ws):
    _chk(item_table, torch.float32, "item_table")
    _chk(dense, torch.float32, "dense")
    _chk(item_seq, torch.int32, "item_seq")
    if tuple(item_seq.shape) != (cfg.B, cfg.L):   # the kernels index [B, L] ids: a shorter tensor would be read past its end
        raise _lib.UnirecAmdError(f"gru_fwd: item_seq has shape {tuple(item_seq.shape)}, the configuration says [{cfg.B}, {cfg.L}]")
    user_emb = torch.empty(cfg.B, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_gru_fwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(user_emb), _p(ws), _stream()),
          "ur_gru_fwd")
    return user_emb


def gru_bwd(cfg, item_table, dense, item_seq, d_user_emb, ws):
    _chk(d_user_emb, torch.float32, "d_user_emb")
    _chk(item_seq, torch.int32, "item_seq")
    if tuple(item_seq.shape) != (cfg.B, cfg.L):
        raise _lib.UnirecAmdError(f"gru_bwd: item_seq has shape {tuple(item_seq.shape)}, the configuration says [{cfg.B}, {cfg.L}]")
    dense_grad = torch.empty_like(dense)
    d_emb_rows = torch.empty(cfg.B * cfg.L, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_gru_bwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(d_user_emb), _p(ws),
                         _p(dense_grad), _p(d_emb_rows), _stream()), "ur_gru_bwd")
    return dense_grad, d_emb_rows


# --------------------------------------------------------------------------------------------- scorer + loss
def loss_cfg(B, G, d, loss_type, tau=1.0, score_clip=-1.0, ccl_w=0.0, ccl_m=0.0, group_size=0) -> UrLossCfg:
    """group_size > 0: the user-item-label row format (G == 1; every group_size consecutive rows form one score row of the loss,
    unirec/model/base/reco_abc.py:233-236)."""
    return UrLossCfg(B, G, d, LOSS_IDS[loss_type], float(tau), float(score_clip if score_clip else -1.0), float(ccl_w), float(ccl_m),
                     int(group_size) if group_size and group_size > 0 else 0)


def gather_dot_loss_fwd(cfg, user_emb, item_table, item_id, label=None, user_bias=None, item_bias=None, user_id=None):
    _chk(user_emb, torch.float32, "user_emb")
    _chk(item_table, torch.float32, "item_table")
    _chk(item_id, torch.int64, "item_id")
    _chk(label, torch.int32, "label", allow_none=True)
    _chk(user_bias, torch.float32, "user_bias", allow_none=True)
    _chk(item_bias, torch.float32, "item_bias", allow_none=True)
    _chk(user_id, torch.int64, "user_id", allow_none=True)
    dev = user_emb.device
    scores = torch.empty(cfg.B, cfg.G, dtype=torch.float32, device=dev)
    loss_rows = torch.empty(2 * cfg.B, dtype=torch.float32, device=dev)
    loss_out = torch.empty(4, dtype=torch.float32, device=dev)   # [loss, count, update guard, -]
    check(lib.ur_gather_dot_loss_fwd(C.byref(cfg), _p(user_emb), _p(item_table), item_table.shape[0], _p(item_id), _p(label),
                                     _p(user_bias), _p(item_bias), _p(user_id), _p(scores), _p(loss_rows), _p(loss_out),
                                     _stream()), "ur_gather_dot_loss_fwd")
    return scores, loss_rows, loss_out


def gather_dot_loss_bwd(cfg, user_emb, item_table, item_id, label, scores, loss_out, d_loss=None, want_user_bias=False):
    dev = user_emb.device
    coef = torch.empty(cfg.B, cfg.G, dtype=torch.float32, device=dev)
    d_user = torch.empty(cfg.B, cfg.d, dtype=torch.float32, device=dev)
    d_ub = torch.empty(cfg.B, dtype=torch.float32, device=dev) if want_user_bias else None
    _chk(d_loss, torch.float32, "d_loss", allow_none=True)
    check(lib.ur_gather_dot_loss_bwd(C.byref(cfg), _p(user_emb), _p(item_table), item_table.shape[0], _p(item_id), _p(label),
                                     _p(scores), _p(loss_out), _p(d_loss), _p(coef), _p(d_user), _p(d_ub), _stream()),
          "ur_gather_dot_loss_bwd")
    return coef, d_user, d_ub


def gather_dot_loss_fwd_bwd(cfg, user_emb, item_table, item_id, label=None, user_bias=None, item_bias=None, user_id=None,
                            want_user_bias=False):
    """scorer + loss + its backward for one training step -> (scores, loss_out, coef, d_user, d_user_bias_rows).  ONE launch
    (ur_gather_dot_loss_fwd_bwd) where the loss allows it (bpr / bce / ccl, G * d * 4 <= 32 KB), else the two entry points."""
    if not lib.ur_gather_dot_loss_fused_supported(C.byref(cfg)):
        scores, _, loss_out = gather_dot_loss_fwd(cfg, user_emb, item_table, item_id, label, user_bias, item_bias, user_id)
        coef, d_user, d_ub = gather_dot_loss_bwd(cfg, user_emb, item_table, item_id, label, scores, loss_out, None, want_user_bias)
        return scores, loss_out, coef, d_user, d_ub
    _chk(user_emb, torch.float32, "user_emb")
    _chk(item_table, torch.float32, "item_table")
    _chk(item_id, torch.int64, "item_id")
    _chk(label, torch.int32, "label", allow_none=True)
    _chk(user_bias, torch.float32, "user_bias", allow_none=True)
    _chk(item_bias, torch.float32, "item_bias", allow_none=True)
    _chk(user_id, torch.int64, "user_id", allow_none=True)
    dev = user_emb.device
    scores = torch.empty(cfg.B, cfg.G, dtype=torch.float32, device=dev)
    loss_rows = torch.empty(2 * cfg.B, dtype=torch.float32, device=dev)
    loss_out = torch.empty(4, dtype=torch.float32, device=dev)   # [loss, count, update guard, -]
    coef = torch.empty(cfg.B, cfg.G, dtype=torch.float32, device=dev)
    d_user = torch.empty(cfg.B, cfg.d, dtype=torch.float32, device=dev)
    d_ub = torch.empty(cfg.B, dtype=torch.float32, device=dev) if want_user_bias else None
    check(lib.ur_gather_dot_loss_fwd_bwd(C.byref(cfg), _p(user_emb), _p(item_table), item_table.shape[0], _p(item_id), _p(label),
                                         _p(user_bias), _p(item_bias), _p(user_id), _p(scores), _p(loss_rows), _p(loss_out), _p(coef),
                                         _p(d_user), _p(d_ub), _stream()), "ur_gather_dot_loss_fwd_bwd")
    return scores, loss_out, coef, d_user, d_ub


# --------------------------------------------------------------------------------------------- sparse rows
_GUARD_OUT = (C.c_int64 * 3)()


def id_guard_check():
    """The host half of the id guard (include/unirec_amd.h: ur_id_guard_state; the reference's nn.Embedding raises IndexError for an id
    outside its table, reco_abc.py:168-170).  A plain load of the host-mapped mirror the plan kernels write -- no synchronisation; called
    at the head of every plan, so a bad id surfaces one or two steps after its batch.  Every step from that batch on has been skipped on
    the device (the guard is sticky).  The batch is planned a step AHEAD, on the plan stream, so the guard can go up while the step in
    front of it is still being applied: that one step may be applied in part (its row update done, its dense update skipped, or -- on
    several ranks -- applied on some ranks only).  The error is fatal for the run: restart from the last checkpoint rather
    than catching it and training on or saving the tables.  ``id_guard_reset()`` clears the guard."""
    if lib.ur_id_guard_state(_GUARD_OUT):
        bad, n_rows = int(_GUARD_OUT[0]), int(_GUARD_OUT[1])
        raise IndexError(f"index {bad} is out of range for an embedding table of {n_rows} rows (unirec_amd id guard: the step that "
                         f"looked it up and every step since were skipped on the device; unirec_amd.ops.id_guard_reset() clears the guard)")


def id_guard_reset():
    """clear this device's id guard (after the IndexError has been handled); synchronises the current stream"""
    check(lib.ur_id_guard_reset(_stream()), "ur_id_guard_reset")


class RowsPlan:
    """Result of ur_rows_plan (all device tensors; n_uniq stays on the device)."""
    __slots__ = ("n", "n_a", "uniq_idx", "seg_start", "sorted_pos", "n_uniq")


def rows_plan_alloc(n, n_a, dev):
    """Output buffers + workspace of a plan over n ids (allocated on the CURRENT stream)."""
    pl = RowsPlan()
    pl.n, pl.n_a = n, n_a
    pl.uniq_idx = torch.empty(n, dtype=torch.int32, device=dev)
    pl.seg_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    pl.sorted_pos = torch.empty(n, dtype=torch.int32, device=dev)
    pl.n_uniq = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(check(lib.ur_rows_plan_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    return pl, ws


def rows_plan(ids_a, ids_b, n_rows, out=None) -> RowsPlan:
    """ids_a: int32 tensor or None, ids_b: int64 tensor or None (flattened internally).
    out: (RowsPlan, workspace) from rows_plan_alloc -- lets a caller that launches the plan on a side stream keep the
    buffers' allocation (and release) on its main stream."""
    dev = (ids_a if ids_a is not None else ids_b).device
    n_a = ids_a.numel() if ids_a is not None else 0
    n_b = ids_b.numel() if ids_b is not None else 0
    _chk(ids_a, torch.int32, "ids_a", allow_none=True)
    _chk(ids_b, torch.int64, "ids_b", allow_none=True)
    n = n_a + n_b
    pl, ws = out if out is not None else rows_plan_alloc(n, n_a, dev)
    assert pl.n == n and pl.n_a == n_a
    id_guard_check()      # (the verdict on EARLIER plans: this one's ids are checked on the device, in its first pass)
    check(lib.ur_rows_plan(_p(ids_a), n_a, _p(ids_b), n_b, int(n_rows), _p(pl.uniq_idx), _p(pl.seg_start), _p(pl.sorted_pos),
                           _p(pl.n_uniq), _p(ws), _stream()), "ur_rows_plan")
    return pl


def rows_plan_merge(ids, run_counts, out=None) -> RowsPlan:
    """plan of int32 `ids` that consist of len(run_counts) concatenated runs, each ascending and unique (one per sending rank):
    same result as rows_plan(ids, None, ...), by a W-way merge instead of a sort (include/unirec_amd.h: ur_rows_plan_merge)."""
    _chk(ids, torch.int32, "ids")
    n = ids.numel()
    starts = [0]
    for c in run_counts:
        starts.append(starts[-1] + int(c))
    if starts[-1] != n:
        raise _lib.UnirecAmdError("rows_plan_merge: run_counts do not sum to the number of ids")
    pl, ws = out if out is not None else rows_plan_alloc(n, n, ids.device)
    arr = (C.c_int32 * len(starts))(*starts)
    check(lib.ur_rows_plan_merge(_p(ids), n, arr, len(run_counts), _p(pl.uniq_idx), _p(pl.seg_start), _p(pl.sorted_pos), _p(pl.n_uniq),
                                 _p(ws), _stream()), "ur_rows_plan_merge")
    return pl


def rows_plan_sharded(ids_a, ids_b, n_rows, world, out=None, want_counts=True):
    """Like rows_plan, but keys are (owner = id % world, local row = id // world); returns (plan, owner_counts[world] int32 dev).
    plan.uniq_idx holds the sharded keys owner * ceil(n_rows/world) + local_row.
    out: ((RowsPlan, workspace), counts) preallocated by the caller (rows_plan_alloc + an int32[world] tensor).
    want_counts=False: no counting pass (one atomic per unique key) -- shard_exchange_ids finds the owner ranges by bisection."""
    dev = (ids_a if ids_a is not None else ids_b).device
    n_a = ids_a.numel() if ids_a is not None else 0
    n_b = ids_b.numel() if ids_b is not None else 0
    _chk(ids_a, torch.int32, "ids_a", allow_none=True)
    _chk(ids_b, torch.int64, "ids_b", allow_none=True)
    n = n_a + n_b
    if out is not None:
        (pl, ws), counts = out
        assert pl.n == n and pl.n_a == n_a and counts.numel() == world
    else:
        pl, ws = rows_plan_alloc(n, n_a, dev)
        counts = torch.empty(world, dtype=torch.int32, device=dev)
    id_guard_check()
    check(lib.ur_rows_plan_sharded(_p(ids_a), n_a, _p(ids_b), n_b, int(n_rows), int(world), _p(pl.uniq_idx), _p(pl.seg_start),
                                   _p(pl.sorted_pos), _p(pl.n_uniq), _p(counts if want_counts else None), _p(ws), _stream()),
          "ur_rows_plan_sharded")
    return pl, counts


def compact_index(pl: RowsPlan, slot_of_uniq=None, out=None):
    """-> (idx_a int32[n_a], idx_b int64[n - n_a]): every lookup as an index into the compact table of unique rows (slot_of_uniq, from
    shard_exchange_ids: into the fixed-capacity [world * cap, d] table instead).  out: preallocated (idx_a, idx_b)."""
    dev = pl.uniq_idx.device
    if out is not None:
        idx_a, idx_b = out
    else:
        idx_a = torch.empty(pl.n_a, dtype=torch.int32, device=dev) if pl.n_a else None
        idx_b = torch.empty(pl.n - pl.n_a, dtype=torch.int64, device=dev) if pl.n > pl.n_a else None
    _chk(slot_of_uniq, torch.int32, "slot_of_uniq", allow_none=True)
    check(lib.ur_compact_index(_p(pl.seg_start), _p(pl.sorted_pos), _p(pl.n_uniq), pl.n, pl.n_a, _p(slot_of_uniq), _p(idx_a), _p(idx_b),
                               _stream()), "ur_compact_index")
    return idx_a, idx_b


# ---- fixed-capacity row exchange of the row-sharded tables (include/unirec_amd.h: ur_shard_exchange_*, ur_comm_*)
def comm_world():
    """-1: no RCCL library in the process, 0: the library's communicator is not initialised, else its size."""
    return int(lib.ur_comm_world())


def comm_count():
    """RCCL's OWN rank count of the library's two communicators (ncclCommCount of each; raises if they disagree with each other or with
    comm_init's arguments); 0: not initialised (the torch.distributed / loopback routes carry the exchange)."""
    if comm_world() <= 0:
        return 0
    out = (C.c_int32 * 4)()
    return check(lib.ur_comm_count(out), "ur_comm_count")


def comm_init(rank, world, group=None):
    """The library's own RCCL communicator, one per process: rank 0 makes the unique id, torch.distributed (any backend) carries it.
    Returns False -- on EVERY rank alike -- when rank 0 could not make the id (no RCCL in the process): the caller then moves the packed
    blocks with torch.distributed instead.  A failure of the collective initialisation itself raises (the other ranks are inside it)."""
    import torch.distributed as dist
    buf = (C.c_char * 256)()
    ok = True
    if rank == 0:
        ok = lib.ur_comm_unique_id(buf) >= 0
    if world > 1:
        box = [bytes(buf.raw) if ok else None]
        dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            return False
        buf = (C.c_char * 256).from_buffer_copy(box[0])
    elif not ok:
        return False
    check(lib.ur_comm_init(buf, int(rank), int(world)), "ur_comm_init")
    return True


def debug_delay(us, stream=None):
    """test aid: a spin kernel of `us` microseconds on `stream` (default: the current one)"""
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(lib.ur_debug_delay(int(us), st), "ur_debug_delay")


def comm_destroy():
    check(lib.ur_comm_destroy(), "ur_comm_destroy")


def comm_all_reduce_sum(t):
    _chk(t, torch.float32, "t")
    check(lib.ur_comm_all_reduce_sum(_p(t), t.numel(), _stream()), "ur_comm_all_reduce_sum")
    return t


def comm_all_to_all(send, recv, world, ahead=False, kind="rows"):
    """equal-split all-to-all of a packed buffer through the library's communicators (ahead: the second one, for work issued a step ahead;
    kind "ids" / "rows" / "grads": the per-collective timer the group is booked on)"""
    assert send.numel() == recv.numel() and send.dtype == recv.dtype and send.numel() % world == 0
    check(lib.ur_comm_all_to_all(_p(send), _p(recv), send.numel() * send.element_size() // world, 1 if ahead else 0,
                                 {"ids": 0, "rows": 1, "grads": 2}[kind], _stream()), "ur_comm_all_to_all")
    return recv


def shard_fixup_plan(recv_ids, world, cap, prev_own, cap2, req2, slot2, flags, counts_ws):
    """(ids only) the slots of the NEXT batch's request list whose row `prev_own` (this step's owner-side plan, or None) updates"""
    _chk(recv_ids, torch.int32, "recv_ids"); _chk(req2, torch.int32, "req2"); _chk(slot2, torch.int32, "slot2"); _chk(flags, torch.int32, "flags")
    assert recv_ids.numel() == world * cap and req2.numel() == world * cap2 == slot2.numel()
    pu, pn, pm = (prev_own.uniq_idx, prev_own.n_uniq, prev_own.n) if prev_own is not None else (None, None, 0)
    _chk(counts_ws, torch.int32, "counts_ws")
    assert counts_ws.numel() >= world
    check(lib.ur_shard_fixup_plan(_p(recv_ids), int(world), int(cap), _p(pu), _p(pn), int(pm), int(cap2), _p(req2), _p(slot2), _p(counts_ws),
                                  _p(flags), _stream()), "ur_shard_fixup_plan")


def shard_fixup_apply(compact, rows2, slot2, world, cap, cap2):
    _chk(compact, torch.float32, "compact"); _chk(rows2, torch.float32, "rows2"); _chk(slot2, torch.int32, "slot2")
    d = compact.shape[-1]
    assert compact.numel() == world * cap * d and rows2.numel() == world * cap2 * d and slot2.numel() == world * cap2
    check(lib.ur_shard_fixup_apply(_p(compact), _p(rows2), _p(slot2), int(world), int(cap), int(cap2), d, _stream()), "ur_shard_fixup_apply")


def rows_split_hot(pl: RowsPlan, last_step, excl: RowsPlan, out=None):
    """pl's unique rows split against excl's (sorted unique): -> (cold, hot) plan-like lists (uniq_idx / n_uniq only, arbitrary order);
    cold = not in excl and (last_step is None or last_step[row] != 0), hot = in both.  out: the result of an earlier call, reused."""
    _chk(last_step, torch.int32, "last_step", allow_none=True)
    dev = pl.uniq_idx.device
    if out is None:
        out = []
        for _ in range(2):
            o = RowsPlan()
            o.n, o.n_a = pl.n, 0
            o.uniq_idx = torch.empty(pl.n, dtype=torch.int32, device=dev)
            o.n_uniq = torch.empty(1, dtype=torch.int32, device=dev)
            o.seg_start = o.sorted_pos = None
            out.append(o)
        out = tuple(out)
    cold, hot = out[0], out[1]
    eu, en, em = (excl.uniq_idx, excl.n_uniq, excl.n) if excl is not None else (None, None, 0)
    check(lib.ur_rows_split_hot(_p(pl.uniq_idx), _p(pl.n_uniq), pl.n, _p(last_step), _p(eu), _p(en), int(em), _p(cold.uniq_idx),
                                _p(cold.n_uniq), _p(hot.uniq_idx), _p(hot.n_uniq), None, None, _stream()), "ur_rows_split_hot")
    return out


def shard_exchange_ids(pl: RowsPlan, counts, n_local, world, cap, send_ids, slot_of_uniq, u_of_slot, flags, recv_ids=None, transport=False):
    """pack (+ RCCL all-to-all when transport): see ur_shard_exchange_ids.  Buffers are the caller's (preallocated once)."""
    for t, nm in ((send_ids, "send_ids"), (slot_of_uniq, "slot_of_uniq"), (u_of_slot, "u_of_slot"), (flags, "flags")):
        _chk(t, torch.int32, nm)
    _chk(counts, torch.int32, "counts", allow_none=True)   # (output: per-owner unique counts, optional)
    assert send_ids.numel() == world * cap and u_of_slot.numel() == world * cap and slot_of_uniq.numel() >= pl.n
    check(lib.ur_shard_exchange_ids(_p(pl.uniq_idx), _p(pl.n_uniq), _p(counts), int(n_local), int(world), int(cap), _p(send_ids),
                                    _p(slot_of_uniq), _p(u_of_slot), _p(flags), _p(recv_ids), 1 if transport else 0, _stream()),
          "ur_shard_exchange_ids")
    return recv_ids if transport else send_ids


def shard_exchange_rows(table, req_ids, world, cap, rows_ws, compact=None, transport=False):
    _chk(table, torch.float32, "table"); _chk(req_ids, torch.int32, "req_ids"); _chk(rows_ws, torch.float32, "rows_ws")
    assert req_ids.numel() == world * cap and rows_ws.numel() == world * cap * table.shape[1]
    check(lib.ur_shard_exchange_rows(_p(table), _p(req_ids), int(world), int(cap), table.shape[1], _p(rows_ws), _p(compact),
                                     1 if transport else 0, _stream()), "ur_shard_exchange_rows")
    return compact if transport else rows_ws


def shard_exchange_grads(uniq_grad, u_of_slot, world, cap, send_ws, grads_in=None, transport=False, loss_out=None, flags=None):
    """loss_out (the loss kernels' [loss, n, guard, .] buffer) / flags (shard_exchange_ids): this rank's step flags, carried in slot 0"""
    _chk(uniq_grad, torch.float32, "uniq_grad", allow_none=True); _chk(u_of_slot, torch.int32, "u_of_slot"); _chk(send_ws, torch.float32, "send_ws")
    _chk(loss_out, torch.float32, "loss_out", allow_none=True); _chk(flags, torch.int32, "flags", allow_none=True)
    d = send_ws.shape[-1]   # (uniq_grad None: the rows are in their slots already -- rows_reduce(out=send_ws, out_rows=slot_of_uniq))
    assert (uniq_grad is None or u_of_slot.numel() == world * cap) and send_ws.numel() == world * cap * d
    check(lib.ur_shard_exchange_grads(_p(uniq_grad), _p(u_of_slot), int(world), int(cap), d, _p(loss_out), _p(flags), _p(send_ws),
                                      _p(grads_in), 1 if transport else 0, _stream()), "ur_shard_exchange_grads")
    return grads_in if transport else send_ws


def shard_step_flags(grads_in, world, cap, out4):
    """-> out4 = [gradient scale (1 / world or -1 = skip), mean loss over the ranks, #NaN ranks, #overflow ranks] (ur_shard_step_flags)"""
    _chk(grads_in, torch.float32, "grads_in"); _chk(out4, torch.float32, "out4")
    check(lib.ur_shard_step_flags(_p(grads_in), int(world), int(cap), grads_in.shape[-1], _p(out4), _stream()), "ur_shard_step_flags")
    return out4


def rows_reduce(pl: RowsPlan, rows_a, coef_b, vec_b, G, d, zero_tail=False, out=None, out_rows=None) -> torch.Tensor:
    """out / out_rows: write the sum of unique id u to out[out_rows[u], :] (a preallocated buffer in the caller's layout, e.g. the
    exchange slots of the multi-GPU step) instead of a fresh [n, d] tensor"""
    _chk(rows_a, torch.float32, "rows_a", allow_none=True)
    _chk(coef_b, torch.float32, "coef_b", allow_none=True)
    _chk(vec_b, torch.float32, "vec_b", allow_none=True)
    _chk(out, torch.float32, "out", allow_none=True); _chk(out_rows, torch.int32, "out_rows", allow_none=True)
    if out_rows is not None and (out is None or zero_tail or out_rows.numel() < pl.n or out.shape[-1] != d):
        raise _lib.UnirecAmdError("rows_reduce: out_rows needs a preallocated `out` of row width d, a map of >= n entries and no zeroed tail")
    if out is None:
        out = torch.empty(pl.n, d, dtype=torch.float32, device=pl.uniq_idx.device)
    flag = out if zero_tail else None  # non-null sumsq pointer = "zero the rows beyond n_uniq"
    check(lib.ur_rows_reduce(_p(pl.uniq_idx), _p(pl.seg_start), _p(pl.sorted_pos), _p(pl.n_uniq), pl.n, _p(rows_a), pl.n_a,
                             _p(coef_b), _p(vec_b), int(G), int(d), _p(out), _p(flag), _p(out_rows), _stream()), "ur_rows_reduce")
    return out


def rows_reduce_riders(pl: RowsPlan, rows_a, coef_b, vec_b, G, d, world, cap, out=None, out_rows=None, zero_tail=False, step_flags_out4=None,
                       flag_rows=None):
    """rows_reduce of the sharded step + riders: step_flags_out4 (owner side; rows_a = the received block): out4 as shard_step_flags;
    flag_rows = (loss_out | None, flags | None) (requester side, out_rows = slots): this rank's flag row into slot 0 of every block of out"""
    _chk(rows_a, torch.float32, "rows_a", allow_none=True); _chk(coef_b, torch.float32, "coef_b", allow_none=True)
    _chk(vec_b, torch.float32, "vec_b", allow_none=True); _chk(out, torch.float32, "out", allow_none=True)
    _chk(out_rows, torch.int32, "out_rows", allow_none=True); _chk(step_flags_out4, torch.float32, "step_flags_out4", allow_none=True)
    if out is None:
        out = torch.empty(pl.n, d, dtype=torch.float32, device=pl.uniq_idx.device)
    loss_out, flags = flag_rows if flag_rows is not None else (None, None)
    check(lib.ur_rows_reduce_riders(_p(pl.uniq_idx), _p(pl.seg_start), _p(pl.sorted_pos), _p(pl.n_uniq), pl.n, _p(rows_a), pl.n_a, _p(coef_b),
                                    _p(vec_b), int(G), int(d), _p(out), _p(out if zero_tail else None), _p(out_rows), int(world), int(cap),
                                    _p(step_flags_out4), 1 if flag_rows is not None else 0, _p(loss_out), _p(flags), _stream()),
          "ur_rows_reduce_riders")
    return out


# --------------------------------------------------------------------------------------------- optimizer
OPT_ALGOS = {"adam": 0, "adamw": 1, "sgd": 2, "adagrad": 3, "rmsprop": 4}
# torch's defaults for what the reference does not pass (trainer.py:134-152): (beta1, beta2 | alpha, eps)
OPT_DEFAULTS = {"adam": (0.9, 0.999, 1e-8), "adamw": (0.9, 0.999, 1e-8), "sgd": (0.0, 0.0, 0.0), "adagrad": (0.0, 0.0, 1e-10),
                "rmsprop": (0.0, 0.99, 1e-8)}


def adam_cfg(lr, step, wd=0.0, b1=None, b2=None, eps=None, algo="adam") -> UrAdamCfg:
    """optimizer-step configuration (`algo`: adam / adamw / sgd / adagrad / rmsprop, torch.optim semantics)"""
    d1, d2, de = OPT_DEFAULTS[algo]
    return UrAdamCfg(float(lr), float(d1 if b1 is None else b1), float(d2 if b2 is None else b2), float(de if eps is None else eps),
                     float(wd), int(step), OPT_ALGOS[algo])


def dense_adam(cfg, param, grad, m, v, grad_scale=None):
    for t, nm in ((param, "param"), (grad, "grad"), (m, "m"), (v, "v")):
        _chk(t, torch.float32, nm)
    check(lib.ur_dense_adam(C.byref(cfg), _p(param), _p(grad), _p(m), _p(v), param.numel(), _p(grad_scale), _stream()),
          "ur_dense_adam")


def sparse_adam_rows(cfg, table, m, v, pl: RowsPlan, uniq_grad, last_step=None, grad_scale=None):
    _chk(table, torch.float32, "table")
    _chk(last_step, torch.int32, "last_step", allow_none=True)
    check(lib.ur_sparse_adam_rows(C.byref(cfg), _p(table), _p(m), _p(v), _p(last_step), _p(pl.uniq_idx), _p(pl.n_uniq), pl.n,
                                  _p(uniq_grad), table.shape[1], _p(grad_scale), _stream()), "ur_sparse_adam_rows")


def rows_reduce_update(cfg, table, m, v, pl: RowsPlan, rows_a, coef_b, vec_b, G, last_step=None, grad_scale=None, next_split=None):
    """rows_reduce + sparse_adam_rows in one launch (ur_rows_reduce_update): no row-gradient tensor comes out.
    next_split: rows_split_hot(next batch's plan, last_step, excl=pl) -- its lazy replay rides in the launch too"""
    cold, hot = next_split if next_split is not None else (None, None)
    _chk(table, torch.float32, "table"); _chk(last_step, torch.int32, "last_step", allow_none=True)
    _chk(rows_a, torch.float32, "rows_a", allow_none=True); _chk(coef_b, torch.float32, "coef_b", allow_none=True)
    _chk(vec_b, torch.float32, "vec_b", allow_none=True)
    check(lib.ur_rows_reduce_update(_p(pl.uniq_idx), _p(pl.seg_start), _p(pl.sorted_pos), _p(pl.n_uniq), pl.n, _p(rows_a), pl.n_a, _p(coef_b),
                                    _p(vec_b), int(G), table.shape[1], C.byref(cfg), _p(table), _p(m), _p(v), _p(last_step), _p(grad_scale),
                                    _p(cold.uniq_idx) if cold is not None else None, _p(cold.n_uniq) if cold is not None else None,
                                    _p(hot.uniq_idx) if hot is not None else None, _p(hot.n_uniq) if hot is not None else None,
                                    int(cold.n if cold is not None else (hot.n if hot is not None else 0)), _stream()), "ur_rows_reduce_update")


def rows_reduce_update_owner(cfg, table, m, v, pl: RowsPlan, recv_rows, world, cap, step_flags_out4, last_step=None):
    """owner side of the sharded step: rows_reduce_riders(step_flags_out4=...) + sparse_adam_rows in one launch (ur_rows_reduce_update_owner)"""
    _chk(table, torch.float32, "table"); _chk(last_step, torch.int32, "last_step", allow_none=True)
    _chk(recv_rows, torch.float32, "recv_rows"); _chk(step_flags_out4, torch.float32, "step_flags_out4")
    assert pl.n == pl.n_a == world * cap and recv_rows.numel() == pl.n * table.shape[1]
    check(lib.ur_rows_reduce_update_owner(_p(pl.uniq_idx), _p(pl.seg_start), _p(pl.sorted_pos), _p(pl.n_uniq), pl.n, _p(recv_rows), table.shape[1],
                                          int(world), int(cap), _p(step_flags_out4), C.byref(cfg), _p(table), _p(m), _p(v), _p(last_step), _stream()),
          "ur_rows_reduce_update_owner")


def rows_filter_touched(pl: RowsPlan, last_step) -> RowsPlan:
    """-> a plan-like list (uniq_idx / n_uniq only, arbitrary order) of pl's rows that were ever updated (ur_rows_filter_touched)"""
    _chk(last_step, torch.int32, "last_step")
    out = RowsPlan()
    out.n, out.n_a = pl.n, 0
    out.uniq_idx = torch.empty(pl.n, dtype=torch.int32, device=last_step.device)
    out.n_uniq = torch.empty(1, dtype=torch.int32, device=last_step.device)
    out.seg_start = out.sorted_pos = None
    check(lib.ur_rows_filter_touched(_p(pl.uniq_idx), _p(pl.n_uniq), pl.n, _p(last_step), _p(out.uniq_idx), _p(out.n_uniq), _stream()),
          "ur_rows_filter_touched")
    return out


def lazy_adam_catchup(cfg, table, m, v, last_step, pl: RowsPlan, background=False):
    """background: the launch sits on a side stream under a step's compute (ur_lazy_adam_catchup_background: small grid, default priority)"""
    _chk(last_step, torch.int32, "last_step")
    fn = lib.ur_lazy_adam_catchup_background if background else lib.ur_lazy_adam_catchup
    check(fn(C.byref(cfg), _p(table), _p(m), _p(v), _p(last_step), _p(pl.uniq_idx), _p(pl.n_uniq), pl.n, table.shape[1], _stream()),
          "ur_lazy_adam_catchup")


def lazy_adam_flush(cfg, table, m, v, last_step, row0=0, n=None):
    n = table.shape[0] - row0 if n is None else n
    check(lib.ur_lazy_adam_flush(C.byref(cfg), _p(table), _p(m), _p(v), _p(last_step), row0, n, table.shape[1], _stream()),
          "ur_lazy_adam_flush")


def sumsq(x, out, accumulate=False, ws=None):
    ws = ws if ws is not None else torch.empty(2048, dtype=torch.float32, device=x.device)
    check(lib.ur_sumsq(_p(x), x.numel(), _p(out), int(accumulate), _p(ws), _stream()), "ur_sumsq")


def clip_coef(sumsq_t, max_norm, out, guard=None):
    """out[0] = min(1, max_norm / (sqrt(sumsq) + 1e-6)); with guard (device float: -1 = the step's loss was NaN) -> -1 passes through."""
    if guard is None:
        check(lib.ur_clip_coef(_p(sumsq_t), float(max_norm), _p(out), _stream()), "ur_clip_coef")
    else:
        check(lib.ur_clip_coef_guarded(_p(sumsq_t), float(max_norm), _p(guard), _p(out), _stream()), "ur_clip_coef_guarded")


# --------------------------------------------------------------------------------------------- full-item ranking
def full_rank(user_emb, item_table, target, user_id=None, hist_ptr=None, hist_sorted=None, user_bias=None, item_bias=None,
              tau=1.0):
    """rank[b] = number of items (not 0, not the target, not in the user's history) scoring above the target.
    Returns (rank int32[B], target_score float32[B]).  See include/unirec_amd.h: ur_full_rank."""
    _chk(user_emb, torch.float32, "user_emb")
    _chk(item_table, torch.float32, "item_table")
    _chk(target, torch.int64, "target")
    _chk(user_id, torch.int64, "user_id", allow_none=True)
    _chk(hist_ptr, torch.int64, "hist_ptr", allow_none=True)
    _chk(hist_sorted, torch.int32, "hist_sorted", allow_none=True)
    _chk(user_bias, torch.float32, "user_bias", allow_none=True)
    _chk(item_bias, torch.float32, "item_bias", allow_none=True)
    B, d = user_emb.shape
    n_items = item_table.shape[0]
    if item_table.shape[1] != d or target.numel() != B:
        raise _lib.UnirecAmdError("full_rank: shape mismatch")
    rank = torch.empty(B, dtype=torch.int32, device=user_emb.device)
    ts = torch.empty(B, dtype=torch.float32, device=user_emb.device)
    thr = torch.empty(B, dtype=torch.float32, device=user_emb.device)
    n_users = hist_ptr.numel() - 1 if hist_ptr is not None else 0
    check(lib.ur_full_rank(_p(user_emb), _p(item_table), n_items, B, d, _p(target), _p(user_id), _p(hist_ptr), _p(hist_sorted),
                           n_users, _p(user_bias), _p(item_bias), float(tau), _p(rank), _p(ts), _p(thr), _stream()), "ur_full_rank")
    # rank = (count over all items, MFMA summation order) - (count over history / non-items, row-dot summation order): a history
    # item within one fp32 rounding of the target's score can be subtracted without having been counted -- a count is never < 0
    return rank.clamp_(min=0), ts


def full_rank_shard(phase, user_emb, shard_table, local_target, thr=None, user_id=None, hist_ptr=None, hist_sorted_local=None,
                    item_bias_local=None, n_rows=None, excl_row=-1):
    """One phase of the row-sharded full-item count on THIS rank's shard (include/unirec_amd.h: ur_full_rank_shard).
    phase 1 -> thr float32[B] (this shard's contribution: the score of the targets it owns, 0 elsewhere; all-reduce it);
    phase 2 (thr = the all-reduced thresholds) -> rank_partial int32[B] (all-reduce it).
    n_rows: rows of the shard that are scanned (default all); excl_row: a local row that is not an item (or -1)."""
    _chk(user_emb, torch.float32, "user_emb")
    _chk(shard_table, torch.float32, "shard_table")
    _chk(local_target, torch.int64, "local_target")
    _chk(user_id, torch.int64, "user_id", allow_none=True)
    _chk(hist_ptr, torch.int64, "hist_ptr", allow_none=True)
    _chk(hist_sorted_local, torch.int32, "hist_sorted_local", allow_none=True)
    _chk(item_bias_local, torch.float32, "item_bias_local", allow_none=True)
    B, d = user_emb.shape
    if phase == 1:
        thr = torch.empty(B, dtype=torch.float32, device=user_emb.device)
        rank = None
    else:
        _chk(thr, torch.float32, "thr")
        rank = torch.empty(B, dtype=torch.int32, device=user_emb.device)
    n_users = hist_ptr.numel() - 1 if hist_ptr is not None else 0
    check(lib.ur_full_rank_shard(int(phase), _p(user_emb), _p(shard_table), int(n_rows) if n_rows is not None else shard_table.shape[0], B, d,
                                 _p(local_target), _p(user_id), _p(hist_ptr), _p(hist_sorted_local), n_users, _p(item_bias_local),
                                 int(excl_row), _p(thr), _p(rank), _stream()),
          "ur_full_rank_shard")
    return thr if phase == 1 else rank


def full_topk(user_emb, item_table, k, user_id=None, hist_ptr=None, hist_sorted=None, user_bias=None, item_bias=None, tau=1.0):
    """-> (scores float32[B,k], ids int64[B,k]): the k best items per row (not item 0, not in the user's history), best
    first.  See include/unirec_amd.h: ur_full_topk."""
    _chk(user_emb, torch.float32, "user_emb")
    _chk(item_table, torch.float32, "item_table")
    _chk(user_id, torch.int64, "user_id", allow_none=True)
    _chk(hist_ptr, torch.int64, "hist_ptr", allow_none=True)
    _chk(hist_sorted, torch.int32, "hist_sorted", allow_none=True)
    _chk(user_bias, torch.float32, "user_bias", allow_none=True)
    _chk(item_bias, torch.float32, "item_bias", allow_none=True)
    B, d = user_emb.shape
    n_items = item_table.shape[0]
    dev = user_emb.device
    scores = torch.empty(B, k, dtype=torch.float32, device=dev)
    ids = torch.empty(B, k, dtype=torch.int64, device=dev)
    ws = torch.empty(check(lib.ur_full_topk_workspace_bytes(B, n_items, k), "ur_full_topk_workspace_bytes"), dtype=torch.uint8, device=dev)
    n_users = hist_ptr.numel() - 1 if hist_ptr is not None else 0
    check(lib.ur_full_topk(_p(user_emb), _p(item_table), n_items, B, d, int(k), _p(user_id), _p(hist_ptr), _p(hist_sorted), n_users,
                           _p(user_bias), _p(item_bias), float(tau), _p(scores), _p(ids), _p(ws), _stream()), "ur_full_topk")
    return scores, ids


# --------------------------------------------------------------------------------------------- pooled history (AvgHist / SVD++)
def pool_rows_fwd(table, item_seq, seq_len, alpha, base=None):
    _chk(table, torch.float32, "table")
    _chk(item_seq, torch.int32, "item_seq")
    _chk(seq_len, torch.int64, "seq_len")
    _chk(base, torch.float32, "base", allow_none=True)
    B, L = item_seq.shape
    d = table.shape[1]
    out = torch.empty(B, d, dtype=torch.float32, device=table.device)
    check(lib.ur_pool_rows_fwd(_p(table), table.shape[0], d, _p(item_seq), _p(seq_len), _p(base), float(alpha), B, L, _p(out), _stream()),
          "ur_pool_rows_fwd")
    return out


def pool_rows_bwd(d_user, seq_len, alpha, L):
    _chk(d_user, torch.float32, "d_user")
    _chk(seq_len, torch.int64, "seq_len")
    B, d = d_user.shape
    rows = torch.empty(B * L, d, dtype=torch.float32, device=d_user.device)
    check(lib.ur_pool_rows_bwd(_p(d_user), _p(seq_len), float(alpha), B, L, d, _p(rows), _stream()), "ur_pool_rows_bwd")
    return rows


# --------------------------------------------------------------------------------------------- fullsoftmax
def full_softmax_fwd(user_emb, item_table, target, user_id=None, user_bias=None, item_bias=None, tau=1.0, score_clip=-1.0):
    """-> (loss_out float32[2] = [mean loss, B], lse float32[B], workspace) -- see include/unirec_amd.h: ur_full_softmax_fwd."""
    _chk(user_emb, torch.float32, "user_emb")
    _chk(item_table, torch.float32, "item_table")
    _chk(target, torch.int64, "target")
    B, d = user_emb.shape
    N = item_table.shape[0]
    dev = user_emb.device
    cfg = loss_cfg(B, 1, d, "bpr", tau, score_clip)
    cfg.loss_type = -1   # UR_LOSS_NONE: the target's score through the same scorer kernel
    ts, _, _ = gather_dot_loss_fwd(cfg, user_emb, item_table, target.view(B, 1).contiguous(), None, user_bias, item_bias,
                                   user_id if user_bias is not None else None)
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    loss_out = torch.empty(4, dtype=torch.float32, device=dev)   # [loss, count, update guard, -]
    ws = torch.empty(check(lib.ur_full_softmax_workspace_bytes(B, d, N), "ur_full_softmax_workspace_bytes"), dtype=torch.uint8, device=dev)
    check(lib.ur_full_softmax_fwd(_p(user_emb), _p(item_table), N, B, d, _p(target), _p(user_id), _p(user_bias), _p(item_bias), float(tau),
                                  float(score_clip if score_clip else -1.0), _p(ts.view(-1)), _p(lse), _p(loss_out), _p(ws), _stream()),
          "ur_full_softmax_fwd")
    return loss_out, lse, ws


def full_softmax_bwd(user_emb, item_table, target, lse, ws, user_id=None, user_bias=None, item_bias=None, tau=1.0, score_clip=-1.0,
                     d_loss=None):
    """-> (d_user [B,d], d_table [N,d] dense, d_item_bias [N] | None)."""
    B, d = user_emb.shape
    N = item_table.shape[0]
    dev = user_emb.device
    d_user = torch.empty(B, d, dtype=torch.float32, device=dev)
    d_table = torch.empty(N, d, dtype=torch.float32, device=dev)
    d_ib = torch.empty(N, dtype=torch.float32, device=dev) if item_bias is not None else None
    check(lib.ur_full_softmax_bwd(_p(user_emb), _p(item_table), N, B, d, _p(target), _p(user_id), _p(user_bias), _p(item_bias), float(tau),
                                  float(score_clip if score_clip else -1.0), _p(lse), _p(d_loss), _p(d_user), _p(d_table), _p(d_ib), _p(ws),
                                  _stream()), "ur_full_softmax_bwd")
    return d_user, d_table, d_ib


def full_softmax_fwd_shard(user_emb_all, shard_rows, target_row, user_id=None, user_bias=None, item_bias_rows=None, tau=1.0, score_clip=-1.0):
    """this rank's partials over the rows it scores -> (part3 float32[3, B_all] = (max, sum-exp, target score), workspace)
    -- include/unirec_amd.h: ur_full_softmax_fwd_shard."""
    _chk(user_emb_all, torch.float32, "user_emb_all")
    _chk(shard_rows, torch.float32, "shard_rows")
    _chk(target_row, torch.int64, "target_row")
    B, d = user_emb_all.shape
    n = shard_rows.shape[0]
    dev = user_emb_all.device
    part3 = torch.empty(3, B, dtype=torch.float32, device=dev)
    ws = torch.empty(check(lib.ur_full_softmax_workspace_bytes(B, d, n), "ur_full_softmax_workspace_bytes"), dtype=torch.uint8, device=dev)
    check(lib.ur_full_softmax_fwd_shard(_p(user_emb_all), _p(shard_rows), n, B, d, _p(target_row), _p(user_id), _p(user_bias),
                                        _p(item_bias_rows), float(tau), float(score_clip if score_clip else -1.0), _p(part3), _p(ws),
                                        _stream()), "ur_full_softmax_fwd_shard")
    return part3, ws


def full_softmax_combine_shards(parts, col0, B_own):
    """parts float32[W, 3, B_all] (the all-gathered part3 of every rank) -> (lse float32[B_all], loss_out float32[4])."""
    _chk(parts, torch.float32, "parts")
    W, _, BA = parts.shape
    lse = torch.empty(BA, dtype=torch.float32, device=parts.device)
    loss_out = torch.empty(4, dtype=torch.float32, device=parts.device)
    check(lib.ur_full_softmax_combine_shards(_p(parts), W, BA, int(col0), int(B_own), _p(lse), _p(loss_out), _stream()),
          "ur_full_softmax_combine_shards")
    return lse, loss_out


def full_softmax_bwd_shard(user_emb_all, shard_rows, target_row, lse, ws, d_shard_rows, user_id=None, user_bias=None, item_bias_rows=None,
                           tau=1.0, score_clip=-1.0, d_loss=None, zero_row0=False):
    """-> (d_user_all [B_all, d] partial over this rank's items, d_item_bias_rows [n] | None); d_shard_rows [n, d] is overwritten."""
    B, d = user_emb_all.shape
    n = shard_rows.shape[0]
    dev = user_emb_all.device
    _chk(d_shard_rows, torch.float32, "d_shard_rows")
    assert tuple(d_shard_rows.shape) == (n, d)
    d_user = torch.empty(B, d, dtype=torch.float32, device=dev)
    d_ib = torch.empty(n, dtype=torch.float32, device=dev) if item_bias_rows is not None else None
    check(lib.ur_full_softmax_bwd_shard(_p(user_emb_all), _p(shard_rows), n, B, d, _p(target_row), _p(user_id), _p(user_bias),
                                        _p(item_bias_rows), float(tau), float(score_clip if score_clip else -1.0), _p(lse), _p(d_loss),
                                        _p(d_user), _p(d_shard_rows), _p(d_ib), 1 if zero_row0 else 0, _p(ws), _stream()),
          "ur_full_softmax_bwd_shard")
    return d_user, d_ib


def rows_scatter_add(pl: RowsPlan, rows, dense):
    check(lib.ur_rows_scatter_add(_p(pl.uniq_idx), _p(pl.n_uniq), pl.n, _p(rows), rows.shape[1], _p(dense), _stream()), "ur_rows_scatter_add")


# --------------------------------------------------------------------------------------------- AttHist
def atthist_cfg(B, L, d, p_drop=0.0, drop_seed=0, drop_step=0):
    return _lib.UrAttHistCfg(B, L, d, float(p_drop), int(drop_seed), int(drop_step))


def atthist_param_layout(cfg):
    offs = (C.c_int64 * 3)()
    total = check(lib.ur_atthist_param_layout(C.byref(cfg), offs), "ur_atthist_param_layout")
    return list(offs), int(total)


def atthist_workspace(cfg, device):
    return torch.empty(check(lib.ur_atthist_workspace_bytes(C.byref(cfg)), "ur_atthist_workspace_bytes"), dtype=torch.uint8, device=device)


def atthist_fwd(cfg, item_table, dense, item_seq, ws):
    _chk(item_table, torch.float32, "item_table")
    _chk(dense, torch.float32, "dense")
    _chk(item_seq, torch.int32, "item_seq")
    user_emb = torch.empty(cfg.B, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_atthist_fwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(user_emb), _p(ws), _stream()),
          "ur_atthist_fwd")
    return user_emb


def atthist_bwd(cfg, item_table, dense, item_seq, d_user_emb, ws):
    _chk(d_user_emb, torch.float32, "d_user_emb")
    dense_grad = torch.empty_like(dense)
    d_emb_rows = torch.empty(cfg.B * cfg.L, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_atthist_bwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(d_user_emb), _p(ws),
                             _p(dense_grad), _p(d_emb_rows), _stream()), "ur_atthist_bwd")
    return dense_grad, d_emb_rows


# --------------------------------------------------------------------------------------------- ConvFormer / FASTConvFormer
PADDING_MODES = {"circular": 0, "reflect": 1, "constant": 2}


def convformer_cfg(B, L, d, inner, n_layers, act, conv_size, padding_mode, fast, seq_merge, eps, seq_decay, p_hidden=0.0, drop_seed=0,
                   drop_step=0):
    pm = PADDING_MODES[padding_mode] if isinstance(padding_mode, str) else int(padding_mode)
    return _lib.UrConvFormerCfg(B, L, d, inner, n_layers, ACT_IDS[act], conv_size, pm, int(bool(fast)), int(bool(seq_merge)), float(eps),
                                float(seq_decay), float(p_hidden), int(drop_seed), int(drop_step))


def convformer_param_layout(cfg):
    offs = (C.c_int64 * (3 + 10 * cfg.n_layers))()
    total = check(lib.ur_convformer_param_layout(C.byref(cfg), offs), "ur_convformer_param_layout")
    return list(offs), int(total)


def convformer_workspace(cfg, device):
    return torch.empty(check(lib.ur_convformer_workspace_bytes(C.byref(cfg)), "ur_convformer_workspace_bytes"), dtype=torch.uint8, device=device)


def convformer_fwd(cfg, item_table, dense, item_seq, seq_len, ws):
    _chk(item_table, torch.float32, "item_table")
    _chk(dense, torch.float32, "dense")
    _chk(item_seq, torch.int32, "item_seq")
    _chk(seq_len, torch.int64, "seq_len", allow_none=True)
    user_emb = torch.empty(cfg.B, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_convformer_fwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(seq_len), _p(user_emb), _p(ws),
                                _stream()), "ur_convformer_fwd")
    return user_emb


def convformer_bwd(cfg, item_table, dense, item_seq, seq_len, d_user_emb, ws):
    _chk(d_user_emb, torch.float32, "d_user_emb")
    dense_grad = torch.empty_like(dense)
    d_emb_rows = torch.empty(cfg.B * cfg.L, cfg.d, dtype=torch.float32, device=dense.device)
    check(lib.ur_convformer_bwd(C.byref(cfg), _p(item_table), item_table.shape[0], _p(dense), _p(item_seq), _p(seq_len), _p(d_user_emb),
                                _p(ws), _p(dense_grad), _p(d_emb_rows), _stream()), "ur_convformer_bwd")
    return dense_grad, d_emb_rows
