"""ctypes binding of the C ABI declared in include/unirec_amd.h.

The product path has NO CPU fallback: importing this module loads the in-tree HIP library and
raises if it is missing (build it with ``python -m unirec_amd.build`` / ``__graft_entry__.build()``).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: the process must use torch's bundled HIP runtime (libamdhip64); loading ours
#                              before torch's would put two runtimes in one process ("no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
# UR_DEBUG_BOUNDS=1: the bounds-checked build (python -m unirec_amd.build --debug-bounds; csrc/common.h: UR_ROW) -- every gathered row
# index is checked where it is used, a bad one prints its site and traps
LIB_PATH = os.path.join(_HERE, "libunirec_amd_dbg.so" if os.environ.get("UR_DEBUG_BOUNDS", "0") not in ("", "0") else "libunirec_amd.so")

UR_MAX_LAYERS = 8
UR_SASREC_N_GLOBAL = 3
UR_SASREC_N_PER_LAYER = 16

ACT_IDS = {"gelu": 0, "relu": 1, "swish": 2, "tanh": 3, "sigmoid": 4}
LOSS_IDS = {"bce": 0, "bpr": 1, "softmax": 2, "ccl": 3, "fullsoftmax": 4}


class UrSasrecCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("d", C.c_int32), ("n_heads", C.c_int32),
                ("inner", C.c_int32), ("n_layers", C.c_int32), ("act", C.c_int32), ("use_pos", C.c_int32),
                ("eps", C.c_float), ("last_only", C.c_int32), ("skip_padding", C.c_int32),
                ("p_hidden", C.c_float), ("p_attn", C.c_float), ("drop_seed", C.c_int64), ("drop_step", C.c_int64),
                ("mfma_arith", C.c_int32), ("reserved_", C.c_int32)]


class UrConvFormerCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("d", C.c_int32), ("inner", C.c_int32), ("n_layers", C.c_int32), ("act", C.c_int32),
                ("conv_size", C.c_int32), ("padding_mode", C.c_int32), ("fast", C.c_int32), ("seq_merge", C.c_int32),
                ("eps", C.c_float), ("seq_decay", C.c_float), ("p_hidden", C.c_float), ("drop_seed", C.c_int64), ("drop_step", C.c_int64)]


class UrAttHistCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("d", C.c_int32), ("p_drop", C.c_float), ("drop_seed", C.c_int64), ("drop_step", C.c_int64)]


class UrLossCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("G", C.c_int32), ("d", C.c_int32), ("loss_type", C.c_int32),
                ("tau", C.c_float), ("score_clip", C.c_float), ("ccl_w", C.c_float), ("ccl_m", C.c_float),
                ("group_size", C.c_int32)]


class UrAdamCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step", C.c_int32), ("algo", C.c_int32)]


class UrGruCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("d", C.c_int32), ("H", C.c_int32),
                ("p_drop", C.c_float), ("drop_seed", C.c_int64), ("drop_step", C.c_int64), ("mfma_arith", C.c_int32), ("reserved_", C.c_int32)]


P = C.c_void_p
I64 = C.c_int64
I32 = C.c_int32

# name -> (restype, argtypes).  Every symbol include/unirec_amd.h declares must be listed here
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    "ur_last_error": (C.c_char_p, []),
    "ur_version": (C.c_int, []),
    "ur_id_guard_state": (C.c_int, [C.POINTER(I64)]),
    "ur_id_guard_reset": (C.c_int, [P]),
    "ur_trace_ranges_pushed": (C.c_int64, []),
    "ur_embedding_gather_f32": (C.c_int, [P, I64, C.c_int, P, C.c_int, I64, P, P]),
    "ur_sasrec_param_layout": (I64, [C.POINTER(UrSasrecCfg), C.POINTER(I64)]),
    "ur_sasrec_workspace_bytes": (I64, [C.POINTER(UrSasrecCfg)]),
    "ur_sasrec_fwd": (C.c_int, [C.POINTER(UrSasrecCfg), P, I64, P, P, P, P, P]),
    "ur_sasrec_bwd": (C.c_int, [C.POINTER(UrSasrecCfg), P, I64, P, P, P, P, P, P, P]),
    "ur_sasrec_bwd_deferred": (C.c_int, [C.POINTER(UrSasrecCfg), P, I64, P, P, P, P, P, P, P]),
    "ur_sasrec_bwd_join": (C.c_int, [P]),
    "ur_stream_wait_stream": (C.c_int, [P, P]),
    "ur_sasrec_side_stream": (P, []),
    "ur_sasrec_side_publish": (C.c_int, [C.c_int]),
    "ur_gru_param_layout": (I64, [C.POINTER(UrGruCfg), C.POINTER(I64)]),
    "ur_gru_workspace_bytes": (I64, [C.POINTER(UrGruCfg)]),
    "ur_gru_fwd": (C.c_int, [C.POINTER(UrGruCfg), P, I64, P, P, P, P, P]),
    "ur_gru_bwd": (C.c_int, [C.POINTER(UrGruCfg), P, I64, P, P, P, P, P, P, P]),
    "ur_gather_dot_loss_fwd": (C.c_int, [C.POINTER(UrLossCfg), P, P, I64, P, P, P, P, P, P, P, P, P]),
    "ur_gather_dot_loss_bwd": (C.c_int, [C.POINTER(UrLossCfg), P, P, I64, P, P, P, P, P, P, P, P, P]),
    "ur_gather_dot_loss_fused_supported": (C.c_int, [C.POINTER(UrLossCfg)]),
    "ur_gather_dot_loss_fwd_bwd": (C.c_int, [C.POINTER(UrLossCfg), P, P, I64, P, P, P, P, P, P, P, P, P, P, P, P]),
    "ur_rows_plan_workspace_bytes": (I64, [I64]),
    "ur_rows_plan": (C.c_int, [P, I64, P, I64, I64, P, P, P, P, P, P]),
    "ur_rows_plan_merge": (C.c_int, [P, I64, P, I32, P, P, P, P, P, P]),
    "ur_rows_plan_sharded": (C.c_int, [P, I64, P, I64, I64, I32, P, P, P, P, P, P, P]),
    "ur_compact_index": (C.c_int, [P, P, P, I64, I64, P, P, P, P]),
    "ur_shard_exchange_ids": (C.c_int, [P, P, P, I64, I32, I32, P, P, P, P, P, I32, P]),
    "ur_shard_exchange_rows": (C.c_int, [P, P, I32, I32, I32, P, P, I32, P]),
    "ur_shard_exchange_grads": (C.c_int, [P, P, I32, I32, I32, P, P, P, P, I32, P]),
    "ur_shard_step_flags": (C.c_int, [P, I32, I32, I32, P, P]),
    "ur_comm_world": (C.c_int, []),
    "ur_comm_count": (C.c_int, [P]),
    "ur_loop_create": (P, [I32]),
    "ur_loop_destroy": (C.c_int, [P]),
    "ur_loop_attach": (C.c_int, [P, I32]),
    "ur_loop_detach": (C.c_int, []),
    "ur_loop_world": (C.c_int, []),
    "ur_loop_post": (C.c_int, [P, I32, P]),
    "ur_loop_all_to_all_pull": (C.c_int, [P, I64, I32, I32, P]),
    "ur_loop_all_reduce_pull": (C.c_int, [I64, I32, P]),
    "ur_loop_finish": (C.c_int, [I32, P, I64, P]),
    "ur_debug_delay": (C.c_int, [I32, P]),
    "ur_comm_unique_id": (C.c_int, [P]),
    "ur_comm_init": (C.c_int, [P, I32, I32]),
    "ur_comm_destroy": (C.c_int, []),
    "ur_comm_all_reduce_sum": (C.c_int, [P, I64, P]),
    "ur_comm_all_to_all": (C.c_int, [P, P, I64, I32, I32, P]),
    "ur_shard_fixup_plan": (C.c_int, [P, I32, I32, P, P, I64, I32, P, P, P, P, P]),
    "ur_shard_fixup_apply": (C.c_int, [P, P, P, I32, I32, I32, I32, P]),
    "ur_rows_split_hot": (C.c_int, [P, P, I64, P, P, P, I64, P, P, P, P, P, P, P]),
    "ur_rows_reduce_riders": (C.c_int, [P, P, P, P, I64, P, I64, P, P, I32, I32, P, P, P, I32, I32, P, I32, P, P, P]),
    "ur_rows_reduce": (C.c_int, [P, P, P, P, I64, P, I64, P, P, I32, I32, P, P, P, P]),
    "ur_dense_adam": (C.c_int, [C.POINTER(UrAdamCfg), P, P, P, P, I64, P, P]),
    "ur_sparse_adam_rows": (C.c_int, [C.POINTER(UrAdamCfg), P, P, P, P, P, P, I64, P, I32, P, P]),
    "ur_rows_reduce_update": (C.c_int, [P, P, P, P, I64, P, I64, P, P, I32, I32, C.POINTER(UrAdamCfg), P, P, P, P, P, P, P, P, P, I64, P]),
    "ur_rows_reduce_update_owner": (C.c_int, [P, P, P, P, I64, P, I32, I32, I32, P, C.POINTER(UrAdamCfg), P, P, P, P, P]),
    "ur_lazy_adam_catchup": (C.c_int, [C.POINTER(UrAdamCfg), P, P, P, P, P, P, I64, I32, P]),
    "ur_lazy_adam_catchup_background": (C.c_int, [C.POINTER(UrAdamCfg), P, P, P, P, P, P, I64, I32, P]),
    "ur_rows_filter_touched": (C.c_int, [P, P, I64, P, P, P, P]),
    "ur_lazy_adam_flush": (C.c_int, [C.POINTER(UrAdamCfg), P, P, P, P, I64, I64, I32, P]),
    "ur_sumsq": (C.c_int, [P, I64, P, C.c_int, P, P]),
    "ur_clip_coef": (C.c_int, [P, C.c_float, P, P]),
    "ur_clip_coef_guarded": (C.c_int, [P, C.c_float, P, P, P]),
    "ur_host_sampler_create": (P, [C.c_uint64]),
    "ur_host_sampler_destroy": (None, [P]),
    "ur_host_sampler_getrandbits": (C.c_uint64, [P, C.c_int]),
    "ur_host_sampler_random": (C.c_double, [P]),
    "ur_host_sampler_state": (C.c_int, [P, P]),
    "ur_mt_workspace_bytes": (I64, [C.c_int32, C.c_int32, I64, C.c_int32]),
    "ur_mt_build_rows": (C.c_int, [P, P, P, C.c_int32, C.c_int32, I64, I64, P, P, P, C.c_int32, C.c_int32, P, P, P, P, P]),
    "ur_device_build_seq_choice": (C.c_int, [P, P, C.c_int32, C.c_int32, I64, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, P, P]),
    "ur_host_sampler_randint": (I64, [P, I64, I64]),
    "ur_host_sampler_set_alias": (C.c_int, [P, P, I64]),
    "ur_host_build_rows": (C.c_int, [P, P, P, I64, I64, I64, I32, P, P, P, I32, I32, I32, I32, P, P, P]),
    "ur_sample_negatives": (C.c_int, [P, P, I32, I32, I64, I64, P, P, C.c_uint64, C.c_uint32, P, P, P]),
    "ur_sample_negatives_pop": (C.c_int, [P, P, I32, I32, I64, I64, P, P, P, P, C.c_uint64, C.c_uint32, P, P, P]),
    "ur_alias_table_build": (C.c_int, [P, I64, P, P]),
    "ur_device_build_seq": (C.c_int, [P, P, C.c_int32, C.c_int32, I64, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                      C.c_uint32, P, P, P]),
    "ur_gemm_nt": (C.c_int, [P, C.c_int, P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P, C.c_int,
                             P, P, C.c_float, P, P, P]),
    "ur_gemm_tn_workspace_floats": (I64, [C.c_int, C.c_int, C.c_int]),
    "ur_gemm_tn_group": (C.c_int, [C.c_int, P, P, P, P, P, P, P, P, C.c_int, P, P, P, P, P]),
    "ur_set_mfma_arith": (C.c_int, [C.c_int]),
    "ur_get_mfma_arith": (C.c_int, []),
    "ur_gemm_tn": (C.c_int, [P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_int, P, P, P]),
    "ur_convformer_param_layout": (I64, [P, P]),
    "ur_convformer_workspace_bytes": (I64, [P]),
    "ur_convformer_fwd": (C.c_int, [P, P, I64, P, P, P, P, P, P]),
    "ur_convformer_bwd": (C.c_int, [P, P, I64, P, P, P, P, P, P, P, P]),
    "ur_atthist_param_layout": (I64, [P, P]),
    "ur_atthist_workspace_bytes": (I64, [P]),
    "ur_atthist_fwd": (C.c_int, [P, P, I64, P, P, P, P, P]),
    "ur_atthist_bwd": (C.c_int, [P, P, I64, P, P, P, P, P, P, P]),
    "ur_pool_rows_fwd": (C.c_int, [P, I64, C.c_int32, P, P, P, C.c_float, C.c_int32, C.c_int32, P, P]),
    "ur_pool_rows_bwd": (C.c_int, [P, P, C.c_float, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "ur_sasrec_set_side_stream": (C.c_int, [C.c_int]),
    "ur_sasrec_set_chain": (C.c_int, [C.c_int]),
    "ur_full_rank": (C.c_int, [P, P, I64, C.c_int32, C.c_int32, P, P, P, P, I64, P, P, C.c_float, P, P, P, P]),
    "ur_full_rank_shard": (C.c_int, [C.c_int32, P, P, I64, C.c_int32, C.c_int32, P, P, P, P, I64, P, I64, P, P, P]),
    "ur_full_topk_workspace_bytes": (I64, [C.c_int32, I64, C.c_int32]),
    "ur_full_topk": (C.c_int, [P, P, I64, C.c_int32, C.c_int32, C.c_int32, P, P, P, I64, P, P, C.c_float, P, P, P, P]),
    "ur_full_softmax_workspace_bytes": (I64, [C.c_int32, C.c_int32, I64]),
    "ur_full_softmax_fwd": (C.c_int, [P, P, I64, C.c_int32, C.c_int32, P, P, P, P, C.c_float, C.c_float, P, P, P, P, P]),
    "ur_full_softmax_bwd": (C.c_int, [P, P, I64, C.c_int32, C.c_int32, P, P, P, P, C.c_float, C.c_float, P, P, P, P, P, P, P]),
    "ur_full_softmax_fwd_shard": (C.c_int, [P, P, I64, C.c_int32, C.c_int32, P, P, P, P, C.c_float, C.c_float, P, P, P]),
    "ur_full_softmax_combine_shards": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, P]),
    "ur_full_softmax_bwd_shard": (C.c_int, [P, P, I64, C.c_int32, C.c_int32, P, P, P, P, C.c_float, C.c_float, P, P, P, P, P, C.c_int32, P, P]),
    "ur_rows_scatter_add": (C.c_int, [P, P, I64, P, C.c_int32, P, P]),
    "ur_prof_enable": (C.c_int, [C.c_int]),
    "ur_prof_set_mask": (C.c_int, [C.c_uint32]),
    "ur_prof_reset": (C.c_int, []),
    "ur_prof_num_classes": (C.c_int, []),
    "ur_prof_class_name": (C.c_char_p, [C.c_int]),
    "ur_prof_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}


class UnirecAmdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise UnirecAmdError(
            f"{LIB_PATH} is missing: the HIP library has not been built (python -m unirec_amd.build). "
            "unirec_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

# ---- one library call at a time (the in-process loopback transport, pgroup.LoopbackGroup: W rank THREADS call into the library, ctypes
# releases the GIL around a foreign call, and the library's lazily initialised per-device state -- zero rows, hand-off counters, event
# rings -- is written for one calling thread).  Off by default: a single-threaded process pays nothing.  The calls are short host-side
# enqueues; the rendezvous of the rank threads happens in Python, outside the lock.
import threading as _threading

_CALL_LOCK = _threading.Lock()
_RAW = {}


_SERIAL_USERS = 0


def serialize_calls(on=True):
    """One library call at a time (a global lock around every entry point) while any LoopbackGroup is open: reference-counted, the raw
    functions are restored when the last group closes (the lock used to stay for the rest of the process)."""
    global _SERIAL_USERS
    with _CALL_LOCK:
        _SERIAL_USERS = max(0, _SERIAL_USERS + (1 if on else -1))
        if (on and _SERIAL_USERS > 1) or (not on and _SERIAL_USERS > 0):
            return
    for name in SIGNATURES:
        if on and name not in _RAW:
            raw = getattr(lib, name)
            _RAW[name] = raw

            def locked(*a, _f=raw):
                with _CALL_LOCK:
                    return _f(*a)
            setattr(lib, name, locked)
        elif not on and name in _RAW:
            setattr(lib, name, _RAW.pop(name))


def check(rc, what=""):
    """Raise UnirecAmdError (a RuntimeError, like TORCH_CHECK would) when a C call failed."""
    if rc < 0:
        msg = lib.ur_last_error()
        raise UnirecAmdError(f"{what or 'unirec_amd'} failed (code {rc}): {msg.decode() if msg else ''}")
    return rc
