"""ctypes binding of libpimemb.so -- the stub a Python maintainer of the reference would write in
place of `CDLL("./emblib.so")` (upmem/c_test.py:15-16,81).  Signatures mirror include/pimemb.h
one to one.  There is no fallback: if the library is missing, loading raises."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

EMB_OK = 0
EMB_ERR_INVALID, EMB_ERR_NOMEM, EMB_ERR_DEVICE, EMB_ERR_UNSUPPORTED, EMB_ERR_RANGE = -1, -2, -3, -4, -5
EMB_F32, EMB_F16, EMB_FIXED32 = 0, 1, 2
EMB_IDX_U32, EMB_IDX_I64 = 0, 1
EMB_MEM_HOST, EMB_MEM_DEVICE = 0, 1
EMB_FLAG_STAGE_TIMING, EMB_FLAG_CHECK_INPUTS, EMB_FLAG_DEFER_CHECK = 1, 2, 4


class EmbConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_tables", C.c_uint32), ("flags", C.c_uint32)]


class EmbLookupDesc(C.Structure):
    _fields_ = [("table_id", C.c_uint32), ("fixed_pooling", C.c_uint32),
                ("indices", C.c_void_p), ("offsets", C.c_void_p),
                ("n_indices", C.c_uint64), ("n_bags", C.c_uint64), ("pooled", C.c_void_p)]


class EmbStats(C.Structure):
    _fields_ = [("n_lookup_calls", C.c_uint64), ("n_kernel_launches", C.c_uint64),
                ("n_bags", C.c_uint64), ("n_indices", C.c_uint64), ("table_bytes", C.c_uint64),
                ("us_copy_in_indices", C.c_double), ("us_copy_in_lengths", C.c_double),
                ("us_launch", C.c_double), ("us_copy_out", C.c_double),
                ("us_post_process", C.c_double), ("us_sync", C.c_double),
                ("n_launches_by_kind", C.c_uint64 * 5)]


class EmbRouteTable(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("offsets", C.c_void_p), ("n_indices", C.c_uint64),
                ("fixed_pooling", C.c_uint32), ("rows_per_shard", C.c_uint32)]


class EmbCommOp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("is_recv", C.c_int32), ("ptr", C.c_void_p), ("bytes", C.c_uint64)]


EMB_PLACE_REPLICATED, EMB_PLACE_WHOLE, EMB_PLACE_ROWS = 0, 1, 2
EMB_SHARD_SELF_VIA_COMM, EMB_SHARD_CHECK_SERVED, EMB_SHARD_PEER_STORES, EMB_SHARD_NO_DIRECT, EMB_SHARD_DEFER_REPORT = 1, 2, 4, 8, 16
EMB_RANGE_OPEN_END = 1 << 63


class EmbShardTable(C.Structure):
    _fields_ = [("placement", C.c_uint32), ("owner", C.c_int32), ("engine_table", C.c_uint32), ("rows_per_shard", C.c_uint32)]


class EmbShardInput(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("offsets", C.c_void_p), ("n_indices", C.c_uint64), ("fixed_pooling", C.c_uint32),
                ("index_type", C.c_uint32), ("pooled", C.c_void_p)]


class EmbShardConfig(C.Structure):
    _fields_ = [("n_tables", C.c_uint32), ("dim", C.c_uint32), ("depth", C.c_uint32), ("flags", C.c_uint32),
                ("tables", C.POINTER(EmbShardTable)), ("peer", C.c_void_p)]


class EmbShardStats(C.Structure):
    _fields_ = [("n_batches", C.c_uint64), ("bytes_to_peers", C.c_uint64), ("bytes_to_self", C.c_uint64),
                ("served_algorithmic_bytes", C.c_uint64), ("local_algorithmic_bytes", C.c_uint64),
                ("served_sub_bags", C.c_uint64), ("served_indices", C.c_uint64),
                ("us_host_submit", C.c_double), ("us_host_wait_counts", C.c_double), ("us_host_wait_served", C.c_double),
                ("us_kernel_route", C.c_double), ("us_kernel_local", C.c_double), ("us_kernel_serve", C.c_double),
                ("us_kernel_unroute", C.c_double), ("n_timed_batches", C.c_uint64), ("us_kernel_direct", C.c_double)]


class EmbTraceEvent(C.Structure):
    _fields_ = [("stage", C.c_uint32), ("call_id", C.c_uint32), ("start_us", C.c_double), ("stop_us", C.c_double)]


class DpuRuntimeTotals(C.Structure):
    """emb_host.h:41-48 / upmem/dputypes.py:67-79"""
    _fields_ = [("execution_time_prepare", C.c_double),
                ("execution_time_populate_copy_in", C.c_double),
                ("execution_time_copy_in", C.c_double), ("execution_time_copy_out", C.c_double),
                ("execution_time_aggregate_result", C.c_double), ("execution_time_launch", C.c_double)]


_vp, _u32, _u64, _i32, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_size_t
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every function declared in include/pimemb.h
SIGNATURES = {
    "emb_last_error": (C.c_char_p, []),
    "emb_version": (C.c_char_p, []),
    "emb_create": (C.c_int, [C.POINTER(EmbConfig), _pp]),
    "emb_destroy": (C.c_int, [_vp]),
    "emb_load_table": (C.c_int, [_vp, _u32, _u64, _u32, C.c_int, _vp, C.c_int]),
    "emb_alloc_table": (C.c_int, [_vp, _u32, _u64, _u32, C.c_int]),
    "emb_load_table_column": (C.c_int, [_vp, _u32, _u32, _vp, _u64]),
    "emb_set_hot_rows": (C.c_int, [_vp, _u32, C.POINTER(_u64), _u32]),
    "emb_learn_hot_rows": (C.c_int, [_vp, _u32, _vp, _u64, C.c_int, C.c_int, _u32, C.c_float, _vp, C.POINTER(_u32), C.POINTER(C.c_float)]),
    "emb_comm_unique_id": (C.c_int, [_vp]),
    "emb_comm_create": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _pp]),
    "emb_comm_all_to_all": (C.c_int, [_vp, _vp, C.POINTER(_u64), _vp, C.POINTER(_u64), _vp]),
    "emb_comm_destroy": (C.c_int, [_vp]),
    "emb_comm_exchange": (C.c_int, [_vp, C.POINTER(EmbCommOp), _u32, _vp]),
    "emb_comm_rank": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "emb_peer_create": (C.c_int, [_vp, C.c_char_p, C.c_int32, C.c_int32, _u64, _pp]),
    "emb_peer_alloc": (C.c_int, [_vp, _u64, _pp]),
    "emb_peer_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _pp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_i32)]),
    "emb_peer_barrier": (C.c_int, [_vp]),
    "emb_peer_destroy": (C.c_int, [_vp]),
    "emb_peer_last_words": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "emb_shard_create": (C.c_int, [_vp, _vp, C.POINTER(EmbShardConfig), _pp]),
    "emb_shard_submit": (C.c_int, [_vp, C.POINTER(EmbShardInput), _u64, _vp, C.POINTER(_u64)]),
    "emb_shard_flush": (C.c_int, [_vp]),
    "emb_shard_wait": (C.c_int, [_vp, _u64, _vp]),
    "emb_shard_lookup": (C.c_int, [_vp, C.POINTER(EmbShardInput), _u64, _vp]),
    "emb_shard_report": (C.c_int, [_vp]),
    "emb_shard_get_stats": (C.c_int, [_vp, C.POINTER(EmbShardStats), C.c_int]),
    "emb_shard_set_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "emb_shard_sent_counts": (C.c_int, [_vp, _u64, C.POINTER(_u32), _u32]),
    "emb_shard_destroy": (C.c_int, [_vp]),
    "emb_table_info": (C.c_int, [_vp, _u32, _pp, C.POINTER(_u64), C.POINTER(_u32), C.POINTER(C.c_int)]),
    "emb_lookup": (C.c_int, [_vp, _u32, _vp, _u64, _vp, _u64, _vp, C.c_int, C.c_int, _vp]),
    "emb_lookup_batched": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.c_int, C.c_int, _vp]),
    "emb_queue_create": (C.c_int, [_vp, C.c_int, C.c_int, _pp]),
    "emb_queue_add": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.POINTER(_u64)]),
    "emb_queue_add_many": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u32), _u32, C.POINTER(_u64)]),
    "emb_queue_flush": (C.c_int, [_vp, _vp, C.POINTER(_u32)]),
    "emb_queue_wait": (C.c_int, [_vp, _u64]),
    "emb_queue_destroy": (C.c_int, [_vp]),
    "emb_lookup_ranged": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u64), _u32, _vp]),
    "emb_plan_create_ranged": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u64), _u32, C.POINTER(_vp)]),
    "emb_lookup_ranged_counted": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u64), C.POINTER(_vp), _u32, _vp]),
    "emb_plan_create_ranged_counted": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u64), C.POINTER(_vp), _u32, C.POINTER(_vp)]),
    "emb_lookup_ranged_typed": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u64), C.POINTER(_vp), _u32, C.c_int, _vp]),
    "emb_plan_create_ranged_typed": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), C.POINTER(_u64), C.POINTER(_vp), _u32, C.c_int, C.POINTER(_vp)]),
    "emb_plan_create": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.c_int, _pp]),
    "emb_plan_launch": (C.c_int, [_vp, _vp]),
    "emb_plan_destroy": (C.c_int, [_vp]),
    "emb_plan_bytes": (C.c_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "emb_plan_signature": (C.c_int, [_vp, C.POINTER(_u64)]),
    "emb_plan_describe": (C.c_int, [_vp, C.c_char_p, _sz]),
    "emb_plan_time": (C.c_int, [_vp, _vp, _u32, _u32, C.POINTER(C.c_float)]),
    "emb_validate_inputs": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.c_int, C.c_int,
                                      C.POINTER(_u64)]),
    "emb_validate_inputs_on": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.c_int, C.c_int, _vp,
                                         C.POINTER(_u64)]),
    "emb_lookup_batched_checked": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.c_int, C.c_int, _vp,
                                             C.POINTER(_u64)]),
    "emb_lookup_batched_checked_deferred": (C.c_int, [_vp, C.POINTER(EmbLookupDesc), _u32, C.c_int, C.c_int, _vp]),
    "emb_check_report": (C.c_int, [_vp, C.POINTER(_u64)]),
    "emb_get_stats": (C.c_int, [_vp, C.POINTER(EmbStats)]),
    "emb_reset_stats": (C.c_int, [_vp]),
    "emb_set_stage_timing": (C.c_int, [_vp, C.c_int]),
    "emb_trace_enable": (C.c_int, [_vp, _u32]),
    "emb_trace_read": (C.c_int, [_vp, C.POINTER(EmbTraceEvent), _u32, C.POINTER(_u32)]),
    "emb_device_alloc": (C.c_int, [_vp, _sz, _pp]),
    "emb_device_free": (C.c_int, [_vp, _vp]),
    "emb_copy_to_device": (C.c_int, [_vp, _vp, _vp, _sz]),
    "emb_copy_to_host": (C.c_int, [_vp, _vp, _vp, _sz]),
    "emb_memset_device": (C.c_int, [_vp, _vp, C.c_int, _sz]),
    "emb_synchronize": (C.c_int, [_vp, _vp]),
    "emb_stream_create": (C.c_int, [_vp, _pp]),
    "emb_stream_destroy": (C.c_int, [_vp, _vp]),
    "emb_device_of": (C.c_int, [_vp, C.POINTER(_i32)]),
    "emb_route_bags_sizes": (C.c_int, [_u32, _u64, _u64, _u32, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64),
                                       C.POINTER(_u64)]),
    "emb_route_bags": (C.c_int, [_vp, C.POINTER(EmbRouteTable), _u32, _u64, _u32, _vp, _vp, _vp, _vp, _vp]),
    "emb_route_bags_typed": (C.c_int, [_vp, C.POINTER(EmbRouteTable), _u32, C.c_int, _u64, _u32, _vp, _vp, _vp, _vp, _vp]),
    "emb_unroute_bags": (C.c_int, [_vp, _vp, _vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp]),
    "emb_route_exchange_sizes": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, C.POINTER(_u64),
                                           C.POINTER(_u64)]),
    "emb_route_serve_descs": (C.c_int, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(EmbLookupDesc), C.POINTER(_u32),
                                        C.POINTER(_u64)]),
    "emb_configure": (C.c_int, [_u32, _u32, _u32, _u32]),
    "populate_mram": (_vp, [_u32, _u64, _u32, _vp, C.POINTER(DpuRuntimeTotals)]),
    "lookup": (_vp, [_pp, _pp, _pp, _vp, C.c_int64]),
    "emb_compat_engine": (_vp, []),
    "emb_compat_reset": (C.c_int, []),
}

_lib = None


class PimembError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"pimemb error {code}: {text}")
        self.code = code


def load(path: str | None = None) -> C.CDLL:
    """Load libpimemb.so (built in-tree by build.py).  Raises if it is missing: the product has no
    CPU or eager fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    try:
        # One HIP runtime per process: if torch is around, map its bundled libamdhip64.so first so
        # libpimemb's DT_NEEDED "libamdhip64.so" resolves to the same runtime (csrc/Makefile note).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = L
    return L


def check(rc: int) -> None:
    if rc != EMB_OK:
        raise PimembError(rc, load().emb_last_error().decode(errors="replace"))
