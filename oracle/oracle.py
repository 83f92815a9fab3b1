"""CPU oracle for the embedding-lookup hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is
the checker, never the thing shipped or measured as the product (see emb_oracle.c header and
DESIGN.md "Oracle").  Two independent restatements live here:

  * `c_*`  : ctypes calls into oracle/libemb_oracle.so (emb_oracle.c, plain C, sequential order)
  * `np_*` : numpy / pure-Python loops for small cases, written separately so the two can be
             checked against each other and against the golden fixtures.

Reference lines followed (relative to /root/reference):
  upmem/src/dpu/emb_dpu_lookup.c:106-116   bag bounds, last bag -> indices_len, empty bag = 0
  upmem/include/emb_host.h:207-212         (float)acc / pow(10,9), [bag][col] layout
  upmem/src/load_generator.c:40-65         row-major `index*nr_cols+t`, tolerance 1000/1e9
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libemb_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile emb_oracle.c -> libemb_oracle.so (gcc, seconds)."""
    src = os.path.join(_HERE, "emb_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libemb_oracle.so"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # spinning idle threads distort the threaded baseline
        L = C.CDLL(_LIB_PATH)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.oracle_bag_sum_f32.argtypes = [vp, u64, u32, vp, u64, vp, u64, C.c_int, vp]
        L.oracle_bag_sum_f32.restype = C.c_int
        L.oracle_bag_sum_f16.argtypes = [vp, u64, u32, vp, u64, vp, u64, C.c_int, vp]
        L.oracle_bag_sum_f16.restype = C.c_int
        L.oracle_dpu_column_i32.argtypes = [vp, u64, vp, u32, vp, u32, vp]
        L.oracle_dpu_column_i32.restype = C.c_int
        L.oracle_post_process.argtypes = [vp, u32, u32, vp]
        L.oracle_post_process.restype = None
        L.oracle_lookup_fixed32.argtypes = [vp, u64, u32, vp, u32, vp, u32, vp]
        L.oracle_lookup_fixed32.restype = C.c_int
        L.oracle_validate_result.argtypes = [vp, u32, vp, u32, vp, u32, vp]
        L.oracle_validate_result.restype = C.c_uint64
        L.oracle_lookup_tables_f32.argtypes = [u32, vp, vp, u32, vp, vp, vp, vp, C.c_int, vp]
        L.oracle_lookup_tables_f32.restype = C.c_int
        L.oracle_lookup_tables_f32_mt.argtypes = [u32, vp, vp, u32, vp, vp, vp, vp, C.c_int, vp, C.c_int]
        L.oracle_lookup_tables_f32_mt.restype = C.c_int
        _lib = L
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def _idx_kind(indices: np.ndarray, offsets: np.ndarray) -> int:
    if indices.dtype == np.int64 and offsets.dtype == np.int64:
        return 1
    if indices.dtype == np.uint32 and offsets.dtype == np.uint32:
        return 0
    raise TypeError("indices/offsets must both be uint32 (reference ABI) or both int64 (torch)")


# ----------------------------------------------------------------------------- C restatement
def c_bag_sum(table: np.ndarray, indices: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """EmbeddingBag(sum): table [N,D] fp32 or fp16 -> pooled [B,D] fp32, sequential order."""
    table = np.ascontiguousarray(table)
    indices = np.ascontiguousarray(indices)
    offsets = np.ascontiguousarray(offsets)
    is64 = _idx_kind(indices, offsets)
    n_rows, dim = table.shape
    n_bags = offsets.shape[0]
    out = np.empty((n_bags, dim), dtype=np.float32)
    if table.dtype == np.float32:
        fn = lib().oracle_bag_sum_f32
    elif table.dtype == np.float16:
        fn = lib().oracle_bag_sum_f16
    else:
        raise TypeError(table.dtype)
    rc = fn(_p(table), n_rows, dim, _p(indices), indices.shape[0], _p(offsets), n_bags, is64, _p(out))
    if rc:
        raise IndexError("index out of range")
    return out


def c_lookup_fixed32(table_i32: np.ndarray, indices: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """Reference arithmetic: int32 row-major table [N,C] -> float32 [B,C] (= sum/1e9)."""
    table_i32 = np.ascontiguousarray(table_i32, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    n_rows, cols = table_i32.shape
    out = np.empty((offsets.shape[0], cols), dtype=np.float32)
    rc = lib().oracle_lookup_fixed32(_p(table_i32), n_rows, cols, _p(indices), indices.shape[0],
                                     _p(offsets), offsets.shape[0], _p(out))
    if rc:
        raise IndexError("index out of range")
    return out


def c_dpu_column(column_i32: np.ndarray, indices: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    column_i32 = np.ascontiguousarray(column_i32, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    res = np.empty(offsets.shape[0], dtype=np.int32)
    rc = lib().oracle_dpu_column_i32(_p(column_i32), column_i32.shape[0], _p(indices),
                                     indices.shape[0], _p(offsets), offsets.shape[0], _p(res))
    if rc:
        raise IndexError("index out of range")
    return res


def c_post_process(tmp_col_major: np.ndarray) -> np.ndarray:
    tmp = np.ascontiguousarray(tmp_col_major, dtype=np.int32)
    cols, nb = tmp.shape
    out = np.empty((nb, cols), dtype=np.float32)
    lib().oracle_post_process(_p(tmp), cols, nb, _p(out))
    return out


def c_validate_result(table_i32, indices, offsets, results) -> int:
    table_i32 = np.ascontiguousarray(table_i32, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    results = np.ascontiguousarray(results, dtype=np.float32)
    return int(lib().oracle_validate_result(_p(table_i32), table_i32.shape[1], _p(indices),
                                            indices.shape[0], _p(offsets), offsets.shape[0],
                                            _p(results)))


def c_lookup_tables(tables, indices, offsets, threads: int = 1, outs=None):
    """Multi-table fp32 lookup with the reference's per-table pointer arrays (emb_host.h:234).
    threads > 1: the bags of every table split over that many OpenMP threads (same bits).
    outs: optional preallocated float32 [B_t, dim] arrays to write into (timing loops reuse them)."""
    T = len(tables)
    tables = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
    indices = [np.ascontiguousarray(i) for i in indices]
    offsets = [np.ascontiguousarray(o) for o in offsets]
    is64 = _idx_kind(indices[0], offsets[0])
    dim = tables[0].shape[1]
    if outs is None:
        outs = [np.empty((o.shape[0], dim), dtype=np.float32) for o in offsets]

    def parr(arrs):
        return (C.c_void_p * T)(*[a.ctypes.data for a in arrs])

    def u64arr(vals):
        return (C.c_uint64 * T)(*vals)

    a = (T, parr(tables), u64arr([t.shape[0] for t in tables]), dim, parr(indices),
         u64arr([i.shape[0] for i in indices]), parr(offsets), u64arr([o.shape[0] for o in offsets]),
         is64, parr(outs))
    rc = lib().oracle_lookup_tables_f32(*a) if threads <= 1 else lib().oracle_lookup_tables_f32_mt(*a, int(threads))
    if rc:
        raise IndexError("index out of range")
    return outs


# ----------------------------------------------------------------------------- numpy restatement
def bag_ends(offsets: np.ndarray, n_idx: int) -> np.ndarray:
    """end[b] = offsets[b+1], last bag ends at n_idx (emb_dpu_lookup.c:109-110)."""
    off = offsets.astype(np.int64)
    return np.concatenate([off[1:], np.array([n_idx], dtype=np.int64)])


def np_bag_sum(table: np.ndarray, indices: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """Sequential-order fp32 sum, vectorised over bags: step j adds every bag's j-th row."""
    n_bags = offsets.shape[0]
    dim = table.shape[1]
    out = np.zeros((n_bags, dim), dtype=np.float32)
    if n_bags == 0:
        return out
    start = offsets.astype(np.int64)
    end = bag_ends(offsets, indices.shape[0])
    lens = end - start
    idx = indices.astype(np.int64)
    for j in range(int(lens.max(initial=0))):
        live = np.nonzero(lens > j)[0]
        rows = idx[start[live] + j]
        out[live] = out[live] + table[rows].astype(np.float32)
    return out


def py_bag_sum(table, indices, offsets):
    """Pure-Python triple loop (tiny cases only): the most literal restatement."""
    n_bags = len(offsets)
    dim = table.shape[1]
    out = np.zeros((n_bags, dim), dtype=np.float32)
    for b in range(n_bags):
        p = int(offsets[b])
        e = int(offsets[b + 1]) if b + 1 < n_bags else len(indices)
        while p < e:
            r = int(indices[p])
            for d in range(dim):
                out[b, d] = np.float32(out[b, d]) + np.float32(table[r, d])
            p += 1
    return out


def np_lookup_fixed32(table_i32: np.ndarray, indices: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """int32 wrap-around sums then (float)acc / 1e9, numpy restatement."""
    n_bags = offsets.shape[0]
    cols = table_i32.shape[1]
    acc = np.zeros((n_bags, cols), dtype=np.uint32)
    start = offsets.astype(np.int64)
    if n_bags:
        start[0] = 0          # emb_dpu_lookup.c:60-63: tasklet 0 starts its first bag at index 0, not at offsets[0]
    end = bag_ends(offsets, indices.shape[0])
    lens = end - start
    idx = indices.astype(np.int64)
    tab = table_i32.view(np.uint32)
    for j in range(int(lens.max(initial=0))):
        live = np.nonzero(lens > j)[0]
        acc[live] = acc[live] + tab[idx[start[live] + j]]  # uint32 wraps like the DPU's int32
    as_f32 = acc.view(np.int32).astype(np.float32)
    return (as_f32.astype(np.float64) / 1e9).astype(np.float32)
