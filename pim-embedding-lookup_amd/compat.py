"""Python driver for the two reference-compatible C entry points, shaped like the reference's own
ctypes harness (upmem/c_test.py:36-74: `populate` then `lookup` on a CDLL handle) but with the
current C signatures (emb_host.h:136, :234) and assertions the reference script never had."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _l


def configure(nr_tables: int, nr_cols: int, max_nr_batches: int, max_indices_per_batch: int) -> None:
    """Runtime stand-in for -DNR_TABLES -DNR_COLS -DMAX_NR_BATCHES -DMAX_INDICES_PER_BATCH
    (upmem/Makefile:69-81)."""
    _l.check(_l.load().emb_configure(nr_tables, nr_cols, max_nr_batches, max_indices_per_batch))


def reset() -> None:
    _l.load().emb_compat_reset()


def populate(tables_i32, runtimes: _l.DpuRuntimeTotals | None = None) -> int:
    """tables_i32: list of row-major int32 [nr_rows, NR_COLS] arrays.  Splits each into columns
    (what alloc_buffers does, emb_host.h:116-118) and calls populate_mram once per (table, col)
    exactly as the PyTorch fork would.  Returns the opaque dpu_set handle."""
    L = _l.load()
    handle = None
    rt = C.byref(runtimes) if runtimes is not None else None
    for t, tab in enumerate(tables_i32):
        tab = np.ascontiguousarray(tab, dtype=np.int32)
        for col in range(tab.shape[1]):
            column = np.ascontiguousarray(tab[:, col])
            handle = L.populate_mram(t, tab.shape[0], col, column.ctypes.data, rt)
            if not handle:
                raise _l.PimembError(_l.EMB_ERR_INVALID, L.emb_last_error().decode())
    return handle


def lookup(handle: int, indices, offsets, nr_cols: int, latency_print: int = 0):
    """indices[t]: uint32[MAX_INDICES_PER_BATCH*MAX_NR_BATCHES], offsets[t]: uint32[MAX_NR_BATCHES].
    Returns final_results as a list of float32 [MAX_NR_BATCHES, NR_COLS] arrays."""
    L = _l.load()
    T = len(indices)
    idx = [np.ascontiguousarray(i, dtype=np.uint32) for i in indices]
    off = [np.ascontiguousarray(o, dtype=np.uint32) for o in offsets]
    res = [np.full((o.shape[0], nr_cols), np.nan, dtype=np.float32) for o in off]
    pi = (C.c_void_p * T)(*[a.ctypes.data for a in idx])
    po = (C.c_void_p * T)(*[a.ctypes.data for a in off])
    pr = (C.c_void_p * T)(*[a.ctypes.data for a in res])
    ret = L.lookup(pi, po, pr, handle, latency_print)
    assert ret is None  # emb_host.h:403: always NULL
    return res
