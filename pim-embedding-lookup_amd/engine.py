"""Host-side mirror of the reference's operator surface over libpimemb.so.

The reference exposes the hot path as two C calls (upmem/include/emb_host.h:136 `populate_mram`,
:234 `lookup`) that a PyTorch fork dispatches into from `nn.EmbeddingBag(mode="sum")`; the Python
here is the equivalent thin layer: it owns no arithmetic, only pointers, shapes and lifetimes.
Every lookup ends in `emb_lookup_batched` / `emb_plan_launch`, i.e. the fused HIP kernel.

Buffers may be
  * numpy arrays            -> EMB_MEM_HOST  (the engine copies in/out, like the reference does)
  * torch CUDA tensors      -> EMB_MEM_DEVICE (zero-copy; enqueued on torch's current stream)
  * `DeviceBuffer` objects  -> EMB_MEM_DEVICE (HBM owned through the C ABI, no torch involved)
"""
from __future__ import annotations

import ctypes as C
import struct
import threading
from typing import Sequence

import numpy as np

from . import lib as _l

_NP_TABLE_DTYPES = {np.dtype(np.float32): _l.EMB_F32, np.dtype(np.float16): _l.EMB_F16,
                    np.dtype(np.int32): _l.EMB_FIXED32}


_MARSHAL = [False]      # False: not looked for yet; None: not built; else the _pimemb_marshal module


def _marshal():
    """The CPython helper that unpacks lists of torch tensors into emb_lookup_desc records in one call
    (csrc/pimemb_torch_marshal.cpp, built next to libpimemb.so).  None when it has not been built: the same unpacking is
    then done in Python, ~1 us per table and call slower.  PIMEMB_NO_MARSHAL=1 switches it off (tests, probes)."""
    m = _MARSHAL[0]
    if m is False:
        import importlib.util
        import os
        m = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_pimemb_marshal.so")
        if os.path.exists(path) and os.environ.get("PIMEMB_NO_MARSHAL") != "1":
            import torch  # noqa: F401  (the helper links against torch's libraries: they must be mapped first)
            try:
                spec = importlib.util.spec_from_file_location("_pimemb_marshal", path)
                m = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(m)
            except (ImportError, OSError) as ex:     # built against another Python / torch: the Python path does the same job
                import warnings
                warnings.warn(f"_pimemb_marshal.so could not be loaded ({ex}); tensor lists are unpacked in Python")
                m = None
        _MARSHAL[0] = m
    return m


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _index_type_of(dtype) -> int:
    name = str(dtype).replace("torch.", "")
    if name in ("uint32", "int32"):  # int32 bits are read as uint32 (indices are non-negative)
        return _l.EMB_IDX_U32
    if name == "int64":
        return _l.EMB_IDX_I64
    raise TypeError(f"indices/offsets must be uint32/int32 or int64, got {dtype}")


class DeviceBuffer:
    """A piece of HBM allocated through the C ABI (emb_device_alloc)."""

    def __init__(self, engine: "EmbeddingEngine", nbytes: int, dtype=np.uint8, shape=None):
        self.engine = engine
        self.nbytes = int(nbytes)
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else (self.nbytes // self.dtype.itemsize,)
        p = C.c_void_p()
        _l.check(engine._L.emb_device_alloc(engine._h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, engine: "EmbeddingEngine", a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(engine, a.nbytes, a.dtype, a.shape)
        if a.nbytes:
            _l.check(engine._L.emb_copy_to_device(engine._h, buf.ptr, a.ctypes.data, a.nbytes))
        return buf

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            _l.check(self.engine._L.emb_copy_to_host(self.engine._h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def fill(self, byte: int = 0) -> None:
        _l.check(self.engine._L.emb_memset_device(self.engine._h, self.ptr, byte, self.nbytes))

    def free(self) -> None:
        if self.ptr and self.engine._h:
            self.engine._L.emb_device_free(self.engine._h, self.ptr)
        self.ptr = None

    def __len__(self):
        return self.shape[0]


class _Arg:
    """Pointer + placement of one caller buffer."""
    __slots__ = ("ptr", "space", "n", "dtype", "keep")

    def __init__(self, x):
        if x is None:
            self.ptr, self.space, self.n, self.dtype, self.keep = None, None, 0, None, None
        elif isinstance(x, DeviceBuffer):
            self.ptr, self.space, self.n, self.dtype, self.keep = x.ptr, _l.EMB_MEM_DEVICE, x.shape[0], x.dtype, x
        elif _is_torch(x):
            if not x.is_contiguous():
                raise ValueError("torch tensors passed to the engine must be contiguous")
            space = _l.EMB_MEM_DEVICE if x.is_cuda else _l.EMB_MEM_HOST
            self.ptr, self.space, self.n, self.dtype, self.keep = x.data_ptr(), space, x.shape[0], x.dtype, x
        else:
            a = np.ascontiguousarray(x)
            self.ptr, self.space, self.n, self.dtype, self.keep = a.ctypes.data, _l.EMB_MEM_HOST, a.shape[0], a.dtype, a


def _current_stream_for(*objs) -> int | None:
    for o in objs:
        if _is_torch(o) and o.is_cuda:
            import torch
            return torch.cuda.current_stream(o.device).cuda_stream
    return None


class Plan:
    """A prepared multi-table lookup over device buffers: one kernel enqueue per launch."""

    def __init__(self, engine: "EmbeddingEngine", handle: int, outputs, keep):
        self.engine, self._p, self.outputs, self._keep = engine, handle, outputs, keep

    def launch(self, stream: int | None = None) -> None:
        _l.check(self.engine._L.emb_plan_launch(self._p, stream))

    def bytes(self) -> tuple[int, int, int]:
        b, nb, ni = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _l.check(self.engine._L.emb_plan_bytes(self._p, C.byref(b), C.byref(nb), C.byref(ni)))
        return b.value, nb.value, ni.value

    def time_us(self, warmup: int = 5, iters: int = 20, stream: int | None = None) -> float:
        """Mean device time of one launch, HIP events on the launch stream (emb_plan_time)."""
        us = C.c_float()
        _l.check(self.engine._L.emb_plan_time(self._p, stream, warmup, iters, C.byref(us)))
        return us.value

    def signature(self) -> int:
        """Hash of what the plan's launches ARE (kernel kind, grid, XCD map, per-descriptor counts; no addresses)."""
        v = C.c_uint64()
        _l.check(self.engine._L.emb_plan_signature(self._p, C.byref(v)))
        return v.value

    def describe(self) -> list[dict]:
        """One dict per kernel launch of the plan: kind, dtype, itype, lanes_per_row, chunks, scalar_lanes, anydim_vec, ranged,
        descs, grid, bags_per_tile (emb_plan_describe)."""
        buf = C.create_string_buffer(4096)
        _l.check(self.engine._L.emb_plan_describe(self._p, buf, len(buf)))
        return [{k: int(v) for k, v in (kv.split("=") for kv in rec.split())} for rec in buf.value.decode().split(";") if rec]

    def destroy(self) -> None:
        if self._p:
            self.engine._L.emb_plan_destroy(self._p)
            self._p = None


class RequestQueue:
    """R pending SMALL lookups -> ONE launch (emb_queue_* in pimemb.h): the reference's serving shapes -- mini-batch 1
    (README.md:6), 32 with MAX_NR_BATCHES = 512 (upmem/run.sh:40-45,119) -- are one ~3.5-us launch each on the GPU, so a
    server that issues them back to back is launch-bound.  add() may be called from any number of threads (the queue's own
    short lock, never the engine's); flush() runs everything pending as one fused launch; every request keeps its own
    buffers and gets the bits a lookup of its own would give.

    space: EMB_MEM_HOST   add(table_ids, indices, offsets, outs) with numpy arrays; wait(ticket) blocks until the request's
                           rows are in `outs`
           EMB_MEM_DEVICE torch CUDA tensors / DeviceBuffer; results complete in stream order behind the flush"""

    def __init__(self, engine: "EmbeddingEngine", itype: int = _l.EMB_IDX_U32, space: int = _l.EMB_MEM_HOST):
        self.engine, self._L, self.itype, self.space = engine, engine._L, itype, space
        h = C.c_void_p()
        _l.check(self._L.emb_queue_create(engine._h, itype, space, C.byref(h)))
        self._h = h
        self._keep = {}

    def descriptors(self, table_ids, indices, offsets, outs, fixed_pooling=0):
        """The request as a reusable descriptor array (a server that re-posts the same buffers builds it once)."""
        arr, n, itype, space, results, keep = self.engine._descs(table_ids, indices, offsets, outs, fixed_pooling)
        if itype != self.itype or space != self.space:
            raise TypeError("request buffers do not match the queue's index width / memory space")
        return arr, n, results, keep

    def add(self, table_ids, indices, offsets, outs, fixed_pooling=0) -> int:
        arr, n, _res, keep = self.descriptors(table_ids, indices, offsets, outs, fixed_pooling)
        t = self.add_descriptors(arr, n)
        self._keep[t] = keep
        return t

    def add_descriptors(self, arr, n: int) -> int:
        t = C.c_uint64()
        _l.check(self._L.emb_queue_add(self._h, arr, n, C.byref(t)))
        return t.value

    def add_many(self, arr, counts) -> int:
        """Several requests in one call: `arr` holds their descriptors back to back (an EmbLookupDesc array), `counts` a
        ctypes uint32 array of descriptors per request.  Returns the first ticket (the others follow consecutively)."""
        t = C.c_uint64()
        _l.check(self._L.emb_queue_add_many(self._h, arr, counts, len(counts), C.byref(t)))
        return t.value

    def flush(self, stream: int | None = None) -> int:
        n = C.c_uint32()
        _l.check(self._L.emb_queue_flush(self._h, stream, C.byref(n)))
        return n.value

    def wait(self, ticket: int) -> None:
        _l.check(self._L.emb_queue_wait(self._h, ticket))
        self._keep.pop(ticket, None)

    def close(self) -> None:
        if self._h:
            self._L.emb_queue_destroy(self._h)
            self._h = None


class NativeExchange:
    """All-to-all of byte ranges issued straight to RCCL on a HIP stream (emb_comm_* in pimemb.h).
    Bootstrap: rank 0 draws the RCCL unique id, `broadcast` (a callable bytes -> bytes that returns rank
    0's value on every rank, e.g. built on torch.distributed.broadcast_object_list) spreads it."""

    def __init__(self, engine: "EmbeddingEngine", rank: int, world: int, broadcast):
        self._L = engine._L
        buf = C.create_string_buffer(128)
        if rank == 0:
            _l.check(self._L.emb_comm_unique_id(buf))
        uid = broadcast(buf.raw)
        self._h = C.c_void_p()
        _l.check(self._L.emb_comm_create(engine._h, C.create_string_buffer(uid, 128), rank, world, C.byref(self._h)))
        self.world = world
        self._offs = {}

    def offsets(self, offsets):
        """world+1 byte offsets as a reusable handle (pass it to all_to_all instead of a list)."""
        return self._arr(offsets)

    def _arr(self, offsets):
        if isinstance(offsets, C.Array):
            return offsets
        key = tuple(int(x) for x in offsets)
        a = self._offs.get(key)
        if a is None:
            if len(key) != self.world + 1:
                raise ValueError("need world+1 byte offsets")
            a = self._offs[key] = (C.c_uint64 * len(key))(*key)
        return a

    def all_to_all(self, send_ptr: int, send_off, recv_ptr: int, recv_off, stream: int | None = None) -> None:
        _l.check(self._L.emb_comm_all_to_all(self._h, send_ptr, self._arr(send_off), recv_ptr, self._arr(recv_off), stream))

    def close(self) -> None:
        if self._h:
            self._L.emb_comm_destroy(self._h)
            self._h = None


class EmbeddingEngine:
    """One engine per GPU: tables resident in HBM, lookups as fused HIP launches."""

    def __init__(self, device: int = -1, max_tables: int = 1024, lib_path: str | None = None,
                 check_inputs: bool | str = False):
        # lib_path: another build of libpimemb.so (e.g. the -DPIMEMB_CLAMP_INPUTS=1 flavour)
        # check_inputs: EMB_FLAG_CHECK_INPUTS -- every plan-less lookup validates its indices / offsets first;
        #   "deferred": + EMB_FLAG_DEFER_CHECK -- a device-pointer call does not wait for its verdict (a later call, check_report()
        #   or close() raises it; the refused call's lookup is kept from running on the GPU all the same)
        self._L = _l.load(lib_path)
        flags = _l.EMB_FLAG_CHECK_INPUTS if check_inputs else 0
        if check_inputs == "deferred":
            flags |= _l.EMB_FLAG_DEFER_CHECK
        cfg = _l.EmbConfig(device, max_tables, flags)
        h = C.c_void_p()
        _l.check(self._L.emb_create(C.byref(cfg), C.byref(h)))
        self._h = h.value
        self._tables: dict[int, tuple[int, int, int]] = {}  # id -> (nr_rows, dim, dtype)
        # Scratch descriptor arrays are PER THREAD: the C call that reads them releases the GIL, so another thread packing
        # its own call into a shared array would be read half-way (lookups may be issued from several threads, pimemb.h).
        self._tls = threading.local()                       # .bufs: n descriptors -> (buffer, typed pointer, address)
                                                            # .stacked: table ids -> descriptor array of lookup_stacked
        self._plan_lock = threading.Lock()                  # a cached plan is not destroyed while another thread launches it
        # plan cache of per-table-list calls (lookup_batched over fresh lists of torch CUDA tensors, as an apply_emb loop
        # makes them): call signature -> prepared plan.  See _lookup_batched_cuda.
        self._plan_cache: dict[tuple, list] = {}            # key -> [Plan, last use]
        self._plan_seen: dict[int, int] = {}                # hash(key) of signatures seen once (a plan is built on the second sighting)
        self._plan_clock = 0                                # cacheable calls so far (hits and misses)
        self._plan_last_evict = -(1 << 30)
        self.plan_cache_size = 16                           # 0 switches the cache off
        # An engine created with check_inputs=True promises that every plan-less lookup validates first; a cached plan
        # launch (emb_plan_launch) never validates, so such an engine never caches plans (the third call with
        # the same buffer addresses used to skip the check).
        self._check_inputs = bool(check_inputs)
        if self._check_inputs:
            self.plan_cache_size = 0
        self._same_dim: dict[tuple, int] = {}               # table ids -> their common dim (0: dims differ)
        self.plan_cache_hits = 0

    # ---- tables (populate_mram's job, emb_host.h:136) ------------------------------------------
    def load_table(self, table_id: int, rows, dtype: int | None = None) -> None:
        """rows: [nr_rows, dim] numpy float32/float16/int32 (int32 = x1e9 fixed point) or a torch
        tensor (CPU or CUDA) of those dtypes."""
        if _is_torch(rows):
            import torch
            tmap = {torch.float32: _l.EMB_F32, torch.float16: _l.EMB_F16, torch.int32: _l.EMB_FIXED32}
            dt = tmap[rows.dtype] if dtype is None else dtype
            if not rows.is_contiguous():
                rows = rows.contiguous()
            space = _l.EMB_MEM_DEVICE if rows.is_cuda else _l.EMB_MEM_HOST
            if rows.is_cuda:
                torch.cuda.current_stream(rows.device).synchronize()
            ptr, shape = rows.data_ptr(), tuple(rows.shape)
        else:
            rows = np.ascontiguousarray(rows)
            dt = _NP_TABLE_DTYPES[rows.dtype] if dtype is None else dtype
            space, ptr, shape = _l.EMB_MEM_HOST, rows.ctypes.data, rows.shape
        if len(shape) != 2:
            raise ValueError("table must be 2-D [nr_rows, dim]")
        _l.check(self._L.emb_load_table(self._h, table_id, shape[0], shape[1], dt, ptr, space))
        self._tables[table_id] = (shape[0], shape[1], dt)
        self._same_dim.clear()

    def alloc_table(self, table_id: int, nr_rows: int, dim: int, dtype: int) -> None:
        _l.check(self._L.emb_alloc_table(self._h, table_id, nr_rows, dim, dtype))
        self._tables[table_id] = (nr_rows, dim, dtype)
        self._same_dim.clear()

    def load_table_column(self, table_id: int, col: int, column: np.ndarray) -> None:
        column = np.ascontiguousarray(column, dtype=np.int32)
        _l.check(self._L.emb_load_table_column(self._h, table_id, col, column.ctypes.data, column.shape[0]))

    def set_hot_rows(self, table_id: int, row_ids) -> None:
        """Hint: the table's hottest rows, hottest first (empty clears).  Pooled launches then serve
        them from LDS (emb_set_hot_rows in pimemb.h); results do not change."""
        ids = np.ascontiguousarray(row_ids, dtype=np.uint64)
        _l.check(self._L.emb_set_hot_rows(self._h, table_id, ids.ctypes.data_as(C.POINTER(C.c_uint64)), ids.shape[0]))

    def learn_hot_rows(self, table_id: int, indices, max_rows: int = 100, min_share: float = 0.05, stream: int | None = None):
        """The engine picks the table's hot rows itself from one batch's indices (torch CUDA tensor, DeviceBuffer or numpy array;
        uint32 bits / int32 or int64): emb_learn_hot_rows counts a sample on the host side of the library and stages the
        `max_rows` most frequent ids -- or clears the set when they cover less than `min_share` of the sample.  Returns
        (rows staged, share of the sample the max_rows most frequent ids cover)."""
        a = _Arg(indices)
        itype = _index_type_of(a.dtype)
        if stream is None and a.space == _l.EMB_MEM_DEVICE:
            stream = _current_stream_for(indices)
        n, sh = C.c_uint32(), C.c_float()
        _l.check(self._L.emb_learn_hot_rows(self._h, table_id, a.ptr, a.n, itype, a.space, int(max_rows), float(min_share), stream,
                                            C.byref(n), C.byref(sh)))
        return n.value, sh.value

    def table_tensor(self, table_id: int):
        """Zero-copy torch view of a table's rows in HBM ([nr_rows, dim], the table's dtype) -- for saving a
        checkpoint or inspecting weights.  Valid until the table is reloaded or the engine closed; writing
        through it changes what lookups return (and leaves any hot-row copy stale)."""
        import torch
        ptr, n, d, dt = self.table_info(table_id)
        typestr = {_l.EMB_F32: "<f4", _l.EMB_F16: "<f2", _l.EMB_FIXED32: "<i4"}[dt]

        class _View:
            __cuda_array_interface__ = {"shape": (n, d), "typestr": typestr, "data": (ptr, False), "version": 2,
                                        "strides": None}
        return torch.as_tensor(_View(), device=torch.device("cuda", self.device))

    def table_info(self, table_id: int):
        ptr, n, d, dt = C.c_void_p(), C.c_uint64(), C.c_uint32(), C.c_int()
        _l.check(self._L.emb_table_info(self._h, table_id, C.byref(ptr), C.byref(n), C.byref(d), C.byref(dt)))
        return ptr.value, n.value, d.value, dt.value

    # ---- lookups (lookup's job, emb_host.h:234) ------------------------------------------------
    def _alloc_out(self, like, n_bags: int, dim: int):
        if isinstance(like, DeviceBuffer):
            return DeviceBuffer(self, n_bags * dim * 4, np.float32, (n_bags, dim))
        if _is_torch(like):
            import torch
            return torch.empty((n_bags, dim), dtype=torch.float32, device=like.device)
        return np.empty((n_bags, dim), dtype=np.float32)

    def _descs(self, table_ids, indices, offsets, outs, fixed_pooling, want_outputs=True):
        """ctypes descriptor array for a batched call.  want_outputs=False (validation) leaves
        `pooled` NULL and allocates nothing."""
        n = len(table_ids)
        if not (len(indices) == n and len(offsets) == n):
            raise ValueError("table_ids, indices and offsets must have equal length")
        arr = (_l.EmbLookupDesc * n)()
        keep, results = [], []
        itype = space = None
        for i, t in enumerate(table_ids):
            if t not in self._tables:
                raise KeyError(f"table {t} is not loaded")
            ia, oa = _Arg(indices[i]), _Arg(offsets[i])
            L = 0
            if oa.ptr is None:
                L = int(fixed_pooling[i] if isinstance(fixed_pooling, (list, tuple)) else fixed_pooling or 0)
                if L <= 0:
                    raise ValueError("offsets=None needs fixed_pooling > 0")
                n_bags = ia.n // L
            else:
                n_bags = oa.n
                if _index_type_of(oa.dtype) != _index_type_of(ia.dtype) or oa.space != ia.space:
                    raise TypeError("indices and offsets must share dtype width and placement")
            it = _index_type_of(ia.dtype)
            if itype is None:
                itype, space = it, ia.space
            elif it != itype or ia.space != space:
                raise TypeError("all tables of one batched lookup must share index width and placement")
            out_ptr = None
            if want_outputs:
                dim = self._tables[t][1]
                out = outs[i] if outs is not None else self._alloc_out(indices[i], n_bags, dim)
                ua = _Arg(out)
                if ua.space != space:
                    raise TypeError("output placement must match the inputs")
                out_ptr = ua.ptr
                keep.append(ua.keep)
                results.append(out)
            arr[i] = _l.EmbLookupDesc(t, L, ia.ptr, oa.ptr, ia.n, n_bags, out_ptr)
            keep += [ia.keep, oa.keep]
        return arr, n, itype, space, results, keep

    _DESC = struct.Struct("<IIQQQQQ")    # emb_lookup_desc: table_id, fixed_pooling, indices, offsets, n_indices, n_bags, pooled

    def _drop_plans(self) -> None:
        with self._plan_lock:
            for plan, _ in self._plan_cache.values():
                plan.destroy()
            self._plan_cache.clear()
            self._plan_seen.clear()

    def _desc_slot(self, n: int):
        """This thread's scratch array of n emb_lookup_desc records: (buffer, typed pointer, address)."""
        try:
            bufs = self._tls.bufs
        except AttributeError:
            bufs = self._tls.bufs = {}
        slot = bufs.get(n)
        if slot is None:
            raw = C.create_string_buffer(self._DESC.size * n)
            slot = bufs[n] = (raw, C.cast(raw, C.POINTER(_l.EmbLookupDesc)), C.addressof(raw))
        return slot

    def _launch_cached(self, key, stream) -> bool:
        """Launch the cached plan of this call signature, if there is one (under the plan lock: see _plan_lock)."""
        if self._check_inputs:               # a checked engine never launches unvalidated (whatever plan_cache_size says)
            return False
        with self._plan_lock:
            self._plan_clock += 1
            ent = self._plan_cache.get(key)
            if ent is None:
                return False
            if self._L.emb_plan_launch(ent[0]._p, stream) == _l.EMB_OK:
                ent[1] = self._plan_clock
                self.plan_cache_hits += 1
                return True
            ent[0].destroy()                 # stale (a table was reloaded): forget it, make the ordinary call
            del self._plan_cache[key]
            return False

    def _raise_range(self, bad):
        raise IndexError(f"{bad} index / offset value(s) out of range for the embedding table(s) "
                         "(checked on the GPU before anything was launched)")

    def _checked_call(self, arr, n, itype, space, stream, check) -> None:
        """One checked plan-less call.  check = True / "sync": the verdict is waited for inside the call (IndexError before
        anything is launched, like nn.EmbeddingBag).  check = "deferred" (device buffers): the call does not wait -- a finding
        disarms the call's lookup on the GPU (outputs untouched) and is raised as IndexError by a LATER checked call of this
        engine, by check_report() or by close(), naming the call it belongs to (emb_lookup_batched_checked_deferred)."""
        if check == "deferred" and space == _l.EMB_MEM_DEVICE:
            rc = self._L.emb_lookup_batched_checked_deferred(self._h, arr, n, itype, space, stream)
            if rc == _l.EMB_ERR_RANGE:
                raise IndexError(self._L.emb_last_error().decode(errors="replace"))
            _l.check(rc)
            return
        bad = C.c_uint64()
        rc = self._L.emb_lookup_batched_checked(self._h, arr, n, itype, space, stream, C.byref(bad))
        if rc == _l.EMB_ERR_RANGE:
            if "EARLIER" in self._L.emb_last_error().decode(errors="replace"):     # a deferred call's finding, met by this one
                raise IndexError(self._L.emb_last_error().decode(errors="replace"))
            self._raise_range(bad.value)
        _l.check(rc)

    def check_report(self) -> None:
        """Wait for the verdicts of every check="deferred" call made so far; IndexError if one of them was refused."""
        bad = C.c_uint64()
        rc = self._L.emb_check_report(self._h, C.byref(bad))
        if rc == _l.EMB_ERR_RANGE:
            raise IndexError(self._L.emb_last_error().decode(errors="replace"))
        _l.check(rc)

    def _lookup_batched_marshal(self, tb, table_ids, indices, offsets, outs, stream, check):
        """_lookup_batched_cuda with the tensor lists unpacked by the C helper (one call instead of ~8 attribute
        reads / method calls per table), same checks, same plan cache -- keyed on the descriptor bytes themselves, which
        ARE the call's signature (table ids, every address and length)."""
        import torch
        n = len(table_ids)
        _raw, buf_ptr, addr = self._desc_slot(n)
        r = tb.pack(addr, table_ids, indices, offsets, outs, self._tables)
        if r is None:
            return None
        itype, _dev, d0, nbs, key, cur_stream = r
        if outs is None:
            if not d0:             # tables of different dims: one allocation per table (the Python path does that)
                return None
            one = torch.empty((sum(nbs), d0), dtype=torch.float32, device=indices[0].device)
            key = tb.fill_pooled(addr, n, one.data_ptr(), d0)
            res = list(one.split(nbs))
        else:
            res = list(outs)
        if stream is None:
            stream = cur_stream
        if check:
            self._checked_call(buf_ptr, n, itype, _l.EMB_MEM_DEVICE, stream, check)
            return res
        if self.plan_cache_size:
            key = (key, itype)
            if self._launch_cached(key, stream):
                return res
        _l.check(self._L.emb_lookup_batched(self._h, buf_ptr, n, itype, _l.EMB_MEM_DEVICE, stream))
        if self.plan_cache_size:
            self._remember_call(key, buf_ptr, n, itype)
        return res

    def _lookup_batched_cuda(self, table_ids, indices, offsets, outs, stream, check=False):
        """Fast path of lookup_batched for torch CUDA tensors with explicit offsets -- what an
        `apply_emb` loop hands over, fresh tensors every batch.  Same C call as the general path, but
        the descriptors are packed without per-buffer helper objects and the outputs of tables that
        share a dim come from ONE allocation (views).  Returns None if the arguments do not qualify.
        26 tables: ~150 us -> ~40 us of Python per call.

        Plan cache.  A training / serving loop hands over NEW tensor objects every batch, but torch's caching
        allocator gives them the addresses of the previous batch's, so the call is byte for byte the one made before:
        same tables, same buffer addresses, same lengths.  Its signature -- table ids, data_ptr / len / dtype of every
        index and offset tensor, the output addresses -- is looked up in a small cache; the second sighting builds a
        prepared plan (emb_plan_create), later ones are ONE emb_plan_launch: no per-table Python, no descriptor
        resolution.  A plan holds addresses, never values: new indices in the same buffers are read by the launch.
        Reloaded tables make the plan stale (emb_plan_launch refuses it; the entry is dropped); a miss is the ordinary
        transient call.  Checked calls (check=True) always take the transient path."""
        import torch
        tb = _marshal()
        if tb is not None and type(table_ids) is list and type(indices) is list and type(offsets) is list:
            res = self._lookup_batched_marshal(tb, table_ids, indices, offsets, outs, stream, check)
            if res is not None:
                return res
        n = len(table_ids)
        i0 = indices[0]
        if type(i0) is not torch.Tensor or not i0.is_cuda or len(indices) != n or len(offsets) != n:
            return None
        dt, dev, tables = i0.dtype, i0.device, self._tables
        if dt is torch.int64:
            itype = _l.EMB_IDX_I64
        elif dt is torch.int32:
            itype = _l.EMB_IDX_U32
        else:
            return None
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        key = one_buffer = None
        if self.plan_cache_size and not check:
            try:
                ptr = torch.Tensor.data_ptr
                ids_t = tuple(table_ids)
                nbs = tuple(map(len, offsets))
                if outs is None:       # fresh outputs: ONE allocation when every table has the same dim (else no caching)
                    d0 = self._same_dim.get(ids_t)
                    if d0 is None:
                        ds = {tables[t][1] for t in table_ids}
                        d0 = self._same_dim[ids_t] = ds.pop() if len(ds) == 1 else 0
                    if d0:
                        one_buffer = torch.empty((sum(nbs), d0), dtype=torch.float32, device=dev)
                        out_key = one_buffer.data_ptr()
                else:
                    out_key = tuple(map(ptr, outs))
                if outs is not None or one_buffer is not None:
                    # (dtypes: the first tensor's stands for its list -- the engine refuses mixed lists on the ordinary path)
                    key = (ids_t, tuple(map(ptr, indices)), tuple(map(ptr, offsets)), tuple(map(len, indices)), nbs,
                           dt, offsets[0].dtype, dev.index, out_key)
            except (TypeError, KeyError):      # not all 1-D torch tensors / a table that is not loaded: the general path sorts it out
                key = None
        if key is not None and self._launch_cached(key, stream):
            return list(one_buffer.split(nbs)) if outs is None else list(outs)
        buf, buf_ptr, _addr = self._desc_slot(n)
        pack, size = self._DESC.pack_into, self._DESC.size
        nbs, dims = [], []
        for t, ia, oa in zip(table_ids, indices, offsets):
            if type(ia) is not torch.Tensor or type(oa) is not torch.Tensor or ia.dtype is not dt or oa.dtype is not dt \
                    or ia.device != dev or oa.device != dev or not ia.is_contiguous() or not oa.is_contiguous() \
                    or ia.dim() != 1 or oa.dim() != 1:
                return None
            nbs.append(oa.numel())
            dims.append(tables[t][1])        # KeyError: table not loaded
        if outs is None:
            d0 = dims[0]
            if one_buffer is not None:       # allocated for the cache key above
                outs = one_buffer.split(nbs)
            elif all(d == d0 for d in dims):   # one allocation, one view per table
                outs = torch.empty((sum(nbs), d0), dtype=torch.float32, device=dev).split(nbs)
            else:
                outs = [torch.empty((b, d), dtype=torch.float32, device=dev) for b, d in zip(nbs, dims)]
        else:
            for o, b, d in zip(outs, nbs, dims):
                if type(o) is not torch.Tensor or o.device != dev or o.dtype is not torch.float32 \
                        or not o.is_contiguous() or o.numel() != b * d:
                    return None
        off = 0
        for t, ia, oa, b, o in zip(table_ids, indices, offsets, nbs, outs):
            pack(buf, off, t, 0, ia.data_ptr(), oa.data_ptr(), ia.numel(), b, o.data_ptr())
            off += size
        if check:
            self._checked_call(buf_ptr, n, itype, _l.EMB_MEM_DEVICE, stream, check)
            return list(outs)
        _l.check(self._L.emb_lookup_batched(self._h, buf_ptr, n, itype, _l.EMB_MEM_DEVICE, stream))
        if key is not None:
            self._remember_call(key, buf_ptr, n, itype)
        return list(outs)

    def _remember_call(self, key, desc_ptr, n, itype) -> None:
        """Second sighting of a call signature: build its plan.  The cache holds plan_cache_size plans; when it is full the
        least recently used one goes -- but at most once per 64 cacheable calls: destroying a plan waits for the device,
        and a caller whose signatures keep changing must not pay that on every call (its recurring signatures keep their
        plans meanwhile)."""
        with self._plan_lock:
            self._remember_locked(key, desc_ptr, n, itype)

    def _remember_locked(self, key, desc_ptr, n, itype) -> None:
        if key in self._plan_cache or self._check_inputs:
            return
        h = hash(key)                         # (admission bookkeeping only: a collision costs one early plan, nothing else)
        if self._plan_seen.pop(h, 0) < 1:     # first sighting: remember it among the last few thousand signatures
            if len(self._plan_seen) >= 4096:
                for old in list(self._plan_seen)[:2048]:      # dicts keep insertion order: drop the oldest half
                    del self._plan_seen[old]
            self._plan_seen[h] = 1
            return
        if len(self._plan_cache) >= self.plan_cache_size:
            if self._plan_clock - self._plan_last_evict < 64:
                return
            self._plan_last_evict = self._plan_clock
            victim = min(self._plan_cache, key=lambda k: self._plan_cache[k][1])
            self._plan_cache.pop(victim)[0].destroy()
        p = C.c_void_p()
        if self._L.emb_plan_create(self._h, desc_ptr, n, itype, C.byref(p)) != _l.EMB_OK:
            return
        self._plan_cache[key] = [Plan(self, p.value, None, None), self._plan_clock]

    _DESC_DT = np.dtype([("table_id", "<u4"), ("fixed_pooling", "<u4"), ("indices", "<u8"), ("offsets", "<u8"),
                         ("n_indices", "<u8"), ("n_bags", "<u8"), ("pooled", "<u8")])

    def lookup_stacked(self, table_ids: Sequence[int], indices, offsets, out=None, stream: int | None = None,
                       check: bool = False):
        """The stacked form DLRM uses for fixed-size batches: `indices` [T, N] and `offsets` [T, B] as two
        2-D torch CUDA tensors (row t belongs to table_ids[t]; all tables share one dim).  Returns ONE
        [T, B, dim] fp32 tensor (out[t] is table t's pooled rows).  The per-table pointers are computed
        arithmetically, so the Python cost does not grow with T (~10 us for 26 tables, against ~40 us for
        lists of per-table tensors)."""
        import torch
        T = len(table_ids)
        if indices.dim() != 2 or offsets.dim() != 2 or indices.shape[0] != T or offsets.shape[0] != T:
            raise ValueError("indices must be [T, N] and offsets [T, B]")
        if not (indices.is_cuda and offsets.is_cuda and indices.is_contiguous() and offsets.is_contiguous()):
            raise ValueError("lookup_stacked needs contiguous CUDA tensors")
        if indices.dtype != offsets.dtype:
            raise TypeError("indices and offsets must share dtype width")
        itype = _index_type_of(indices.dtype)
        key = tuple(table_ids)
        try:
            stacked = self._tls.stacked
        except AttributeError:
            stacked = self._tls.stacked = {}
        cached = stacked.get(key)
        if cached is None:
            dims = {self._tables[t][1] for t in table_ids}        # KeyError: table not loaded
            if len(dims) != 1:
                raise ValueError("lookup_stacked needs tables of one dim")
            arr = np.zeros(T, dtype=self._DESC_DT)
            arr["table_id"] = np.asarray(table_ids, dtype=np.uint32)
            cached = stacked[key] = (arr, np.arange(T, dtype=np.uint64), dims.pop(),
                                           C.cast(arr.ctypes.data, C.POINTER(_l.EmbLookupDesc)))
        arr, steps, dim, arr_ptr = cached
        N, B = indices.shape[1], offsets.shape[1]
        if out is None:
            out = torch.empty((T, B, dim), dtype=torch.float32, device=indices.device)
        elif tuple(out.shape) != (T, B, dim) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("out must be a contiguous float32 [T, B, dim] tensor")
        if stream is None:
            stream = torch.cuda.current_stream(indices.device).cuda_stream
        i_ptr, o_ptr, out_ptr = indices.data_ptr(), offsets.data_ptr(), out.data_ptr()
        pkey = None
        if self.plan_cache_size and not check:       # the same three buffers as an earlier call: one launch of its plan
            pkey = ("stacked", key, i_ptr, o_ptr, out_ptr, N, B, itype)
            if self._launch_cached(pkey, stream):
                return out
        esz = indices.element_size()
        arr["indices"] = i_ptr + steps * np.uint64(N * esz)
        arr["offsets"] = o_ptr + steps * np.uint64(B * esz)
        arr["pooled"] = out_ptr + steps * np.uint64(B * dim * 4)
        arr["n_indices"] = N
        arr["n_bags"] = B
        if check:
            self._checked_call(arr_ptr, T, itype, _l.EMB_MEM_DEVICE, stream, check)
            return out
        _l.check(self._L.emb_lookup_batched(self._h, arr_ptr, T, itype, _l.EMB_MEM_DEVICE, stream))
        if pkey is not None:
            self._remember_call(pkey, arr_ptr, T, itype)
        return out

    def lookup_descs(self, descs: np.ndarray, itype: int = _l.EMB_IDX_U32, stream: int | None = None) -> None:
        """One fused launch over a numpy array of emb_lookup_desc records (dtype EmbeddingEngine._DESC_DT) holding DEVICE
        pointers -- for callers that compute their buffer addresses arithmetically (the sharded exchange serves N x K
        request pieces per step this way).  The caller keeps the buffers alive; nothing is allocated here."""
        if descs.dtype != self._DESC_DT or not descs.flags["C_CONTIGUOUS"]:
            raise TypeError("descs must be a contiguous array of EmbeddingEngine._DESC_DT records")
        _l.check(self._L.emb_lookup_batched(self._h, C.cast(descs.ctypes.data, C.POINTER(_l.EmbLookupDesc)), len(descs),
                                            itype, _l.EMB_MEM_DEVICE, stream))

    def lookup_batched(self, table_ids: Sequence[int], indices: Sequence, offsets: Sequence,
                       outs: Sequence | None = None, fixed_pooling=0, stream: int | None = None, check: bool | str = False):
        """All tables in one fused launch; returns the list of pooled [B_t, D] outputs
        (the `apply_emb` contract: one [B, D] per table).  check=True: indices / offsets are validated on the GPU
        first (emb_lookup_batched_checked) and IndexError is raised, like nn.EmbeddingBag, before anything is launched.
        check="deferred": validated all the same and a finding still keeps the lookup from running, but the call does not wait
        for the verdict (see _checked_call / check_report)."""
        if not fixed_pooling and len(table_ids) and _is_torch(indices[0]):
            res = self._lookup_batched_cuda(table_ids, indices, offsets, outs, stream, check)
            if res is not None:
                return res
        arr, n, itype, space, results, _keep = self._descs(table_ids, indices, offsets, outs, fixed_pooling)
        if stream is None and space == _l.EMB_MEM_DEVICE:
            stream = _current_stream_for(*indices)
        if check:
            self._checked_call(arr, n, itype, space, stream, check)
            return results
        _l.check(self._L.emb_lookup_batched(self._h, arr, n, itype, space, stream))
        return results

    def lookup(self, table_id: int, indices, offsets, out=None, fixed_pooling: int = 0,
               stream: int | None = None, check: bool = False):
        """lookup(table_id, offsets, indices) -> pooled_rows for one table."""
        return self.lookup_batched([table_id], [indices], [offsets], None if out is None else [out],
                                   fixed_pooling, stream, check)[0]

    def plan(self, table_ids, indices, offsets, outs=None, fixed_pooling=0) -> Plan:
        arr, n, itype, space, results, keep = self._descs(table_ids, indices, offsets, outs, fixed_pooling)
        if space != _l.EMB_MEM_DEVICE:
            raise TypeError("plans need device-resident buffers (torch CUDA tensors or DeviceBuffer)")
        p = C.c_void_p()
        _l.check(self._L.emb_plan_create(self._h, arr, n, itype, C.byref(p)))
        return Plan(self, p.value, results, keep)

    def validate(self, table_ids, indices, offsets, fixed_pooling=0) -> int:
        """Debug check: number of out-of-range indices / broken offsets (0 = clean)."""
        arr, n, itype, space, _r, _k = self._descs(table_ids, indices, offsets, None, fixed_pooling,
                                                   want_outputs=False)
        bad = C.c_uint64()
        stream = _current_stream_for(*indices) if space == _l.EMB_MEM_DEVICE else None   # behind the indices' producer
        rc = self._L.emb_validate_inputs_on(self._h, arr, n, itype, space, stream, C.byref(bad))
        if rc not in (_l.EMB_OK, _l.EMB_ERR_RANGE):
            _l.check(rc)
        return bad.value

    # ---- multi-GPU routing helpers (row-range shards, variable-length bags: counts first) ---------------
    def route_bags_sizes(self, n_tables: int, n_bags: int, total_indices: int, n_shards: int) -> dict:
        """Bytes of the four device buffers emb_route_bags works with: send, meta, slots, work."""
        v = [C.c_uint64() for _ in range(4)]
        _l.check(self._L.emb_route_bags_sizes(n_tables, n_bags, total_indices, n_shards, *[C.byref(x) for x in v]))
        return dict(zip(("send", "meta", "slots", "work"), (x.value for x in v)))

    def route_bags(self, tables, n_bags: int, n_shards: int, send_ptr: int, meta_ptr: int, slots_ptr: int,
                   work_ptr: int, stream: int | None = None, itype: int = _l.EMB_IDX_U32) -> None:
        """tables: sequence of (indices_ptr, offsets_ptr or None, n_indices, fixed_pooling, rows_per_shard) or a
        prepared array from route_tables().  itype: width of the index / offset arrays (EMB_IDX_I64: torch's int64, read in
        place; the request pieces are uint32 local row ids either way).  Enqueue only (emb_route_bags_typed in pimemb.h)."""
        arr = tables if isinstance(tables, C.Array) else self.route_tables(tables)
        _l.check(self._L.emb_route_bags_typed(self._h, arr, len(arr), itype, n_bags, n_shards, send_ptr, meta_ptr, slots_ptr,
                                              work_ptr, stream))

    @staticmethod
    def route_tables(tables):
        arr = (_l.EmbRouteTable * len(tables))()
        for k, (ip, op, n, L, rps) in enumerate(tables):
            arr[k] = _l.EmbRouteTable(ip, op, int(n), int(L), int(rps))
        return arr

    def route_exchange_sizes(self, counts_host_ptr: int, n_tables: int, n_shards: int, dim: int):
        """Split sizes of one exchange step from the counts this rank sent / received: counts_host_ptr addresses
        uint32 [2][N][K+1][2] on the HOST ([0] sent, [1] received).  Returns (req_out_words, req_in_words,
        ret_rows_back, ret_rows_served) as lists of N ints and the job's largest request / return piece in bytes
        (emb_route_exchange_sizes in pimemb.h)."""
        N = n_shards
        buf = self._tls.__dict__.get("xsz")
        if buf is None or len(buf[0]) != 4 * N:
            buf = self._tls.xsz = ((C.c_uint64 * (4 * N))(), C.c_uint64(), C.c_uint64())
        arr, pr, pt = buf
        base = C.addressof(arr)
        _l.check(self._L.emb_route_exchange_sizes(counts_host_ptr, counts_host_ptr + 8 * N * (n_tables + 1), n_tables, N, dim,
                                                  base, base + 8 * N, base + 16 * N, base + 24 * N, C.byref(pr), C.byref(pt)))
        v = arr[:]
        return v[:N], v[N:2 * N], v[2 * N:3 * N], v[3 * N:], pr.value, pt.value

    def lookup_served(self, received_host_ptr: int, n_tables: int, n_shards: int, dim: int, shard_table_ids,
                      req_recv_ptr: int, ret_send_ptr: int, stream: int | None = None) -> int:
        """ONE fused lookup over every request piece a rank received (emb_route_serve_descs + emb_lookup_batched):
        received_host_ptr addresses the received counts, uint32 [N][K+1][2] on the host.  Returns the launch's
        algorithmic bytes (0: nothing to serve)."""
        key = (n_tables, n_shards)
        cache = self._tls.__dict__.get("served")
        if cache is None or cache[0] != key or cache[1] != tuple(shard_table_ids):
            ids = (C.c_uint32 * n_tables)(*[int(x) for x in shard_table_ids])
            cache = self._tls.served = (key, tuple(shard_table_ids), ids, (_l.EmbLookupDesc * (n_tables * n_shards))(),
                                        C.c_uint32(), C.c_uint64())
        _k, _ids_t, ids, descs, n, nbytes = cache
        _l.check(self._L.emb_route_serve_descs(received_host_ptr, n_tables, n_shards, dim, ids, req_recv_ptr, ret_send_ptr,
                                               descs, C.byref(n), C.byref(nbytes)))
        if n.value:
            _l.check(self._L.emb_lookup_batched(self._h, descs, n.value, _l.EMB_IDX_U32, _l.EMB_MEM_DEVICE, stream))
        return nbytes.value

    def unroute_bags(self, recv_ptr: int, meta_ptr: int, slots_ptr: int, n_tables: int, n_bags: int, n_shards: int,
                     dim: int, pooled_ptr: int, stream: int | None = None) -> None:
        _l.check(self._L.emb_unroute_bags(self._h, recv_ptr, meta_ptr, slots_ptr, n_tables, n_bags, n_shards, dim,
                                          pooled_ptr, stream))

    # ---- misc --------------------------------------------------------------------------------------
    def stats(self) -> dict:
        s = _l.EmbStats()
        _l.check(self._L.emb_get_stats(self._h, C.byref(s)))
        d = {k: getattr(s, k) for k, _ in s._fields_}
        d["n_launches_by_kind"] = list(d["n_launches_by_kind"])
        return d

    def reset_stats(self) -> None:
        _l.check(self._L.emb_reset_stats(self._h))

    def synchronize(self, stream: int | None = None) -> None:
        _l.check(self._L.emb_synchronize(self._h, stream))

    @property
    def device(self) -> int:
        d = C.c_int32()
        _l.check(self._L.emb_device_of(self._h, C.byref(d)))
        return d.value

    def close(self) -> None:
        if self._h:
            self._drop_plans()                       # the plan cache's own plans
            pending = None
            try:
                self.check_report()                  # a deferred verdict nobody met is raised here, after the engine is gone
            except (IndexError, _l.PimembError) as ex:
                pending = ex
            _l.check(self._L.emb_destroy(self._h))   # raises while plans are alive
            self._h = None
            if pending is not None:
                raise pending

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
