"""Multi-GPU embedding lookup: one process per GPU, tables sharded over the ranks of one node, indices in / pooled rows
out exchanged between the ranks (RCCL over xGMI) -- the Python face of `emb_shard_*` in include/pimemb.h.

Counterpart of the reference's only "distribution" mechanism -- one DPU per (table, column) with the
indices broadcast to a table's DPUs and the per-DPU results gathered back by the host, ALL inside one lookup() call
(upmem/include/emb_host.h:167 DPU id = table*NR_COLS + col; :258-270 index/offset push; :297 launch; :312-321
result pull) -- re-thought for 8 GPUs with 288 GB each:

  * inputs are data-parallel: rank r holds its own B bags for every table;
  * tables are model-parallel.  The planner (`plan_shards`) places each table as
      - REPLICATED  (<= replicate_bytes): every rank holds it, no exchange at all;
      - WHOLE       : one owner rank (greedy bin-packing on bytes); the bags' indices travel to the owner straight out of
                      the caller's buffers, the pooled rows arrive straight in the caller's output;
      - ROW-SPLIT   (> split_bytes): contiguous row ranges over all ranks; each rank returns a
                      partial pooled sum, the sample owner adds the partials in shard order
                      (deterministic).  Column (D) splitting as in the reference is NOT carried over:
                      rows of <= 1 KiB are already smaller than an efficient transfer unit.
  * one batch = ONE library call (`ShardedEmbeddingBags.forward`, or `submit` / `wait` for the software-pipelined form):
    routing on the GPU, counts first, payload second, ONE fused lookup over everything a rank serves, partial rows added in
    shard order.  The step itself lives in csrc/pimemb_shard.cpp; nothing here computes a lookup or moves a byte."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Sequence


REPLICATED, WHOLE, ROW_SPLIT = "replicated", "whole", "row_split"


@dataclass(frozen=True)
class Unit:
    """A contiguous row range of one table served by one rank."""
    table: int
    owner: int          # -1: every rank (replicated)
    row_lo: int
    row_hi: int
    uid: int            # dense id, also the local engine table id on the owner


@dataclass
class ShardPlan:
    world: int
    rows: list[int]
    dim: int
    elem_bytes: int
    kinds: list[str]
    units: list[Unit]
    units_of_table: list[list[int]] = field(default_factory=list)
    notes: list[str] = field(default_factory=list)       # per table: the rule that placed it

    def owned_units(self, rank: int) -> list[Unit]:
        return [u for u in self.units if u.owner == rank]

    def replicated_units(self) -> list[Unit]:
        return [u for u in self.units if u.owner < 0]

    def bytes_on(self, rank: int) -> int:
        return sum((u.row_hi - u.row_lo) * self.dim * self.elem_bytes
                   for u in self.units if u.owner in (rank, -1))

    def describe(self) -> str:
        n = {k: self.kinds.count(k) for k in (REPLICATED, WHOLE, ROW_SPLIT)}
        return (f"{len(self.rows)} tables over {self.world} ranks: {n[REPLICATED]} replicated, "
                f"{n[WHOLE]} whole, {n[ROW_SPLIT]} row-split; bytes/rank "
                f"{[self.bytes_on(r) for r in range(self.world)]}")


def plan_shards(rows: Sequence[int], dim: int, elem_bytes: int, world: int,
                replicate_bytes: int = 64 << 20, split_bytes: int | None = None,
                pooling: float | Sequence[float] = 1.0, capacity_bytes: int | None = None,
                split_single_rank: bool = False) -> ShardPlan:
    """Greedy placement.  split_bytes=None: split tables larger than 1/world of all sharded bytes.

    Return volume.  A row-split table returns one PARTIAL row per (bag, shard that holds one of the bag's rows) -- up to
    min(pooling, world) rows per bag, against exactly one for a table held whole (un-routing 8 partial rows per bag at
    pooling 32 costs 108 us where the whole-table return costs nothing, profiles/r03/route_probe.log).  So a pooled table
    (`pooling` >= 2 expected indices per bag; one number or one per table) above the split threshold is still placed WHOLE
    while the rank it lands on stays under `capacity_bytes` (default: no limit other than balance -- at most twice the mean
    load); only tables that must be split (capacity, or one index per bag where splitting costs nothing extra) are.
    `plan.notes[t]` says which rule placed a table.  split_single_rank: let a world of ONE rank row-split too (one shard
    holding every row: the whole routed path, rehearsed on one GPU)."""
    rows = [int(r) for r in rows]
    T = len(rows)
    pool = [float(pooling)] * T if not hasattr(pooling, "__len__") else [float(x) for x in pooling]
    size = [r * dim * elem_bytes for r in rows]
    kinds = [REPLICATED if s <= replicate_bytes else WHOLE for s in size]
    notes = ["<= replicate_bytes" if k == REPLICATED else "" for k in kinds]
    sharded = [t for t, k in enumerate(kinds) if k == WHOLE]
    total = sum(size[t] for t in sharded)
    if split_bytes is None:
        split_bytes = max(1, total // max(world, 1))
    cap = capacity_bytes if capacity_bytes is not None else max(1, 2 * total // max(world, 1))
    load = [0] * world
    owner = {}
    for t in sorted(sharded, key=lambda t: -size[t]):          # big first: candidates for splitting
        if not ((world > 1 or split_single_rank) and size[t] > split_bytes and rows[t] >= world):
            continue
        r = min(range(world), key=lambda r: (load[r], r))
        if pool[t] >= 2.0 and load[r] + size[t] <= cap:
            owner[t] = r                                        # pooled and it fits: whole (one returned row per bag)
            load[r] += size[t]
            notes[t] = "pooled (%.0f indices/bag): whole although > split_bytes -- a split would return up to %d partial rows per bag" % (
                pool[t], min(int(pool[t]), world))
        else:
            kinds[t] = ROW_SPLIT
            notes[t] = "> split_bytes" + (": one index per bag, a split returns one row per bag too" if pool[t] < 2.0 else
                                          ": pooled, but no rank has room for it whole")
            for q in range(world):
                load[q] += size[t] // world
    for t in sorted((t for t in sharded if kinds[t] == WHOLE and t not in owner), key=lambda t: -size[t]):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[t] = r
        load[r] += size[t]
        notes[t] = "whole: least loaded rank"
    units, units_of_table = [], []
    for t, k in enumerate(kinds):
        ids = []
        if k == REPLICATED:
            ids.append(len(units)); units.append(Unit(t, -1, 0, rows[t], len(units)))
        elif k == WHOLE:
            ids.append(len(units)); units.append(Unit(t, owner[t], 0, rows[t], len(units)))
        else:
            per = -(-rows[t] // world)
            for r in range(world):
                lo, hi = min(r * per, rows[t]), min((r + 1) * per, rows[t])
                ids.append(len(units)); units.append(Unit(t, r, lo, hi, len(units)))
        units_of_table.append(ids)
    plan = ShardPlan(world, rows, dim, elem_bytes, kinds, units, units_of_table)
    plan.notes = notes
    return plan


def hbm_budget(plan: ShardPlan, rank: int, bags: int, pooling: int, n_slots: int = 8, depth: int = 3, transport: str = "rccl",
               checked: bool = False, balance: float = 1.0, index_bytes: int = 4) -> dict:
    """HBM one rank needs for a sharded job, by category, in bytes -- host arithmetic only (emb_route_bags_sizes through the C
    ABI), so that a layout can be checked against 288 GB BEFORE an 8-GPU node is booked (and by bench.py before it allocates):

      tables     what the rank holds: its shards, its whole tables, every replicated table;
      batches    the caller's `n_slots` rotating batch slots: indices (`index_bytes` each: 4 for uint32, 8 for torch's int64,
                 which the library uses in place) + pooled rows of ALL tables for its own bags (peer stores: carved from the
                 arena instead -- counted there);
      staging    the library's ring of 6 batch slots (pimemb_shard.cpp kRing), grow-only with 25 % headroom: routed requests,
                 slot maps, an int64 job's widened copy of the pieces it serves, what arrives from the peers (request pieces, whole tables' index arrays) and what goes back (partial
                 rows, pooled rows of whole tables owned here) -- the last two only over RCCL: with peer stores the owner gathers
                 and stores in place; batches rotate through ALL six slots whatever the depth, and a slot keeps what it grew to;
      plans      64 cached plans (emb_shard's kPlanCache): descriptors (128 B) + the XCD map (8 B per workgroup);
      maps       the engine's cache of XCD maps of plan-less launches (the routed step's): 32 live entries of up to 8 MB each
                 (8 B per workgroup, <= 2^20 workgroups expanded) + evicted ones awaiting their events, capped at 64 MB
                 (pimemb_engine.cpp kXmapCacheEntries / kXmapGraveBytes) -- round 5 left these inside the flat runtime term;
      counters   a checked shard's served-bag counters (16 KiB per descriptor and ring slot);
      arena      peer stores: what emb_peer_create allocates (bench.py's formula);
      runtime    3 GB flat: the HIP context, code objects, torch's allocator slack, RCCL's buffers -- the world-1 runs of
                 profiles/r05/dist_world1.md hold 1.6 ... 4.3 GB more than the other terms add up to.

    `balance` scales what a rank RECEIVES: 1.0 = every shard gets 1/world of the row-split requests (uniform indices), world =
    every index of every rank names rows of this one shard (the worst skew)."""
    from . import lib as _l
    import ctypes as C
    N, T, dim, elem = plan.world, len(plan.rows), plan.dim, plan.elem_bytes
    B, L = int(bags), int(pooling)
    split = [t for t, k in enumerate(plan.kinds) if k == ROW_SPLIT]
    whole_here = [t for t, k in enumerate(plan.kinds) if k == WHOLE and plan.units[plan.units_of_table[t][0]].owner == rank]
    whole_else = [t for t, k in enumerate(plan.kinds) if k == WHOLE and plan.units[plan.units_of_table[t][0]].owner != rank]
    rep = [t for t, k in enumerate(plan.kinds) if k == REPLICATED]
    Kr, M = len(split), len(whole_here)
    grow = lambda b: int(b + b // 4 + 256)          # pimemb_shard.cpp ensure()
    live = 6                                         # (batch seq uses slot seq % 6: every slot ends up with buffers of its own)
    out = {"tables": plan.bytes_on(rank)}
    ib = int(index_bytes)
    per_slot = T * B * L * ib + T * B * dim * 4
    out["batches"] = 0 if transport == "peer" else n_slots * per_slot
    stg = 0
    one_hot_direct = L == 1 and (transport == "peer" or N == 1)           # the direct path: nothing is routed, nothing staged
    if Kr and not one_hot_direct:
        sb, mb, lb, wb = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _l.check(_l.load().emb_route_bags_sizes(Kr, max(B, 1), Kr * B * L, N, C.byref(sb), C.byref(mb), C.byref(lb), C.byref(wb)))
        sub = min(L, N)                                # partial rows a bag can come back as
        stg += live * (grow(sb.value) + grow(mb.value) + grow(lb.value)) + grow(wb.value)
        stg += live * grow(Kr * B * sub * dim * 4)                                              # ret_recv
        if transport != "peer":                        # pieces arrive in / leave from staging only when RCCL carries them
            recv_idx = int(balance * Kr * B * L)                                                # indices this shard is asked for, all sources
            recv_sub = int(min(balance * Kr * B * sub, recv_idx))
            stg += live * (grow((recv_idx + recv_sub) * 4 + 64 * N * Kr) + grow(recv_sub * dim * 4))     # req_recv, ret_send
        if ib == 8:                                    # an int64 job widens the pieces it serves once (Batch.wide), whatever carried them
            stg += live * grow((int(balance * Kr * B * L) + int(min(balance * Kr * B * sub, balance * Kr * B * L))) * 8 + 128 * N * Kr)
    if M and transport != "peer" and N > 1:            # whole tables owned here: the other ranks' index arrays in, their pooled rows out
        stg += live * (grow((N - 1) * M * B * (L + 1) * ib) + grow((N - 1) * M * B * dim * 4))       # (whole tables' arrays travel as they are)
    out["staging"] = stg
    n_desc = N * (Kr + M) + len(rep)
    tiles = -(-B // 64)
    out["plans"] = 64 * (n_desc * 128 + (n_desc * tiles + 8 * n_desc) * 8)
    out["maps"] = 32 * min(8 << 20, (n_desc * tiles + 8 * n_desc) * 8) + (64 << 20)
    out["counters"] = live * n_desc * 16384 if checked else 0
    out["arena"] = 0
    if transport == "peer":
        out["arena"] = int(1.25 * n_slots * T * B * (L * ib + dim * 4)) + 8 * 2 * Kr * B * (L * 8 + min(L, N) * dim * 4 * 2) + (256 << 20)
    out["runtime"] = 3 * 10**9
    out["total"] = sum(out.values())
    return out


def fit_to_hbm(rows: Sequence[int], hbm_bytes: int, make_plan: Callable[[Sequence[int]], ShardPlan], budget_of: Callable[[ShardPlan, int], dict],
               headroom: float = 0.94):
    """Largest uniform row scale <= 1 at which the worst rank's hbm_budget stays within headroom x hbm_bytes:
    (scale, rows, plan, worst budget).  make_plan(rows) -> ShardPlan; budget_of(plan, rank) -> hbm_budget dict.  BASELINE
    configs[4] (512 x 50M x dim 64 fp16) does not fit 8 x 288 GB as written; at 30M rows the TABLES fit (245.8 GB per rank) but
    tables + the batch slots + the RCCL staging of 64 whole tables per owner do not (293 GB): the bench shrinks the rows by a few
    per cent more and says so, instead of running out of memory half-way through its first real 8-GPU run."""
    scale, cur = 1.0, [int(r) for r in rows]
    for _ in range(8):
        plan = make_plan(cur)
        worst = max((budget_of(plan, r) for r in range(plan.world)), key=lambda d: d["total"])
        if worst["total"] <= headroom * hbm_bytes:
            return scale, cur, plan, worst
        room = headroom * hbm_bytes - (worst["total"] - worst["tables"])
        if room <= 0:
            raise MemoryError("the batch slots and staging buffers alone (%.1f GB) exceed %.0f %% of %.1f GB of HBM: fewer bags per batch or "
                              "fewer rotating slots" % ((worst["total"] - worst["tables"]) / 1e9, headroom * 100, hbm_bytes / 1e9))
        scale *= 0.995 * room / worst["tables"]
        cur = [max(1, int(r * scale)) for r in rows]
    raise MemoryError("no row scale found that fits %.1f GB" % (hbm_bytes / 1e9))


# ------------------------------------------------------------------------------------------------
class ShardedEmbeddingBags:
    """`apply_emb` over tables sharded across the ranks of one node: forward(lS_o, lS_i) -> [B, dim] per table, ONE
    library call per batch (emb_shard_* in include/pimemb.h; the reference serves all its devices from one lookup() call
    too, emb_host.h:258-270, :297, :312-321).

        sh = ShardedEmbeddingBags(plan, engine, rank, comm)        # comm: engine.NativeExchange (RCCL), None for one rank
        sh.load_tables(lambda t, lo, hi: rows)                     # only this rank's shards are requested
        outs = sh.forward(lS_o, lS_i)                              # synchronous form: submit + flush + wait

    Software-pipelined form (depth 1 .. 3): `seq, outs = sh.submit(...)` every batch, `sh.wait(seq)` once the batch is
    `depth` submits old (or after `sh.flush()`); a batch's tensors belong to the library until then.  submit / flush are
    COLLECTIVE: every rank makes the same calls in the same order (a rank with nothing to look up passes empty tensors).

    Indices / offsets: torch CUDA int32 (uint32 bits, the reference's width) or int64 (DLRM's dtype) -- BOTH are handed to the
    library in place (emb_shard_input.index_type): no narrowing pass, no copy; an id outside its table stays what it is for the
    checks to refuse.  One dtype per batch.

    check=True (default, like nn.EmbeddingBag): indices are not trusted.  Routed batches: every fused lookup validates what it
    serves first and the SERVING rank raises IndexError after the batch has gone through all its stages -- nobody is left in a
    transfer.  One-index batches on the direct path: every launch counts the bags it serves and the REQUESTING rank compares.
    Either way an offending bag pools to a ZERO row.  When the requester's comparison is made:
      check=True / "deferred"  by a LATER forward / submit / wait(seq) -- the first one that finds the batch's counts arrived
                               (they only look, never wait) -- or by flush(), report(), close(): the call that completes a
                               batch never waits for the GPU (the default: an IndexError arrives a call or a few late, like a
                               device-side assert of torch's own CUDA EmbeddingBag, and names its batch);
      check="sync"             inside the call that completes the batch (forward() then waits for its own launch: +13 us on a
                               59-us step, profiles/r05/dist_world1.md);
      check=False              nothing is checked (the reference's behaviour, emb_dpu_lookup.c:113)."""

    def __init__(self, plan: ShardPlan, engine, rank: int, comm=None, depth: int = 0, check: "bool | str" = True,
                 self_via_comm: bool = False, peer: "PeerGroup | None" = None):
        import ctypes as C
        import torch
        from . import lib as _l
        self.torch, self._l, self._C = torch, _l, C
        self.plan, self.engine, self.rank, self.world, self.comm = plan, engine, int(rank), plan.world, comm
        self.depth, self.dim, self.T = int(depth), plan.dim, len(plan.rows)
        self.peer = peer           # the collective-free exchange (EMB_SHARD_PEER_STORES): buffers of tables other ranks hold come from peer.empty()
        if comm is None and peer is None and plan.world != 1:
            raise ValueError("a world of %d ranks needs a communicator (engine.NativeExchange) or a PeerGroup" % plan.world)
        self._L = engine._L
        tabs = (_l.EmbShardTable * self.T)()
        for t, k in enumerate(plan.kinds):
            us = [plan.units[i] for i in plan.units_of_table[t]]
            if k == REPLICATED:
                tabs[t] = _l.EmbShardTable(_l.EMB_PLACE_REPLICATED, -1, us[0].uid, 0)
            elif k == WHOLE:
                tabs[t] = _l.EmbShardTable(_l.EMB_PLACE_WHOLE, us[0].owner, us[0].uid, 0)
            else:
                tabs[t] = _l.EmbShardTable(_l.EMB_PLACE_ROWS, -1, us[self.rank].uid, -(-plan.rows[t] // plan.world))
        self._tabs = tabs
        if check not in (True, False, "deferred", "sync"):
            raise ValueError("check is True / 'deferred', 'sync' or False")
        self.check = "sync" if check == "sync" else ("deferred" if check else False)
        self._flags = (_l.EMB_SHARD_CHECK_SERVED if check else 0) | (_l.EMB_SHARD_SELF_VIA_COMM if self_via_comm else 0) | \
                      (_l.EMB_SHARD_PEER_STORES if peer is not None else 0) | (_l.EMB_SHARD_DEFER_REPORT if self.check == "deferred" else 0)
        self._h = None
        self._live = {}            # seq -> tensors kept alive until the batch is waited for

    # ---- tables -------------------------------------------------------------------------------
    def load_tables(self, table_rows: Callable[[int, int, int], object]) -> None:
        """table_rows(table, row_lo, row_hi) -> [row_hi-row_lo, dim] rows (torch tensor or numpy); only the shards this
        rank serves are requested.  Creates the shard object (it checks the engine's tables against the plan)."""
        for u in self.plan.owned_units(self.rank) + self.plan.replicated_units():
            self.engine.load_table(u.uid, table_rows(u.table, u.row_lo, u.row_hi))
        self._create()

    def _create(self) -> None:
        C, _l = self._C, self._l
        if self._h is not None:
            return
        cfg = _l.EmbShardConfig(self.T, self.dim, self.depth, self._flags, self._tabs, self.peer._h if self.peer is not None else None)
        h = C.c_void_p()
        _l.check(self._L.emb_shard_create(self.engine._h, self.comm._h if self.comm is not None else None, C.byref(cfg), C.byref(h)))
        self._h = h

    # ---- one batch ----------------------------------------------------------------------------
    def _ids(self, x, want=None):
        """One index / offset tensor as the library takes it: contiguous int32 (uint32 bits) or int64, in place."""
        t = self.torch
        if x.dtype not in (t.int32, t.int64):
            raise TypeError(f"indices/offsets must be int32 (uint32 bits) or int64 CUDA tensors, got {x.dtype}")
        if want is not None and x.dtype != want:
            raise TypeError(f"one index dtype per batch: got {x.dtype} next to {want}")
        return x if x.is_contiguous() else x.contiguous()

    def prepare(self, indices: Sequence, offsets: Sequence | None = None, fixed_pooling: int = 0, outs: Sequence | None = None):
        """Descriptor array of one batch (reusable while the tensors stay where they are): (array, n_bags, outs, keep).
        outs=None allocates the pooled rows: a fresh torch tensor -- or, with a peer group, a fresh block of the ARENA, which is
        a bump allocator (nothing is freed before PeerGroup.close): a long-running peer-store caller allocates its batch slots
        once (PeerGroup.empty for indices, offsets and outs) and passes them in, as bench.py --exchange peer does."""
        t = self.torch
        if len(indices) != self.T or (offsets is not None and len(offsets) != self.T):
            raise ValueError("one index (and offset) tensor per table")
        fast = self._prepare_fast(indices, offsets, fixed_pooling, outs)
        if fast is not None:
            return fast
        want = indices[0].dtype
        idx = [self._ids(i, want) for i in indices]
        off = [self._ids(o, want) for o in offsets] if offsets is not None else [None] * self.T
        itype = self._l.EMB_IDX_I64 if want == t.int64 else self._l.EMB_IDX_U32
        if offsets is not None:
            n_bags = int(off[0].numel())
            if any(int(o.numel()) != n_bags for o in off):
                raise ValueError("every table needs the same number of bags in one batch")
        else:
            if fixed_pooling <= 0:
                raise ValueError("offsets=None needs fixed_pooling > 0")
            n_bags = int(idx[0].numel()) // int(fixed_pooling)
        dev = idx[0].device
        self._same_device(dev)
        if outs is None:
            one = (self.peer.empty if self.peer is not None else lambda sh_, dtype: t.empty(sh_, dtype=dtype, device=dev))(
                (self.T, n_bags, self.dim), t.float32)
            outs = [one[k] for k in range(self.T)]
        arr = (self._l.EmbShardInput * self.T)()
        for k in range(self.T):
            o = outs[k]
            if o.dtype != t.float32 or not o.is_contiguous() or o.numel() != n_bags * self.dim:
                raise ValueError("outs[%d] must be a contiguous float32 [n_bags, dim] tensor" % k)
            if idx[k].device != dev or o.device != dev or (off[k] is not None and off[k].device != dev):
                raise ValueError("table %d: every tensor of a batch must live on one device" % k)
            arr[k] = self._l.EmbShardInput(idx[k].data_ptr(), off[k].data_ptr() if off[k] is not None else None, idx[k].numel(),
                                           0 if off[k] is not None else int(fixed_pooling), itype, o.data_ptr())
        return arr, n_bags, list(outs), (idx, off, outs)

    def _same_device(self, dev) -> None:
        """The tensors of a batch are raw pointers to the library, which works on the ENGINE's device: tensors of another GPU
        would be a wild access there, not a Python error -- refuse them here."""
        index = dev if isinstance(dev, int) else (dev.index if dev.index is not None else self.torch.cuda.current_device())
        if getattr(dev, "type", "cuda") != "cuda" or int(index) != int(self.engine.device):
            raise ValueError(f"the batch's tensors live on {dev}, the engine on cuda:{self.engine.device}")

    def _prepare_fast(self, indices, offsets, fixed_pooling, outs):
        """The descriptor array through the C helper (`_pimemb_marshal.pack_shard`: int32 or int64 CUDA tensors as they are, one
        call instead of ~1.2 us of Python per table -- 32 us for 26 tables against a 60-us step, tools/shard_py_overhead_probe.py).
        None when the helper is not built or the tensors are not what it takes (mixed dtypes, odd layouts): the general path
        then does the work, or words the error."""
        from .engine import _marshal
        tb = _marshal()
        if tb is None or not hasattr(tb, "pack_shard") or self.peer is not None and outs is None:
            return None
        t = self.torch
        indices = indices if isinstance(indices, list) else list(indices)
        offsets = None if offsets is None else (offsets if isinstance(offsets, list) else list(offsets))
        if outs is None:
            first = indices[0]
            if not (isinstance(first, t.Tensor) and first.is_cuda and first.dtype in (t.int32, t.int64)):
                return None
            if offsets is not None:
                n_bags = int(offsets[0].numel())
            elif fixed_pooling > 0:
                n_bags = int(first.numel()) // int(fixed_pooling)
            else:
                return None
            outs = t.empty((self.T, n_bags, self.dim), dtype=t.float32, device=first.device).unbind(0)
        arr = (self._l.EmbShardInput * self.T)()
        res = tb.pack_shard(self._C.addressof(arr), indices, offsets, outs, int(fixed_pooling), self.dim)
        if res is None:
            return None
        if int(res[1]) != int(self.engine.device):       # (raw pointers of another GPU would be a wild access on the engine's)
            raise ValueError(f"the batch's tensors live on cuda:{int(res[1])}, the engine on cuda:{self.engine.device}")
        return arr, int(res[0]), list(outs), (indices, offsets, outs)

    def _stream(self, stream):
        return stream if stream is not None else self.torch.cuda.current_stream(self.engine.device).cuda_stream

    def _check(self, rc):
        if rc == self._l.EMB_ERR_RANGE:
            raise IndexError("ShardedEmbeddingBags: " + self._L.emb_last_error().decode(errors="replace"))
        self._l.check(rc)

    def submit_prepared(self, prepared, stream: int | None = None) -> int:
        arr, n_bags, _outs, keep = prepared
        seq = self._C.c_uint64()
        rc = self._L.emb_shard_submit(self._h, arr, n_bags, self._stream(stream), self._C.byref(seq))
        if rc in (self._l.EMB_OK, self._l.EMB_ERR_RANGE):
            self._live[seq.value] = keep
            for old in [q for q in self._live if q + 8 <= seq.value]:
                del self._live[old]
        self._check(rc)
        return seq.value

    def submit(self, indices, offsets=None, fixed_pooling: int = 0, outs=None, stream: int | None = None):
        """Hand one batch over.  Returns (seq, outs); outs[t] is valid once wait(seq) has been ordered."""
        self._create()
        prepared = self.prepare(indices, offsets, fixed_pooling, outs)
        return self.submit_prepared(prepared, stream), prepared[2]

    def wait(self, seq: int, stream: int | None = None) -> None:
        self._l.check(self._L.emb_shard_wait(self._h, seq, self._stream(stream)))

    def flush(self) -> None:
        self._check(self._L.emb_shard_flush(self._h))

    def report(self) -> None:
        """check="deferred": compare the served counts of every completed batch not looked at yet, NOW (waits for their
        kernels); raises IndexError naming the batch.  A no-op otherwise."""
        if self._h is not None:
            self._check(self._L.emb_shard_report(self._h))

    def forward(self, lS_o: Sequence | None, lS_i: Sequence, fixed_pooling: int = 0, outs=None):
        """The `apply_emb` contract: offsets per table, indices per table -> [B, dim] fp32 per table (this rank's bags)."""
        self._create()
        arr, n_bags, res, keep = self.prepare(lS_i, lS_o, fixed_pooling, outs)
        self._check(self._L.emb_shard_lookup(self._h, arr, n_bags, self._stream(None)))
        self._keep = keep          # until the next call: the stream may still be reading them
        return res

    __call__ = forward

    def stats(self, reset: bool = False) -> dict:
        st = self._l.EmbShardStats()
        self._l.check(self._L.emb_shard_get_stats(self._h, self._C.byref(st), int(reset)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def set_kernel_timing(self, on: bool) -> None:
        self._l.check(self._L.emb_shard_set_kernel_timing(self._h, int(on)))

    def sent_counts(self, seq: int):
        """{sub-bags, indices} per (peer, row-split table) of the requests this rank sent for batch seq: [N][Kr][2]."""
        import numpy as np
        kr = sum(1 for k in self.plan.kinds if k == ROW_SPLIT)
        buf = (self._C.c_uint32 * max(1, self.world * kr * 2))()
        self._l.check(self._L.emb_shard_sent_counts(self._h, seq, buf, len(buf)))
        return np.ctypeslib.as_array(buf)[:self.world * kr * 2].reshape(self.world, kr, 2).astype(np.int64)

    def close(self) -> None:
        """Destroys the shard object.  A deferred finding nobody collected is raised here (after the object is gone)."""
        if self._h is not None:
            rc = self._L.emb_shard_report(self._h) if self.check == "deferred" else self._l.EMB_OK
            msg = self._L.emb_last_error().decode(errors="replace") if rc == self._l.EMB_ERR_RANGE else ""
            self._L.emb_shard_destroy(self._h)
            self._h = None
            self._live.clear()
            if rc == self._l.EMB_ERR_RANGE:
                raise IndexError("ShardedEmbeddingBags: " + msg)
        self._live.clear()


class PeerGroup:
    """Ranks that read and write each other's HBM directly (emb_peer_* in pimemb.h): the substrate of the collective-free
    exchange (`ShardedEmbeddingBags(..., peer=group)`).  Every rank creates it with the same `tag` (any string unique to the
    job) and `world`; set-up goes through one POSIX shared-memory segment and HIP IPC handles -- no RCCL, no torch.  Everything
    a peer may touch (indices / offsets / outputs of tables other ranks hold) must be carved from the arena: `empty()`."""

    def __init__(self, engine, tag: str, rank: int, world: int, arena_bytes: int = 256 << 20):
        import ctypes as C
        from . import lib as _l
        self._C, self._l, self._L, self.engine = C, _l, engine._L, engine
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        _l.check(self._L.emb_peer_create(engine._h, tag.encode(), rank, world, int(arena_bytes), C.byref(h)))
        self._h = h

    def alloc(self, nbytes: int) -> int:
        p = self._C.c_void_p()
        self._l.check(self._L.emb_peer_alloc(self._h, int(nbytes), self._C.byref(p)))
        return p.value

    def empty(self, shape, dtype):
        """A torch CUDA tensor living in this rank's arena (peers can address it)."""
        import torch
        shape = tuple(int(x) for x in (shape if hasattr(shape, "__len__") else (shape,)))
        n = 1
        for x in shape:
            n *= x
        esz = torch.empty(0, dtype=dtype).element_size()
        ptr = self.alloc(max(n * esz, 16))
        typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8", torch.float16: "<f2", torch.uint8: "|u1"}[dtype]

        class _View:
            __cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}
        if n == 0:
            return torch.empty(shape, dtype=dtype, device=torch.device("cuda", self.engine.device))
        return torch.as_tensor(_View(), device=torch.device("cuda", self.engine.device))

    def info(self) -> dict:
        C = self._C
        r, w, fg = C.c_int32(), C.c_int32(), C.c_int32()
        a, nb, used = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._l.check(self._L.emb_peer_info(self._h, C.byref(r), C.byref(w), C.byref(a), C.byref(nb), C.byref(used), C.byref(fg)))
        return dict(rank=r.value, world=w.value, arena=a.value, arena_bytes=nb.value, used=used.value, fine_grained=bool(fg.value))

    def barrier(self) -> None:
        self._l.check(self._L.emb_peer_barrier(self._h))

    def close(self) -> None:
        if self._h:
            try:
                self.barrier()          # a peer may still be gathering from / storing into this rank's arena
            finally:
                self._L.emb_peer_destroy(self._h)
                self._h = None


def native_comm(engine, rank: int, world: int, always: bool = False):
    """An engine.NativeExchange (RCCL communicator of the C side) bootstrapped over an initialised torch.distributed
    group: rank 0 draws the id, broadcast_object_list spreads it.  None for a world of one rank unless `always` (a
    one-rank communicator: what self_via_comm rehearses RCCL's grouped send / receive with)."""
    if world == 1 and not always:
        return None
    from .engine import NativeExchange

    def bcast(raw: bytes) -> bytes:
        if world == 1:
            return raw
        import torch.distributed as dist
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    return NativeExchange(engine, rank, world, bcast)
