"""Multi-GPU embedding lookup: one process per GPU, tables sharded over the ranks of one node, indices
in / pooled rows out exchanged with all-to-all (RCCL over xGMI on GPUs; gloo in the CPU tests).

Counterpart of the reference's only "distribution" mechanism -- one DPU per (table, column) with the
indices broadcast to a table's DPUs and the per-DPU results gathered back by the host
(upmem/include/emb_host.h:167 DPU id = table*NR_COLS + col; :258-270 index/offset push; :312-321
result pull) -- re-thought for 8 GPUs with 288 GB each:

  * inputs are data-parallel: rank r holds its own B bags for every table;
  * tables are model-parallel.  The planner places each table as
      - REPLICATED  (<= replicate_bytes): every rank holds it, no exchange at all;
      - WHOLE       : one owner rank (greedy bin-packing on bytes);
      - ROW-SPLIT   (> split_bytes): contiguous row ranges over all ranks; each rank returns a
                      partial pooled sum, the sample owner adds the partials in shard order
                      (deterministic).  Column (D) splitting as in the reference is NOT carried over:
                      rows of <= 1 KiB are already smaller than an efficient transfer unit.
  * one step = three collectives: a small head message (piece sizes + what all ranks must agree on: the job's largest
    pieces, a bad-input flag), all_to_all(per unit: header, bag lengths, indices -- one piece per destination) -> ONE
    fused local lookup (HIP engine) over all units this rank serves -> all_to_all(pooled rows) -> per-table [B, D].

Nothing here computes a lookup: the local step is delegated to a backend (`EngineBackend` = the HIP
engine).  Tests inject their own backend to check the routing on CPU."""
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
                replicate_bytes: int = 64 << 20, split_bytes: int | None = None) -> ShardPlan:
    """Greedy placement.  split_bytes=None: split tables larger than 1/world of all sharded bytes."""
    rows = [int(r) for r in rows]
    size = [r * dim * elem_bytes for r in rows]
    kinds = [REPLICATED if s <= replicate_bytes else WHOLE for s in size]
    sharded = [t for t, k in enumerate(kinds) if k == WHOLE]
    if split_bytes is None:
        split_bytes = max(1, sum(size[t] for t in sharded) // max(world, 1))
    for t in sharded:
        if world > 1 and size[t] > split_bytes and rows[t] >= world:
            kinds[t] = ROW_SPLIT
    load = [0] * world
    for t in sharded:                       # row-split tables load every rank equally
        if kinds[t] == ROW_SPLIT:
            for r in range(world):
                load[r] += size[t] // world
    owner = {}
    for t in sorted((t for t in sharded if kinds[t] == WHOLE), key=lambda t: -size[t]):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[t] = r
        load[r] += size[t]
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
    return ShardPlan(world, rows, dim, elem_bytes, kinds, units, units_of_table)


# ------------------------------------------------------------------------------------------------
class EngineBackend:
    """Local step on the HIP engine: unit uid -> engine table uid; torch CUDA tensors, zero-copy."""

    def __init__(self, engine):
        self.engine = engine

    def load(self, uid: int, rows) -> None:
        self.engine.load_table(uid, rows)

    def lookup(self, uids, indices, offsets, outs):
        """One fused launch over every unit this rank serves."""
        return self.engine.lookup_batched(uids, indices, offsets, outs)


class ShardedLookup:
    """Data-parallel in, model-parallel tables, all-to-all both ways.  `torch.distributed` must be
    initialised (backend nccl = RCCL on GPUs; gloo for CPU tests).  comm_device: where collective
    buffers live ('cuda:N' for RCCL; 'cpu' stages through host for gloo)."""

    def __init__(self, plan: ShardPlan, rank: int, backend, device, comm_device=None, group=None,
                 trusted_inputs: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.plan, self.rank, self.world = plan, rank, plan.world
        self.backend, self.device, self.group = backend, torch.device(device), group
        self.trusted_inputs = trusted_inputs       # False: row ids of row-split tables are range-checked before they are narrowed
        self.comm_device = torch.device(comm_device) if comm_device is not None else self.device
        self.served = plan.owned_units(rank)              # units whose rows live here (sharded)
        self.local = plan.replicated_units()              # units every rank holds
        # units I must send requests for, grouped by destination rank, in a fixed global order
        self.send_units = [[u for u in plan.units if u.owner == d] for d in range(self.world)]
        # Row-split tables over the HIP engine take the GPU router + counts-first exchange (RowRangeExchange); the
        # host-side routing below then only sees replicated and whole tables.  Other backends (the CPU tests'
        # stand-in) keep the host-routed path for every table.
        self.split_tables = [t for t, k in enumerate(plan.kinds) if k == ROW_SPLIT]
        self._rr = None
        if self.split_tables and isinstance(backend, EngineBackend) and self.device.type == "cuda":
            mine = {u.table: u.uid for u in self.served}
            per = [-(-plan.rows[t] // self.world) for t in self.split_tables]
            self._rr = RowRangeExchange(backend.engine, [mine[t] for t in self.split_tables], per, plan.dim, rank,
                                        self.world, self.device, group=group,
                                        stage_cpu=self.comm_device.type != "cuda")
            self.send_units = [[u for u in us if plan.kinds[u.table] != ROW_SPLIT] for us in self.send_units]
            self.served = [u for u in self.served if plan.kinds[u.table] != ROW_SPLIT]

    # ---- tables -------------------------------------------------------------------------------
    def load_tables(self, table_rows: Callable[[int, int, int], object]) -> None:
        """table_rows(table, row_lo, row_hi) -> [row_hi-row_lo, dim] rows (torch tensor or numpy);
        only the shards this rank serves are requested."""
        for u in self.plan.owned_units(self.rank) + self.local:
            self.backend.load(u.uid, table_rows(u.table, u.row_lo, u.row_hi))

    # ---- one step -----------------------------------------------------------------------------
    def _lens(self, offsets, n_idx: int):
        t = self.torch
        end = t.cat([offsets[1:], offsets.new_tensor([n_idx])]) if offsets.numel() else offsets
        return end - offsets

    def _route(self, u: Unit, idx, lens):
        """Indices of my bags that fall into unit u (rebased to the unit's first row) + per-bag
        counts.  Whole tables pass through untouched."""
        t = self.torch
        if u.row_lo == 0 and u.row_hi == self.plan.rows[u.table]:
            return idx, lens
        keep = (idx >= u.row_lo) & (idx < u.row_hi)
        bag_of = t.repeat_interleave(t.arange(lens.numel(), device=idx.device), lens)
        new_lens = t.bincount(bag_of[keep], minlength=lens.numel()).to(lens.dtype)
        return idx[keep] - u.row_lo, new_lens

    def _head(self, rows):
        """The FIRST message of a step: one small int64 row per destination, equal length everywhere (all_to_all with equal
        splits).  It carries the element count of the payload piece that follows and what every rank must agree on before
        it enters the next collectives (largest pieces, the bad-input flag).  The one host read of the step."""
        t, dist = self.torch, self.dist
        c = t.tensor(rows, dtype=t.int64, device=self.comm_device)
        r = t.empty_like(c)
        dist.all_to_all_single(r, c, group=self.group)
        return r.cpu()

    def _move(self, send_parts, dtype, recv_counts, worst_elems: int):
        """all_to_all of one 1-D tensor per destination with known receive counts.  worst_elems: the JOB's largest piece
        (every rank passes the same number, learnt from the head messages), which sets the number of rounds over RCCL --
        all ranks enter the same collectives."""
        t, dist = self.torch, self.dist
        counts = [int(p.numel()) for p in send_parts]
        send = t.cat([p.reshape(-1).to(dtype) for p in send_parts]).to(self.comm_device)
        recv = t.empty(int(sum(recv_counts)), dtype=dtype, device=self.comm_device)
        if self.comm_device.type == "cuda":
            all_to_all_rounds(dist, recv, send, list(recv_counts), counts, rounds_for(worst_elems * send.element_size()),
                              group=self.group).wait()
        else:            # gloo: no piece-size limit (and no list all_to_all)
            dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=counts,
                                   group=self.group)
        return list(recv.split(list(recv_counts)))

    def _bad_inputs(self, indices) -> bool:
        """Row ids of the row-split tables are narrowed to uint32 for the GPU router: a negative or >= nr_rows id must
        not wrap silently (nn.EmbeddingBag raises IndexError).  One device-side min / max per table, one host read."""
        t = self.torch
        lo = [indices[k].min() if indices[k].numel() else indices[k].new_zeros(()) for k in self.split_tables]
        hi = [indices[k].max() - self.plan.rows[k] if indices[k].numel() else indices[k].new_full((), -1) for k in self.split_tables]
        flags = t.stack([t.stack(lo).min() < 0, t.stack(hi).max() >= 0])
        return bool(flags.any().item())

    def forward(self, indices: Sequence, offsets: Sequence):
        """indices[t], offsets[t]: this rank's bags for table t (torch int64/int32 tensors on
        `device`).  Returns [B_t, dim] fp32 per table -- the `apply_emb` contract."""
        t = self.torch
        T = len(self.plan.rows)
        assert len(indices) == T and len(offsets) == T
        idx_dtype = indices[0].dtype
        bad = int(self._rr is not None and not self.trusted_inputs and self._bad_inputs(indices))
        lens = [self._lens(offsets[i], indices[i].numel()) for i in range(T)]
        n_bags = [int(l.numel()) for l in lens]

        # 1. requests per destination, ONE payload piece each: for every unit the destination owns a {n_bags, n_indices}
        #    header, then all bag lengths, then all indices (int64 on the wire).  Three collectives per step in all:
        #    head (counts + what all ranks must agree on), payload, pooled rows back.
        D = self.plan.dim
        pieces, n_units = [], []
        for d in range(self.world):
            meta, ls, ix = [], [], []
            for u in self.send_units[d]:
                i_u, l_u = self._route(u, indices[u.table], lens[u.table])
                meta += [l_u.numel(), i_u.numel()]
                ls.append(l_u.to(t.int64))
                ix.append(i_u.to(t.int64))
            pieces.append(t.cat([t.tensor(meta, dtype=t.int64, device=indices[0].device)] + ls + ix) if meta
                          else t.empty(0, dtype=t.int64, device=indices[0].device))
            n_units.append(len(self.send_units[d]))
        out_counts = [sum(n_bags[u.table] for u in self.send_units[d]) * D for d in range(self.world)]   # rows coming back to me
        worst_piece = max((int(p.numel()) for p in pieces), default=0)
        head = self._head([[int(pieces[d].numel()), worst_piece, max(out_counts, default=0), bad] for d in range(self.world)])
        culprits = [s for s in range(self.world) if int(head[s][3])]
        if culprits:         # every rank learns it from the same message and raises together: nobody hangs in a collective
            raise IndexError(f"ShardedLookup: rank(s) {culprits} passed row ids outside [0, nr_rows) of a row-split table")
        worst_in, worst_out = int(head[:, 1].max()), int(head[:, 2].max())          # the job's largest pieces (elements)
        rr_out = None
        if self._rr is not None:     # row-split tables: GPU routing, counts first (uint32 row ids at that boundary)
            rr_out = self._rr.forward([indices[k].to(t.int32).contiguous() for k in self.split_tables],
                                      [offsets[k].to(t.int32).contiguous() for k in self.split_tables])
        got = self._move(pieces, t.int64, [int(x) for x in head[:, 0].tolist()], worst_in)
        K = len(self.served)
        meta_in, lens_in, idx_in = [], [], []
        heads = t.stack([g[:2 * K] for g in got]).cpu() if K else None      # one host read for all sources
        for s_ in range(self.world):       # every source sent me {header for my K units | lens | indices}
            m = heads[s_] if K else got[s_][:0]
            nb_all, ni_all = (int(m[0::2].sum()), int(m[1::2].sum())) if K else (0, 0)
            meta_in.append(m)
            lens_in.append(got[s_][2 * K:2 * K + nb_all])
            idx_in.append(got[s_][2 * K + nb_all:2 * K + nb_all + ni_all].to(idx_dtype))

        # 2. ONE fused local lookup: served units over the bags of every source rank (in rank order)
        #    + replicated units over my own bags
        uids, l_idx, l_off, src_bags = [], [], [], []
        cursor_l = [0] * self.world
        cursor_i = [0] * self.world
        for k, u in enumerate(self.served):
            parts_l, parts_i, bags_per_src = [], [], []
            for s in range(self.world):
                nb, ni = int(meta_in[s][2 * k]), int(meta_in[s][2 * k + 1])
                parts_l.append(lens_in[s][cursor_l[s]:cursor_l[s] + nb])
                parts_i.append(idx_in[s][cursor_i[s]:cursor_i[s] + ni])
                cursor_l[s] += nb
                cursor_i[s] += ni
                bags_per_src.append(nb)
            all_l = t.cat(parts_l).to(self.device)
            off = (t.cumsum(all_l, 0) - all_l).to(idx_dtype)
            uids.append(u.uid)
            l_idx.append(t.cat(parts_i).to(self.device).contiguous())
            l_off.append(off.contiguous())
            src_bags.append(bags_per_src)
        for u in self.local:
            uids.append(u.uid)
            l_idx.append(indices[u.table].contiguous())
            l_off.append(offsets[u.table].contiguous())
        outs = self.backend.lookup(uids, l_idx, l_off, None) if uids else []

        # 3. pooled rows back to the ranks that own the bags
        send_out = []
        for s in range(self.world):
            parts = []
            for k in range(len(self.served)):
                lo = sum(src_bags[k][:s])
                parts.append(outs[k][lo:lo + src_bags[k][s]].reshape(-1))
            send_out.append(t.cat(parts) if parts else t.empty(0, dtype=t.float32))
        out_in = self._move(send_out, t.float32, out_counts, worst_out)

        # 4. assemble per table; row-split partials are added in shard (rank) order
        result = [None] * T
        if rr_out is not None:
            for j, k in enumerate(self.split_tables):
                result[k] = rr_out[j]
        for j, u in enumerate(self.local):
            result[u.table] = outs[len(self.served) + j]
        for d in range(self.world):
            cur = 0
            for u in self.send_units[d]:
                nb = n_bags[u.table]
                part = out_in[d][cur:cur + nb * D].reshape(nb, D).to(self.device)
                cur += nb * D
                result[u.table] = part if result[u.table] is None else result[u.table] + part
        return result


# ------------------------------------------------------------------------------------------------
class RowRangeExchange:
    """Pooled lookups over tables split by ROW RANGE across the ranks of a torch.distributed group: routing on the
    GPU, counts first, payload second (include/pimemb.h: emb_route_bags / emb_unroute_bags; SURVEY.md section 8 row E;
    the reference sends its lengths before every launch, emb_host.h:280-287).

    One step, four phases (each only enqueues work, except the one host wait in `send_requests`):

        route(slot, spec)      every bag cut into per-shard sub-bags; the per-(peer, table) counts leave FIRST
        send_requests(slot)    host reads the counts, request pieces travel as all_to_all with split sizes from them
        serve(slot)            ONE fused engine lookup over every piece received -> one partial row per sub-bag,
                               returned with split sizes from the same counts
        finish(slot, out)      partial rows added in shard order from +0 into out[k][b][:]

    `forward` runs them back to back; a pipelined caller (dist_bench.run_rows) interleaves the phases of consecutive
    batches over several slots.  Nothing has a capacity that skewed indices could overflow.  Shard k of this rank is
    engine table `shard_table_ids[k]` holding rows [rank*rows_per_shard[k], (rank+1)*rows_per_shard[k])."""

    def __init__(self, engine, shard_table_ids, rows_per_shard, dim: int, rank: int, world: int, device,
                 n_slots: int = 1, group=None, stage_cpu: bool = False, native=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine, self.ids, self.rps = engine, list(shard_table_ids), [int(r) for r in rows_per_shard]
        self.K, self.N, self.dim, self.rank = len(self.ids), int(world), int(dim), int(rank)
        self.device, self.group, self.stage_cpu, self.native = torch.device(device), group, stage_cpu, native
        self.side = torch.cuda.Stream(self.device)
        self.slots = [dict() for _ in range(n_slots)]
        self.work = None
        self._streams = {}          # every stream the exchange's buffers have been used on (cuda_stream handle -> Stream)
        self.compute = None         # the caller's compute stream (a torch.cuda.Stream), when it has told us: a pipelined
                                    # caller sets it once; forward() sets it per call.  None: asked from torch every time
                                    # (torch.cuda.current_stream costs ~6 us a call, three of them per step)
        if self.K == 0 or self.K > 64:
            raise ValueError("1..64 row-split tables per exchange")

    # ---- buffers ------------------------------------------------------------------------------------
    def _cur(self):
        return self.compute if self.compute is not None else self.torch.cuda.current_stream(self.device)

    def _retire(self, buf):
        """A buffer about to be dropped may still be read or written by work queued on ANY stream the exchange uses
        (router stream, compute stream, the host-copy side stream): tell the caching allocator, so the block is not
        handed out again before that work has run."""
        if buf is not None and buf.is_cuda:
            for st in self._streams.values():
                buf.record_stream(st)

    def _use_stream(self, stream):
        self._streams.setdefault(stream.cuda_stream, stream)

    def _grown(self, sl, name, n, dtype, tail=()):
        """Grow-only device buffer of >= n leading elements (25 % headroom): sized by the counts, never fixed."""
        buf = sl.get(name)
        if buf is None or buf.shape[0] < n:
            self._retire(buf)
            sl[name] = buf = self.torch.empty((n + n // 4 + 16,) + tuple(tail), dtype=dtype, device=self.device)
        return buf

    def _prepare(self, sl, n_bags, total_indices):
        """Buffers of one route call.  Runs on the stream the router runs on (see route): the zero fills of a slot's
        first use are then ordered before the router kernels and the counts exchange that write the same words."""
        t = self.torch
        sz = self.engine.route_bags_sizes(self.K, max(n_bags, 1), total_indices, self.N)
        self._grown(sl, "req_send", sz["send"] // 4, t.int32)
        self._grown(sl, "slotmap", sz["slots"] // 4, t.int32)
        if "meta" not in sl:
            sl["meta"] = t.zeros(sz["meta"] // 4, dtype=t.int32, device=self.device)
            sl["counts_in"] = t.zeros((self.N, self.K + 1, 2), dtype=t.int32, device=self.device)
            sl["counts_host"] = t.zeros((2, self.N, self.K + 1, 2), dtype=t.int32).pin_memory()   # [0] sent, [1] received
            sl["counts_ev"] = t.cuda.Event()
        if self.work is None or self.work.numel() < sz["work"]:
            self._retire(self.work)
            self.work = t.empty(sz["work"] + sz["work"] // 4, dtype=t.uint8, device=self.device)   # scratch of one route call

    def _exchange(self, recv, send, out_splits, in_splits, rounds: int = 1):
        """all_to_all of leading-dimension ranges (RCCL; gloo stages through the host).  Work handle or None.
        rounds: how many transfers the largest piece of the JOB needs (every rank passes the same number: it comes from
        the peaks in the counts messages, see send_requests)."""
        t, dist = self.torch, self.dist
        n_out, n_in = int(sum(out_splits)), int(sum(in_splits))
        if self.native is not None:      # stream-ordered on the compute stream (emb_comm_all_to_all cuts large pieces itself)
            import ctypes as C
            esz = send.element_size()
            for d in send.shape[1:]:
                esz *= int(d)
            offs = lambda sp: (C.c_uint64 * (self.N + 1))(*([0] + [int(x) * esz for x in _cumsum(sp)]))
            self.native.all_to_all(send.data_ptr(), offs(in_splits), recv.data_ptr(), offs(out_splits),
                                   self._cur().cuda_stream)
            return None
        if self.stage_cpu:
            r = t.empty((n_out,) + tuple(recv.shape[1:]), dtype=recv.dtype)
            dist.all_to_all_single(r, send[:n_in].cpu(), output_split_sizes=list(out_splits),
                                   input_split_sizes=list(in_splits), group=self.group)
            recv[:n_out].copy_(r)
            return None
        return all_to_all_rounds(dist, recv[:n_out], send[:n_in], out_splits, in_splits, rounds, group=self.group)

    # ---- phases -------------------------------------------------------------------------------------
    def route(self, slot: int, spec, n_bags: int, total_indices: int, stream=None) -> None:
        """spec: EmbeddingEngine.route_tables([...]) over this rank's K index arrays (uint32 row ids).
        stream: a torch.cuda.Stream to run the router and the counts exchange on (they depend on nothing of the current
        step, so a pipelined caller lets them overlap the lookups); the caller orders it behind the slot's last reader.
        The slot's buffers are allocated AND zero-filled on that stream, so their first use is ordered too; a buffer
        that is outgrown is retired with record_stream on every stream the exchange has used.
        n_bags == 0 (a rank with an empty batch) is a valid participant: it sends zero counts and still serves."""
        t, sl = self.torch, self.slots[slot]
        cur = self._cur()
        run_on = stream if stream is not None else cur
        for st in (cur, run_on, self.side):
            if st.cuda_stream not in self._streams:
                self._use_stream(st)
        sl["n_bags"] = n_bags

        def enqueue():
            self._prepare(sl, n_bags, total_indices)
            h = run_on.cuda_stream
            if n_bags:
                self.engine.route_bags(spec, n_bags, self.N, sl["req_send"].data_ptr(), sl["meta"].data_ptr(),
                                       sl["slotmap"].data_ptr(), self.work.data_ptr(), h)
            else:
                sl["meta"].zero_()           # nothing to ask for: all counts (and peaks) zero
            counts_out = sl["meta"][:2 * self.N * (self.K + 1)].view(self.N, self.K + 1, 2)
            sl["counts_work"] = self._exchange(sl["counts_in"], counts_out, [1] * self.N, [1] * self.N)

        sl["routed_ev"] = None
        if stream is None:
            enqueue()
        else:
            with t.cuda.stream(stream):
                enqueue()
                sl["routed_ev"] = t.cuda.Event()
                sl["routed_ev"].record(stream)

    def send_requests(self, slot: int) -> None:
        """The one host wait of a step: learn the counts, then send the request pieces sized by them."""
        t, sl = self.torch, self.slots[slot]
        K = self.K
        counts_out = sl["meta"][:2 * self.N * (K + 1)].view(self.N, K + 1, 2)
        if sl.get("counts_work") is not None:
            with t.cuda.stream(self.side):    # off the compute stream: the host waits for the router + counts only
                sl["counts_work"].wait()
                sl["counts_host"][0].copy_(counts_out, non_blocking=True)
                sl["counts_host"][1].copy_(sl["counts_in"], non_blocking=True)
                sl["counts_ev"].record(self.side)
            sl["counts_ev"].synchronize()
            sl["counts_work"] = None
        else:
            if sl.get("routed_ev") is not None:     # the router ran on another stream (no collective handle to wait on)
                self._cur().wait_event(sl["routed_ev"])
            sl["counts_host"][0].copy_(counts_out, non_blocking=True)
            sl["counts_host"][1].copy_(sl["counts_in"], non_blocking=True)
            self._cur().synchronize()
        # split sizes and the job's largest pieces from the counts, in the library (emb_route_exchange_sizes): every rank
        # put its largest piece in every counts message, so the maximum over the messages received (my own included) is
        # the JOB's largest piece -- the same number of rounds on every rank, nobody left in a collective
        out_w, in_w, back, served, peak_req, peak_ret = self.engine.route_exchange_sizes(
            sl["counts_host"].data_ptr(), K, self.N, self.dim)
        sl["req_out_words"], sl["req_in_words"] = out_w, in_w
        sl["ret_rows_back"], sl["ret_rows_served"] = back, served      # partial rows each shard returns to me / I return
        sl["req_rounds"], sl["ret_rounds"] = rounds_for(peak_req), rounds_for(peak_ret)
        recv = self._grown(sl, "req_recv", sum(in_w), t.int32)
        sl["req_work"] = self._exchange(recv, sl["req_send"], sl["req_in_words"], sl["req_out_words"], sl["req_rounds"])

    def lookup_received(self, slot: int) -> int:
        """The fused lookup over every request piece received for this slot (no exchange).  Returns its algorithmic
        bytes.  The N x K descriptors are laid out from the received counts inside the library (emb_route_serve_descs)."""
        t, sl = self.torch, self.slots[slot]
        ret = self._grown(sl, "ret_send", int(sum(sl["ret_rows_served"])), t.float32, (self.dim,))
        received = sl["counts_host"].data_ptr() + 8 * self.N * (self.K + 1)          # counts_host[1]
        return self.engine.lookup_served(received, self.K, self.N, self.dim, self.ids, sl["req_recv"].data_ptr(),
                                         ret.data_ptr(), self._cur().cuda_stream)

    def sent_counts(self, slot: int):
        """{n_sub, n_idx} per (peer, table) of the requests this rank sent for the slot: int64 [N][K][2] (reporting)."""
        return self.slots[slot]["counts_host"][0].numpy().view("uint32")[:, :self.K, :].astype("int64")

    def serve(self, slot: int) -> int:
        t, sl = self.torch, self.slots[slot]
        if sl.get("req_work") is not None:
            sl["req_work"].wait()
            sl["req_work"] = None
        nbytes = self.lookup_received(slot)
        back = self._grown(sl, "ret_recv", int(sum(sl["ret_rows_back"])), t.float32, (self.dim,))
        sl["ret_work"] = self._exchange(back, sl["ret_send"], sl["ret_rows_back"], sl["ret_rows_served"], sl["ret_rounds"])
        return nbytes

    def finish(self, slot: int, out) -> None:
        """out: float32 [K, n_bags, dim] on the device."""
        sl = self.slots[slot]
        if sl.get("ret_work") is not None:
            sl["ret_work"].wait()
            sl["ret_work"] = None
        if not sl["n_bags"]:
            return
        self.engine.unroute_bags(sl["ret_recv"].data_ptr(), sl["meta"].data_ptr(), sl["slotmap"].data_ptr(), self.K,
                                 sl["n_bags"], self.N, self.dim, out.data_ptr(),
                                 self._cur().cuda_stream)

    def wait_requests(self, slot: int) -> None:
        sl = self.slots[slot]
        if sl.get("req_work") is not None:
            sl["req_work"].wait()
            sl["req_work"] = None

    # ---- one whole step -----------------------------------------------------------------------------
    def forward(self, indices, offsets=None, fixed_pooling: int = 0, out=None, slot: int = 0):
        """indices[k]: this rank's index array of row-split table k (CUDA int32 = uint32 row ids); offsets[k]: bag starts
        (int32, same bag count for every table) or None with fixed_pooling.  Returns float32 [K, n_bags, dim]."""
        t = self.torch
        if offsets is not None:
            n_bags = int(offsets[0].numel())
            if any(int(o.numel()) != n_bags for o in offsets):
                raise ValueError("every row-split table needs the same number of bags in one exchange (one batch)")
            spec = [(i.data_ptr(), o.data_ptr(), i.numel(), 0, r) for i, o, r in zip(indices, offsets, self.rps)]
        else:
            n_bags = int(indices[0].numel()) // int(fixed_pooling)
            spec = [(i.data_ptr(), None, i.numel(), fixed_pooling, r) for i, r in zip(indices, self.rps)]
        if out is None:
            out = t.empty((self.K, n_bags, self.dim), dtype=t.float32, device=self.device)
        pinned = self.compute
        if pinned is None:
            self.compute = t.cuda.current_stream(self.device)      # asked once for the four phases
        try:
            self.route(slot, self.engine.route_tables(spec), n_bags, sum(int(i.numel()) for i in indices))
            self.send_requests(slot)
            self.serve(slot)
            self.finish(slot, out)
        finally:
            self.compute = pinned
        return out


# RCCL 2.26.6 (bundled with torch 2.10) delivers only the first half of a single send / receive above 1 GiB
# (tools/a2a_size_probe.py: intact at 1.0 GiB, corrupt from 1.1 GiB, whatever the element type).  The native exchange
# (emb_comm_all_to_all) cuts every pair's transfer into 512-MiB pieces itself; through torch.distributed a large piece is
# moved in several rounds (all_to_all_rounds).  The number of rounds is a JOB-wide decision -- every rank must enter the
# same collectives -- so it is derived from a number all ranks share: static shapes (dist_bench.run_whole), or the largest
# piece of the job, which every rank learns from the counts messages (RowRangeExchange: the peaks entry of emb_route_bags'
# counts; ShardedLookup: the head message of every step).  check_piece_sizes is the last line of defence.
A2A_MAX_PIECE_BYTES = 1 << 30
A2A_ROUND_BYTES = 512 << 20
try:        # PIMEMB_A2A_ROUND_BYTES: a small round size makes ordinary payloads take several agreed rounds (tests)
    import os as _os
    if _os.environ.get("PIMEMB_A2A_ROUND_BYTES"):
        A2A_ROUND_BYTES = max(16, int(_os.environ["PIMEMB_A2A_ROUND_BYTES"]))
except ValueError:
    pass


def check_piece_sizes(splits, bytes_per_item: int, what: str = "all_to_all") -> None:
    worst = max((int(x) for x in splits), default=0) * int(bytes_per_item)
    if worst > A2A_MAX_PIECE_BYTES:
        raise RuntimeError(f"{what}: a {worst / 2**30:.2f}-GiB piece for one peer -- RCCL 2.26 corrupts single transfers above "
                           "1 GiB; use the native exchange (emb_comm_all_to_all cuts them), more ranks or a smaller batch")


def rounds_for(max_piece_bytes: int) -> int:
    """Rounds all_to_all_rounds needs so that no piece of a round exceeds A2A_ROUND_BYTES.  Every rank must arrive at the
    SAME number: compute it from shapes all ranks share, never from one rank's own counts."""
    return max(1, -(-int(max_piece_bytes) // A2A_ROUND_BYTES))


class _Works:
    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


def _item_bytes(x) -> int:
    n = x.element_size()
    for d in x.shape[1:]:
        n *= int(d)
    return n


def all_to_all_rounds(dist, recv, send, out_splits, in_splits, rounds: int, group=None):
    """all_to_all over leading-dimension ranges (splits count leading-dimension items) in `rounds` rounds: round r moves
    bytes [r*A2A_ROUND_BYTES, (r+1)*A2A_ROUND_BYTES) of every pair's piece.  rounds == 1 is one all_to_all_single.  Every
    rank of the group must pass the SAME `rounds` (rounds_for of a number all ranks share)."""
    item = _item_bytes(send)
    if rounds <= 1:
        check_piece_sizes(list(out_splits) + list(in_splits), item, "all_to_all_single")
        return dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits),
                                      group=group, async_op=True)
    in_off, out_off = [0] + _cumsum(in_splits), [0] + _cumsum(out_splits)
    step = max(1, A2A_ROUND_BYTES // item)
    works = []
    for r in range(rounds):
        ins = [send[in_off[p] + min(r * step, n):in_off[p] + min((r + 1) * step, n)] for p, n in enumerate(in_splits)]
        outs = [recv[out_off[p] + min(r * step, n):out_off[p] + min((r + 1) * step, n)] for p, n in enumerate(out_splits)]
        works.append(dist.all_to_all(outs, ins, group=group, async_op=True))
    return _Works(works)


def _cumsum(xs):
    acc, out = 0, []
    for x in xs:
        acc += int(x)
        out.append(acc)
    return out
