"""bench.py's N > 1 leg: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI),
weak scaling -- every rank owns B bags per table, so the global batch is N * B.

Placement policy (`run`): tables are REPLICATED while the whole table set fits a quarter of one
GPU's HBM (the 26 Kaggle tables, 2.16 GB, do): `run_dp` -- the single-GPU fused launch on every rank,
no data-path collective.  Larger sets (or --replicate-mb N) SHARD and exchange:

`run_whole` (static shapes, one index per bag): tables <= the threshold replicated, the rest placed
whole on owner ranks by the shard planner.  The two exchanges of a lookup -- indices in, pooled rows
out -- are software-pipelined over consecutive batches so that ONE all_to_all per step carries both
the pooled rows of batch i and the indices of batch i+1 (byte payloads with static splits):

    launch A(i): fused lookup of the replicated tables        HIP engine plan, local bags, no dependency
    wait collective(i-1)                                      stream-level, the CPU never blocks
    launch B(i): fused lookup of the tables served here       HIP engine plan over the bags of ALL ranks;
                 indices are read from collective(i-1)'s receive buffer, pooled rows are written
                 straight into collective(i)'s send buffer
    collective(i) = all_to_all([pooled rows of batch i | indices of batch i+1])   RCCL over xGMI
    outputs(i): replicated tables -> own buffers; sharded tables -> views of the receive buffer

`run_rows`: the big tables split by ROW RANGE over all ranks, any number of indices per bag.  Every bag
is cut into per-shard sub-bags on the GPU (emb_route_bags); the per-(peer, table) counts are exchanged
FIRST and the request pieces / partial rows travel as all_to_all with split sizes taken from them
(alltoallv -- nothing has a capacity that skewed indices could overflow); the bag's owner adds the
partial rows in shard order (emb_unroute_bags).  Pipelined over consecutive batches like `run_whole`.

With the auto policy the sharded exchange is still measured in the same run as a secondary leg
(`sharded_exchange` in the JSON line).  All buffers and engine plans are created once per rotating
batch slot."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np


TABLE_SCALE: dict[int, float] = {}     # table id -> 2/sqrt(rows): DLRM's U(-sqrt(1/n), sqrt(1/n)) range, set by run()
TABLE_F16 = [False]                    # run(): the table set is stored as fp16 (c5); expected values round the same way


XGMI_EGRESS_GBS = 7 * 153.0     # MI355X_MICROARCH.md / SURVEY.md section 8 row E: 7 xGMI links x ~153 GB/s per GPU


def step_fractions(alg_bytes_per_rank: int, bytes_out_per_rank: int, ms_per_step: float, hbm_peak_gbs: float) -> dict:
    """Step-level yardsticks of a sharded leg: what the whole step -- routing, collectives, un-routing
    included -- achieves against the two bounds it can hit.  alg_bytes_per_rank: the lookup's algorithmic bytes for this
    rank's own bags (as if the tables were local: the exchange is overhead, not payload);
    bytes_out_per_rank: bytes this rank hands to OTHER ranks per step (requests + returned rows + counts)."""
    t = ms_per_step * 1e-3
    out = {"step_algorithmic_bytes_per_rank": int(alg_bytes_per_rank), "bytes_out_per_rank_per_step": int(bytes_out_per_rank)}
    if t > 0:
        out["step_GBps"] = alg_bytes_per_rank / t / 1e9
        out["step_frac"] = out["step_GBps"] / hbm_peak_gbs
        out["xgmi_GBps"] = bytes_out_per_rank / t / 1e9
        out["xgmi_frac"] = out["xgmi_GBps"] / XGMI_EGRESS_GBS
        out["xgmi_peak_GBps"] = XGMI_EGRESS_GBS
    return out


def lookup_bytes(T: int, B: int, L: int, dim: int, elem: int) -> int:
    """Algorithmic bytes of one rank's step: T tables x B bags x (L rows + L u32 indices + one u32 offset + one fp32 row)."""
    return T * B * (L * (dim * elem + 4) + 4 + dim * 4)


def job_times(torch, dist, t0: float, t_event: float, stage_cpu: bool, dev):
    """Both closing brackets of a timed region, MAX over ranks (N = 1 and N > 1 on the same clocks).
    t_event: perf_counter when the event behind this rank's K-th step had fired (polled / waited on the compute stream);
    the device-wide torch.cuda.synchronize() -- the contract's bracket -- is taken here, inside the second clock.
    Returns (event seconds, sync seconds)."""
    import time as _t
    torch.cuda.synchronize()
    t_sync = _t.perf_counter()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([t_event - t0, t_sync - t0], dtype=torch.float64, device="cpu" if stage_cpu else dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)   # job time = the slowest rank's, all ranks having started together
    return float(el[0].item()), float(el[1].item())


def clock_fields(wall_event: float, wall_sync: float, steps: int, units: float) -> dict:
    """ms_per_step / value on the primary clock plus both clocks spelled out.  Primary: `sync`, the contract's bracket
    (the clock stops after the device-wide torch.cuda.synchronize()), as on bench.py's N = 1 line, so that the per-N
    values a SCALE record compares sit on ONE clock.  `event` stops when the event behind the K-th step has fired: the
    difference is the runtime's synchronize latency (24 us of a 0.42-ms region with RCCL loaded and one rank, round 3;
    round 2 once saw 0.8 ms), not the path's -- PIMEMB_CLOCK=event makes it the primary one."""
    primary = os.environ.get("PIMEMB_CLOCK", "sync")
    wall = wall_sync if primary == "sync" else wall_event
    return {"value": units / wall, "ms_per_step": wall * 1000.0 / steps, "clock": primary,
            "ms_per_step_event": wall_event * 1000.0 / steps, "ms_per_step_sync": wall_sync * 1000.0 / steps,
            "value_event": units / wall_event, "value_sync": units / wall_sync}


def table_values(torch, t: int, row_lo: int, row_hi: int, dim: int, device):
    """Deterministic table contents any rank can recompute: W_t[r][c] from a hash of (t, r, c), in DLRM's
    U(-sqrt(1/n), sqrt(1/n)) range (the scale the 1e-6 tolerance of load_generator.c:58 is meant for).
    Chunked to bound temporaries."""
    out = torch.empty((row_hi - row_lo, dim), dtype=torch.float32, device=device)
    step = 1 << 22
    scale = TABLE_SCALE.get(t, 1.0)
    for lo in range(row_lo, row_hi, step):
        hi = min(lo + step, row_hi)
        e = torch.arange(lo * dim, hi * dim, dtype=torch.int64, device=device)
        h = (e * 2654435761 + (t + 1) * 40503) % 2147483647
        out[lo - row_lo:hi - row_lo] = ((h.to(torch.float32) / 2147483647.0 - 0.5) * scale).reshape(hi - lo, dim)
    return out.to(torch.float16) if TABLE_F16[0] else out


def expected_rows(torch, t: int, idx, dim: int):
    """Rows idx of table t recomputed from the formula (fp32, same ops as table_values)."""
    e = idx.to(torch.int64)[:, None] * dim + torch.arange(dim, dtype=torch.int64, device=idx.device)[None, :]
    h = (e * 2654435761 + (t + 1) * 40503) % 2147483647
    v = (h.to(torch.float32) / 2147483647.0 - 0.5) * TABLE_SCALE.get(t, 1.0)
    return v.to(torch.float16).to(torch.float32) if TABLE_F16[0] else v     # fp16 rows, fp32 accumulate


def expected_pooled(torch, t: int, idx, dim: int, L: int):
    """Pooled rows of fixed-size bags, summed in index order in fp32 (what the kernel does): bit-exact."""
    rows = expected_rows(torch, t, idx, dim).view(-1, L, dim)
    acc = rows[:, 0, :] + 0.0
    for j in range(1, L):
        acc = acc + rows[:, j, :]
    return acc


def native_exchange(pel, args, eng, ctx):
    """--collective native: the all-to-all goes straight to RCCL on the compute stream (emb_comm_*),
    no torch work object, no extra waits.  None for the default (torch.distributed)."""
    if getattr(args, "collective", "torch") != "native":
        return None
    if ctx["stage_cpu"]:
        raise SystemExit("--collective native needs the nccl backend (RCCL), not " + ctx["backend"])
    import torch.distributed as dist

    def bcast(raw: bytes) -> bytes:
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    return pel.NativeExchange(eng, ctx["rank"], ctx["world"], bcast)


def table_set_of(pel, args):
    """rows, dim, default batch, label of the table set the N > 1 legs run (--workload c2 | c4)."""
    name = getattr(args, "workload", "c2")
    if name not in ("c2", "c4", "c5"):
        raise SystemExit("bench.py --gpus N > 1 runs --workload c2, c4 or c5 (c3 is a single-GPU line)")
    return pel.workloads.table_set(name, float(getattr(args, "rows_scale", 1.0) or 1.0))


def run_whole(args, hbm_peak_gbs: float, ctx, rep_bytes: int):
    import torch
    import torch.distributed as dist
    import pim_embedding_lookup_amd as pel
    from importlib import import_module
    sh = import_module("pim-embedding-lookup_amd.sharding")

    rank, world, dev, backend, stage_cpu = ctx["rank"], ctx["world"], ctx["dev"], ctx["backend"], ctx["stage_cpu"]

    rows_list, dim, B0, label = table_set_of(pel, args)
    B = args.batch or B0
    L = pooling_of(pel, args)               # indices per bag (fixed pooling)
    Bp = (B * L + 3) // 4 * 4               # index slots per (table, rank): keeps every piece 16-B aligned
    T = len(rows_list)
    # how many batches ahead the indices travel: with 1 (default), launch B(i+1) needs collective(i) -- a chain
    # B -> collective -> B -> ...; with 2 it needs collective(i-1), which has had a whole step to finish.  Tried in round 3
    # (PIMEMB_WHOLE_DEPTH=2 / 3): 71 / 63 us per step against 63 with one RCCL rank -- the step is bound by the host time
    # of all_to_all_single, not by that chain (profiles/r02/REJECTED_EXPERIMENTS.md), so the shallower pipeline stays
    DEPTH = max(1, int(os.environ.get("PIMEMB_WHOLE_DEPTH", "1")))
    NBATCH = max(DEPTH + 1, args.nbatch)
    plan = sh.plan_shards(rows_list, dim, 2 if TABLE_F16[0] else 4, world, replicate_bytes=rep_bytes, split_bytes=1 << 62)
    served = plan.owned_units(rank)
    local = plan.replicated_units()
    send_units = [[u for u in plan.units if u.owner == d] for d in range(world)]
    n_send = [len(x) for x in send_units]           # sharded tables owned by each destination
    n_sharded = sum(n_send)
    K = len(served)

    eng = pel.EmbeddingEngine(device=dev.index, max_tables=len(plan.units) + 1)
    for u in served + local:
        w = table_values(torch, u.table, u.row_lo, u.row_hi, dim, dev)
        eng.load_table(u.uid, w)
        del w
    torch.cuda.empty_cache()

    # ---- byte layout of the fused collective ----------------------------------------------------
    # to destination d  : [ K tables x B x dim fp32 pooled rows of d's bags | n_send[d] tables x Bp u32 indices ]
    # from source s     : [ n_send[s] tables x B x dim fp32 pooled rows of MY bags | K tables x Bp u32 indices of s's bags ]
    row_b = dim * 4
    in_split = [K * B * row_b + n_send[d] * Bp * 4 for d in range(world)]
    out_split = [n_send[s] * B * row_b + K * Bp * 4 for s in range(world)]
    in_off = np.concatenate([[0], np.cumsum(in_split)]).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum(out_split)]).astype(np.int64)
    # the largest piece any rank sends any peer, from the plan every rank holds (so all ranks agree on the rounds)
    owned = [sum(1 for u in plan.units if u.owner == r) for r in range(world)]
    a2a_rounds = sh.rounds_for(max(owned[r] * B * row_b + owned[d] * Bp * 4 for r in range(world) for d in range(world)))

    def f32_view(buf, byte_off, n_rows):
        return buf[byte_off:byte_off + n_rows * row_b].view(torch.float32).view(n_rows, dim)

    def i32_view(buf, byte_off, n):
        return buf[byte_off:byte_off + n * 4].view(torch.int32)

    rng = np.random.default_rng(1 + rank)
    gen, dist_name = index_generator(pel, args)
    off_dev = torch.arange(B, dtype=torch.int32, device=dev) * L
    idx_host = [[gen(rng, n, B * L, t).view(np.int32) for t, n in enumerate(rows_list)] for _ in range(NBATCH)]
    slots = []
    for j in range(NBATCH):
        send = torch.zeros(max(int(in_off[-1]), 16), dtype=torch.uint8, device=dev)
        recv = torch.zeros(max(int(out_off[-1]), 16), dtype=torch.uint8, device=dev)   # zeros: index 0 is valid
        slots.append(dict(send=send, recv=recv))
    for j in range(NBATCH):
        sl, nxt = slots[j], (j + DEPTH) % NBATCH
        # indices of the batch DEPTH steps ahead ride in this slot's send buffer (static, written once)
        for d in range(world):
            base = int(in_off[d]) + K * B * row_b
            for q, u in enumerate(send_units[d]):
                i32_view(sl["send"], base + q * Bp * 4, B * L).copy_(torch.from_numpy(idx_host[nxt][u.table]))
        sl["idx_local"] = {u.table: torch.from_numpy(idx_host[j][u.table]).to(dev) for u in local}
        sl["out_local"] = {u.table: torch.empty((B, dim), dtype=torch.float32, device=dev) for u in local}
        sl["plan_a"] = None
        if local:
            sl["plan_a"] = eng.plan([u.uid for u in local], [sl["idx_local"][u.table] for u in local],
                                    [off_dev] * len(local), [sl["out_local"][u.table] for u in local])
    for j in range(NBATCH):
        sl, prev = slots[j], slots[(j - DEPTH) % NBATCH]
        sl["plan_b"] = None
        if K:   # indices of batch j arrived with collective(j-DEPTH); pooled rows go into collective(j)
            ids, ii, oo, uu = [], [], [], []
            for s in range(world):
                for k, u in enumerate(served):
                    ids.append(u.uid)
                    ii.append(i32_view(prev["recv"], int(out_off[s]) + n_send[s] * B * row_b + k * Bp * 4, B * L))
                    oo.append(off_dev)
                    uu.append(f32_view(sl["send"], int(in_off[s]) + k * B * row_b, B))
            sl["plan_b"] = eng.plan(ids, ii, oo, uu)

    stream = torch.cuda.current_stream(dev)
    sh_handle = stream.cuda_stream
    from collections import deque
    pending = deque()            # work handles of the collectives in flight, oldest first (at most DEPTH)
    native = native_exchange(pel, args, eng, ctx)
    if native is not None:
        a_in, a_out = native.offsets(in_off), native.offsets(out_off)

    def collective(sl):
        if n_sharded == 0:
            return None
        if native is not None:      # stream-ordered: the next launch on this stream sees the received bytes
            native.all_to_all(sl["send"].data_ptr(), a_in, sl["recv"].data_ptr(), a_out, sh_handle)
            return None
        if stage_cpu:
            r, s_ = torch.empty(sl["recv"].shape, dtype=torch.uint8), sl["send"].cpu()
            dist.all_to_all_single(r, s_, output_split_sizes=out_split, input_split_sizes=in_split)
            sl["recv"].copy_(r)
            return None
        return sh.all_to_all_rounds(dist, sl["recv"], sl["send"], out_split, in_split, a2a_rounds)

    def step(i):
        sl = slots[i % NBATCH]
        if sl["plan_a"] is not None:
            sl["plan_a"].launch(sh_handle)          # independent of the exchange
        if len(pending) >= DEPTH:
            w = pending.popleft()
            if w is not None:
                w.wait()                            # compute stream waits for collective(i-DEPTH)
        if sl["plan_b"] is not None:
            sl["plan_b"].launch(sh_handle)
        pending.append(collective(sl))

    done_ev = torch.cuda.Event()

    def drain():
        while pending:
            w = pending.popleft()
            if w is not None:
                w.wait()
        done_ev.record(stream)       # everything of the loop is ordered before this event on the compute stream
        done_ev.synchronize()        # (a device-wide synchronize returns up to a millisecond later once RCCL is loaded)

    def outputs(sl):
        res = [None] * T
        for u in local:
            res[u.table] = sl["out_local"][u.table]
        for s in range(world):
            for q, u in enumerate(send_units[s]):
                res[u.table] = f32_view(sl["recv"], int(out_off[s]) + q * B * row_b, B)
        return res

    # ---- prime the pipeline (one full rotation), then check two consecutive steps bit-exactly:
    #      every table on every rank (one-hot => pooled row == table row) ---------------------------
    for i in range(NBATCH):
        step(i)
    for i in (NBATCH, NBATCH + 1):
        step(i)
        drain()
        res = outputs(slots[i % NBATCH])
        for t in range(T):
            idx = torch.from_numpy(idx_host[i % NBATCH][t]).to(dev)
            if not torch.equal(res[t], expected_pooled(torch, t, idx, dim, L)):
                raise AssertionError(f"rank {rank}: step {i} table {t} ({plan.kinds[t]}) differs from the expected rows")
    n_primed = NBATCH + 2

    # ---- kernel-only time of this rank's two launches, rotating over the batch slots (roofline) ----
    kernel_us, alg_bytes = 0.0, 0
    for key in ("plan_a", "plan_b"):
        if slots[0][key] is None:
            continue
        alg_bytes += slots[0][key].bytes()[0]
        for i in range(8):
            slots[i % NBATCH][key].launch(sh_handle)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_rep = 64
        e0.record(stream)
        for i in range(n_rep):
            slots[i % NBATCH][key].launch(sh_handle)
        e1.record(stream)
        torch.cuda.synchronize()
        kernel_us += e0.elapsed_time(e1) * 1000.0 / n_rep

    # NOTE: steps are enqueued eagerly.  The engine's launches are hipGraph-capturable (tests/
    # test_gpu_parity.py::test_plan_launch_is_graph_capturable), but capturing RCCL's all_to_all with
    # this torch 2.10 / RCCL 2.26 build segfaults in capture_end (tools/graph_probe.py), so the
    # collective keeps the step out of a graph.
    it = n_primed
    for _ in range(args.warmup):
        step(it)
        it += 1
    drain()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(it)
        it += 1
    drain()
    wall_ev, wall_sync = job_times(torch, dist, t0, time.perf_counter(), stage_cpu, dev)   # drain waited for this rank's K-th step
    # what the timed loop left behind: the last step's 26 outputs, bit for bit
    res = outputs(slots[(it - 1) % NBATCH])
    for t in range(T):
        idx = torch.from_numpy(idx_host[(it - 1) % NBATCH][t]).to(dev)
        if not torch.equal(res[t], expected_pooled(torch, t, idx, dim, L)):
            raise AssertionError(f"rank {rank}: last timed step, table {t} ({plan.kinds[t]}) differs from the expected rows")

    result = None
    if rank == 0:
        ach = alg_bytes / (kernel_us * 1e-6) / 1e9 if kernel_us > 0 else 0.0
        clk = clock_fields(wall_ev, wall_sync, args.steps, world * args.steps * T * B)
        bytes_out = int(sum(in_split[d] for d in range(world) if d != rank))
        fr = step_fractions(lookup_bytes(T, B, L, dim, 2 if TABLE_F16[0] else 4), bytes_out, clk["ms_per_step"], hbm_peak_gbs)
        result = ({
            "metric": "pooled-lookups/sec + achieved HBM GB/s, 26-table dim-16 Kaggle, 1/2/4/8 GPU",
            "unit": "pooled-lookups/s", **clk,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if TABLE_F16[0] else "f32", "data": "synthetic",
            "config": {"workload": "%s sharded, dim %d %s, B=%d bags/table PER RANK, "
                                   "L=%d, %s indices, %d rotating batches; %s" % (label, dim, "fp16" if TABLE_F16[0] else "fp32", B, L, dist_name, NBATCH, plan.describe()),
                       "tables": T, "dim": dim, "bags_per_table_per_rank": B, "global_bags_per_table": world * B,
                       "pooling": L,
                       "parallelism": "tables sharded by id (replicate <= %d MiB); one all_to_all per step carries "
                                      "pooled rows of batch i + indices of batch i+%d (%d B out / %d B in per rank); "
                                      "backend %s, %s, eager steps" %
                                      (rep_bytes >> 20, DEPTH, int(in_off[-1]), int(out_off[-1]), backend,
                                       "collective issued natively to RCCL on the compute stream" if native is not None
                                       else "torch.distributed.all_to_all_single"),
                       "exchange": {"mode": "whole", "value": clk["value"], "ms_per_step": clk["ms_per_step"], "verified": True,
                                    "bytes_out_per_rank_per_step": bytes_out}},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
                         "frac": ach / hbm_peak_gbs, "traffic": None, "kernel_us": kernel_us,
                         "algorithmic_bytes": alg_bytes,
                         "basis": "algorithmic bytes (no PMC profile of the N > 1 legs; a pooled launch is partly cache-served and can exceed 1: the same kernels at N = 1 are priced on measured bytes, profiles/traffic.json)", "note": "rank 0's two local launches (replicated + served tables), kernel-only; the whole step (collective included) against HBM and xGMI: roofline.exchange",
                         "exchange": fr},
        })
    dist.barrier()
    for sl in slots:
        for p in (sl["plan_a"], sl["plan_b"]):
            if p is not None:
                p.destroy()
    if native is not None:
        torch.cuda.synchronize()
        native.close()
    eng.close()
    return result


def expected_row_split(torch, t: int, idx, dim: int, L: int, rps: int, n_shards: int):
    """What the row-range sharded path returns for fixed-size bags, bit for bit: per shard d the rows of the bag
    that live in d, summed in index order from +0 (the serving rank's kernel), the per-shard partial sums then
    added in shard order from +0 (emb_unroute_bags).  Adding +0.0 where a row belongs to another shard leaves
    every intermediate value unchanged, so the masked adds below reproduce the kernels' arithmetic exactly."""
    rows = expected_rows(torch, t, idx, dim).view(-1, L, dim)
    dest = torch.clamp(idx.to(torch.int64) // rps, max=n_shards - 1).view(-1, L)
    acc = torch.zeros_like(rows[:, 0, :])
    zero = torch.zeros_like(acc)
    for d in range(n_shards):
        part = torch.zeros_like(acc)
        for j in range(L):
            part = part + torch.where((dest[:, j] == d)[:, None], rows[:, j, :], zero)
        acc = acc + part
    return acc


def index_generator(pel, args):
    """gen(rng, n_rows, count, table=t) and the distribution's name: uniform, zipf, or the table set's default
    ("mixed": Zipf(1.2) on even tables, uniform on odd ones)."""
    extras = pel.workloads.TABLE_SET_EXTRAS[getattr(args, "workload", "c2")]
    dist_name = getattr(args, "index_dist", None) or extras["dist"]

    def gen(rng, n, count, table=0):
        zipf = dist_name == "zipf" or (dist_name == "mixed" and table % 2 == 0)
        return (pel.workloads.zipf_indices if zipf else pel.workloads.uniform_indices)(rng, n, count)

    return gen, dist_name


def pooling_of(pel, args) -> int:
    extras = pel.workloads.TABLE_SET_EXTRAS[getattr(args, "workload", "c2")]
    return max(1, int(getattr(args, "pooling", None) or extras["pooling"]))


def dump_row_split(torch, eng, args, gen, rank, world, dev, rows_list, sharded, rps, T, B, L, dim, NBATCH, last, idx_host, slots):
    """PIMEMB_DUMP_ROWSPLIT=<prefix>: leave what a TEST needs to put the oracle behind the sharded path (nothing under the
    package may import it).  Every rank writes <prefix>.rank<r>.npz with, per row-split table, the rows of ITS shard that
    rank 0's bags of the last timed step name -- read back from the engine's table in HBM -- and rank 0 adds its index
    arrays (global row ids) and the pooled rows the exchange returned.  Other ranks regenerate rank 0's indices from its seed."""
    prefix = os.environ.get("PIMEMB_DUMP_ROWSPLIT")
    if not prefix or not sharded:
        return
    if rank == 0:
        idx0 = idx_host[last]
    else:
        rng0 = np.random.default_rng(1 + 0)
        idx0 = [[gen(rng0, n, B * L, t).view(np.int32) for t, n in enumerate(rows_list)] for _ in range(NBATCH)][last]
    rec = {"tables": np.asarray(sharded), "pooling": np.asarray(L), "bags": np.asarray(B), "dim": np.asarray(dim),
           "rows_per_shard": np.asarray(rps), "world": np.asarray(world)}
    for k, t in enumerate(sharded):
        ids = np.unique(idx0[t].view(np.uint32).astype(np.int64))
        lo = rank * rps[k]
        hi = rows_list[t] if rank == world - 1 else min(lo + rps[k], rows_list[t])
        mine = ids[(ids >= lo) & (ids < hi)]
        w = eng.table_tensor(T + k)
        rec["ids_%d" % t] = mine
        rec["rows_%d" % t] = (w[torch.from_numpy(mine - lo).to(dev)].float().cpu().numpy() if mine.size
                              else np.zeros((0, dim), np.float32))
        if rank == 0:
            rec["idx_%d" % t] = idx0[t].view(np.uint32)
            rec["out_%d" % t] = slots[last]["out_sh"][k].cpu().numpy()
    np.savez("%s.rank%d.npz" % (prefix, rank), **rec)


def run_rows(args, hbm_peak_gbs: float, ctx, rep_bytes: int):
    """Big tables split by ROW RANGE over all ranks, any number of indices per bag.  Counts first, payload second
    (the reference sends its lengths before every launch, emb_host.h:280-287):

        route(i+2)      GPU: every bag of a row-split table cut into per-shard sub-bags (emb_route_bags)
        counts(i+2)     all_to_all of {sub-bags, indices} per (peer, table)             -- small, first, two batches ahead
        local(i)        fused lookup of the replicated tables (prepared plan)
        serve(i)        fused lookup over the request pieces received for batch i -> one partial row per sub-bag
        return(i)       all_to_all of the partial rows, split sizes from counts(i)
        requests(i+1)   all_to_all of the request pieces, split sizes from counts(i+1)  -- the only host wait
        finish(i)       partial rows added in shard order into [B, dim] per table (emb_unroute_bags) -- enqueued one step
                        later, in front of serve(i+1), so the compute stream never waits out return(i)

    Nothing has a capacity that skewed indices could overflow: the payload is sized by the counts."""
    import torch
    import torch.distributed as dist
    import pim_embedding_lookup_amd as pel
    from importlib import import_module
    sh = import_module("pim-embedding-lookup_amd.sharding")

    rank, world, dev, backend, stage_cpu = ctx["rank"], ctx["world"], ctx["dev"], ctx["backend"], ctx["stage_cpu"]

    rows_list, dim, B0, label = table_set_of(pel, args)
    L = pooling_of(pel, args)
    gen, dist_name = index_generator(pel, args)
    row_b = dim * 4
    B = args.batch or B0
    T = len(rows_list)
    N = world
    NBATCH = max(4, args.nbatch)            # slots: the router runs two batches ahead of the un-router
    sharded = [t for t in range(T) if rows_list[t] * dim * (2 if TABLE_F16[0] else 4) > rep_bytes and rows_list[t] >= world]
    local = [t for t in range(T) if t not in sharded]
    K = len(sharded)
    rps = [-(-rows_list[t] // world) for t in sharded]

    eng = pel.EmbeddingEngine(device=dev.index, max_tables=T + K + 1)
    for t in local:
        eng.load_table(t, table_values(torch, t, 0, rows_list[t], dim, dev))
    for k, t in enumerate(sharded):
        lo, hi = min(rank * rps[k], rows_list[t]), min((rank + 1) * rps[k], rows_list[t])
        eng.load_table(T + k, table_values(torch, t, lo, hi, dim, dev))
    torch.cuda.empty_cache()

    rng = np.random.default_rng(1 + rank)
    idx_host = [[gen(rng, n, B * L, t).view(np.int32) for t, n in enumerate(rows_list)] for _ in range(NBATCH)]
    off_b = torch.arange(B, dtype=torch.int32, device=dev) * L
    stream = torch.cuda.current_stream(dev)
    h = stream.cuda_stream
    native = native_exchange(pel, args, eng, ctx)
    # the exchange itself is library code (sharding.RowRangeExchange); this leg only pipelines its four phases
    ex = sh.RowRangeExchange(eng, [T + k for k in range(K)], rps, dim, rank, world, dev, n_slots=NBATCH,
                             stage_cpu=stage_cpu, native=native) if K else None
    if ex is not None:
        ex.compute = stream          # every phase of this leg is issued from this (the current) stream

    slots = []
    for j in range(NBATCH):
        sl = dict(idx_local=[torch.from_numpy(idx_host[j][t]).to(dev) for t in local],
                  out_local=[torch.empty((B, dim), dtype=torch.float32, device=dev) for _ in local])
        sl["plan_a"] = eng.plan(local, sl["idx_local"], [off_b] * len(local), sl["out_local"]) if local else None
        if K:
            sl["idx_sh"] = torch.from_numpy(np.stack([idx_host[j][t] for t in sharded])).to(dev)     # [K, B*L]
            sl["route_spec"] = eng.route_tables([(sl["idx_sh"][k].data_ptr(), None, B * L, L, rps[k]) for k in range(K)])
            sl["out_sh"] = torch.zeros((K, B, dim), dtype=torch.float32, device=dev)
        slots.append(sl)

    prof = {} if os.environ.get("PIMEMB_DIST_PROFILE") == "1" else None

    def timed(name, fn, *a):
        if prof is None:
            return fn(*a)
        t = time.perf_counter_ns()
        r = fn(*a)
        prof[name] = prof.get(name, 0) + time.perf_counter_ns() - t
        return r

    unfinished = [None]        # slot whose partial rows are on their way back (its un-routing is the next step's job)
    # the router and its counts exchange depend on nothing of the step they are issued in: they run on their own stream,
    # ordered behind the last reader of the slot they fill (the un-router of NBATCH batches ago)
    route_stream = torch.cuda.Stream(dev) if (K and os.environ.get("PIMEMB_ROUTE_STREAM", "1") != "0" and native is None) else None
    slot_free = [torch.cuda.Event() for _ in range(NBATCH)]

    def route_into(slot):
        if route_stream is not None:
            route_stream.wait_event(slot_free[slot])
        ex.route(slot, slots[slot]["route_spec"], B, K * B * L, stream=route_stream)

    def finish_pending():
        if unfinished[0] is not None:
            timed("finish", ex.finish, unfinished[0], slots[unfinished[0]]["out_sh"])
            slot_free[unfinished[0]].record(stream)
            unfinished[0] = None

    def step(i):
        """One pipelined step.  On the compute stream: route(i+2), local(i), finish(i-1), serve(i).
          * the router runs TWO batches ahead: the counts the host needs for batch i+1 were sent a whole step ago, so its
            one wait per step (send_requests) does not depend on how far the slowest peer has got in THIS step.  (With one
            rank it changes nothing -- 246 vs 262 us per step at the C4 shape, inside the noise: there the step is the
            sum of its kernels, all in one hardware queue, the 50-us self-copy of the return collective included.)
          * the partial rows of batch i-1 have had a whole step to come back, so un-routing them never stalls the stream
            behind a collective."""
        j, nxt, nxt2 = i % NBATCH, (i + 1) % NBATCH, (i + 2) % NBATCH
        sl = slots[j]
        if K:
            timed("route+counts", route_into, nxt2)
        if sl["plan_a"] is not None:
            timed("local", sl["plan_a"].launch, h)
        if K:
            finish_pending()
            timed("serve+return", ex.serve, j)
            timed("wait counts+requests", ex.send_requests, nxt)
            unfinished[0] = j

    def prologue(i):           # before step(i): batch i's requests on their way, batch i+1 routed and its counts sent
        if K:
            route_into(i % NBATCH)
            ex.send_requests(i % NBATCH)
            route_into((i + 1) % NBATCH)

    done_ev = torch.cuda.Event()

    def drain(next_i):         # un-route the last batch; the requests of the batch after it are in flight: let them land
        if K:
            finish_pending()
            ex.wait_requests(next_i % NBATCH)
        done_ev.record(stream)       # everything of the loop is ordered before this event on the compute stream
        done_ev.synchronize()        # (a device-wide synchronize returns up to a millisecond later once RCCL is loaded)

    def outputs(j):
        res = [None] * T
        for q, t in enumerate(local):
            res[t] = slots[j]["out_local"][q]
        for k, t in enumerate(sharded):
            res[t] = slots[j]["out_sh"][k]
        return res

    def verify(i, what):
        res = outputs(i % NBATCH)
        for t in range(T):
            idx = torch.from_numpy(idx_host[i % NBATCH][t]).to(dev)
            plain = expected_pooled(torch, t, idx, dim, L)
            want = plain if t in local else expected_row_split(torch, t, idx, dim, L, rps[sharded.index(t)], N)
            if not torch.equal(res[t], want):
                raise AssertionError(f"rank {rank}: {what} step {i} table {t} differs from the expected rows")
            if float((res[t] - plain).abs().max()) > 1e-6:
                raise AssertionError(f"rank {rank}: {what} step {i} table {t} is more than 1e-6 from the unsharded sum")

    # ---- the first rotation, fully pipelined (every slot's buffers are used for the first time here: ADVICE r2 --
    #      an un-ordered first use shows up in these steps and in no later one), checked afterwards from the slots' own
    #      output buffers; then two consecutive pipelined steps, each checked on its own -------------------------------
    prologue(0)
    for i in range(NBATCH):
        step(i)
    drain(NBATCH)
    for i in range(NBATCH):
        verify(i, "first rotation")
    for i in (NBATCH, NBATCH + 1):
        step(i)
        drain(i + 1)
        verify(i, "pipelined")
    it = NBATCH + 2

    # ---- kernel-only time of this rank's two lookups (replicated tables + served request pieces) ----------
    kernel_us, alg_bytes = 0.0, 0
    if slots[0]["plan_a"] is not None:
        alg_bytes += slots[0]["plan_a"].bytes()[0]
    if K:                                # requests for batch `it` were issued by the last step: let them land
        ex.wait_requests(it % NBATCH)
        torch.cuda.synchronize()
    serve_bytes = 0
    for key in ("local", "serve"):
        if (key == "local" and slots[0]["plan_a"] is None) or (key == "serve" and not K):
            continue
        fn = (lambda q: slots[q % NBATCH]["plan_a"].launch(h)) if key == "local" else (lambda q: ex.lookup_received(it % NBATCH))
        for q in range(4):
            serve_bytes = fn(q) or serve_bytes
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for q in range(32):
            fn(q)
        e1.record(stream)
        torch.cuda.synchronize()
        kernel_us += e0.elapsed_time(e1) * 1000.0 / 32
    alg_bytes += serve_bytes

    if prof is not None:
        prof.clear()               # (the priming steps allocate and pin the slots' buffers: not what a step costs)
    for _ in range(args.warmup):
        step(it)
        it += 1
    drain(it)
    dist.barrier()
    torch.cuda.synchronize()
    check_every = int(os.environ.get("PIMEMB_VERIFY_EVERY", "0"))   # soak mode: verify inside the loop (times mean nothing then)
    t0 = time.perf_counter()
    for n in range(args.steps):
        step(it)
        it += 1
        if check_every and (n + 1) % check_every == 0:
            drain(it)
            verify(it - 1, "soak")
    drain(it)
    wall_ev, wall_sync = job_times(torch, dist, t0, time.perf_counter(), stage_cpu, dev)   # drain waited for this rank's K-th step
    verify(it - 1, "last timed")                     # what the timed loop left behind
    dump_row_split(torch, eng, args, gen, rank, world, dev, rows_list, sharded, rps, T, B, L, dim, NBATCH, (it - 1) % NBATCH,
                   idx_host, slots)
    digest = None
    if rank == 0 and K:                              # bits of rank 0's row-split outputs of that step: two runs must agree
        import hashlib
        hsh = hashlib.sha1()
        for k in range(K):
            hsh.update(slots[(it - 1) % NBATCH]["out_sh"][k][:4096].contiguous().cpu().numpy().tobytes())
        digest = hsh.hexdigest()
    if prof is not None and rank == 0:
        n_calls = args.steps + args.warmup
        print("[dist_bench] host microseconds per step by call:",
              {k: round(v / 1e3 / n_calls, 1) for k, v in prof.items()}, flush=True)

    result = None
    if rank == 0:
        ach = alg_bytes / (kernel_us * 1e-6) / 1e9 if kernel_us > 0 else 0.0
        sent = ex.sent_counts((it - 1) % NBATCH) if K else np.zeros((N, 1, 2), np.int64)
        clk = clock_fields(wall_ev, wall_sync, args.steps, world * args.steps * T * B)
        bytes_out = 0
        if K:        # what rank 0 handed to OTHER ranks in the last step: counts message, request pieces, returned partial rows
            last = ex.slots[(it - 1) % NBATCH]
            bytes_out = sum(8 * (K + 1) + 4 * int(last["req_out_words"][d]) + row_b * int(last["ret_rows_served"][d])
                            for d in range(N) if d != rank)
        fr = step_fractions(lookup_bytes(T, B, L, dim, 2 if TABLE_F16[0] else 4), bytes_out, clk["ms_per_step"], hbm_peak_gbs)
        result = ({
            "metric": "pooled-lookups/sec + achieved HBM GB/s, 26-table dim-16 Kaggle, 1/2/4/8 GPU",
            "unit": "pooled-lookups/s", **clk,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if TABLE_F16[0] else "f32", "data": "synthetic",
            "config": {"workload": "%s sharded, dim %d %s, B=%d bags/table PER RANK, L=%d, %s indices, "
                                   "%d rotating batches; %d tables replicated (<= %d MiB), %d row-range sharded over "
                                   "%d ranks" % (label, dim, "fp16" if TABLE_F16[0] else "fp32", B, L, dist_name, NBATCH, len(local), rep_bytes >> 20, K, world),
                       "tables": T, "dim": dim, "bags_per_table_per_rank": B, "global_bags_per_table": world * B,
                       "pooling": L, "index_dist": dist_name,
                       "parallelism": "row-range shards; bags cut into per-shard sub-bags on the GPU; counts first "
                                      "(all_to_all of {sub-bags, indices} per peer and table), then the request pieces "
                                      "and the partial rows as all_to_all with split sizes from the counts; partial "
                                      "rows added in shard order; backend %s, %s, eager steps" %
                                      (backend, "collectives issued natively to RCCL on the compute stream"
                                       if native is not None else "torch.distributed.all_to_all_single"),
                       "last_step_outputs_sha1": digest,
                       "last_step_request_rows_per_peer": sent[:, :, 0].sum(axis=1).tolist(),
                       "last_step_request_indices_per_peer": sent[:, :, 1].sum(axis=1).tolist(),
                       "exchange": {"mode": "rows", "value": clk["value"], "ms_per_step": clk["ms_per_step"], "verified": True,
                                    "bytes_out_per_rank_per_step": bytes_out}},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
                         "frac": ach / hbm_peak_gbs, "traffic": None, "kernel_us": kernel_us,
                         "algorithmic_bytes": alg_bytes,
                         "basis": "algorithmic bytes (no PMC profile of the N > 1 legs; a pooled launch is partly cache-served and can exceed 1: the same kernels at N = 1 are priced on measured bytes, profiles/traffic.json)", "note": "rank 0's two lookup launches (replicated tables + served request pieces), kernel-only; the whole step (routing, three collectives, un-routing) against HBM and xGMI: roofline.exchange",
                         "exchange": fr},
        })
    dist.barrier()
    for sl in slots:
        if sl["plan_a"] is not None:
            sl["plan_a"].destroy()
    if native is not None:
        torch.cuda.synchronize()
        native.close()
    eng.close()
    return result


def run_dp(args, hbm_peak_gbs: float, ctx):
    """Every table replicated on every rank: each rank looks its own B bags up locally -- the
    single-GPU fused launch, N times, no data-path collective."""
    import torch
    import torch.distributed as dist
    import pim_embedding_lookup_amd as pel
    rank, world, dev, stage_cpu = ctx["rank"], ctx["world"], ctx["dev"], ctx["stage_cpu"]
    rows_list, dim, B0, label = table_set_of(pel, args)
    B = args.batch or B0
    L = pooling_of(pel, args)
    T = len(rows_list)
    NBATCH = max(2, args.nbatch)
    eng = pel.EmbeddingEngine(device=dev.index, max_tables=T)
    for t in range(T):
        eng.load_table(t, table_values(torch, t, 0, rows_list[t], dim, dev))
    torch.cuda.empty_cache()
    rng = np.random.default_rng(1 + rank)
    off = torch.arange(B, dtype=torch.int32, device=dev) * L
    gen, dist_name = index_generator(pel, args)
    idx_host = [[gen(rng, n, B * L, t).view(np.int32) for t, n in enumerate(rows_list)] for _ in range(NBATCH)]
    plans = []
    for j in range(NBATCH):      # one offsets array PER TABLE, as the reference passes them and as the N = 1 run holds them
        plans.append(eng.plan(list(range(T)), [torch.from_numpy(i).to(dev) for i in idx_host[j]],
                              [off.clone() for _ in range(T)]))
    stream = torch.cuda.current_stream(dev)
    h = stream.cuda_stream
    plans[0].launch(h)
    torch.cuda.synchronize()
    for t in range(T):       # parity on every rank: one-hot => pooled row == table row
        idx = torch.from_numpy(idx_host[0][t]).to(dev)
        if not torch.equal(plans[0].outputs[t], expected_pooled(torch, t, idx, dim, L)):
            raise AssertionError(f"rank {rank}: table {t} differs from the expected rows")
    # device pre-warm (untimed, before the W warm-up steps), as in the single-GPU run: a fresh process starts with idle clocks
    prewarm_ms, n_pre, t_pre = float(getattr(args, "prewarm_ms", 250.0) or 0.0), 0, time.perf_counter()
    while time.perf_counter() - t_pre < prewarm_ms * 1e-3:
        for _ in range(64):
            plans[n_pre % NBATCH].launch(h)
            n_pre += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        plans[i % NBATCH].launch(h)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dbg = [] if os.environ.get("PIMEMB_DIST_PROFILE") == "1" else None
    t0 = time.perf_counter()
    e0.record(stream)
    if dbg is not None: dbg.append(("e0.record", time.perf_counter() - t0))
    for i in range(args.steps):
        plans[i % NBATCH].launch(h)
        if dbg is not None and i in (0, 1, args.steps - 1): dbg.append((f"launch {i}", time.perf_counter() - t0))
    e1.record(stream)
    if dbg is not None: dbg.append(("e1.record", time.perf_counter() - t0))
    while not e1.query():                   # polled: a blocking wait is woken ~20 us late
        pass
    t_event = time.perf_counter()           # this rank's K steps are complete (the event after the K-th launch has fired)
    wall_ev, wall_sync = job_times(torch, dist, t0, t_event, stage_cpu, dev)
    if dbg is not None and rank == 0:
        print("[dist_bench] run_dp timed region, seconds since t0:", dbg, "event", t_event - t0, "job event / sync",
              wall_ev, wall_sync, flush=True)
    kernel_us = e0.elapsed_time(e1) * 1000.0 / args.steps
    alg_bytes = plans[0].bytes()[0]
    last = (args.steps - 1) % NBATCH if args.steps > 0 else 0
    for t in range(T):       # what was just timed: the last batch's outputs on every rank, bit for bit
        idx = torch.from_numpy(idx_host[last][t]).to(dev)
        if not torch.equal(plans[last].outputs[t], expected_pooled(torch, t, idx, dim, L)):
            raise AssertionError(f"rank {rank}: last timed batch, table {t} differs from the expected rows")
    result = None
    if rank == 0:
        ach = alg_bytes / (kernel_us * 1e-6) / 1e9
        clk = clock_fields(wall_ev, wall_sync, args.steps, world * args.steps * T * B)
        result = {
            "metric": "pooled-lookups/sec + achieved HBM GB/s, 26-table dim-16 Kaggle, 1/2/4/8 GPU",
            "unit": "pooled-lookups/s", **clk,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if TABLE_F16[0] else "f32", "data": "synthetic",
            "config": {"workload": "%s, dim %d %s, B=%d bags/table PER RANK, L=%d, u32 "
                                   "indices+offsets, %s indices, %d rotating batches" % (label, dim, "fp16" if TABLE_F16[0] else "fp32", B, L, dist_name, NBATCH),
                       "tables": T, "dim": dim, "bags_per_table_per_rank": B, "global_bags_per_table": world * B,
                       "prewarm_ms": prewarm_ms, "prewarm_launches": n_pre,
                       "parallelism": "all %d tables (%.2f GB) replicated on every rank (they fit the per-GPU "
                                      "replication budget); bags data-parallel, no data-path collective"
                                      % (T, sum(rows_list) * dim * (2 if TABLE_F16[0] else 4) / 1e9)},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
                         "frac": ach / hbm_peak_gbs, "traffic": None, "kernel_us": kernel_us,
                         "algorithmic_bytes": alg_bytes, "basis": "algorithmic bytes (no PMC profile of the N > 1 legs; a pooled launch is partly cache-served and can exceed 1: the same kernels at N = 1 are priced on measured bytes, profiles/traffic.json)", "note": "rank 0's fused launch, HIP events over the timed region"},
        }
    for p in plans:
        p.destroy()
    eng.close()
    return result


def run(args, hbm_peak_gbs: float) -> None:
    """N > 1 entry.  Placement policy (--replicate-mb, default auto): tables are replicated while the
    whole set fits a quarter of one GPU's HBM -- the 26 Kaggle tables (2.16 GB) do, so the metric's
    config runs data-parallel with no exchange; anything larger shards (table id / row range) and
    exchanges indices and pooled rows with all_to_all.  With the auto policy the exchange path is
    still MEASURED in the same run, as a secondary leg forced to shard the five big tables
    (`sharded_exchange` in the JSON line): that is the xGMI all-to-all curve, link- and host-bound
    for one index per bag."""
    import threading
    import torch
    import torch.distributed as dist
    import pim_embedding_lookup_amd as pel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend = os.environ.get("PIMEMB_DIST_BACKEND", "nccl")
    # The job's stdout is for ONE JSON line.  RCCL prints a five-line banner ("RCCL version : ...") to stdout when a
    # communicator is created, and under torch.distributed.run a rank's stdout IS the job's: keep the real stdout aside
    # for the line and point file descriptor 1 at stderr for everything else (libraries included).
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # A rank that never comes back (a peer died before the rendezvous, a collective that cannot complete) must not hold
    # the job until the driver's own limit: after PIMEMB_RUN_TIMEOUT seconds (default 1800) it says so and leaves.
    def _give_up():
        sys.stderr.write("[dist_bench] rank %d: no result after %.0f s -- giving up\n" % (rank, run_limit))
        sys.stderr.flush()
        os._exit(5)
    run_limit = float(os.environ.get("PIMEMB_RUN_TIMEOUT", "1800"))
    run_dog = threading.Timer(run_limit, _give_up)
    run_dog.daemon = True
    run_dog.start()
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    one_gpu = os.environ.get("PIMEMB_RCCL_ONE_GPU") == "1"
    if one_gpu and backend == "nccl":
        # Rehearsal of the REAL RCCL path with several ranks on ONE GPU: RCCL refuses two ranks on a device of the same
        # host ("duplicate GPU"), so every rank claims a host of its own (NCCL_HOSTID) and, being "remote" to its peers, talks
        # to them through RCCL's socket transport over loopback -- slow, but the same communicator, the same grouped
        # send / receive calls and the same torch code path as on an 8-GPU node (tools/rccl_one_gpu_ranks_probe.py).
        os.environ.update(NCCL_HOSTID="pimemb-rank%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                          NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", NCCL_NET_GDR_LEVEL="0")
    if backend == "nccl" and n_dev < local_world and not one_gpu:
        raise SystemExit(f"bench.py --gpus {world}: {local_world} ranks on this node but {n_dev} GPU(s) visible -- RCCL needs one GPU "
                         "per rank (PIMEMB_RCCL_ONE_GPU=1 rehearses the RCCL path, PIMEMB_DIST_BACKEND=gloo the N > 1 code "
                         "with host-staged collectives, with several ranks on one GPU)")
    dev = torch.device("cuda", local_rank % max(n_dev, 1))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend, rank=rank, world_size=world,
                            **({"device_id": dev} if backend == "nccl" else {}))
    ctx = dict(rank=rank, world=world, dev=dev, backend=backend, stage_cpu=backend != "nccl")
    dist.barrier()       # the first collective sets the communicator's channels up (milliseconds): not right before a clock starts

    rows_list, dim, _, _ = table_set_of(pel, args)
    TABLE_SCALE.update({t: float(np.float32(2.0 / np.sqrt(n))) for t, n in enumerate(rows_list)})
    TABLE_F16[0] = pel.workloads.TABLE_SET_EXTRAS[getattr(args, "workload", "c2")]["dtype"] == "f16"
    total_bytes = sum(rows_list) * dim * (2 if TABLE_F16[0] else 4)
    hbm = torch.cuda.get_device_properties(dev).total_memory
    auto = getattr(args, "replicate_mb", None) is None
    mode = getattr(args, "shard_mode", None) or ("rows" if getattr(args, "workload", "c2") == "c4" else "whole")
    shard_leg = run_rows if mode == "rows" else run_whole
    state = {"printed": False, "dog": None, "primary": None}

    def emit(res):
        if rank == 0 and res is not None and not state["printed"]:
            state["printed"] = True
            os.write(json_fd, (json.dumps(res) + "\n").encode())

    def finish(res):
        """backend / rank count on every N > 1 line, so a SCALE record shows what RCCL saw."""
        if res is not None:
            res.setdefault("verified", True)     # every leg compares its outputs bit for bit before AND after timing
            res["config"]["backend"] = backend
            res["config"]["rccl_ranks"] = world if backend == "nccl" else 0
            if one_gpu and backend == "nccl":
                res["config"]["rccl_transport"] = "sockets over loopback, %d ranks on %d GPU(s) (PIMEMB_RCCL_ONE_GPU=1)" % (world, n_dev)
            res["config"]["world_size"] = world
        return res

    def die(code, note):
        """The exchange leg failed or hung on this rank: print what there is (rank 0), end NON-ZERO.  Peers that
        wait for this rank in a collective are ended by the launcher (bench.py's self-launch / torchrun)."""
        if rank == 0 and state.get("primary") is not None:
            state["primary"]["sharded_exchange"] = {"failed": note}
            state["primary"]["config"]["exchange"] = {"failed": note, "verified": False}
            state["primary"]["verified"] = False
            emit(finish(state["primary"]))
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)

    if auto and total_bytes <= hbm // 4:
        result = run_dp(args, hbm_peak_gbs, ctx)
        state["primary"] = result
        if not getattr(args, "no_exchange_leg", False):
            # secondary leg: the sharded exchange (the xGMI all-to-all curve), fewer steps.  A parity failure, an
            # exception or a hang in it FAILS the run: the primary line is still printed, the exit status is not 0.
            import copy
            a2 = copy.copy(args)
            a2.steps, a2.warmup = min(args.steps, 400), min(args.warmup, 40)
            limit = float(os.environ.get("PIMEMB_EXCHANGE_TIMEOUT", "180"))
            state["dog"] = threading.Timer(limit, die, (3, "timed out after %.0f s" % limit))
            state["dog"].daemon = True
            state["dog"].start()
            try:
                sec = shard_leg(a2, hbm_peak_gbs, ctx, 64 << 20)
            except BaseException as ex:  # noqa: BLE001 -- SystemExit from a leg included
                import traceback
                traceback.print_exc()
                die(4, f"{type(ex).__name__}: {ex}")
            state["dog"].cancel()
            if rank == 0:
                result["sharded_exchange"] = {k: sec[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_event",
                                                                  "ms_per_step_sync", "clock", "steps", "roofline")}
                result["sharded_exchange"]["verified"] = True
                result["sharded_exchange"]["config"] = sec["config"]["workload"] + "; " + sec["config"]["parallelism"]
                # the same numbers inside the two objects a SCALE record keeps (config / roofline): the all-to-all leg's
                # value is the xGMI curve north_star asks for, next to the replica curve in `value`
                result["config"]["exchange"] = dict(sec["config"]["exchange"], steps=sec["steps"],
                                                    what="secondary leg of the same run: " + sec["config"]["workload"])
                result["roofline"]["exchange"] = sec["roofline"]["exchange"]
    else:
        rep_mb = 64 if auto else int(args.replicate_mb)
        result = shard_leg(args, hbm_peak_gbs, ctx, rep_mb << 20)
    emit(finish(result))
    dist.barrier()
    dist.destroy_process_group()
    run_dog.cancel()
