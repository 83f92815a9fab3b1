"""bench.py's N > 1 leg: one process per GPU, weak scaling -- every rank owns B bags per table, so the global batch is
N * B.  torch.distributed (nccl = RCCL, or gloo) carries the bootstrap, the barriers and the job clock; the data path of
the sharded legs is the library's own (emb_shard_*: RCCL groups issued from C, csrc/pimemb_shard.cpp).

Placement policy (`run`): tables are REPLICATED while the whole table set fits a quarter of one GPU's HBM (the 26 Kaggle
tables, 2.16 GB, do): `run_dp` -- the single-GPU fused launch on every rank, no data-path collective.  Larger sets (or
--replicate-mb N) SHARD: `run_sharded` hands every batch to `ShardedEmbeddingBags.submit` -- ONE library call per batch,
as the reference's lookup() serves all its devices from one call (emb_host.h:258-270, 297, 312-321):

    --shard-mode whole   tables above the threshold placed whole on owner ranks (table-id sharding): the bags' indices travel
                         to the owner straight out of the caller's buffers, the pooled rows arrive straight in its output
    --shard-mode rows    ... split by ROW RANGE over all ranks, any number of indices per bag: every bag cut into per-shard
                         sub-bags on the GPU (emb_route_bags), the per-(peer, table) counts exchanged FIRST, the payload sized
                         from them (nothing has a capacity skew could overflow), partial rows added in shard order
    --shard-mode plan    what `plan_shards` decides (its return-volume term keeps pooled tables whole when they fit)

With the auto policy the sharded exchange is still measured in the same run as a secondary leg and reported at the top
level of the JSON line (`value_exchange`, `ms_per_step_exchange`) next to the replica curve in `value`."""
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


def lookup_bytes(T: int, B: int, L: int, dim: int, elem: int, isz: int = 4) -> int:
    """Algorithmic bytes of one rank's step: T tables x B bags x (L rows + L indices + one offset + one fp32 row)."""
    return T * B * (L * (dim * elem + isz) + isz + dim * 4)


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


def table_set_of(pel, args):
    """rows, dim, default batch, label of the table set the N > 1 legs run (--workload c2 | c4)."""
    name = getattr(args, "workload", "c2")
    if name not in ("c2", "c4", "c5"):
        raise SystemExit("bench.py --gpus N > 1 runs --workload c2, c4 or c5 (c3 is a single-GPU line)")
    return pel.workloads.table_set(name, float(getattr(args, "rows_scale", 1.0) or 1.0))


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


def dump_row_split(torch, S, plan, sh, gen, rank, world, dev, rows_list, T, B, L, dim, NBATCH, last, idx_host, outs):
    """PIMEMB_DUMP_ROWSPLIT=<prefix>: leave what a TEST needs to put the oracle behind the sharded path (nothing under the
    package may import it).  Every rank writes <prefix>.rank<r>.npz with, per row-split table, the rows of ITS shard that
    rank 0's bags of the last timed step name -- read back from the engine's table in HBM -- and rank 0 adds its index
    arrays (global row ids) and the pooled rows the exchange returned.  Other ranks regenerate rank 0's indices from its seed."""
    prefix = os.environ.get("PIMEMB_DUMP_ROWSPLIT")
    sharded = [t for t, k in enumerate(plan.kinds) if k == sh.ROW_SPLIT]
    if not prefix or not sharded:
        return
    if rank == 0:
        idx0 = idx_host[last]
    else:
        rng0 = np.random.default_rng(1 + 0)
        idx0 = [[gen(rng0, n, B * L, t).view(np.int32) for t, n in enumerate(rows_list)] for _ in range(NBATCH)][last]
    rps = [-(-rows_list[t] // world) for t in sharded]
    rec = {"tables": np.asarray(sharded), "pooling": np.asarray(L), "bags": np.asarray(B), "dim": np.asarray(dim),
           "rows_per_shard": np.asarray(rps), "world": np.asarray(world)}
    for k, t in enumerate(sharded):
        ids = np.unique(idx0[t].view(np.uint32).astype(np.int64))
        u = plan.units[plan.units_of_table[t][rank]]
        mine = ids[(ids >= u.row_lo) & (ids < u.row_hi)]
        w = S.engine.table_tensor(u.uid) if u.row_hi > u.row_lo else None
        rec["ids_%d" % t] = mine
        rec["rows_%d" % t] = (w[torch.from_numpy(mine - u.row_lo).to(dev)].float().cpu().numpy() if mine.size
                              else np.zeros((0, dim), np.float32))
        if rank == 0:
            rec["idx_%d" % t] = idx0[t].view(np.uint32)
            rec["out_%d" % t] = outs[t].cpu().numpy()
    np.savez("%s.rank%d.npz" % (prefix, rank), **rec)


def run_sharded(args, hbm_peak_gbs: float, ctx, rep_bytes: int, mode: str):
    """The sharded step, ONE library call per batch (ShardedEmbeddingBags over emb_shard_*: csrc/pimemb_shard.cpp; the
    reference's lookup() likewise serves every device from one call, emb_host.h:258-270, 297, 312-321).  This leg only
    builds the placement, rotates NBATCH static batches through `submit` (depth 3) and checks what comes out:

        mode "whole": tables above the replication threshold placed whole on owner ranks (table-id sharding);
        mode "rows" : ... split by ROW RANGE over all ranks (GPU routing, counts first, partial rows added in shard order);
        mode "plan" : whatever plan_shards decides (row-split only what must be; pooled tables that fit stay whole).

    All transfers are RCCL groups issued from the C side; torch.distributed carries the bootstrap, the barriers and the
    job clock only."""
    import torch
    import torch.distributed as dist
    import pim_embedding_lookup_amd as pel
    from importlib import import_module
    sh = import_module("pim-embedding-lookup_amd.sharding")

    rank, world, dev, backend = ctx["rank"], ctx["world"], ctx["dev"], ctx["backend"]
    rows_list, dim, B0, label = table_set_of(pel, args)
    L = pooling_of(pel, args)
    gen, dist_name = index_generator(pel, args)
    elem = 2 if TABLE_F16[0] else 4
    row_b = dim * 4
    B = args.batch or B0
    T = len(rows_list)
    NBATCH = max(6, args.nbatch)
    depth = int(os.environ.get("PIMEMB_SHARD_DEPTH", "3"))
    def make_plan(rows_):
        if mode == "whole":
            return sh.plan_shards(rows_, dim, elem, world, replicate_bytes=rep_bytes, split_bytes=1 << 62)
        if mode == "rows":
            return sh.plan_shards(rows_, dim, elem, world, replicate_bytes=rep_bytes, split_bytes=rep_bytes, pooling=1.0, split_single_rank=True)
        return sh.plan_shards(rows_, dim, elem, world, replicate_bytes=rep_bytes, pooling=float(L), split_single_rank=True)
    # Pre-flight: tables + batch slots + the library's staging against THIS GPU's HBM, on the worst rank -- before anything is
    # allocated.  A layout that does not fit is shrunk (rows, uniformly) and the line says so; C5 at 8 ranks needs it.
    transport = "peer" if getattr(args, "exchange", "rccl") == "peer" else "rccl"
    checked = bool(getattr(args, "checked", False))
    hbm_total = int(torch.cuda.get_device_properties(dev).total_memory)
    # (--exchange both: one layout for the two legs -- their outputs are compared bit for bit -- so the needier transport decides)
    fit_for = ("rccl", "peer") if getattr(args, "exchange_both", False) else (transport,)
    fit_scale, rows_list, plan, hbm_need = sh.fit_to_hbm(
        rows_list, hbm_total, make_plan,
        lambda p_, r_: max((sh.hbm_budget(p_, r_, B, L, NBATCH, depth, tr_, checked, index_bytes=8 if getattr(args, "ids", "uint32") == "int64" else 4)
                            for tr_ in fit_for), key=lambda d_: d_["total"]))
    if fit_scale < 1.0:
        label += " (rows x %.3f more: tables + batch slots + staging must fit %.0f GB of HBM)" % (fit_scale, hbm_total / 1e9)
        if rank == 0:
            print("[dist_bench] the layout as configured needs more than this GPU's HBM on its fullest rank: rows scaled by %.3f" % fit_scale,
                  file=sys.stderr)
    split = [t for t, k in enumerate(plan.kinds) if k == sh.ROW_SPLIT]
    whole = [t for t, k in enumerate(plan.kinds) if k == sh.WHOLE]
    rps = {t: -(-rows_list[t] // world) for t in split}

    eng = pel.EmbeddingEngine(device=dev.index, max_tables=len(plan.units) + 1)
    via = os.environ.get("PIMEMB_SHARD_SELF_VIA_COMM") == "1"      # rehearsal / A-B: self pieces through RCCL like any other
    use_peer = getattr(args, "exchange", "rccl") == "peer"
    peer = comm = None
    if use_peer:                  # the collective-free exchange: no RCCL communicator in the data path at all
        box = [os.urandom(6).hex()]
        if world > 1:
            dist.broadcast_object_list(box, src=0)         # one job tag for the shared-memory segment
        n_split = sum(1 for k in plan.kinds if k == sh.ROW_SPLIT)
        arena = int(1.25 * NBATCH * T * B * (L * 4 + dim * 4)) + 8 * 2 * n_split * B * (L * 8 + min(L, world) * dim * 4 * 2) + (256 << 20)
        arena = int(os.environ.get("PIMEMB_BENCH_PEER_ARENA_BYTES", arena))         # (test hook: an arena nobody can allocate)
        peer = sh.PeerGroup(eng, "bench-" + box[0], rank, world, arena_bytes=arena)
    else:
        comm = sh.native_comm(eng, rank, world, always=via)
    # --checked: EMB_SHARD_CHECK_SERVED -- what ShardedEmbeddingBags does by default for tensors it does not trust.  One-index
    # batches keep the direct path and COUNT what every shard serves (the requester compares); routed batches validate first.
    # (--checked: the deferred report, what check=True means; PIMEMB_BENCH_CHECK=sync compares inside the completing call, as round 5 did)
    S = sh.ShardedEmbeddingBags(plan, eng, rank, comm, depth=depth, check=(os.environ.get("PIMEMB_BENCH_CHECK", "deferred") if checked else False),
                                self_via_comm=via and not use_peer, peer=peer)
    S.load_tables(lambda t, lo, hi: table_values(torch, t, lo, hi, dim, dev))
    torch.cuda.empty_cache()

    rng = np.random.default_rng(1 + rank)
    # --ids int64: DLRM's dtype, handed to the library in place (emb_shard_input.index_type); default: the reference's uint32
    ids64 = getattr(args, "ids", "uint32") == "int64"
    isz, idt = (8, torch.int64) if ids64 else (4, torch.int32)
    idx_host = [[gen(rng, n, B * L, t).view(np.int32) for t, n in enumerate(rows_list)] for _ in range(NBATCH)]
    if ids64:
        idx_host = [[x.view(np.uint32).astype(np.int64) for x in b_] for b_ in idx_host]
    stream = torch.cuda.current_stream(dev)
    h = stream.cuda_stream
    slots = []
    for j in range(NBATCH):
        d_idx = [torch.from_numpy(idx_host[j][t]).to(dev) for t in range(T)]
        if peer is not None:      # peers gather from / store into these in place: they live in this rank's arena (one
            a_idx = [peer.empty(x.shape, idt) for x in d_idx]       # allocation per table: a chunk of it holds 1 GiB)
            for a, x in zip(a_idx, d_idx):
                a.copy_(x)
            d_idx = a_idx
            outs = [peer.empty((B, dim), torch.float32) for _ in range(T)]
            for o in outs:
                o.zero_()
        else:
            one = torch.zeros((T, B, dim), dtype=torch.float32, device=dev)
            outs = [one[t] for t in range(T)]
        slots.append(dict(prep=S.prepare(d_idx, None, L, outs), outs=outs))
    torch.cuda.synchronize()

    seqs = {}
    # Where the pooled rows are consumed.  "same" (default): on the caller's stream, as DLRM's interaction layer consumes
    # apply_emb's outputs -- stream order is the hand-over, emb_shard_wait has nothing to enqueue.  "other": a second stream
    # (a consumer that overlaps with the next batches' lookups): emb_shard_wait then records an event behind the batch's last
    # kernel and makes that stream wait for it -- one event between two kernels of the caller's stream per step, which
    # costs GPU time of its own (a few us on this runtime) whatever the sharding does.
    consumer_mode = os.environ.get("PIMEMB_BENCH_CONSUMER", "same")
    consumer = torch.cuda.Stream(dev) if consumer_mode == "other" else None
    hc = consumer.cuda_stream if consumer is not None else h

    def step(i):
        """One call per batch; the consumer's stream is made to wait for the batch that is `depth` submits old."""
        seqs[i] = S.submit_prepared(slots[i % NBATCH]["prep"], h)
        if i - depth in seqs:
            S.wait(seqs.pop(i - depth), hc)

    done_ev = torch.cuda.Event()

    def drain():
        S.flush()
        for i in sorted(seqs):
            S.wait(seqs[i], h)
        seqs.clear()
        done_ev.record(stream)       # everything of the loop is ordered before this event on the caller's stream
        done_ev.synchronize()        # (a device-wide synchronize returns up to a millisecond later once RCCL is loaded)

    def verify(i, what):
        res = slots[i % NBATCH]["outs"]
        for t in range(T):
            idx = torch.from_numpy(idx_host[i % NBATCH][t]).to(dev)
            plain = expected_pooled(torch, t, idx, dim, L)
            want = expected_row_split(torch, t, idx, dim, L, rps[t], world) if t in rps else plain
            if not torch.equal(res[t], want):
                raise AssertionError(f"rank {rank}: {what} step {i} table {t} ({plan.kinds[t]}) differs from the expected rows")
            if float((res[t] - plain).abs().max()) > 1e-6:
                raise AssertionError(f"rank {rank}: {what} step {i} table {t} is more than 1e-6 from the unsharded sum")

    # ---- the first rotation, fully pipelined (every slot's buffers are used for the first time here), checked afterwards
    #      from the slots' own output buffers; then two more pipelined steps, each checked on its own -------------------------
    it = 0
    for _ in range(NBATCH):
        step(it)
        it += 1
    drain()
    for i in range(NBATCH):
        verify(i, "first rotation")
        for o in slots[i]["outs"]:
            o.zero_()
    for _ in range(2):
        step(it)
        it += 1
        drain()
        verify(it - 1, "pipelined")

    # untimed pre-warm before the W warm-up steps, as in the replica leg and the single-GPU run (disclosed in config): a fresh
    # process starts with idle clocks, and a recurring launch is served by a prepared plan only from its third sighting on
    # (three rotations of the batch slots).  Rank 0's clock decides for everyone: submit / flush are collective.
    prewarm_ms, n_pre, t_pre = float(getattr(args, "prewarm_ms", 250.0) or 0.0), 0, time.perf_counter()
    while prewarm_ms > 0:
        for _ in range(4 * NBATCH + 1):        # (+ 1: the pipeline fills and drains at every slot phase in turn, as the timed region's will)
            step(it)
            it += 1
        n_pre += 4 * NBATCH + 1
        drain()
        go = torch.tensor([1 if time.perf_counter() - t_pre < prewarm_ms * 1e-3 else 0], dtype=torch.int32,
                          device=dev if backend == "nccl" else "cpu")
        dist.broadcast(go, 0)
        if int(go.item()) == 0:
            break
    for _ in range(args.warmup):
        step(it)
        it += 1
    drain()
    S.stats(reset=True)
    dist.barrier()
    torch.cuda.synchronize()
    check_every = int(os.environ.get("PIMEMB_VERIFY_EVERY", "0"))   # soak mode: verify inside the loop (times mean nothing then)
    step_clock = [] if os.environ.get("PIMEMB_BENCH_STEP_TIMES") == "1" else None     # where a short timed region's time goes (stderr)
    t0 = time.perf_counter()
    for n in range(args.steps):
        step(it)
        it += 1
        if step_clock is not None:
            step_clock.append(time.perf_counter())
        if check_every and (n + 1) % check_every == 0:
            drain()
            verify(it - 1, "soak")
    drain()
    if step_clock is not None and rank == 0:
        t_end = time.perf_counter()
        print("host us per submit: " + " ".join("%.1f" % ((b - a) * 1e6) for a, b in zip([t0] + step_clock[:-1], step_clock)) +
              " | drain %.1f us" % ((t_end - step_clock[-1]) * 1e6), file=sys.stderr)
    wall_ev, wall_sync = job_times(torch, dist, t0, time.perf_counter(), backend != "nccl", dev)   # drain waited for this rank's K-th step
    st = S.stats(reset=True)
    n_verified = 0
    for i in range(max(it - NBATCH, it - args.steps), it):         # what the timed loop left behind: EVERY rotating slot
        verify(i, "timed")
        n_verified += 1
    # The digests below are over ONE batch slot's outputs.  A slot's inputs never change, so what it holds does not depend on
    # how many steps ran; the slot is the one a run without pre-warm ends on (NBATCH + 2 checked steps, W, K), so the digest
    # of a command is the same whatever the (time-based) pre-warm did -- and comparable with earlier rounds' lines.
    last = (NBATCH + 2 + args.warmup + args.steps - 1) % NBATCH
    dump_row_split(torch, S, plan, sh, gen, rank, world, dev, rows_list, T, B, L, dim, NBATCH, last, idx_host, slots[last]["outs"])
    digest = None
    digest_all = None
    if rank == 0:                                    # bits of rank 0's outputs of that step: two runs / two transports must agree
        import hashlib
        if split:
            hsh = hashlib.sha1()
            for t in split:
                hsh.update(slots[last]["outs"][t][:4096].contiguous().cpu().numpy().tobytes())
            digest = hsh.hexdigest()
        hsh = hashlib.sha1()
        for t in split + whole:
            hsh.update(slots[last]["outs"][t].contiguous().cpu().numpy().tobytes())
        digest_all = hsh.hexdigest()
    sent = S.sent_counts(it - 1) if split else np.zeros((world, 1, 2), np.int64)      # (seq == step index: one submit per step)

    # ---- the kernels' own time: the same steps with the library's kernel brackets on (events cost GPU time: not in the timed loop)
    S.set_kernel_timing(True)
    n_k = 16
    for _ in range(n_k):
        step(it)
        it += 1
    drain()
    kst = S.stats(reset=True)
    S.set_kernel_timing(False)
    nt = max(int(kst["n_timed_batches"]), 1)
    k_route, k_local, k_serve, k_un = (kst["us_kernel_route"] / nt, kst["us_kernel_local"] / nt, kst["us_kernel_serve"] / nt,
                                       kst["us_kernel_unroute"] / nt)
    k_direct = kst["us_kernel_direct"] / nt          # the ranged one-hot lookups of the direct path (no router / un-router then)
    nb = max(int(kst["n_batches"]), 1)
    serve_bytes, local_bytes = kst["served_algorithmic_bytes"] / nb, kst["local_algorithmic_bytes"] / nb
    kernel_us, alg_bytes = k_local + k_serve + k_direct, int(serve_bytes + local_bytes)
    direct = k_direct > 0 and k_route == 0

    free_b, total_b = torch.cuda.mem_get_info(dev)        # what the GPU holds now, all processes on it: next to the pre-flight's estimate
    result = None
    if rank == 0:
        ach = alg_bytes / (kernel_us * 1e-6) / 1e9 if kernel_us > 0 else 0.0
        clk = clock_fields(wall_ev, wall_sync, args.steps, world * args.steps * T * B)
        n_st = max(int(st["n_batches"]), 1)
        bytes_out = int(st["bytes_to_peers"] / n_st)
        fr = step_fractions(lookup_bytes(T, B, L, dim, elem, isz), bytes_out, clk["ms_per_step"], hbm_peak_gbs)
        fr["host_us_per_step"] = st["us_host_submit"] / n_st
        fr["host_wait_counts_us_per_step"] = st["us_host_wait_counts"] / n_st
        fr["host_wait_served_us_per_step"] = st["us_host_wait_served"] / n_st
        # (L of the newest batch and S of the one being served share ONE launch in the steady state: priced together)
        kernels = {"router_us": k_route, "lookup_us": kernel_us, "unrouter_us": k_un, "direct_lookup_us": k_direct,
                   "lookup_algorithmic_bytes": alg_bytes, "served_algorithmic_bytes": int(serve_bytes),
                   "local_algorithmic_bytes": int(local_bytes),
                   "lookup_GBps": ach,
                   # un-router: reads one partial row per sub-bag + the slot words, writes one pooled row per bag
                   "unrouter_bytes": int(len(split) * B * row_b + (sent[:, :, 0].sum() if split else 0) * row_b),
                   # router: reads every index once, writes one offset + one row id per sub-bag and one slot word per bag
                   "router_bytes": int(len(split) * B * L * 4 + (sent[:, :, 0].sum() + sent[:, :, 1].sum() if split else 0) * 4 + len(split) * B * 4),
                   "timed_batches": nt}
        if k_un > 0:
            kernels["unrouter_GBps"] = kernels["unrouter_bytes"] / (k_un * 1e-6) / 1e9
        if k_route > 0:
            kernels["router_GBps"] = kernels["router_bytes"] / (k_route * 1e-6) / 1e9
        traffic = sharded_traffic_entry(args, mode, world, direct)
        if world == 1:       # every table is looked up here: compulsory table bytes of one step = every DISTINCT row once
            uniq = sum(int(np.unique(idx_host[0][t]).shape[0]) for t in range(T)) * dim * elem
        result = ({
            "metric": "pooled-lookups/sec + achieved HBM GB/s, 26-table dim-16 Kaggle, 1/2/4/8 GPU",
            "unit": "pooled-lookups/s", **clk,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if TABLE_F16[0] else "f32", "data": "synthetic",
            "verify": {"timed_batches_checked_bit_for_bit": n_verified, "what": "every table of every rotating slot the timed loop "
                       "wrote last, on every rank, after the timed region; the first rotation and two pipelined steps before it"},
            "config": {"workload": "%s sharded, dim %d %s, B=%d bags/table PER RANK, L=%d, %s indices, %d rotating batches; %s"
                                   % (label, dim, "fp16" if TABLE_F16[0] else "fp32", B, L, dist_name, NBATCH, plan.describe()),
                       "tables": T, "dim": dim, "bags_per_table_per_rank": B, "global_bags_per_table": world * B,
                       "pooling": L, "index_dist": dist_name, "shard_mode": mode, "pipeline_depth": depth,
                       "index_type": "int64" if ids64 else "uint32",
                       # one index per bag and no peer behind RCCL: row-split tables are not routed -- every shard scans the
                       # requesters' raw index arrays and serves the bags whose row it holds (PIMEMB_SHARD_DIRECT=0: always route)
                       "direct_one_hot_path": bool(direct),
                       "checked": checked,      # EMB_SHARD_CHECK_SERVED: served bags counted (direct path) / pieces validated (routed)
                       # what the fullest rank holds, GB (sharding.hbm_budget, computed before anything was allocated)
                       "hbm_budget_GB": {k: round(v / 1e9, 3) for k, v in hbm_need.items()}, "hbm_total_GB": round(hbm_total / 1e9, 1),
                       "rows_scale_to_fit": fit_scale, "hbm_in_use_GB_measured": round((total_b - free_b) / 1e9, 3),
                       # where the loop consumes a finished batch: the caller's stream (stream order is the hand-over) or a second
                       # stream (emb_shard_wait records an event between two kernels of the caller's stream every step)
                       "consumer_stream": consumer_mode,
                       "prewarm_ms": prewarm_ms, "prewarm_steps": n_pre,
                       "placement": {"replicated": plan.kinds.count(sh.REPLICATED), "whole": len(whole), "row_split": len(split),
                                     "rules": sorted(set(n for n in plan.notes if n))},
                       "parallelism": "ONE library call per batch (emb_shard_submit, depth %d): whole tables travel straight out "
                                      "of / into the caller's buffers, row-split tables are cut into per-shard sub-bags on the GPU, "
                                      "counts first, ONE fused lookup over everything a rank serves, partial rows added in shard "
                                      "order; transfers = %s, self pieces %s; control "
                                      "plane (bootstrap, barriers, job clock) torch.distributed/%s"
                                      % (depth, "NONE -- the owner's fused lookup gathers a requester's indices in place and stores its pooled "
                                                "rows straight into the requester's HBM (EMB_SHARD_PEER_STORES: HIP IPC mappings, handshake through "
                                                "a shared-memory segment)" if peer is not None else "RCCL groups issued from C (emb_comm_exchange)",
                                         "through RCCL too" if S._flags & 1 else "served in place", backend),
                       "last_step_outputs_sha1": digest, "last_step_sharded_outputs_sha1": digest_all,
                       "exchange_transport": "peer stores (HIP IPC, no RCCL in the data path; %s arena)" % ("fine-grained" if peer.info()["fine_grained"] else "ordinary device memory")
                                             if peer is not None else "RCCL groups issued from C",
                       "last_step_request_rows_per_peer": sent[:, :, 0].sum(axis=1).tolist(),
                       "last_step_request_indices_per_peer": sent[:, :, 1].sum(axis=1).tolist(),
                       "exchange": {"mode": mode, "value": clk["value"], "ms_per_step": clk["ms_per_step"], "verified": True,
                                    "bytes_out_per_rank_per_step": bytes_out, "host_us_per_step": fr["host_us_per_step"]}},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
                         "frac": ach / hbm_peak_gbs, "traffic": None, "kernel_us": kernel_us,
                         "algorithmic_bytes": alg_bytes, "basis": "algorithmic bytes",
                         "note": "rank 0's lookup launch per step (replicated tables of the newest batch + every piece received for "
                                 "the batch being served: ONE fused launch), HIP events inside the library (emb_shard_set_kernel_timing) over %d extra steps after "
                                 "the timed region (the brackets cost GPU time themselves: where a step is ONE launch -- the direct path, whole tables "
                                 "on their owner -- ms_per_step is the tighter bound on that launch); the whole step (routing, transfers, "
                                 "un-routing) against HBM and xGMI: roofline.exchange" % nt,
                         "kernels": kernels, "exchange": fr},
        })
        if world == 1:
            result["roofline"]["unique_row_bytes"], result["roofline"]["index_bytes"] = uniq, T * B * L * 4
        apply_sharded_traffic(result["roofline"], traffic, kernel_us, k_route, k_un)
    dist.barrier()
    torch.cuda.synchronize()
    S.close()
    if peer is not None:
        peer.close()
    if comm is not None:
        comm.close()
    eng.close()
    return result


def sharded_traffic_entry(args, mode, world, direct=False):
    """profiles/traffic.json entry of this sharded command (world-1 PMC passes: the ROUTED step's fused lookup, router and
    un-router, or -- key suffix -direct -- the direct path's one ranged launch; see profiles/collect_dist_pmc.sh), or None."""
    if world != 1 or args.batch is not None:
        return None          # the counters were collected with one rank: other pieces / other kernels otherwise
    rep_mb = getattr(args, "replicate_mb", None)
    if rep_mb is not None and int(rep_mb) != 64:          # (not given: the auto leg, which shards tables above 64 MiB)
        return None          # (... and with tables above 64 MiB sharded: another threshold is another placement)
    if getattr(args, "workload", "c2") == "c4" and abs(float(getattr(args, "rows_scale", 1.0) or 1.0) - 0.125) > 1e-9:
        return None          # (C4 is profiled at one of 8 ranks' share of the rows)
    key = "dist-%s-%s-l%d" % (getattr(args, "workload", "c2"), mode, int(getattr(args, "pooling", None) or 0) or 1)
    if getattr(args, "index_dist", None):
        key += "-" + args.index_dist
    if direct:
        key += "-direct"
    try:
        import bench
        return bench.measured_traffic(key, "bag_sum", lambda e: bench.shard_identity(e.get("shard_kernel_names")))
    except Exception:  # noqa: BLE001
        return None


def apply_sharded_traffic(roof, entry, k_serve, k_route, k_un):
    """HBM-side bytes of the sharded leg's three kernel families from the committed PMC passes, priced with THIS run's
    kernel times."""
    if entry and entry.get("dropped"):
        roof["traffic_dropped"] = "profiles/traffic.json entry (%s) not used: %s" % (entry.get("source"), entry["dropped"])
        entry = None
    if not entry:
        roof["traffic_note"] = "no usable PMC profile of this exact sharded command (profiles/traffic.json holds the world-1 ones)"
        return
    roof["traffic"] = entry.get("traffic_bytes_per_launch")
    roof["traffic_source"] = entry.get("source")
    if roof["traffic"] and k_serve > 0:
        roof["achieved_measured"] = roof["traffic"] / (k_serve * 1e-6) / 1e9
        roof["frac_measured"] = roof["achieved_measured"] / roof["peak"]
    for name, us in (("router", k_route), ("unrouter", k_un)):
        sub = entry.get(name)
        if sub and us > 0:
            roof["kernels"][name + "_traffic_bytes"] = sub.get("traffic_bytes_per_launch")
            roof["kernels"][name + "_measured_GBps"] = sub.get("traffic_bytes_per_launch", 0) / (us * 1e-6) / 1e9
    if roof.get("unique_row_bytes") and entry.get("read_bytes"):      # table bytes read (indices taken out) per distinct-row byte
        roof["read_over_unique_rows"] = max(entry["read_bytes"] - roof.get("index_bytes", 0), 0) / roof["unique_row_bytes"]


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


_SCALAR = (int, float, str, bool, type(None))
# the keys of bench.py's contract: they close the line (a record that keeps only its tail still holds them)
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "verified")


def headline_from_legs(replica: dict, sec: dict, mode: str) -> dict:
    """The N > 1 line of the metric's config (26 Kaggle tables, which fit every GPU many times over): `value` / `ms_per_step`
    are the SHARDED leg over RCCL -- tables above 64 MiB placed on owner ranks, indices in / pooled rows out through grouped
    ncclSend / ncclRecv issued from C: the path `north_star` names, the one number of this run that can see a link.  The
    replica leg of the same run (every table on every rank, no data-path transfer: it scales with N by construction and says
    nothing about xGMI) rides along as `config.replica_value` / `value_replica` and the `replica` object.  Both legs time
    EXACTLY K steps after W warm-up steps on the same clocks.  The reference serves all its devices from one lookup() call and
    prints its per-stage figures flat on every call (upmem/include/emb_host.h:258-321, :395-402)."""
    res = dict(sec)
    res["config"] = dict(sec["config"])
    res["roofline"] = dict(sec["roofline"])
    res["headline"] = "sharded-rccl" if "RCCL" in str(sec["config"].get("exchange_transport", "")) else "sharded-peer"
    res["config"]["workload"] = (sec["config"]["workload"] + " -- HEADLINE (value, ms_per_step): this sharded leg, shard mode '%s', "
                                 "transport: %s; the replica leg of the same run (all tables on every rank, no transfer) is "
                                 "config.replica_value" % (mode, sec["config"].get("exchange_transport")))
    res["replica"] = {k: replica[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_event", "ms_per_step_sync", "clock", "steps")
                      if k in replica}
    res["replica"]["verified"] = bool(replica.get("verified", True))
    res["replica"]["roofline"] = replica.get("roofline")
    res["replica"]["config"] = "%s; %s" % (replica["config"].get("workload"), replica["config"].get("parallelism"))
    res["value_replica"], res["ms_per_step_replica"] = replica["value"], replica["ms_per_step"]
    res["config"]["replica_value"], res["config"]["replica_ms_per_step"] = replica["value"], replica["ms_per_step"]
    res["config"]["replica_prewarm_ms"] = replica["config"].get("prewarm_ms")
    rr = replica.get("roofline") or {}
    res["roofline"]["replica_frac"], res["roofline"]["replica_kernel_us"], res["roofline"]["replica_achieved"] = \
        rr.get("frac"), rr.get("kernel_us"), rr.get("achieved")
    res["value_exchange"], res["ms_per_step_exchange"], res["exchange_mode"] = res["value"], res["ms_per_step"], mode
    res["exchange_transport"] = sec["config"].get("exchange_transport")
    return res


def driver_proof(result: dict) -> dict:
    """Make the line survive a record that keeps only the SCALAR members of `config` / `roofline` / `cpu_baseline`, the names of
    other keys, and the last ~2 kB of the text (what BENCH / SCALE records kept of round 5's line: config.exchange{...},
    roofline.exchange{...} and value_exchange were dropped or reduced to their names).  Every figure of the sharded legs -- RCCL
    and peer stores -- is mirrored as a scalar member of `config` / `roofline`; the nested objects stay for human readers.  Key
    order: objects and lists first, scalars after them, the contract's keys last -- at the top level and inside both objects."""
    cfg, roof = result.get("config") or {}, result.get("roofline") or {}
    ex, rx = cfg.get("exchange") or {}, roof.get("exchange") or {}
    if ex:
        cfg["exchange_value"] = ex.get("value", result.get("value_exchange"))
        cfg["exchange_ms_per_step"] = ex.get("ms_per_step", result.get("ms_per_step_exchange"))
        cfg["exchange_mode"] = ex.get("mode", result.get("exchange_mode"))
        cfg["exchange_transport"] = ex.get("transport", cfg.get("exchange_transport", result.get("exchange_transport")))
        cfg["exchange_verified"] = bool(ex.get("verified", False))
        cfg["exchange_steps"] = ex.get("steps", result.get("steps"))
        cfg["exchange_bytes_out_per_rank_per_step"] = ex.get("bytes_out_per_rank_per_step")
        cfg["exchange_host_us_per_step"] = ex.get("host_us_per_step")
        cfg["exchange_sha1"] = ex.get("last_step_sharded_outputs_sha1", cfg.get("last_step_sharded_outputs_sha1"))
        if ex.get("failed"):
            cfg["exchange_failed"] = str(ex["failed"])
    if "value_replica" in result:
        cfg.setdefault("replica_value", result["value_replica"])
        cfg.setdefault("replica_ms_per_step", result.get("ms_per_step_replica"))
    pe = cfg.get("exchange_peer") or {}
    if pe:
        if "value" in pe:
            cfg["exchange_peer_value"], cfg["exchange_peer_ms_per_step"] = pe["value"], pe.get("ms_per_step")
            cfg["exchange_peer_transport"], cfg["exchange_peer_verified"] = pe.get("transport"), bool(pe.get("verified", False))
            cfg["exchange_same_bits"] = bool(pe.get("same_bits_as_rccl_leg", result.get("exchange_same_bits", False)))
            cfg["exchange_peer_direct_one_hot_path"] = pe.get("direct_one_hot_path")
            cfg["exchange_peer_bytes_out_per_rank_per_step"] = pe.get("bytes_out_per_rank_per_step")
        for k in ("skipped", "failed"):
            if pe.get(k):
                cfg["exchange_peer_" + k] = str(pe[k])
    for name, obj in (("exchange", rx), ("exchange_peer", roof.get("exchange_peer") or {})):
        for k in ("step_frac", "step_GBps", "xgmi_GBps", "xgmi_frac", "xgmi_peak_GBps", "host_us_per_step", "host_wait_counts_us_per_step",
                  "host_wait_served_us_per_step", "bytes_out_per_rank_per_step"):
            if k in obj:
                roof["%s_%s" % (name, k)] = obj[k]
    ks = roof.get("kernels") or {}
    for k in ("router_us", "lookup_us", "unrouter_us", "direct_lookup_us", "lookup_GBps"):
        if k in ks:
            roof["kernel_" + k] = ks[k]

    def ordered(d, first_scalar=None):
        nested = {k: v for k, v in d.items() if not isinstance(v, _SCALAR)}
        flat = {k: v for k, v in d.items() if isinstance(v, _SCALAR) and k != first_scalar}
        head = {first_scalar: d[first_scalar]} if first_scalar in d else {}
        return {**nested, **head, **flat}

    out = dict(result)
    # (config.workload is the longest string of the line: it leads config's scalars, so that the figures -- not the prose -- are
    #  what sits nearest the end)
    out["config"], out["roofline"] = ordered(cfg, "workload"), ordered(roof)
    top_nested = {k: v for k, v in out.items() if not isinstance(v, _SCALAR) and k not in ("config", "roofline", "cpu_baseline")}
    top_flat = {k: v for k, v in out.items() if isinstance(v, _SCALAR) and k not in CONTRACT_KEYS}
    return {**top_nested, **top_flat, **{k: out[k] for k in ("cpu_baseline", "config", "roofline") if k in out},
            **{k: out[k] for k in CONTRACT_KEYS if k in out}}


def driver_record_stand_in(line: str, tail_bytes: int = 2300) -> dict:
    """What a BENCH / SCALE record keeps of a bench line, as the round-5 records show it: the contract's top-level keys, the SCALAR
    members of config / roofline / cpu_baseline (lists and objects dropped), the NAMES of every other key, the last ~2.3 kB of the
    text.  tests/test_bench_contract.py runs the N > 1 line through this and asserts that every sharded-leg figure survives."""
    d = json.loads(line)
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    parsed = {k: d.get(k) for k in keep}
    for obj in ("config", "roofline", "cpu_baseline"):
        if isinstance(d.get(obj), dict):
            parsed[obj] = {k: v for k, v in d[obj].items() if isinstance(v, _SCALAR)}
    parsed["extra_keys"] = sorted(k for k in d if k not in keep and k not in ("config", "roofline", "cpu_baseline"))
    return {"parsed": parsed, "tail": line[-tail_bytes:]}


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
    if os.environ.get("PIMEMB_DUMP_AFTER"):       # where is this rank stuck?  Python stacks of all threads to stderr after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PIMEMB_DUMP_AFTER"]), repeat=False, file=sys.stderr)
    run_limit = float(os.environ.get("PIMEMB_RUN_TIMEOUT", "1800"))
    run_dog = threading.Timer(run_limit, _give_up)
    run_dog.daemon = True
    run_dog.start()
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # Several ranks on ONE GPU (rehearsal): RCCL refuses two ranks on a device of the same host ("duplicate GPU"), so every
    # rank claims a host of its own (NCCL_HOSTID) and, being "remote" to its peers, talks to them through RCCL's socket
    # transport over loopback -- slow, but the same communicator, the same grouped send / receive calls as on an 8-GPU node
    # (tools/rccl_one_gpu_ranks_probe.py).  Asked for with PIMEMB_RCCL_ONE_GPU=1; implied by the gloo control plane, whose
    # whole point is more ranks than GPUs (the sharded legs' data path is the C side's RCCL communicator either way).
    one_gpu = os.environ.get("PIMEMB_RCCL_ONE_GPU") == "1" or (backend != "nccl" and n_dev < local_world)
    if one_gpu and world > 1:
        os.environ.update(NCCL_HOSTID="pimemb-rank%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                          NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", NCCL_NET_GDR_LEVEL="0")
    if backend == "nccl" and n_dev < local_world and not one_gpu:
        raise SystemExit(f"bench.py --gpus {world}: {local_world} ranks on this node but {n_dev} GPU(s) visible -- RCCL needs one GPU "
                         "per rank (PIMEMB_RCCL_ONE_GPU=1 rehearses the RCCL path with several ranks on one GPU)")
    dev = torch.device("cuda", local_rank % max(n_dev, 1))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend, rank=rank, world_size=world,
                            **({"device_id": dev} if backend == "nccl" else {}))
    ctx = dict(rank=rank, world=world, dev=dev, backend=backend, stage_cpu=backend != "nccl", json_fd=json_fd)
    dist.barrier()       # the first collective sets the communicator's channels up (milliseconds): not right before a clock starts

    rows_list, dim, _, _ = table_set_of(pel, args)
    TABLE_SCALE.update({t: float(np.float32(2.0 / np.sqrt(n))) for t, n in enumerate(rows_list)})
    TABLE_F16[0] = pel.workloads.TABLE_SET_EXTRAS[getattr(args, "workload", "c2")]["dtype"] == "f16"
    total_bytes = sum(rows_list) * dim * (2 if TABLE_F16[0] else 4)
    hbm = torch.cuda.get_device_properties(dev).total_memory
    auto = getattr(args, "replicate_mb", None) is None
    mode = getattr(args, "shard_mode", None) or ("rows" if getattr(args, "workload", "c2") == "c4" else "whole")

    # --exchange: rccl | peer | both.  "both" (the default for N > 1): the sharded leg runs over RCCL first -- that is
    # value_exchange, and a failure there fails the run -- then once more over peer stores, under a deadline (peer_leg below).
    exchange = getattr(args, "exchange", None) or ("both" if world > 1 else "rccl")
    first_transport = "peer" if exchange == "peer" else "rccl"

    def shard_leg(a, peak, c, rep, transport=None):
        import copy
        a = copy.copy(a)
        a.exchange = transport or first_transport
        a.exchange_both = exchange == "both"
        return run_sharded(a, peak, c, rep, mode)
    state = {"printed": False, "dog": None, "primary": None}

    def emit(res):
        if rank == 0 and res is not None and not state["printed"]:
            state["printed"] = True
            os.write(json_fd, (json.dumps(res) + "\n").encode())

    def finish(res):
        """backend / rank count on every N > 1 line, so a SCALE record shows what RCCL saw; then the scalar mirrors and the key
        order that let the line survive the driver's record (driver_proof)."""
        if res is not None:
            res.setdefault("verified", True)     # every leg compares its outputs bit for bit before AND after timing
            res["config"]["backend"] = backend
            res["config"]["rccl_ranks"] = world          # the sharded legs' transfers are the C side's RCCL communicator on every backend
            if one_gpu and world > 1:
                res["config"]["rccl_transport"] = "sockets over loopback, %d ranks on %d GPU(s) (PIMEMB_RCCL_ONE_GPU=1)" % (world, n_dev)
            res["config"]["world_size"] = world
            res = driver_proof(res)
        return res

    def die(code, note):
        """The exchange leg failed or hung on this rank: print what there is (rank 0), end NON-ZERO.  Peers that
        wait for this rank in a collective are ended by the launcher (bench.py's self-launch / torchrun)."""
        if rank == 0 and state.get("primary") is not None:
            state["primary"]["sharded_exchange"] = {"failed": note}
            state["primary"]["headline"] = "replica (the sharded leg failed: %s)" % note
            state["primary"]["config"]["exchange"] = {"failed": note, "verified": False}
            state["primary"]["verified"] = False
            emit(finish(state["primary"]))
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)

    if auto and total_bytes <= hbm // 4:
        # The metric's tables fit every GPU many times over.  Two legs, EXACTLY K timed steps each: the replica leg (every table on
        # every rank: the placement policy's own choice, no data-path transfer) and the sharded leg over RCCL (tables above 64 MiB on
        # owner ranks / split by row range: indices in, pooled rows out).  The SHARDED leg is the line's value -- the path
        # north_star names and the only curve of the run that can see a link; the replica figure rides in config.replica_value.
        result = run_dp(args, hbm_peak_gbs, ctx)
        state["primary"] = result
        if not getattr(args, "no_exchange_leg", False):
            # A parity failure, an exception or a hang in the sharded leg FAILS the run: what there is (the replica line, with
            # the failure noted) is still printed, the exit status is not 0.
            limit = float(os.environ.get("PIMEMB_EXCHANGE_TIMEOUT", "180"))
            state["dog"] = threading.Timer(limit, die, (3, "timed out after %.0f s" % limit))
            state["dog"].daemon = True
            state["dog"].start()
            # ... and so does a process that DIES in it (abort() is the runtime's answer to a GPU memory fault): rank 0 leaves the
            # replica line with the failure noted on its way out (emb_peer_last_words), status 5
            from . import lib as _lib
            if rank == 0 and result is not None:
                import copy
                snap = copy.deepcopy(result)
                note = "the process received a fatal signal inside the sharded leg (stderr has the runtime's message)"
                snap["sharded_exchange"] = {"failed": note}
                snap["headline"] = "replica (the sharded leg failed: %s)" % note
                snap["config"]["exchange"] = {"failed": note, "verified": False}
                snap["verified"] = False
                sys.stdout.flush()
                _lib.load().emb_peer_last_words(json.dumps(finish(snap)).encode(), json_fd, 5)
            if os.environ.get("PIMEMB_BENCH_TEST_ABORT") == "rccl:%d" % rank:       # (test hook: a rank dies in the sharded leg)
                os.abort()
            try:
                sec = shard_leg(args, hbm_peak_gbs, ctx, 64 << 20)
            except BaseException as ex:  # noqa: BLE001 -- SystemExit from a leg included
                import traceback
                traceback.print_exc()
                die(4, f"{type(ex).__name__}: {ex}")
            state["dog"].cancel()
            if rank == 0:
                _lib.load().emb_peer_last_words(None, 1, 0)
            if rank == 0:
                sec["config"]["exchange"] = dict(sec["config"]["exchange"], steps=sec["steps"], transport=sec["config"]["exchange_transport"],
                                                 last_step_sharded_outputs_sha1=sec["config"].get("last_step_sharded_outputs_sha1"))
                result = headline_from_legs(result, sec, mode)
        elif rank == 0 and result is not None:
            result["headline"] = "replica"
            result["config"]["workload"] += " -- --no-exchange-leg: value is the REPLICA leg (no data-path transfer; scales with N by construction)"
    else:
        rep_mb = 64 if auto else int(args.replicate_mb)
        result = shard_leg(args, hbm_peak_gbs, ctx, rep_mb << 20)
        if rank == 0 and result is not None:       # the primary leg IS the exchange: one number, two names
            result["value_exchange"], result["ms_per_step_exchange"], result["exchange_mode"] = result["value"], result["ms_per_step"], mode
            result["exchange_transport"] = result["config"]["exchange_transport"]
            result["headline"] = "sharded-rccl" if "RCCL" in result["exchange_transport"] else "sharded-peer"
    two_legs = auto and total_bytes <= hbm // 4           # (replica + sharded: --no-exchange-leg leaves nothing for a peer leg to be compared with)
    if exchange == "both" and not (two_legs and getattr(args, "no_exchange_leg", False)):
        peer_leg(args, (64 if auto else int(args.replicate_mb)) << 20, result, emit, finish, ctx, shard_leg, hbm_peak_gbs, False)
    emit(finish(result))
    # teardown under a deadline: a peer leg that lost a rank must not keep the survivors in a barrier nobody will complete
    bye = threading.Timer(float(os.environ.get("PIMEMB_TEARDOWN_TIMEOUT", "60")), lambda: os._exit(0))
    bye.daemon = True
    bye.start()
    dist.barrier()
    dist.destroy_process_group()
    bye.cancel()
    run_dog.cancel()


def peer_leg(args, rep_bytes, result, emit, finish, ctx, shard_leg, hbm_peak_gbs, secondary):
    """--exchange both: the sharded leg once more over PEER STORES, after the RCCL leg has been timed and verified.  The faster
    design on one GPU (58 vs 144 us per step at world 1, round 4) has never crossed a link: whenever a multi-GPU node runs
    this file, ONE run measures both transports.  Reported on the same line as value_exchange_peer / ms_per_step_exchange_peer
    / exchange_transport_peer (+ config.exchange_peer, roofline.exchange_peer), verified against the same expected rows, and
    its outputs must carry the RCCL leg's digest (exchange_same_bits).

    What can go wrong here must not cost the RCCL numbers.  A peer leg that cannot come up -- no fine-grained arena
    (emb_peer_create refuses to fall back silently), an IPC mapping that does not return (the library's watchdog reports, this
    leg's deadline ends the wait), a rank that lost its peers -- is recorded as {"skipped": reason}: the line is printed, the
    exit status stays 0.  A peer leg that comes up and leaves WRONG rows fails the run like any parity failure."""
    import copy
    import threading
    import torch
    import torch.distributed as dist
    rank, world, dev, backend = ctx["rank"], ctx["world"], ctx["dev"], ctx["backend"]
    limit = float(os.environ.get("PIMEMB_PEER_LEG_TIMEOUT", "150"))

    def record(obj):
        if rank == 0 and result is not None:
            result["exchange_peer"] = obj
            result["config"]["exchange_peer"] = obj
    def expire():
        record({"skipped": "no result within %.0f s: a rank is stuck in the peer group's set-up or in a wait for a peer (stderr has the library's message)" % limit})
        emit(finish(result))
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    dog = threading.Timer(limit, expire)
    dog.daemon = True
    dog.start()
    # ... and a process that DIES in here (a GPU memory fault across links nobody has run this leg over ends in abort()) says the
    # same thing on its way out: rank 0 leaves the line it has, every rank ends with status 0 (emb_peer_last_words)
    from . import lib as _lib
    last = ""
    if rank == 0 and result is not None:
        snap = copy.deepcopy(result)
        snap["exchange_peer"] = snap["config"]["exchange_peer"] = {
            "skipped": "the process received a fatal signal inside the peer-store leg (stderr has the runtime's message); "
                       "the RCCL leg's numbers on this line were measured before it"}
        last = json.dumps(finish(snap))
    sys.stdout.flush()
    _lib.load().emb_peer_last_words(last.encode(), int(ctx.get("json_fd", 1)), 0)
    if os.environ.get("PIMEMB_BENCH_TEST_ABORT") == "peer:%d" % rank:       # (test hook: what a GPU memory fault in this leg ends in)
        os.abort()
    a3 = copy.copy(args)
    if secondary:            # the same W / K as the RCCL leg it is compared with (the digest is over the slot the loop ends on)
        a3.steps, a3.warmup = min(args.steps, 400), min(args.warmup, 40)
    saved = {k: os.environ.get(k) for k in ("PIMEMB_SHARD_TIMEOUT_S", "PIMEMB_PEER_WATCHDOG")}
    os.environ["PIMEMB_SHARD_TIMEOUT_S"] = os.environ.get("PIMEMB_PEER_LEG_WAIT_S", "30")      # a missing peer ends a wait after this long
    os.environ["PIMEMB_PEER_WATCHDOG"] = "report"            # a mapping call that hangs: say so, this leg's deadline ends the process (status 0)
    status, note, out = 0, "", None
    try:
        out = shard_leg(a3, hbm_peak_gbs, ctx, rep_bytes, "peer")
    except AssertionError as ex:                # a leg that came up and left wrong rows is a parity failure, not a skip
        status, note = (2 if ("differs from the expected" in str(ex) or "more than 1e-6" in str(ex)) else 1), "%s: %s" % (type(ex).__name__, ex)
    except BaseException as ex:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        status, note = 1, "%s: %s" % (type(ex).__name__, ex)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    worst = torch.tensor([status], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    try:
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)      # every rank ends this leg the same way (under the same deadline)
        worst = int(worst.item())
    except BaseException as ex:  # noqa: BLE001 -- a rank that died in the leg takes the collective with it (gloo raises, RCCL hangs until the deadline)
        dog.cancel()
        record({"skipped": "a rank left the job inside the peer-store leg (%s: %s); %s" % (type(ex).__name__, str(ex)[:200], note or "its stderr says why")})
        emit(finish(result))
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    dog.cancel()
    _lib.load().emb_peer_last_words(None, 1, 0)
    if worst == 2:
        record({"failed": note or "another rank's peer-store outputs differ from the expected rows", "verified": False})
        if rank == 0 and result is not None:
            result["verified"] = False
        emit(finish(result))
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
    if worst == 1:
        record({"skipped": note or "another rank could not bring the peer-store leg up (its stderr says why)"})
        return
    if rank == 0 and result is not None and out is not None:
        same = out["config"].get("last_step_sharded_outputs_sha1") == result["config"].get("last_step_sharded_outputs_sha1",
                                                                                         (result["config"].get("exchange") or {}).get("last_step_sharded_outputs_sha1"))
        obj = {"value": out["value"], "ms_per_step": out["ms_per_step"], "ms_per_step_event": out["ms_per_step_event"], "steps": out["steps"],
               "verified": True, "transport": out["config"]["exchange_transport"], "direct_one_hot_path": out["config"]["direct_one_hot_path"],
               "bytes_out_per_rank_per_step": out["config"]["exchange"]["bytes_out_per_rank_per_step"],
               "host_us_per_step": out["config"]["exchange"]["host_us_per_step"],
               "last_step_sharded_outputs_sha1": out["config"].get("last_step_sharded_outputs_sha1"), "same_bits_as_rccl_leg": bool(same)}
        record(obj)
        result["roofline"]["exchange_peer"] = out["roofline"]["exchange"]
        result["value_exchange_peer"] = out["value"]
        result["ms_per_step_exchange_peer"] = out["ms_per_step"]
        result["exchange_transport_peer"] = out["config"]["exchange_transport"]
        result["exchange_same_bits"] = bool(same)
        if not same:
            result["verified"] = False
            emit(finish(result))
            sys.stdout.flush()
            os._exit(1)
