"""CPU: shard planner (host logic) and the N > 1 exchange PROTOCOL with world_size-2 / -3 gloo processes.

The sharded step itself is csrc/pimemb_shard.cpp -- HIP kernels and RCCL groups, nothing a CPU can run (the product has
no CPU path; tests/test_gpu_shard.py runs it with 1-4 real RCCL ranks).  What is covered here is everything of the N > 1
path that is plain host arithmetic in the C ABI, driven through a real multi-process exchange: the message formats of
include/pimemb.h (counts first: [peer][table]{sub-bags, indices} + peaks; request pieces: offsets[pad4] then row
ids[pad4] per table; partial rows back to back), the split sizes every rank derives from the counts it sent and received
(emb_route_exchange_sizes), the descriptors of the one fused lookup over everything received (emb_route_serve_descs), and
the shard-order addition of the partial rows.  The GPU's three kernels -- router, bag-sum, un-router -- are stood in for
by numpy restatements of their documented rules and by the oracle; gloo carries the bytes."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sharding():
    from importlib import import_module
    sys.path.insert(0, ROOT)
    import pim_embedding_lookup_amd  # noqa: F401
    return import_module("pim-embedding-lookup_amd.sharding")


def test_planner_kaggle_8_ranks(pel):
    sh = _sharding()
    rows = pel.workloads.KAGGLE_ROWS
    plan = sh.plan_shards(rows, 16, 4, world=8)
    # tables <= 64 MiB are replicated (21 of them), the 5 big ones are sharded
    assert plan.kinds.count(sh.REPLICATED) == 21
    big = [t for t, k in enumerate(plan.kinds) if k != sh.REPLICATED]
    assert sorted(rows[t] for t in big) == [2202608, 5461306, 7046547, 8351593, 10131227]
    # every row of every table is served exactly once (or by everyone, if replicated)
    for t, ids in enumerate(plan.units_of_table):
        us = [plan.units[i] for i in ids]
        assert us[0].row_lo == 0 and us[-1].row_hi == rows[t]
        for a, b in zip(us, us[1:]):
            assert a.row_hi == b.row_lo
        if plan.kinds[t] == sh.ROW_SPLIT:
            assert [u.owner for u in us] == list(range(8))
    loads = [plan.bytes_on(r) for r in range(8)]
    assert max(loads) < 2.0 * (sum(loads) / 8)
    assert "21 replicated" in plan.describe()


def test_planner_terabyte_shaped_8_ranks(pel):
    """BASELINE configs[3]: 26 Terabyte-shaped tables at dim 128 (452 GB) over 8 ranks: the tables
    above 1/8 of the sharded bytes are row-split over all ranks, the load is balanced, every rank's
    share fits its 288 GB HBM with room to spare."""
    sh = _sharding()
    rows, dim, _, _ = pel.workloads.table_set("c4")
    assert len(rows) == 26 and sum(rows) == 882_774_559 and max(rows) == 292_775_614
    plan = sh.plan_shards(rows, dim, 4, world=8)
    split = [t for t, k in enumerate(plan.kinds) if k == sh.ROW_SPLIT]
    assert sorted(rows[t] for t in split) == [130229467, 187188510, 227605432, 292775614]
    assert plan.kinds[20] in (sh.WHOLE, sh.ROW_SPLIT)           # the 40.8M-row table (20.9 GB)
    for t, ids in enumerate(plan.units_of_table):
        us = [plan.units[i] for i in ids]
        assert us[0].row_lo == 0 and us[-1].row_hi == rows[t]
        assert all(a.row_hi == b.row_lo for a, b in zip(us, us[1:]))
    loads = [plan.bytes_on(r) for r in range(8)]
    assert max(loads) < 90e9 and max(loads) < 1.5 * (sum(loads) / 8)
    # scaled copies keep the shape (used to rehearse the layout on fewer GPUs)
    small, _, _, label = pel.workloads.table_set("c4", 1 / 256)
    assert small[19] == 292775614 // 256 and min(small) == 1 and "rows x" in label


def test_planner_policies():
    sh = _sharding()
    # world 1: small tables local; a big one is "owned" by rank 0 (self exchange, used to rehearse RCCL)
    p1 = sh.plan_shards([10, 1000, 2_000_000], 16, 4, world=1)
    assert p1.kinds == [sh.REPLICATED, sh.REPLICATED, sh.WHOLE] and p1.units[2].owner == 0
    # nothing replicated, giant table row-split, others whole and balanced
    rows = [200_000_000, 40_000_000, 30_000_000, 20_000_000, 10_000_000, 5]
    p = sh.plan_shards(rows, 128, 4, world=8, replicate_bytes=0)
    # > 1/8 of the sharded bytes (19.2 GB) -> row-split: the 200M- and the 40M-row tables
    assert p.kinds[:2] == [sh.ROW_SPLIT, sh.ROW_SPLIT] and p.kinds[2:] == [sh.WHOLE] * 4
    assert len(p.units_of_table[0]) == 8
    owners = [p.units[p.units_of_table[t][0]].owner for t in range(2, 6)]
    assert len(set(owners)) == 4                         # whole tables land on distinct ranks
    # a table with fewer rows than ranks is never row-split
    q = sh.plan_shards([3], 16, 4, world=8, replicate_bytes=0, split_bytes=1)
    assert q.kinds == [sh.WHOLE]


PAD4 = lambda v: (int(v) + 3) // 4 * 4  # noqa: E731


def _route(idx, off, rps, N):
    """numpy restatement of emb_route_bags' rule for one table: per shard d (sub-bag start offsets, local row ids, slot of
    every bag or -1)."""
    B = off.shape[0]
    end = np.concatenate([off[1:], [idx.shape[0]]]).astype(np.int64)
    dest = np.minimum(idx.astype(np.int64) // rps, N - 1)
    out = []
    for d in range(N):
        sub_off, lists, slots, pos = [], [], np.full(B, -1, dtype=np.int64), 0
        for b in range(B):
            sel = idx[off[b]:end[b]][dest[off[b]:end[b]] == d]
            if sel.size:
                slots[b] = len(sub_off)
                sub_off.append(pos)
                lists.append(sel.astype(np.int64) - d * rps)
                pos += sel.size
        out.append((np.array(sub_off, dtype=np.uint32), np.concatenate(lists).astype(np.uint32) if lists else np.zeros(0, np.uint32), slots))
    return out


def _worker(rank, world, port, cfg, q):
    import ctypes as C
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import pim_embedding_lookup_amd as pel
    from oracle import oracle
    sh = _sharding()
    L = pel.lib.load()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, dim, N = cfg["rows"], cfg["dim"], world
        plan = sh.plan_shards(rows, dim, 4, world, replicate_bytes=cfg["rep"], split_bytes=cfg["split"], pooling=cfg.get("plan_pooling", 1.0))
        assert set(plan.kinds) == set(cfg["expect_kinds"]), plan.kinds
        if cfg.get("expect_row_split") is not None:
            assert plan.kinds.count(sh.ROW_SPLIT) == cfg["expect_row_split"], plan.kinds
        tabs = [np.random.default_rng(100 + t).standard_normal((n, dim)).astype(np.float32) for t, n in enumerate(rows)]
        split = [t for t, k in enumerate(plan.kinds) if k == sh.ROW_SPLIT]
        whole_of = [[t for t, k in enumerate(plan.kinds) if k == sh.WHOLE and plan.units[plan.units_of_table[t][0]].owner == p]
                    for p in range(N)]
        mine = whole_of[rank]
        K = len(split)
        rps = [-(-rows[t] // N) for t in split]
        rng = np.random.default_rng(1000 + rank)               # every rank has its OWN bags
        B = cfg["bags"] + rank                                   # ragged: ranks differ in bag count
        idx, off = [], []
        for n in rows:
            lens = rng.integers(0, cfg["max_len"] + 1, size=B)
            o = np.zeros(B, dtype=np.int64)
            o[1:] = np.cumsum(lens)[:-1]
            off.append(o)
            idx.append(rng.integers(0, n, size=int(lens.sum())).astype(np.int64))

        def a2a(send_parts, recv_sizes, dtype):
            wire = np.int32 if dtype == np.uint32 else dtype            # gloo has no uint32: the same bits as int32
            flat = np.concatenate([np.asarray(p_, dtype=dtype).reshape(-1) for p_ in send_parts]) if send_parts else np.zeros(0, dtype)
            send = torch.from_numpy(np.ascontiguousarray(flat).view(wire))
            recv = torch.empty(int(sum(recv_sizes)), dtype=send.dtype)
            dist.all_to_all_single(recv, send, output_split_sizes=[int(x) for x in recv_sizes],
                                   input_split_sizes=[int(np.asarray(p_).size) for p_ in send_parts])
            return np.split(recv.numpy().view(dtype), np.cumsum(recv_sizes)[:-1].astype(np.int64)) if len(recv_sizes) else []

        # ---- R: route (model of the kernel), counts message [N][K+1][2] with the peaks entry
        routed = [_route(idx[t], off[t], rps[k], N) for k, t in enumerate(split)]
        sent = np.zeros((N, K + 1, 2), dtype=np.uint32)
        pieces = []
        for d in range(N):
            words = []
            for k in range(K):
                so, ids, _ = routed[k][d]
                sent[d, k] = (so.size, ids.size)
                words.append(np.concatenate([so, np.zeros(PAD4(so.size) - so.size, np.uint32), ids, np.zeros(PAD4(ids.size) - ids.size, np.uint32)]))
            pieces.append(np.concatenate(words) if words else np.zeros(0, np.uint32))
        if K:
            sent[:, K, 0] = max(p_.size for p_ in pieces)
            sent[:, K, 1] = max(int(sent[d, :K, 0].sum()) for d in range(N))
        # ---- counts FIRST
        received = np.stack(a2a([sent[d] for d in range(N)], [(K + 1) * 2] * N, np.uint32)).reshape(N, K + 1, 2).astype(np.uint32) if K else sent
        wsent = [[(B, idx[t].size, 0, 0) for t in whole_of[p]] for p in range(N)]          # whole tables: {bags, indices, pooling 0 = offsets travel, 0}
        wrecv = a2a([np.array(w, dtype=np.uint32) for w in wsent], [len(mine) * 4] * N, np.uint32)
        wrecv = [w.reshape(len(mine), 4) for w in wrecv]
        # ---- sizes from the counts: the LIBRARY's arithmetic
        if K:
            both = np.ascontiguousarray(np.stack([sent, received]))
            out = (C.c_uint64 * (4 * N))()
            pr, pt = C.c_uint64(), C.c_uint64()
            base = C.addressof(out)
            assert L.emb_route_exchange_sizes(both.ctypes.data, both.ctypes.data + sent.nbytes, K, N, dim, base, base + 8 * N,
                                              base + 16 * N, base + 24 * N, C.byref(pr), C.byref(pt)) == 0
            v = np.array(out[:], dtype=np.int64).reshape(4, N)
            out_w, in_w, back, served = v
            assert [p_.size for p_ in pieces] == out_w.tolist()
            # every rank derives the SAME job-wide peaks
            peaks = torch.tensor([pr.value, pt.value], dtype=torch.int64)
            lo, hi = peaks.clone(), peaks.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi)
            # ---- Q: request pieces with split sizes from the counts
            req_recv = np.ascontiguousarray(np.concatenate(a2a(pieces, in_w, np.uint32))) if in_w.sum() else np.zeros(4, np.uint32)
            # ---- S: the descriptors of the ONE fused lookup -- library arithmetic, host pointers here -- run by the oracle
            ids_arr = (C.c_uint32 * K)(*range(K))
            descs = (pel.lib.EmbLookupDesc * (N * K))()
            nd, nbytes = C.c_uint32(), C.c_uint64()
            ret_send = np.zeros((max(int(served.sum()), 1), dim), dtype=np.float32)
            assert L.emb_route_serve_descs(received.ctypes.data, K, N, dim, ids_arr, req_recv.ctypes.data, ret_send.ctypes.data,
                                           descs, C.byref(nd), C.byref(nbytes)) == 0
            for j in range(nd.value):
                dsc = descs[j]
                t = split[dsc.table_id]
                u = plan.units[plan.units_of_table[t][rank]]
                o_ = np.ctypeslib.as_array(C.cast(dsc.offsets, C.POINTER(C.c_uint32)), (dsc.n_bags,)).astype(np.int64)
                i_ = np.ctypeslib.as_array(C.cast(dsc.indices, C.POINTER(C.c_uint32)), (dsc.n_indices,)).astype(np.int64)
                row0 = (dsc.pooled - ret_send.ctypes.data) // (4 * dim)
                ret_send[row0:row0 + dsc.n_bags] = oracle.c_bag_sum(tabs[t][u.row_lo:u.row_hi], i_, o_)
            # ---- T: partial rows back, split sizes from the same counts
            parts = [ret_send[int(served[:s_].sum()):int(served[:s_ + 1].sum())] for s_ in range(N)]
            back_rows = a2a(parts, back * dim, np.float32)
            ret_recv = np.concatenate(back_rows).reshape(-1, dim) if back.sum() else np.zeros((0, dim), np.float32)
        # whole tables: indices / offsets travel as they are, pooled rows come straight back
        wreq = a2a([np.concatenate([np.concatenate([off[t].astype(np.uint32), idx[t].astype(np.uint32)]) for t in whole_of[p]])
                    if whole_of[p] else np.zeros(0, np.uint32) for p in range(N)],
                   [int(sum(int(c[0]) + int(c[1]) for c in wrecv[s_])) for s_ in range(N)], np.uint32)
        wret = []
        for s_ in range(N):
            cur, rows_out = 0, []
            for j, t in enumerate(mine):
                nb, ni = int(wrecv[s_][j][0]), int(wrecv[s_][j][1])
                o_ = wreq[s_][cur:cur + nb].astype(np.int64); cur += nb
                i_ = wreq[s_][cur:cur + ni].astype(np.int64); cur += ni
                rows_out.append(oracle.c_bag_sum(tabs[t], i_, o_) if nb else np.zeros((0, dim), np.float32))
            wret.append(np.concatenate(rows_out).reshape(-1) if rows_out else np.zeros(0, np.float32))
        wback = a2a(wret, [len(whole_of[p]) * B * dim for p in range(N)], np.float32)

        # ---- U + assembly, then the single-process oracle over the whole tables
        worst = 0.0
        for t in range(len(rows)):
            want = oracle.c_bag_sum(tabs[t], idx[t], off[t])
            if plan.kinds[t] == sh.REPLICATED:
                got = want          # (looked up locally by the engine in the product: nothing to exchange)
            elif plan.kinds[t] == sh.WHOLE:
                p = plan.units[plan.units_of_table[t][0]].owner
                j = whole_of[p].index(t)
                got = wback[p][j * B * dim:(j + 1) * B * dim].reshape(B, dim)
                assert np.array_equal(got, want), f"table {t} (whole)"
            else:
                k = split.index(t)
                got = np.zeros((B, dim), np.float32)
                for d in range(N):                    # shard order, from +0: emb_unroute_bags' rule
                    row0 = int(back[:d].sum()) + int(sent[d, :k, 0].sum())
                    slots = routed[k][d][2]
                    has = slots >= 0
                    part = np.zeros((B, dim), np.float32)
                    part[has] = ret_recv[row0 + slots[has]]
                    got = got + part
                # the same thing restated per shard with the oracle: bit for bit; and 1e-5 from the sequential sum
                exact = np.zeros_like(want)
                bag_of = np.repeat(np.arange(B), np.diff(np.append(off[t], len(idx[t]))))
                for u in (plan.units[i] for i in plan.units_of_table[t]):
                    keep = (idx[t] >= u.row_lo) & (idx[t] < u.row_hi)
                    l2 = np.bincount(bag_of[keep], minlength=B)
                    o2 = np.zeros(B, np.int64); o2[1:] = np.cumsum(l2)[:-1]
                    if u.row_hi > u.row_lo:
                        exact = exact + oracle.c_bag_sum(tabs[t][u.row_lo:u.row_hi], idx[t][keep] - u.row_lo, o2)
                assert np.array_equal(got, exact), f"table {t} (row-split)"
                assert np.abs(got - want).max() <= 1e-5
                worst = max(worst, float(np.abs(got - want).max()))
        q.put((rank, "ok", worst))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("cfg", [
    dict(rows=[7, 300, 5000, 64, 2000], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=37, max_len=1,
         expect_kinds=["replicated", "whole", "row_split"]),
    dict(rows=[7, 300, 5000, 64, 2000], dim=64, rep=64 * 64 * 4, split=3000 * 64 * 4, bags=20, max_len=9,
         expect_kinds=["replicated", "whole", "row_split"]),
    dict(rows=[100, 200, 300], dim=16, rep=0, split=10**12, bags=11, max_len=4, expect_kinds=["whole"]),
])
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_exchange_protocol_matches_single_process_oracle(cfg, world):
    """world 3: row ranges that do not divide evenly, a rank that owns no whole table."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}:\n{info}"


@pytest.mark.parametrize("pooling", [1, 32])
def test_gloo_exchange_protocol_world_8_terabyte_placement(pel, pooling):
    """World EIGHT: the planner's own placement of the 26 Terabyte-shaped tables (BASELINE configs[3]; rows
    and the replication threshold scaled by 1/20000 so that eight CPU processes hold them) -- 4 row-split tables at one index per
    bag, fewer at pooling 32 where the return-volume term keeps pooled tables whole -- through a real eight-process exchange:
    counts[8][K+1][2] first, request pieces sized by the library's own arithmetic (emb_route_exchange_sizes), 8 x K serve
    descriptors (emb_route_serve_descs), partial rows back and added in shard order; whole tables on eight owners.  Every table
    of every rank against the single-process oracle.  The reference fans one lookup() out to all its devices the same way
    (upmem/include/emb_host.h:155-160, 258-321)."""
    import torch.multiprocessing as mp
    scale = 1.0 / 20000
    rows, dim, _, _ = pel.workloads.table_set("c4", rows_scale=scale)
    sh = _sharding()
    plan = sh.plan_shards(rows, dim, 4, 8, replicate_bytes=int((64 << 20) * scale), pooling=float(pooling))
    n_split = plan.kinds.count(sh.ROW_SPLIT)
    assert n_split == 4 if pooling == 1 else 0 < n_split < 4, plan.kinds
    cfg = dict(rows=rows, dim=dim, rep=int((64 << 20) * scale), split=None, plan_pooling=float(pooling), bags=9, max_len=pooling,
               expect_kinds=sorted(set(plan.kinds)), expect_row_split=n_split)
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=420) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}:\n{info}"


def test_hbm_budget_of_the_8_rank_layouts_fits_288_gb(pel):
    """per-rank HBM of BASELINE configs[3] (C4) and configs[4] (C5, 30M rows) at EIGHT ranks -- tables +
    the caller's rotating batch slots (or the arena) + the library's staging ring + 64 cached plans + a checked shard's
    counters -- against 288 GB, for both transports, by host arithmetic alone.  C4 is far inside.  C5's TABLES fit (245.8 GB a
    rank) but tables + slots + the RCCL staging of 64 whole tables per owner do NOT (293 GB): bench.py's pre-flight
    (sharding.fit_to_hbm) shrinks the rows a few per cent and the line says so -- checked here, before a node is booked."""
    sh = _sharding()
    HBM = 288 * 10**9
    for name, L, elem in (("c4", 1, 4), ("c4", 32, 4), ("c5", 32, 2)):
        rows, dim, B, _ = pel.workloads.table_set(name)
        make = lambda rows_: sh.plan_shards(rows_, dim, elem, 8, replicate_bytes=64 << 20, pooling=float(L))   # noqa: E731
        for transport in ("rccl", "peer"):
            bud = lambda p_, r_: sh.hbm_budget(p_, r_, B, L, n_slots=8, depth=3, transport=transport, checked=True)   # noqa: E731
            plan = make(rows)
            worst = max((bud(plan, r) for r in range(8)), key=lambda d: d["total"])
            assert worst["total"] == sum(v for k, v in worst.items() if k != "total") and worst["tables"] == max(plan.bytes_on(r) for r in range(8))
            if name == "c4":
                assert worst["total"] < 0.5 * HBM, (name, L, transport, worst)
                assert sh.fit_to_hbm(rows, HBM, make, bud)[0] == 1.0
            else:
                assert 240e9 < worst["tables"] < 250e9
                if transport == "rccl":
                    assert worst["total"] > HBM, worst            # as configured it would run out of memory on the node
                    assert worst["staging"] > 10e9 and worst["batches"] > 20e9
                scale, rows2, plan2, worst2 = sh.fit_to_hbm(rows, HBM, make, bud)
                assert 0.8 < scale <= 1.0 and worst2["total"] <= 0.94 * HBM and all(r2 <= r for r2, r in zip(rows2, rows))
                assert (scale < 1.0) == (worst["total"] > 0.94 * HBM)
    # the worst skew (every index of every rank names ONE shard's rows) at C4, pooling 32: still inside
    rows, dim, B, _ = pel.workloads.table_set("c4")
    plan = sh.plan_shards(rows, dim, 4, 8, replicate_bytes=64 << 20, pooling=1.0)
    skew = max(sh.hbm_budget(plan, r, B, 32, transport="rccl", balance=8.0)["total"] for r in range(8))
    assert skew < 0.6 * HBM


def test_planner_return_volume_term(pel):
    """pooled tables that fit a rank are placed WHOLE rather than row-split (a split returns up to
    min(pooling, world) partial rows per bag); one-index-per-bag tables and tables no rank has room for still split."""
    sh = _sharding()
    rows, dim, _, _ = pel.workloads.table_set("c4")
    one_hot = sh.plan_shards(rows, dim, 4, world=8)
    pooled = sh.plan_shards(rows, dim, 4, world=8, pooling=32)
    assert one_hot.kinds.count(sh.ROW_SPLIT) == 4
    assert pooled.kinds.count(sh.ROW_SPLIT) < 4 and pooled.kinds.count(sh.WHOLE) > one_hot.kinds.count(sh.WHOLE)
    kept = [t for t in range(len(rows)) if one_hot.kinds[t] == sh.ROW_SPLIT and pooled.kinds[t] == sh.WHOLE]
    assert kept and all("pooled" in pooled.notes[t] and "whole although" in pooled.notes[t] for t in kept)
    assert max(pooled.bytes_on(r) for r in range(8)) < 288e9 * 0.5                     # still far inside one GPU's HBM
    still = [t for t in range(len(rows)) if pooled.kinds[t] == sh.ROW_SPLIT]
    assert all("no rank has room" in pooled.notes[t] for t in still)
    # an explicit capacity forces the split again
    tight = sh.plan_shards(rows, dim, 4, world=8, pooling=32, capacity_bytes=60 << 30)
    assert tight.kinds.count(sh.ROW_SPLIT) >= pooled.kinds.count(sh.ROW_SPLIT)
    # a world of one rank splits only when asked to (the routed path rehearsed on one GPU)
    assert sh.plan_shards([10, 5_000_000], 16, 4, world=1).kinds == [sh.REPLICATED, sh.WHOLE]
    assert sh.plan_shards([10, 5_000_000], 16, 4, world=1, split_bytes=1 << 20, split_single_rank=True).kinds == [sh.REPLICATED, sh.ROW_SPLIT]
