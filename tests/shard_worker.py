"""One rank of a sharded-lookup test job (tests/test_gpu_shard.py starts N of these; the oracle is the checker).

Every rank: builds the plan, loads its shards, hands its OWN ragged batches to `ShardedEmbeddingBags` -- synchronous
`forward` and the software-pipelined `submit` / `wait` form -- and compares every table with the CPU oracle over the
whole tables.  Row-split tables with more than one index per bag are also compared bit for bit with the oracle's
per-shard partial sums added in shard order (what emb_unroute_bags does)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_batch(rng, rows, n_bags, max_len, fixed):
    idx, off = [], []
    for n in rows:
        if fixed:
            lens = np.full(n_bags, max_len, dtype=np.int64)
        else:
            lens = rng.integers(0, max_len + 1, size=n_bags)
        o = np.zeros(n_bags, dtype=np.int64)
        if n_bags:
            o[1:] = np.cumsum(lens)[:-1]
        off.append(o)
        idx.append(rng.integers(0, n, size=int(lens.sum())).astype(np.int64))
    return idx, off


def expect(oracle, sh, plan, tabs, idx, off, t):
    """(exact, sequential): what the sharded path must return bit for bit, and the unsharded in-order sum."""
    seq = oracle.c_bag_sum(tabs[t], idx[t], off[t]) if len(off[t]) else np.zeros((0, tabs[t].shape[1]), np.float32)
    if plan.kinds[t] != sh.ROW_SPLIT or not len(off[t]):
        return seq, seq
    exact = np.zeros_like(seq)
    bag_of = np.repeat(np.arange(len(off[t])), np.diff(np.append(off[t], len(idx[t]))))
    for u in (plan.units[i] for i in plan.units_of_table[t]):
        keep = (idx[t] >= u.row_lo) & (idx[t] < u.row_hi)
        l2 = np.bincount(bag_of[keep], minlength=len(off[t]))
        o2 = np.zeros(len(off[t]), np.int64)
        o2[1:] = np.cumsum(l2)[:-1]
        if u.row_hi > u.row_lo:
            part = oracle.c_bag_sum(tabs[t][u.row_lo:u.row_hi], idx[t][keep] - u.row_lo, o2)
            exact = exact + part
    return exact, seq


def main():
    cfg = json.loads(sys.argv[1])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:      # several RCCL ranks on the one GPU: every rank claims a host of its own (sockets over loopback)
        os.environ.update(NCCL_HOSTID="pimemb-rank%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                          NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", NCCL_NET_GDR_LEVEL="0")
    import torch
    import torch.distributed as dist
    import pim_embedding_lookup_amd as pel
    from importlib import import_module
    from oracle import oracle
    sh = import_module("pim-embedding-lookup_amd.sharding")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sys.stdout.flush()
    os.dup2(2, 1)              # RCCL's banner goes to stderr with everything else
    if world > 1 and not cfg.get("peer"):
        dist.init_process_group("gloo", rank=rank, world_size=world)      # bootstrap only (the unique id): the data path is the C side's RCCL
    rows, dim = cfg["rows"], cfg["dim"]
    f16 = cfg.get("f16", False)
    if cfg.get("kinds"):        # placement given by the test (a world of one rank never splits by itself)
        kinds = cfg["kinds"]
        units, uot = [], []
        owner_rr = 0
        for t, k in enumerate(kinds):
            ids = []
            if k == sh.REPLICATED:
                ids.append(len(units)); units.append(sh.Unit(t, -1, 0, rows[t], len(units)))
            elif k == sh.WHOLE:
                ids.append(len(units)); units.append(sh.Unit(t, owner_rr % world, 0, rows[t], len(units)))
                owner_rr += 1
            else:
                per = -(-rows[t] // world)
                for r in range(world):
                    ids.append(len(units)); units.append(sh.Unit(t, r, min(r * per, rows[t]), min((r + 1) * per, rows[t]), len(units)))
            uot.append(ids)
        plan = sh.ShardPlan(world, rows, dim, 2 if f16 else 4, kinds, units, uot)
    else:
        plan = sh.plan_shards(rows, dim, 2 if f16 else 4, world, replicate_bytes=cfg["rep"], split_bytes=cfg["split"],
                              pooling=cfg.get("plan_pooling", 1.0))
    if cfg.get("expect_kinds"):
        assert sorted(set(plan.kinds)) == sorted(cfg["expect_kinds"]), plan.kinds
    tabs = [(np.random.default_rng(100 + t).standard_normal((n, dim)) * 0.05).astype(np.float16 if f16 else np.float32)
            for t, n in enumerate(rows)]
    tabs32 = [w.astype(np.float32) for w in tabs]
    eng = pel.EmbeddingEngine(device=0, max_tables=len(plan.units) + 1)
    peer = sh.PeerGroup(eng, cfg["peer_tag"], rank, world, arena_bytes=cfg.get("arena_bytes", 256 << 20)) if cfg.get("peer") else None
    if peer is not None and cfg.get("arena_filler"):       # push what follows into the arena's later chunks (each chunk is its own IPC mapping)
        for nbytes in cfg["arena_filler"]:
            peer.alloc(nbytes)
    comm = sh.native_comm(eng, rank, world) if peer is None else None
    status = {"rank": rank, "ok": False}
    if peer is not None:
        status["peer"] = peer.info()
    try:
        for depth in cfg.get("depths", [0, 2, 3]):
            S = sh.ShardedEmbeddingBags(plan, eng, rank, comm, depth=depth, check=cfg.get("check", True),
                                        self_via_comm=cfg.get("self_via_comm", False), peer=peer)
            S.load_tables(lambda t, lo, hi: torch.from_numpy(tabs[t][lo:hi]).to(dev))
            rng = np.random.default_rng(1000 + rank)           # every rank has its OWN bags
            n_batches = cfg.get("batches", 6)
            batches = []
            for j in range(n_batches):
                B = cfg["bags"] + (rank if cfg.get("ragged_ranks", True) else 0) + 3 * j
                if cfg.get("empty_rank") == rank and j % 2 == 1:
                    B = 0
                batches.append(make_batch(rng, rows, B, cfg["max_len"], cfg.get("fixed", False)))
            dt = torch.int64 if cfg.get("int64", False) else torch.int32

            def to_dev(a):
                x = torch.from_numpy(a).to(dev).to(dt)
                if peer is None:
                    return x
                y = peer.empty(x.shape, dt)                   # peers gather from it in place: it must live in the arena
                y.copy_(x)
                return y

            def dev_batch(b):
                return [to_dev(i) for i in b[0]], [to_dev(o) for o in b[1]]

            def check(outs, b, what):
                torch.cuda.synchronize()
                for t in range(len(rows)):
                    got = outs[t].cpu().numpy()
                    exact, seq = expect(oracle, sh, plan, tabs32, b[0], b[1], t)
                    assert got.shape == exact.shape, (what, t, got.shape, exact.shape)
                    assert np.array_equal(got, exact), f"{what}: table {t} ({plan.kinds[t]}) differs from the oracle"
                    if got.size:
                        assert np.abs(got - seq).max() <= 1e-5

            if depth == 0:
                for j, b in enumerate(batches):
                    di, do = dev_batch(b)
                    if cfg.get("fixed", False):
                        outs = S.forward(None, di, fixed_pooling=cfg["max_len"])
                    else:
                        outs = S.forward(do, di)
                    check(outs, b, f"depth 0 batch {j}")
            else:
                pending = []
                for j, b in enumerate(batches):
                    di, do = dev_batch(b)
                    if cfg.get("fixed", False):
                        seq, outs = S.submit(di, None, fixed_pooling=cfg["max_len"])
                    else:
                        seq, outs = S.submit(di, do)
                    pending.append((seq, outs, b))
                    if j >= depth:
                        q, o, bb = pending[j - depth]
                        S.wait(q)
                        check(o, bb, f"depth {depth} batch {j - depth}")
                S.flush()
                for q, o, bb in pending[max(0, n_batches - depth):]:
                    S.wait(q)
                    check(o, bb, f"depth {depth} drained batch {q}")
            st = S.stats()
            if cfg.get("expect_direct"):          # the direct one-hot path ran: nothing was routed, nothing un-routed
                S.set_kernel_timing(True)
                b = batches[0]
                di, _do = dev_batch(b)
                outs = S.forward(None, di, fixed_pooling=cfg["max_len"]) if depth == 0 else None
                if depth:
                    q, outs = S.submit(di, None, fixed_pooling=cfg["max_len"])
                    S.flush()
                    S.wait(q)
                check(outs, b, "timed direct batch")
                kst = S.stats()
                assert kst["us_kernel_direct"] > 0 and kst["us_kernel_route"] == 0 and kst["us_kernel_unroute"] == 0, kst
                S.set_kernel_timing(False)
            status["stats_depth%d" % depth] = {k: (int(v) if isinstance(v, int) else float(v)) for k, v in st.items()}
            if cfg.get("bad_index"):           # a row id outside the table: the SERVING rank raises, nobody hangs
                b = make_batch(rng, rows, cfg["bags"], max(1, cfg["max_len"]), True)
                t_bad = cfg["bad_index"]["table"]
                bad_value = {"beyond": rows[t_bad] + 5, "negative": -3, "huge": (1 << 32) + 2}[cfg["bad_index"].get("value", "beyond")]
                if rank == cfg["bad_index"]["rank"]:
                    b[0][t_bad][3] = bad_value
                di = [to_dev(i) for i in b[0]]
                n_raised, outs = 0, None
                try:                       # check="sync": the call that completes the batch raises; "deferred" (the default): the
                    if depth == 0:         # NEXT call does -- report() here, so that both forms are counted in one place
                        outs = S.forward(None, di, fixed_pooling=max(1, cfg["max_len"]))
                    else:
                        q, outs = S.submit(di, None, fixed_pooling=max(1, cfg["max_len"]))
                        S.flush()
                except IndexError:
                    n_raised += 1
                try:
                    S.report()
                except IndexError:
                    n_raised += 1
                status["raised_depth%d" % depth] = n_raised > 0
                status["n_raised_depth%d" % depth] = n_raised
                torch.cuda.synchronize()
                if outs is not None and rank == cfg["bad_index"]["rank"] and cfg.get("check", True):
                    # whichever mechanism refused it: the offending bag pooled to a ZERO row, never to stale bytes
                    row = outs[t_bad][3 // max(1, cfg["max_len"])].cpu().numpy()
                    assert not row.any(), ("the bag of the refused index holds", row[:4])
                if cfg.get("good_after_bad"):      # the shard object is still usable: a clean batch after the refused one, against the oracle
                    torch.cuda.synchronize()
                    g = make_batch(rng, rows, cfg["bags"] + 1, max(1, cfg["max_len"]), True)
                    outs = S.forward(None, [to_dev(i) for i in g[0]], fixed_pooling=max(1, cfg["max_len"])) if depth == 0 else None
                    if depth:
                        q, outs = S.submit([to_dev(i) for i in g[0]], None, fixed_pooling=max(1, cfg["max_len"]))
                        S.flush()
                        S.wait(q)
                    check(outs, g, "clean batch after a refused one")
            if cfg.get("bad_pipeline"):        # ONE bad batch in the middle of a pipelined stream of clean ones: how many calls raise?
                t_bad = cfg["bad_pipeline"]["table"]
                n_raised = 0
                keep = []
                for j in range(8):
                    b = make_batch(rng, rows, cfg["bags"], max(1, cfg["max_len"]), cfg.get("fixed", False))
                    if j == 2 and rank == cfg["bad_pipeline"]["rank"]:
                        b[0][t_bad][min(3, len(b[0][t_bad]) - 1)] = rows[t_bad] + 5
                    di, do = dev_batch(b)
                    keep.append((di, do))
                    try:
                        if cfg.get("fixed", False):
                            S.submit(di, None, fixed_pooling=cfg["max_len"])
                        else:
                            S.submit(di, do)
                    except IndexError:
                        n_raised += 1
                for final in (S.flush, S.report):
                    try:
                        final()
                    except IndexError:
                        n_raised += 1
                status["pipeline_raised_depth%d" % depth] = n_raised
            torch.cuda.synchronize()
            S.close()
        if cfg.get("harness"):        # the apply_emb-shaped module over the sharded call: its own engine, the same communicator
            hz = import_module("pim-embedding-lookup_amd.dlrm_harness")
            ebc = hz.ShardedEmbeddingBagCollection(rows, dim, rank, world, comm=comm, peer=peer, weights=tabs32,
                                                   replicate_bytes=cfg["rep"])
            assert set(ebc.plan.kinds) >= {"replicated"} and len(set(ebc.plan.kinds)) >= 2, ebc.plan.kinds
            rng = np.random.default_rng(4000 + rank)
            for j in range(cfg.get("harness_batches", 3)):
                b = make_batch(rng, rows, cfg["bags"] + rank + j % 7, cfg["max_len"], False)
                ly = ebc.apply_emb([torch.from_numpy(o).to(dev) for o in b[1]], [torch.from_numpy(i).to(dev) for i in b[0]])   # int64, as DLRM passes them
                torch.cuda.synchronize()
                for t in range(len(rows)):
                    want = oracle.c_bag_sum(tabs32[t], b[0][t], b[1][t])
                    got = ly[t].cpu().numpy()
                    assert got.shape == want.shape and np.abs(got - want).max(initial=0.0) <= 1e-5, (t, ebc.plan.kinds[t])
                    if ebc.plan.kinds[t] != "row_split":
                        assert np.array_equal(got, want), (t, ebc.plan.kinds[t])
            ebc.close()
            status["harness"] = ebc.plan.describe()
        status["ok"] = True
    except Exception:  # noqa: BLE001
        import traceback
        status["error"] = traceback.format_exc()
    finally:
        torch.cuda.synchronize()
        if peer is not None:
            peer.close()
        if comm is not None:
            comm.close()
        eng.close()
        if world > 1 and not cfg.get("peer"):
            dist.destroy_process_group()
    with open(cfg["out"] + ".rank%d.json" % rank, "w") as f:
        json.dump(status, f)
    sys.exit(0 if status["ok"] else 1)


if __name__ == "__main__":
    main()
