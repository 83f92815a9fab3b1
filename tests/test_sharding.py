"""CPU: shard planner (host logic) and the N>1 exchange path with world_size-2 gloo processes.
The local lookup step is the HIP engine in the product; here a test backend built on the oracle
stands in for it so the ROUTING (bucketing, all_to_all both ways, partial-sum order) is what is
checked."""
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


class OracleBackend:
    """TEST stand-in for the HIP engine's local step (numpy in, numpy out via the CPU oracle)."""

    def __init__(self):
        self.tables = {}

    def load(self, uid, rows):
        self.tables[uid] = np.ascontiguousarray(rows, dtype=np.float32)

    def lookup(self, uids, indices, offsets, outs):
        import torch
        from oracle import oracle
        res = []
        for u, i, o in zip(uids, indices, offsets):
            res.append(torch.from_numpy(oracle.c_bag_sum(self.tables[u], i.numpy().astype(np.int64),
                                                         o.numpy().astype(np.int64))))
        return res


def _worker(rank, world, port, cfg, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import oracle
    sh = _sharding()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, dim = cfg["rows"], cfg["dim"]
        plan = sh.plan_shards(rows, dim, 4, world, replicate_bytes=cfg["rep"], split_bytes=cfg["split"])
        assert set(plan.kinds) == set(cfg["expect_kinds"]), plan.kinds
        tabs = [np.random.default_rng(100 + t).standard_normal((n, dim)).astype(np.float32)
                for t, n in enumerate(rows)]
        sl = sh.ShardedLookup(plan, rank, OracleBackend(), device="cpu")
        sl.load_tables(lambda t, lo, hi: tabs[t][lo:hi])
        rng = np.random.default_rng(1000 + rank)               # every rank has its OWN bags
        idx, off = [], []
        for n in rows:
            B = cfg["bags"] + rank                               # ragged: ranks differ in bag count
            lens = rng.integers(0, cfg["max_len"] + 1, size=B)
            o = np.zeros(B, dtype=np.int64)
            o[1:] = np.cumsum(lens)[:-1]
            off.append(o)
            idx.append(rng.integers(0, n, size=int(lens.sum())).astype(np.int64))
        outs = sl.forward([torch.from_numpy(i) for i in idx], [torch.from_numpy(o) for o in off])
        worst = 0.0
        for t in range(len(rows)):
            want = oracle.c_bag_sum(tabs[t], idx[t], off[t])
            got = outs[t].numpy()
            assert got.shape == want.shape
            if plan.kinds[t] == sh.ROW_SPLIT and cfg["max_len"] > 1:
                # partial sums per shard added in shard order: re-derive exactly, and 1e-6 vs sequential
                exact = np.zeros_like(want)
                for u in (plan.units[i] for i in plan.units_of_table[t]):
                    keep = (idx[t] >= u.row_lo) & (idx[t] < u.row_hi)
                    bag_of = np.repeat(np.arange(len(off[t])), np.diff(np.append(off[t], len(idx[t]))))
                    l2 = np.bincount(bag_of[keep], minlength=len(off[t]))
                    o2 = np.zeros(len(off[t]), np.int64); o2[1:] = np.cumsum(l2)[:-1]
                    part = oracle.c_bag_sum(tabs[t][u.row_lo:u.row_hi], idx[t][keep] - u.row_lo, o2)
                    exact = exact + part if u.owner else part
                assert np.array_equal(got, exact)
                assert np.abs(got - want).max() <= 1e-5
                worst = max(worst, float(np.abs(got - want).max()))
            else:
                assert np.array_equal(got, want), f"table {t} ({plan.kinds[t]})"
        q.put((rank, "ok", worst))
    except Exception as e:  # pragma: no cover
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
def test_gloo_exchange_matches_single_process_oracle(cfg, world):
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
