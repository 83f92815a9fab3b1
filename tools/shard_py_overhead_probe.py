"""What the PYTHON face of the sharded call costs per batch (world 1, the C4 one-of-8 share, one index per bag): `forward()` with
per-table tensor lists (what a DLRM loop calls) against `submit_prepared` of a descriptor array built once (what bench.py times).
usage: python tools/shard_py_overhead_probe.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pim_embedding_lookup_amd as pel
from importlib import import_module
sh = import_module("pim-embedding-lookup_amd.sharding")
dev = torch.device("cuda", 0)
rows, dim, B, _ = pel.workloads.table_set("c4", rows_scale=0.125)
plan = sh.plan_shards(rows, dim, 4, 1, replicate_bytes=64 << 20, split_bytes=64 << 20, pooling=1.0, split_single_rank=True)
for check in (False, "deferred", "sync"):       # "deferred" is what check=True means (the default)
    eng = pel.EmbeddingEngine(device=0, max_tables=len(plan.units) + 1)
    S = sh.ShardedEmbeddingBags(plan, eng, 0, None, depth=0, check=check)
    S.load_tables(lambda t, lo, hi: torch.zeros((hi - lo, dim), device=dev))
    rng = np.random.default_rng(0)
    slots = []
    for _ in range(4):
        d_i = [torch.from_numpy(rng.integers(0, n, size=B).astype(np.int32)).to(dev) for n in rows]
        outs = [torch.empty((B, dim), device=dev) for _ in rows]
        slots.append((d_i, outs, S.prepare(d_i, None, 1, outs)))
    torch.cuda.synchronize()
    def timed(fn, n=300):
        for i in range(20): fn(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): fn(i)
        host = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        return host, (time.perf_counter() - t0) / n * 1e6
    def fwd(i):
        d_i, outs, _ = slots[i % 4]
        S.forward(None, d_i, fixed_pooling=1, outs=outs)
    def fwd_alloc(i):
        d_i, _, _ = slots[i % 4]
        S.forward(None, d_i, fixed_pooling=1)
    def prepared(i):
        S._check(S._L.emb_shard_lookup(S._h, slots[i % 4][2][0], B, S._stream(None)))
    i64 = [[x.to(torch.int64) for x in sl[0]] for sl in slots]
    def fwd_i64(i):          # what DLRM passes: int64 ids, handed to the library IN PLACE (round 5 narrowed them on the GPU first: 76 / 114 us)
        S.forward(None, i64[i % 4], fixed_pooling=1, outs=slots[i % 4][1])
    def fwd_i64_alloc(i):
        S.forward(None, i64[i % 4], fixed_pooling=1)
    for name, fn in (("forward(lists, outs given)", fwd), ("forward(lists, outs allocated)", fwd_alloc), ("forward(int64 lists, outs given)", fwd_i64),
                     ("forward(int64 lists, outs allocated)", fwd_i64_alloc), ("emb_shard_lookup on a prepared array", prepared)):
        h, w = timed(fn)
        print("check=%-8s %-38s host %.1f us per call, wall %.1f us per call" % (check, name, h, w), flush=True)
    S.report()
    S.close(); eng.close()
