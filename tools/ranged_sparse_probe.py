"""What one rank's direct-path launch costs when N ranks share the row-split tables: 20 replicated tables (dense: every bag
is served) + 6 shards x N sources (each source's raw index array scanned, 1/N of its bags served).  One GPU: the sources'
arrays all live here, so this prices the SCAN (instructions, index reads), not the links.
usage: python tools/ranged_sparse_probe.py [N ...]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pim_embedding_lookup_amd as pel  # noqa: E402

dev = torch.device("cuda", 0)
dim, B, n_rep, n_split, rows = 128, 16384, 20, 6, 400_000
L = pel.lib.load()
eng = pel.EmbeddingEngine(device=0, max_tables=n_rep + n_split + 1)
g = torch.Generator(device=dev).manual_seed(1)
for t in range(n_rep + n_split):
    eng.load_table(t, torch.randn(rows, dim, device=dev, generator=g) * 0.05)


def run(N, reps=300):
    descs, lo, keep = [], [], []
    for t in range(n_rep):
        idx = torch.randint(0, rows, (B,), device=dev, dtype=torch.int32)
        out = torch.empty(B, dim, device=dev)
        keep += [idx, out]
        descs.append(pel.lib.EmbLookupDesc(t, 1, idx.data_ptr(), None, B, B, out.data_ptr()))
        lo.append(0)
    me = N // 2
    for p in range(N):
        for k in range(n_split):
            idx = torch.randint(0, rows * N, (B,), device=dev, dtype=torch.int32)      # the table has rows*N rows; this rank holds shard `me`
            out = torch.empty(B, dim, device=dev)
            keep += [idx, out]
            descs.append(pel.lib.EmbLookupDesc(n_rep + k, 1, idx.data_ptr(), None, B, B, out.data_ptr()))
            lo.append(me * rows)
    n = len(descs)
    arr = (pel.lib.EmbLookupDesc * n)(*descs)
    los = (C.c_uint64 * n)(*lo)
    plan = C.c_void_p()
    pel.lib.check(L.emb_plan_create_ranged(eng._h, arr, los, n, C.byref(plan)))
    for _ in range(50):
        pel.lib.check(L.emb_plan_launch(plan, None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pel.lib.check(L.emb_plan_launch(plan, None))
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    pel.lib.check(L.emb_plan_destroy(plan))
    served = (n_rep + n_split) * B
    print(f"N = {N}: {n} descriptors, {served} rows served of {n * B} bags scanned: {us:.1f} us / launch", flush=True)


for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    run(N)
eng.close()
