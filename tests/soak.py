"""Soak: minutes of randomised lookups -- every path (host / device / plan / checked engine / checked call / per-table
lists through the plan cache / bag router, general and one-index-per-bag, + partial-sum un-router emulating N shards in
one process), every result compared with the CPU oracle -- watching HBM and host memory.
    python tests/soak.py [seconds]     (lives under tests/: it uses the oracle as its checker)"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402
from oracle import oracle  # noqa: E402  (tool, not product: the checker)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda", 0)
eng = pel.EmbeddingEngine(device=0, max_tables=64)
chk = pel.EmbeddingEngine(device=0, max_tables=8, check_inputs=True)
rng = np.random.default_rng(0)
shapes = [(5000, 16, np.float32), (300, 128, np.float32), (70000, 10, np.float32), (9000, 64, np.float16), (40000, 32, np.float32)]
tabs = [rng.standard_normal((r, d)).astype(t) * np.float32(0.01) for r, d, t in shapes]
tabs = [w.astype(shapes[i][2]) for i, w in enumerate(tabs)]
for t, w in enumerate(tabs):
    eng.load_table(t, w)
    chk.load_table(t, w) if t < 8 else None
N = 4
rr_tables = [0, 4]                                        # routed as row-split tables over N emulated shards
rps = {t: -(-tabs[t].shape[0] // N) for t in rr_tables}
for t in rr_tables:
    for d in range(N):
        lo, hi = min(d * rps[t], tabs[t].shape[0]), min((d + 1) * rps[t], tabs[t].shape[0])
        eng.load_table(20 + t * N + d, tabs[t][lo:hi])
list_bufs = {}
t0, it, n_checked, free0, rss0 = time.time(), 0, 0, None, None
while time.time() - t0 < budget:
    t = it % len(tabs)
    nb = int(rng.integers(1, 3000))
    off, n = pel.workloads.ragged_offsets(rng, nb, 6, dtype=np.int64)
    idx = rng.integers(0, tabs[t].shape[0], size=n).astype(np.int64)
    want = oracle.c_bag_sum(tabs[t], idx, off)
    mode = it % 7
    if mode == 5:                                         # checked call on device tensors (validated, then looked up)
        got = eng.lookup(t, torch.from_numpy(idx).to(dev), torch.from_numpy(off).to(dev), check=True).cpu().numpy()
    elif mode == 6:                                       # per-table lists, a handful of recurring shapes: the plan cache
        nb = (64, 257, 1000)[(it // 7) % 3]
        off = (np.arange(nb, dtype=np.int64) * 3)
        idx = rng.integers(0, tabs[t].shape[0], size=3 * nb).astype(np.int64)
        want = oracle.c_bag_sum(tabs[t], idx, off)
        key = (t, nb)
        if key not in list_bufs:           # recurring buffers (as a serving loop has them): same addresses, new values
            list_bufs[key] = (torch.empty(3 * nb, dtype=torch.int64, device=dev), torch.from_numpy(off).to(dev),
                              torch.empty((nb, tabs[t].shape[1]), dtype=torch.float32, device=dev))
        list_bufs[key][0].copy_(torch.from_numpy(idx))
        list_bufs[key][2].fill_(float("nan"))
        got = eng.lookup_batched([t], [list_bufs[key][0].view(-1)], [list_bufs[key][1].view(-1)],
                                 [list_bufs[key][2]])[0].cpu().numpy()
    elif mode == 0:
        got = eng.lookup(t, idx, off)
    elif mode == 1:
        got = eng.lookup(t, torch.from_numpy(idx).to(dev), torch.from_numpy(off).to(dev)).cpu().numpy()
    elif mode == 2:
        p = eng.plan([t], [torch.from_numpy(idx).to(dev)], [torch.from_numpy(off).to(dev)])
        p.launch(); torch.cuda.synchronize(); got = p.outputs[0].cpu().numpy(); p.destroy()
    elif mode == 3:
        got = chk.lookup(t, idx.astype(np.uint32), off.astype(np.uint32))
    else:                                                 # router: bags -> N shards -> partial rows -> shard-ordered sum
        t = rr_tables[(it // 7) % 2]
        onehot = (it // 14) % 2 == 1                      # every other router call: one index per bag (the fast path)
        if onehot:
            off, n = np.arange(nb, dtype=np.int64), nb
        idx = rng.integers(0, tabs[t].shape[0], size=n).astype(np.int64)
        want = oracle.c_bag_sum(tabs[t], idx, off)
        dim = tabs[t].shape[1]
        d_i, d_o = torch.from_numpy(idx.astype(np.int32)).to(dev), torch.from_numpy(off.astype(np.int32)).to(dev)
        sz = eng.route_bags_sizes(1, nb, max(n, 1), N)
        bufs = {k: torch.zeros(max(v, 16), dtype=torch.uint8, device=dev) for k, v in sz.items()}
        spec = (d_i.data_ptr(), None, n, 1, rps[t]) if onehot else (d_i.data_ptr() if n else 0, d_o.data_ptr(), n, 0, rps[t])
        eng.route_bags([spec], nb, N, bufs["send"].data_ptr(),
                       bufs["meta"].data_ptr(), bufs["slots"].data_ptr(), bufs["work"].data_ptr())
        torch.cuda.synchronize()
        m = bufs["meta"].view(torch.int32).cpu().numpy().view(np.uint32)      # K = 1: counts[N][2][2] | base[N][1][2] | ...
        counts, base = m[:4 * N].reshape(N, 2, 2)[:, 0, :], m[4 * N:6 * N].reshape(N, 2)
        assert int(m[6 * N + N + 1 + N]) == (1 if onehot else 0)              # slots encoding the router chose
        words = bufs["send"].view(torch.int32)
        rets = []
        for d in range(N):
            ns, ni = int(counts[d, 0]), int(counts[d, 1])
            if ns:
                rets.append(eng.lookup(20 + t * N + d, words[int(base[d, 1]):int(base[d, 1]) + ni],
                                       words[int(base[d, 0]):int(base[d, 0]) + ns]))
        recv = torch.cat(rets) if rets else torch.zeros((1, dim), device=dev)
        pooled = torch.empty((1, nb, dim), device=dev)
        eng.unroute_bags(recv.data_ptr(), bufs["meta"].data_ptr(), bufs["slots"].data_ptr(), 1, nb, N, dim, pooled.data_ptr())
        torch.cuda.synchronize()
        got = pooled[0].cpu().numpy()
        if onehot:
            assert np.array_equal(got, want), (it, "router, one index per bag")
        assert float(np.abs(got - want).max()) <= 1e-6, (it, "router")
        got = want                                        # (re-associated: tolerance above, not bits)
    assert np.array_equal(got, want), (it, mode, t)
    n_checked += nb
    it += 1
    if it == 500:
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free0, rss0 = torch.cuda.mem_get_info()[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    if it % 2000 == 0:
        print("  %6d calls, %.0f s" % (it, time.time() - t0), flush=True)
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1, rss1 = torch.cuda.mem_get_info()[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("soak: %d calls, %d bags checked against the oracle in %.0f s -- all equal" % (it, n_checked, time.time() - t0))
print("free HBM after 500 calls %d MiB, at the end %d MiB (delta %d KiB); max RSS %d -> %d MiB"
      % (free0 >> 20, free1 >> 20, (free0 - free1) >> 10, rss0 >> 10, rss1 >> 10))
print("engine stats:", eng.stats()["n_lookup_calls"], "lookups,", eng.stats()["n_kernel_launches"], "launches,",
      eng.plan_cache_hits, "plan-cache hits")
eng.close(); chk.close()
