"""Device time of emb_route_bags / emb_unroute_bags at the C4 shape (8 row-split tables, B = 16384 bags, 8 shards).
L = 1 runs twice: the one-index-per-bag path (default) and, in a child process with PIMEMB_ROUTE_ONEHOT=0, the general
per-(bag, shard) path it replaces for that shape."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

only = [int(x) for x in sys.argv[1:]] or [1, 32]
dev = torch.device("cuda", 0)
eng = pel.EmbeddingEngine(device=0, max_tables=4)
K, N, B, dim = 8, 8, 16384, 128
rows = [227605432, 130229467, 3067956, 405282, 292775614, 40790948, 187188510, 590152]
rps = [-(-r // N) for r in rows]
rng = np.random.default_rng(0)
tag = "general path forced" if os.environ.get("PIMEMB_ROUTE_ONEHOT") == "0" else "default"
for L in only:
    idx = [torch.from_numpy(pel.workloads.uniform_indices(rng, r, B * L).view(np.int32)).to(dev) for r in rows]
    sz = eng.route_bags_sizes(K, B, K * B * L, N)
    u8 = lambda n: torch.zeros(max(n, 16), dtype=torch.uint8, device=dev)
    send, meta, slots, work = u8(sz["send"]), u8(sz["meta"]), u8(sz["slots"]), u8(sz["work"])
    spec = eng.route_tables([(idx[k].data_ptr(), None, B * L, L, rps[k]) for k in range(K)])
    h = torch.cuda.current_stream().cuda_stream

    def route():
        eng.route_bags(spec, B, N, send.data_ptr(), meta.data_ptr(), slots.data_ptr(), work.data_ptr(), h)

    route()
    torch.cuda.synchronize()
    m = meta.view(torch.int32).cpu().numpy().view(np.uint32)
    n_sub = int(m[:2 * N * (K + 1)].reshape(N, K + 1, 2)[:, :K, 0].sum())
    mode = int(m[2 * N * (K + 1) + 3 * N * K + N + 1])
    recv = torch.zeros((n_sub, dim), dtype=torch.float32, device=dev)
    out = torch.empty((K, B, dim), dtype=torch.float32, device=dev)

    def unroute():
        eng.unroute_bags(recv.data_ptr(), meta.data_ptr(), slots.data_ptr(), K, B, N, dim, out.data_ptr(), h)

    total = 0.0
    for name, fn in (("route_bags", route), ("unroute_bags", unroute)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        total += us
        print("L=%2d  %-14s %8.1f us   (%s, slots mode %d, sub-bags %d, %.1f MB of partial rows)"
              % (L, name, us, tag, mode, n_sub, n_sub * dim * 4 / 1e6), flush=True)
    print("L=%2d  router + un-router %6.1f us" % (L, total), flush=True)
eng.close()
if 1 in only and os.environ.get("PIMEMB_ROUTE_ONEHOT") != "0":
    subprocess.run([sys.executable, os.path.abspath(__file__), "1"], env=dict(os.environ, PIMEMB_ROUTE_ONEHOT="0"))
