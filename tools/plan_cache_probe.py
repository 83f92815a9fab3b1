"""Developer probe: the plan cache under a mix of recurring call signatures and one-off ones (how often do the recurring
ones hit?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import pim_embedding_lookup_amd as pel
dev = torch.device("cuda", 0)
eng = pel.EmbeddingEngine(device=0, max_tables=8)
rng = np.random.default_rng(0)
tabs = [rng.standard_normal((r, 16)).astype(np.float32) for r in (5000, 300, 70000)]
for t, w in enumerate(tabs):
    eng.load_table(t, w)
bufs = {}
noise = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for it in range(20000):
    if it % (noise + 1) == noise:                      # recurring: 9 signatures, same buffers every time
        t, nb = (it // (noise + 1)) % 3, (64, 257, 1000)[(it // (3 * (noise + 1))) % 3]
        if (t, nb) not in bufs:
            bufs[(t, nb)] = (torch.zeros(3 * nb, dtype=torch.int64, device=dev), torch.arange(0, 3 * nb, 3, device=dev),
                             torch.empty((nb, 16), device=dev))
        i, o, out = bufs[(t, nb)]
        eng.lookup_batched([t], [i.view(-1)], [o.view(-1)], [out])
    else:                                              # one-off: fresh tensors of a random size
        nb = int(rng.integers(1, 3000))
        eng.lookup(it % 3, torch.zeros(2 * nb, dtype=torch.int64, device=dev), torch.arange(0, 2 * nb, 2, device=dev))
    if it in (100, 1000, 5000, 19999):
        print(it, "hits", eng.plan_cache_hits, "plans", len(eng._plan_cache), "seen", len(eng._plan_seen), "clock", eng._plan_clock, flush=True)
torch.cuda.synchronize()
eng.close()
