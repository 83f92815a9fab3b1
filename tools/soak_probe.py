import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import pim_embedding_lookup_amd as pel
dev = torch.device("cuda", 0)
eng = pel.EmbeddingEngine(device=0, max_tables=8)
rng = np.random.default_rng(0)
tabs = [rng.standard_normal((r, d)).astype(np.float32) for r, d in ((5000, 16), (300, 128), (70000, 10))]
for t, w in enumerate(tabs):
    eng.load_table(t, w)
free0 = None
for it in range(6000):
    t = it % 3
    nb = int(rng.integers(1, 3000))
    off, n = pel.workloads.ragged_offsets(rng, nb, 6, dtype=np.int64)
    idx = rng.integers(0, tabs[t].shape[0], size=n).astype(np.int64)
    if it % 2:
        out = eng.lookup(t, torch.from_numpy(idx).to(dev), torch.from_numpy(off).to(dev))
    else:
        out = eng.lookup(t, idx, off)
    if it % 7 == 0:
        p = eng.plan([t], [torch.from_numpy(idx).to(dev)], [torch.from_numpy(off).to(dev)])
        p.launch(); torch.cuda.synchronize(); p.destroy()
    if it == 1000:
        torch.cuda.synchronize(); torch.cuda.empty_cache(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("free HBM after 1000 calls: %d MiB, after 6000: %d MiB, delta %d KiB" % (free0 >> 20, free1 >> 20, (free0 - free1) >> 10))
import resource
print("max RSS MiB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)
eng.close()
