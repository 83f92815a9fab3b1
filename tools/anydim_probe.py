"""Device time of the element-per-thread kernel (rows that are not 16-byte multiples): 26 Kaggle-sized
fp32 tables, B = 39292, one and eight indices per bag."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import pim_embedding_lookup_amd as pel
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS; B = 39292
rng = np.random.default_rng(1)
for dim in (2, 6, 10, 30):
    for L in (1, 8):
        eng = pel.EmbeddingEngine(device=0, max_tables=26)
        for t, n in enumerate(rows):
            eng.load_table(t, torch.rand((n, dim), device=dev))
        idx = [torch.from_numpy(rng.integers(0, n, size=B * L).astype(np.int32)).to(dev) for n in rows]
        off = torch.arange(B, dtype=torch.int32, device=dev) * L
        p = eng.plan(list(range(26)), idx, [off] * 26)
        us = p.time_us(3, 20)
        print("dim %3d L=%d: %.1f us, %.2f TB/s algorithmic, %.2e gathers/s" % (dim, L, us, p.bytes()[0] / us / 1e6, 26 * B * L / us * 1e6), flush=True)
        p.destroy(); eng.close()
