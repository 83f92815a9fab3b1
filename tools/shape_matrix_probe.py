"""Device time per fused launch over a matrix of table dtypes, dims, pooling factors and index types
(26 Kaggle-sized tables, B bags per table): algorithmic TB/s per cell, to spot combinations that fall
off the curve.   python shape_matrix_probe.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(1)
print("%-8s %4s %3s %4s %10s %8s  %s" % ("dtype", "dim", "L", "idx", "us", "TB/s", "kinds"))
for dname, tdt, mk in (("f32", torch.float32, None), ("f16", torch.float16, None), ("fixed32", torch.int32, pel.EMB_FIXED32)):
    for dim in (16, 64, 128, 256):
        eng = pel.EmbeddingEngine(device=0, max_tables=26)
        for t, n in enumerate(rows):
            if tdt == torch.int32:
                eng.load_table(t, torch.randint(-2**31, 2**31 - 1, (n, dim), dtype=torch.int32, device=dev), dtype=mk)
            else:
                eng.load_table(t, (torch.rand((n, dim), device=dev) - 0.5).to(tdt))
        for L in (1, 32):
            for iname, idt in (("u32", np.uint32), ("i64", np.int64)):
                if L == 32 and iname == "i64" and dim > 64:
                    continue
                plans = []
                for j in range(2):
                    idx = [torch.from_numpy(rng.integers(0, n, size=B * L).astype(idt).view(np.int32 if idt == np.uint32 else np.int64)).to(dev)
                           for n in rows]
                    off = torch.from_numpy((np.arange(B) * L).astype(idt).view(np.int32 if idt == np.uint32 else np.int64)).to(dev)
                    plans.append(eng.plan(list(range(26)), idx, [off] * 26))
                eng.reset_stats()
                us = float(np.mean([p.time_us(3, 10 if L > 1 else 40) for p in plans]))
                print("%-8s %4d %3d %4s %10.1f %8.2f  %s" % (dname, dim, L, iname, us, plans[0].bytes()[0] / us / 1e6,
                                                           [int(x > 0) for x in eng.stats()["n_launches_by_kind"]]), flush=True)
                for p in plans:
                    p.destroy()
        eng.close()
        torch.cuda.empty_cache()
