"""One-hot lookups on fp16 tables (26 Kaggle-sized tables, B = 39292): device time per fused launch
for a given build of the library.   python f16_onehot_probe.py [lib_path]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

lib_path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
B = int(sys.argv[2]) if len(sys.argv) > 2 else pel.workloads.KAGGLE_BATCH
rng = np.random.default_rng(1)
for dim in (16, 32, 64, 128):
    eng = pel.EmbeddingEngine(device=0, max_tables=26, lib_path=lib_path)
    for t, n in enumerate(rows):
        eng.load_table(t, (torch.rand((n, dim), device=dev) - 0.5).to(torch.float16))
    off = torch.arange(B, dtype=torch.int32, device=dev)
    plans = []
    for j in range(4):
        idx = [torch.from_numpy(pel.workloads.uniform_indices(rng, n, B).view(np.int32)).to(dev) for n in rows]
        plans.append(eng.plan(list(range(26)), idx, [off] * 26))
    us = min(np.mean([p.time_us(5, 40) for p in plans]) for _ in range(3))
    alg = plans[0].bytes()[0]
    print("fp16 dim %3d: %.2f us per launch, %.0f GB/s algorithmic, kinds %s" % (dim, us, alg / us / 1e3,
                                                                             eng.stats()["n_launches_by_kind"]), flush=True)
    for p in plans:
        p.destroy()
    eng.close()
    torch.cuda.empty_cache()
