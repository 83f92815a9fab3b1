"""Python-side cost of EmbeddingEngine.lookup_batched with torch CUDA tensors (26 tables) against the
raw C call with a prebuilt descriptor array and a prepared plan."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = pel.EmbeddingEngine(device=0, max_tables=26)
for t, n in enumerate(rows):
    eng.load_table(t, torch.rand((n, 16), device=dev))
rng = np.random.default_rng(1)
idx = [torch.from_numpy(rng.integers(0, n, size=B)).to(dev) for n in rows]          # int64, as torch hands them over
off = [torch.arange(B, dtype=torch.int64, device=dev) for _ in rows]
ids = list(range(26))


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


print("B=%d  lookup_batched (fresh outputs)  : %.1f us/call host" % (B, timed(lambda: eng.lookup_batched(ids, idx, off))))
outs = eng.lookup_batched(ids, idx, off)
print("B=%d  lookup_batched (outs= reused)    : %.1f us/call host" % (B, timed(lambda: eng.lookup_batched(ids, idx, off, outs))))
si, so = torch.stack(idx), torch.stack(off)
print("B=%d  lookup_stacked [T,N] / [T,B]       : %.1f us/call host" % (B, timed(lambda: eng.lookup_stacked(ids, si, so))))
assert all(torch.equal(a, b) for a, b in zip(eng.lookup_stacked(ids, si, so).unbind(0), outs))
plan = eng.plan(ids, idx, off, outs)
print("B=%d  plan.launch                      : %.1f us/call host" % (B, timed(lambda: plan.launch())))
plan.destroy()
eng.close()
