"""The C4 one-of-8 share (26 Terabyte-shaped tables at 1/8 of their rows, dim 128 fp32 = 512-byte rows, 16 384 bags, ONE index per
bag) as a prepared plan, HIP events: us per launch of the library at $PIMEMB_PROBE_LIB (default: the shipped one).  The streaming
floor for this launch's measured bytes (79 MB read, 218 MB written) is 46.4 us (tools/bin/stream_floor 79371110 218125004).
usage: python tools/onehot_inflight_probe.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pim_embedding_lookup_amd as pel
dev = torch.device("cuda", 0)
rows, dim, B, _ = pel.workloads.table_set("c4", rows_scale=0.125)
eng = pel.EmbeddingEngine(device=0, max_tables=len(rows), lib_path=os.environ.get("PIMEMB_PROBE_LIB"))
for t, n in enumerate(rows):
    eng.alloc_table(t, n, dim, pel.EMB_F32)
rng = np.random.default_rng(0)
plans = []
for _ in range(8):          # eight batches rotated, like bench.py: one plan launched over and over finds its rows in the 256-MB last-level cache
    idx = [torch.from_numpy(pel.workloads.uniform_indices(rng, n, B).view(np.int32)).to(dev) for n in rows]
    off = [torch.arange(B, dtype=torch.int32, device=dev) for _ in rows]
    outs = [torch.empty((B, dim), device=dev) for _ in rows]
    plans.append(eng.plan(list(range(len(rows))), idx, off, outs))
s = torch.cuda.current_stream(dev).cuda_stream
us = []
for rep in range(6):
    for i in range(40):
        plans[i % 8].launch(s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(400):
        plans[i % 8].launch(s)
    b.record()
    torch.cuda.synchronize()
    us.append(a.elapsed_time(b) * 1e3 / 400)
print("%s: %.2f us per launch (min %.2f, max %.2f over 6 timings of 400 launches, 8 batches rotated)" % (os.environ.get("PIMEMB_PROBE_TAG", "shipped"), float(np.mean(us)), min(us), max(us)), flush=True)
for p in plans:
    p.destroy()
eng.close()
