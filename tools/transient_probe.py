"""Host cost of the transient (no prepared plan) device-pointer emb_lookup_batched at the C2 shape,
against emb_plan_launch of the same work.  python transient_probe.py [bags_per_table]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else pel.workloads.KAGGLE_BATCH
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
eng = pel.EmbeddingEngine(device=0, max_tables=26)
tabs = [torch.rand((n, 16), device=dev) for n in rows]
for t in range(26):
    eng.load_table(t, tabs[t])
rng = np.random.default_rng(1)
idx = [torch.from_numpy(pel.workloads.uniform_indices(rng, n, B).view(np.int32)).to(dev) for n in rows]
off = torch.arange(B, dtype=torch.int32, device=dev)
out = [torch.empty((B, 16), device=dev) for _ in rows]
L = pel.lib.load()
descs = (pel.lib.EmbLookupDesc * 26)()
for t in range(26):
    descs[t] = pel.lib.EmbLookupDesc(t, 0, idx[t].data_ptr(), off.data_ptr(), B, B, out[t].data_ptr())
stream = torch.cuda.current_stream(dev).cuda_stream
plan = eng.plan(list(range(26)), idx, [off] * 26, out)


def timed(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def transient():
    rc = L.emb_lookup_batched(eng._h, descs, 26, pel.lib.EMB_IDX_U32, pel.lib.EMB_MEM_DEVICE, stream)
    assert rc == 0


for o in out:
    o.zero_()
transient()
torch.cuda.synchronize()
assert all(torch.equal(out[t], tabs[t][idx[t].long()]) for t in range(26)), "transient launch: wrong rows"
print("B=%d  transient emb_lookup_batched: host %.1f us/call, wall %.1f us/call" % ((B,) + timed(transient)))
print("B=%d  emb_plan_launch            : host %.1f us/call, wall %.1f us/call" % ((B,) + timed(lambda: plan.launch(stream))))
for p in (plan,):
    p.destroy()
eng.close()
