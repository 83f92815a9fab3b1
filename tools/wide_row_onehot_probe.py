"""One-hot lookups over WIDE rows (dim 256 fp32 = 1 KiB, the widest the lane-piece kernels take; dim 128 beside it): 26 Kaggle-sized
tables, B = 39 292, prepared plan, HIP events.  The dim-256 wave-batch instantiation spills two registers under the 64-VGPR cap
(codeobj.kernel_resources); PIMEMB_LPR64_ONEHOT_INFLIGHT (a build flag) picks how many rounds it keeps in flight.
usage: python tools/wide_row_onehot_probe.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pim_embedding_lookup_amd as pel
from importlib import import_module
codeobj = import_module("pim-embedding-lookup_amd.codeobj")
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
B = 39292
for dim in (128, 256):
    eng = pel.EmbeddingEngine(device=0, max_tables=len(rows))
    for t, n in enumerate(rows):
        eng.load_table(t, torch.randn((n, dim), device=dev) * 0.05)
    rng = np.random.default_rng(0)
    plans = []
    for _ in range(4):
        idx = [torch.from_numpy(pel.workloads.uniform_indices(rng, n, B).view(np.int32)).to(dev) for n in rows]
        off = [torch.arange(B, dtype=torch.int32, device=dev) for _ in rows]
        outs = [torch.empty((B, dim), device=dev) for _ in rows]
        plans.append(eng.plan(list(range(len(rows))), idx, off, outs))
    s = torch.cuda.current_stream(dev).cuda_stream
    for p in plans:
        p.launch(s)
    torch.cuda.synchronize()
    us = [p.time_us(10, 50, s) for p in plans]
    alg = plans[0].bytes()[0]
    launch = plans[0].describe()[0]
    sym, _ = codeobj.kernel_of_launch(os.path.join(ROOT, "pim-embedding-lookup_amd", "lib", "libpimemb.so"), launch)
    res = codeobj.kernel_resources(os.path.join(ROOT, "pim-embedding-lookup_amd", "lib", "libpimemb.so"))[sym]
    print("dim %d: %.1f us per launch (min of 4 plans %.1f), %.2f TB/s algorithmic; kernel vgpr %d spill %d scratch %d" % (
        dim, float(np.mean(us)), min(us), alg / (min(us) * 1e-6) / 1e12, res["vgpr"], res["vgpr_spill"], res["scratch"]), flush=True)
    for p in plans:
        p.destroy()
    eng.close()
