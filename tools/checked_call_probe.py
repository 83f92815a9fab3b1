"""A checked 26-table call (Kaggle row counts, int64 ids as torch hands them over, one id per bag) N times back to back: what the
validation costs on the GPU.  Run under `rocprofv3 --kernel-trace --stats` to see validate_kernel / validate_publish_kernel next
to the lookup kernel.  usage: python tools/checked_call_probe.py [bags per table] [sync|deferred|off] [calls]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 39292
mode = sys.argv[2] if len(sys.argv) > 2 else "deferred"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
eng = pel.EmbeddingEngine(device=0, max_tables=32, lib_path=os.environ.get("PIMEMB_PROBE_LIB"))     # (another build: tools/validate_variants.sh)
for t, n in enumerate(rows):
    eng.load_table(t, torch.rand((n, 16), device=dev))
rng = np.random.default_rng(1)
idx = [torch.from_numpy(rng.integers(0, n, size=B)).to(dev) for n in rows]
off = [torch.arange(B, dtype=torch.int64, device=dev) for _ in rows]
ids = list(range(26))
outs = eng.lookup_batched(ids, idx, off)
check = {"sync": True, "deferred": "deferred", "off": False}[mode]
eng.plan_cache_size = 0
for _ in range(20):
    eng.lookup_batched(ids, idx, off, outs, check=check)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    eng.lookup_batched(ids, idx, off, outs, check=check)
host = (time.perf_counter() - t0) / calls * 1e6
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / calls * 1e6
eng.check_report()
print("B=%d check=%s: host %.1f us per call, wall %.1f us per call" % (B, mode, host, wall))
eng.close()
