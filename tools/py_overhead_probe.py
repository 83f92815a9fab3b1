"""Python-side cost of the operator surface over the engine (26 Kaggle tables, torch CUDA tensors):
lookup_batched over per-table lists with the plan cache on / off, lookup_stacked, a prepared plan, the checked calls
(emb_lookup_batched_checked) against the unchecked ones, and nn.EmbeddingBag-shaped single-table forwards."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402
from importlib import import_module  # noqa: E402

dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = pel.EmbeddingEngine(device=0, max_tables=32)
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
    torch.cuda.synchronize()          # inside the clock: GPU-bound calls show their kernel time, host-bound ones their host time
    t1 = time.perf_counter()
    return (t1 - t0) / n * 1e6


def fresh_lists():                   # what an apply_emb loop does: new list objects every batch (the tensors come from its loader)
    return eng.lookup_batched(ids, list(idx), list(off))


def fresh_tensors():                 # ... and new tensor OBJECTS over the same memory (the 52 .view() calls are in the time)
    return eng.lookup_batched(ids, [t.view(-1) for t in idx], [t.view(-1) for t in off])


engine_mod = import_module("pim-embedding-lookup_amd.engine")
print("marshalling helper (_pimemb_marshal):", "loaded" if engine_mod._marshal() is not None else "ABSENT -> Python unpacking")


print("B=%d  us per call, host + device, %d calls back to back" % (B, 300))
eng.plan_cache_size = 0
print("  lookup_batched, fresh lists + fresh outputs, plan cache OFF : %6.1f" % timed(fresh_lists))
eng.plan_cache_size = 16
print("  lookup_batched, fresh lists + fresh outputs, plan cache ON  : %6.1f   (hits %d)" % (timed(fresh_lists), eng.plan_cache_hits))
print("  ... with 52 new tensor objects made per call (.view)        : %6.1f" % timed(fresh_tensors))
saved = engine_mod._MARSHAL[0]
engine_mod._MARSHAL[0] = None
print("  lookup_batched, fresh lists, Python unpacking, cache ON      : %6.1f" % timed(fresh_lists))
eng.plan_cache_size = 0
print("  lookup_batched, fresh lists, Python unpacking, cache OFF     : %6.1f" % timed(fresh_lists))
eng.plan_cache_size = 16
engine_mod._MARSHAL[0] = saved
outs = eng.lookup_batched(ids, idx, off)
print("  lookup_batched, outs= reused                (cache ON)      : %6.1f" % timed(lambda: eng.lookup_batched(ids, idx, off, outs)))
print("  lookup_batched, check=True (validated first)                : %6.1f" % timed(lambda: eng.lookup_batched(ids, idx, off, outs, check=True)))
si, so = torch.stack(idx), torch.stack(off)
print("  lookup_batched, check=\"deferred\" (verdict read by a later call) : %6.1f" % timed(lambda: eng.lookup_batched(ids, idx, off, outs, check="deferred")))
eng.check_report()
print("  lookup_stacked [T,N] / [T,B]                                : %6.1f" % timed(lambda: eng.lookup_stacked(ids, si, so)))
print("  lookup_stacked, check=True                                  : %6.1f" % timed(lambda: eng.lookup_stacked(ids, si, so, check=True)))
print("  lookup_stacked, check=\"deferred\"                            : %6.1f" % timed(lambda: eng.lookup_stacked(ids, si, so, check="deferred")))
eng.check_report()
assert all(torch.equal(a, b) for a, b in zip(eng.lookup_stacked(ids, si, so).unbind(0), outs))
assert all(torch.equal(a, b) for a, b in zip(fresh_lists(), outs))
plan = eng.plan(ids, idx, off, outs)
print("  plan.launch                                                 : %6.1f" % timed(lambda: plan.launch()))
plan.destroy()

tm = import_module("pim-embedding-lookup_amd.torch_module")
bag = tm.EmbeddingBag(rows[2], 16, engine=eng, table_id=30, _weight=torch.rand((rows[2], 16), device=dev))
i2, o2 = idx[2], off[2]
bag.trusted_inputs = True
t_tr = timed(lambda: bag(i2, o2))
bag.trusted_inputs = False
t_ck = timed(lambda: bag(i2, o2))
bag.deferred_check = True
t_df = timed(lambda: bag(i2, o2))
eng.check_report()
print("  EmbeddingBag.forward (one table), trusted / checked / deferred: %6.1f / %6.1f / %6.1f" % (t_tr, t_ck, t_df))
hz = import_module("pim-embedding-lookup_amd.dlrm_harness")
eng.close()
ebc = hz.EmbeddingBagCollection(list(rows), 16, device=0)
lS_i = [torch.from_numpy(rng.integers(0, n, size=B)).to(dev) for n in rows]
lS_o = [torch.arange(B, dtype=torch.int64, device=dev) for _ in rows]
ebc.trusted_inputs = True
t_tr = timed(lambda: ebc.apply_emb(list(lS_o), list(lS_i)))
ebc.trusted_inputs = False
t_ck = timed(lambda: ebc.apply_emb(list(lS_o), list(lS_i)))
ebc.deferred_check = True
t_df = timed(lambda: ebc.apply_emb(list(lS_o), list(lS_i)))
print("  harness apply_emb (26 tables, per-table lists), trusted / checked / deferred : %6.1f / %6.1f / %6.1f" % (t_tr, t_ck, t_df))
ebc.close()
