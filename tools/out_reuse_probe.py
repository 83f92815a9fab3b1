"""Does it matter where the pooled rows land?  The bench rotates 8 batches, each with its own 65 MB of outputs (524 MB:
larger than the 256 MB memory-side cache).  A DLRM loop gets the SAME output addresses back from torch's caching
allocator every batch.  This probe times the C2 launch over 8 rotating index batches writing (a) 8 output sets,
(b) one output set, (c) one output set that a consumer kernel reads between launches (the interaction layer's read).
python out_reuse_probe.py [bags_per_table]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else pel.workloads.KAGGLE_BATCH
NB = 8
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS
eng = pel.EmbeddingEngine(device=0, max_tables=26)
tabs = [torch.rand((n, 16), device=dev) for n in rows]
for t in range(26):
    eng.load_table(t, tabs[t])
rng = np.random.default_rng(1)
idx = [[torch.from_numpy(pel.workloads.uniform_indices(rng, n, B).view(np.int32)).to(dev) for n in rows] for _ in range(NB)]
off = torch.arange(B, dtype=torch.int32, device=dev)
outs = [torch.empty((26, B, 16), device=dev) for _ in range(NB)]
stream = torch.cuda.current_stream(dev).cuda_stream
ids = list(range(26))
plans_own = [eng.plan(ids, idx[b], [off] * 26, list(outs[b].unbind(0))) for b in range(NB)]
plans_one = [eng.plan(ids, idx[b], [off] * 26, list(outs[0].unbind(0))) for b in range(NB)]

for b in range(NB):          # what is timed is right
    outs[0].zero_()
    plans_one[b].launch(stream)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0][t], tabs[t][idx[b][t].long()]) for t in range(26))


def timed(plans, between=None, n=2000):
    for i in range(200):
        plans[i % NB].launch(stream)
        if between:
            between()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        plans[i % NB].launch(stream)
        if between:
            between()
    z.record()
    z.synchronize()
    return a.elapsed_time(z) / n * 1e3


acc = torch.empty((B, 16), device=dev)


def consumer():              # reads every pooled row once (65 MB), writes 2.5 MB
    torch.sum(outs[0], dim=0, out=acc)


t_cons = timed([type("P", (), {"launch": staticmethod(lambda s: None)})()] * NB, consumer)
print("B=%d, 8 rotating index batches" % B)
print("  own outputs per batch (8 x %.1f MB) : %.2f us / launch" % (outs[0].numel() * 4 / 1e6, timed(plans_own)))
print("  one output set                       : %.2f us / launch" % timed(plans_one))
print("  one output set + consumer read       : %.2f us / step (consumer alone %.2f us)" % (timed(plans_one, consumer), t_cons))
for p in plans_own + plans_one:
    p.destroy()
eng.close()
