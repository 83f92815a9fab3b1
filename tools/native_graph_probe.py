"""Can one hipGraph hold [engine launch, native RCCL all-to-all, engine launch]?  (torch's own
all_to_all_single could not be captured: graph_probe.py.)  One rank, self exchange."""
import faulthandler
import os
import sys
import time

faulthandler.enable()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
eng = pel.EmbeddingEngine(device=0, max_tables=4)
w = torch.randn(100000, 16, device=dev)
eng.load_table(0, w)
B = 200000
idx = torch.randint(0, 100000, (B,), dtype=torch.int32, device=dev)
off = torch.arange(B, dtype=torch.int32, device=dev)
send = torch.empty((B, 16), device=dev)
recv = torch.zeros((B, 16), device=dev)
idx2 = torch.randint(0, 100000, (B,), dtype=torch.int32, device=dev)
out2 = torch.empty((B, 16), device=dev)
plan1 = eng.plan([0], [idx], [off], [send])          # pooled rows -> send buffer
plan2 = eng.plan([0], [idx2], [off], [out2])
ex = pel.NativeExchange(eng, 0, 1, lambda raw: raw)
offs = ex.offsets([0, send.numel() * 4])
s = torch.cuda.Stream(dev)


def step(h):
    plan1.launch(h)
    ex.all_to_all(send.data_ptr(), offs, recv.data_ptr(), offs, h)
    plan2.launch(h)


with torch.cuda.stream(s):
    step(s.cuda_stream)
s.synchronize()
assert torch.equal(recv, w[idx.long()]) and torch.equal(out2, w[idx2.long()])
print("eager step ok", flush=True)
recv.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    for _ in range(8):
        step(torch.cuda.current_stream().cuda_stream)
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
assert torch.equal(recv, w[idx.long()])
print("replayed, results ok", flush=True)
for name, fn, n in (("eager", lambda: [step(s.cuda_stream) for _ in range(8)], 50), ("graph", g.replay, 50)):
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    print("%s: %.1f us per step" % (name, (time.perf_counter() - t0) / n / 8 * 1e6), flush=True)
# Teardown, ONE ordered attempt (round 2): RCCL keeps a reference per captured graph (persistent refs released by
# a user-object destructor when the graph is destroyed) and ncclCommDestroy waits for them to drop -- so: all work
# drained, graph exec + graph destroyed FIRST, device drained again, THEN the communicator.  If this still hangs,
# faulthandler prints the Python frame it hangs in after 40 s and ends the process.
torch.cuda.synchronize()
faulthandler.dump_traceback_later(40, exit=True)
print("teardown: graph reset", flush=True)
g.reset()
del g
torch.cuda.synchronize()
print("teardown: graph gone, destroying the communicator", flush=True)
ex.close()
print("teardown: communicator gone", flush=True)
plan1.destroy(); plan2.destroy(); eng.close()
faulthandler.cancel_dump_traceback_later()
print("teardown complete", flush=True)
