"""Developer probe: eager plan launches vs one hipGraph holding 8 of them, small batches (launch-bound)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import pim_embedding_lookup_amd as pel
dev = torch.device("cuda", 0)
rows = pel.workloads.KAGGLE_ROWS; T = len(rows)
eng = pel.EmbeddingEngine(device=0, max_tables=T)
g = torch.Generator(device=dev); g.manual_seed(0)
for t, n in enumerate(rows):
    eng.load_table(t, torch.empty((n, 16), device=dev).uniform_(-1, 1, generator=g))
for B in (1, 64, 512, 4096):
    rng = np.random.default_rng(B)
    off = torch.arange(B, dtype=torch.int32, device=dev)
    plans = []
    for j in range(8):
        idx = [torch.from_numpy(pel.workloads.uniform_indices(rng, n, B).view(np.int32)).to(dev) for n in rows]
        plans.append(eng.plan(list(range(T)), idx, [off] * T))
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        h = s.cuda_stream
        for p in plans: p.launch(h)
        torch.cuda.synchronize()
        n = 4000
        t0 = time.perf_counter()
        for i in range(n): plans[i % 8].launch(h)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e6
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for p in plans: p.launch(torch.cuda.current_stream().cuda_stream)
        gr.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n // 8): gr.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n * 1e6
    print(f"B={B:5d}: eager {eager:6.2f} us/launch, graph of 8 {graph:6.2f} us/launch")
    for p in plans: p.destroy()
eng.close()
