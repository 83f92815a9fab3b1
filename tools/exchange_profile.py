"""Where the host time of one counts-first exchange step goes (RowRangeExchange, ONE RCCL rank, C2's five big tables):
microseconds per step, then a cProfile of 300 steps."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29591")
import numpy as np, torch, torch.distributed as dist
import pim_embedding_lookup_amd as pel
from importlib import import_module
sh = import_module("pim-embedding-lookup_amd.sharding")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
eng = pel.EmbeddingEngine(device=0, max_tables=16)
K, B, dim, L = 5, 39292, 16, 1
rows = [10131227, 2202608, 8351593, 5461306, 7046547]
for k, r in enumerate(rows):
    eng.load_table(k, torch.randn(r, dim, device=dev))
ex = sh.RowRangeExchange(eng, list(range(K)), rows, dim, 0, 1, dev, n_slots=3)
rng = np.random.default_rng(0)
idx = [torch.from_numpy(np.stack([pel.workloads.uniform_indices(rng, r, B*L).view(np.int32) for r in rows])).to(dev) for _ in range(3)]
specs = [eng.route_tables([(idx[j][k].data_ptr(), None, B*L, L, rows[k]) for k in range(K)]) for j in range(3)]
outs = [torch.empty((K, B, dim), device=dev) for _ in range(3)]
def step(i):
    j, n = i % 3, (i+1) % 3
    ex.route(n, specs[n], B, K*B*L)
    ex.serve(j)
    ex.send_requests(n)
    ex.finish(j, outs[j])
ex.route(0, specs[0], B, K*B*L); ex.send_requests(0)
for i in range(50): step(i)
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(50, 350): step(i)
torch.cuda.synchronize()
print("us/step", (time.perf_counter()-t0)/300*1e6)
pr = cProfile.Profile(); pr.enable()
for i in range(350, 650): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
