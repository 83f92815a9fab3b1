import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pim_embedding_lookup_amd as pel
which = sys.argv[1]
dev = torch.device("cuda", 0)
if which in ("kernel", "both"):
    eng = pel.EmbeddingEngine(device=0, max_tables=4)
    w = torch.randn(1000, 16, device=dev)
    eng.load_table(0, w)
    idx = torch.randint(0, 1000, (4096,), dtype=torch.int32, device=dev)
    off = torch.arange(4096, dtype=torch.int32, device=dev)
    plan = eng.plan([0], [idx], [off])
    plan.launch(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
if which in ("nccl", "both"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    a = torch.arange(1024, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    dist.all_to_all_single(b, a); torch.cuda.synchronize()
s = torch.cuda.Stream(dev)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    if which in ("kernel", "both"):
        plan.launch(torch.cuda.current_stream().cuda_stream)
    if which in ("nccl", "both"):
        wk = dist.all_to_all_single(b, a, async_op=True); wk.wait()
print("captured", which, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", which, flush=True)
if which in ("kernel", "both"):
    assert torch.equal(plan.outputs[0], w[idx.long()])
    print("kernel output ok")
