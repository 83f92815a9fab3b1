"""Developer probe: can two ranks share one GPU under RCCL (needed to rehearse multi-rank collectives
on a one-GPU box)?"""
import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
a = torch.full((world * 4,), float(rank), device=dev); b = torch.empty_like(a)
dist.all_to_all_single(b, a)
torch.cuda.synchronize()
print("rank", rank, "got", b.tolist(), flush=True)
dist.destroy_process_group()
