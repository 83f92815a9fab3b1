"""Developer probe: can several RCCL ranks share ONE GPU if every rank claims to be on a different host (NCCL_HOSTID), so that
RCCL neither sees a duplicate GPU nor tries P2P / shared memory and falls back to its socket transport over loopback?
usage: python tools/rccl_one_gpu_ranks_probe.py [world]     (parent spawns the ranks)"""
import os
import subprocess
import sys

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if "RANK" not in os.environ:
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29621",
                   NCCL_HOSTID="fakehost%d" % r, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                   NCCL_SHM_DISABLE="1", NCCL_NET_GDR_LEVEL="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), str(world)], env=env))
    rc = [p.wait(timeout=280) for p in procs]
    print("exit codes", rc)
    sys.exit(max(rc))

import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
# unequal splits: rank r sends (p + 1) * (r + 1) * 1000 bytes to peer p
send_counts = [(p + 1) * (rank + 1) * 1000 for p in range(world)]
recv_counts = [(rank + 1) * (p + 1) * 1000 for p in range(world)]
send = torch.cat([torch.full((n,), 16 * rank + p, dtype=torch.uint8, device=dev) for p, n in enumerate(send_counts)])
recv = torch.zeros(sum(recv_counts), dtype=torch.uint8, device=dev)
dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts)
torch.cuda.synchronize()
off = 0
for p, n in enumerate(recv_counts):
    assert bool((recv[off:off + n] == 16 * p + rank).all()), (rank, p)
    off += n
x = torch.tensor([float(rank)], device=dev)
dist.all_reduce(x)
assert float(x) == sum(range(world))
print("rank", rank, "of", world, "on one GPU: all_to_all_single with unequal splits + all_reduce over RCCL OK", flush=True)
dist.destroy_process_group()
