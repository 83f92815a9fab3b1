"""Developer probe: host-side cost of torch.distributed.all_to_all_single over RCCL (world size 1),
idle GPU vs queued work, equal splits vs explicit split lists."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for nbytes in (1 << 16, 13 << 20):
    a = torch.zeros(nbytes, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    for mode in ("equal", "lists"):
        kw = {} if mode == "equal" else dict(output_split_sizes=[nbytes], input_split_sizes=[nbytes])
        for sync_each in (True, False):
            for _ in range(20):
                dist.all_to_all_single(b, a, async_op=True, **kw).wait()
            torch.cuda.synchronize()
            n = 300; call = 0.0; t0 = time.perf_counter()
            for _ in range(n):
                t = time.perf_counter()
                w = dist.all_to_all_single(b, a, async_op=True, **kw)
                call += time.perf_counter() - t
                w.wait()
                if sync_each: torch.cuda.synchronize()
            torch.cuda.synchronize()
            tot = time.perf_counter() - t0
            print(f"{nbytes>>10:6d} KiB {mode:6s} sync_each={sync_each!s:5s}: call {call/n*1e6:6.1f} us, period {tot/n*1e6:6.1f} us")
dist.destroy_process_group()

# lower-level entry: the ProcessGroup object itself (skips the Python wrapper's checks)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pg = dist.distributed_c10d._get_default_group()
a = torch.zeros(13 << 20, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
for name, fn in (("wrapper", lambda: dist.all_to_all_single(b, a, async_op=True)),
                 ("pg.alltoall_base", lambda: pg.alltoall_base(b, a, [], []))):
    for _ in range(20):
        fn().wait()
    torch.cuda.synchronize()
    n = 500; call = 0.0; t0 = time.perf_counter()
    for _ in range(n):
        t = time.perf_counter(); w = fn(); call += time.perf_counter() - t; w.wait()
    torch.cuda.synchronize()
    print(f"13312 KiB {name:18s}: call {call/n*1e6:6.1f} us, period {(time.perf_counter()-t0)/n*1e6:6.1f} us")
dist.destroy_process_group()
