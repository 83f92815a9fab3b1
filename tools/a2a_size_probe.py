"""Does one RCCL rank move large self all-to-all payloads intact?  (c5's one-rank rehearsal sends 3.2 GB to itself.)
Checks all_to_all_single by payload size and element type: uint8 vs the same bytes viewed as int64."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29592")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for gb in (0.5, 0.9, 1.0, 1.1, 1.5, 2.2, 3.3):
    n = int(gb * (1 << 30)) // 16 * 16
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    src.view(torch.int64).copy_(torch.arange(n // 8, device=dev, dtype=torch.int64) * 2654435761)
    for name, view in (("uint8", lambda t: t), ("int64", lambda t: t.view(torch.int64))):
        dst = torch.zeros(n, dtype=torch.uint8, device=dev)
        s, d = view(src), view(dst)
        dist.all_to_all_single(d, s, output_split_sizes=[s.numel()], input_split_sizes=[s.numel()])
        torch.cuda.synchronize()
        ok = bool(torch.equal(dst, src))
        first_bad = -1
        if not ok:
            first_bad = int((dst.view(torch.int64) != src.view(torch.int64)).nonzero()[0].item()) * 8
        print("%.1f GiB as %-5s: %s%s" % (gb, name, "intact" if ok else "CORRUPT", "" if ok else " (first bad byte %d)" % first_bad), flush=True)
        del dst
    del src
dist.destroy_process_group() if False else None

# the native exchange (emb_comm_all_to_all) cuts every pair's transfer into 512-MiB sends / receives
import pim_embedding_lookup_amd as pel  # noqa: E402
eng = pel.EmbeddingEngine(device=0, max_tables=2)
ex = pel.NativeExchange(eng, 0, 1, lambda raw: raw)
for gb in (0.9, 1.1, 2.2, 3.3):
    n = int(gb * (1 << 30)) // 16 * 16
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    src.view(torch.int64).copy_(torch.arange(n // 8, device=dev, dtype=torch.int64) * 2654435761)
    dst = torch.zeros(n, dtype=torch.uint8, device=dev)
    ex.all_to_all(src.data_ptr(), [0, n], dst.data_ptr(), [0, n], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("%.1f GiB through emb_comm_all_to_all (512-MiB pieces): %s" % (gb, "intact" if torch.equal(dst, src) else "CORRUPT"), flush=True)
    del src, dst
ex.close()
eng.close()
