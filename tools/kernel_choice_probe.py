"""ONE-INDEX-PER-BAG launches of few bags: the lane-group kernel against the wave-batch kernel (choose_kernel's threshold).
T tables of 2^20 rows, B bags each, dims 16 / 64 / 128 / 256 fp32, uint32 ids; device time per prepared launch with the kernel
forced either way (PIMEMB_FORCE_KERNEL, read once per process: the script runs itself once per kernel).
usage: python tools/kernel_choice_probe.py            (prints the table)"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
DIMS = (16, 64, 128, 256)
SHAPES = [(6, 2048), (26, 2048), (6, 8192), (6, 16384), (26, 8192), (6, 32768), (12, 32768), (26, 16384), (6, 131072), (26, 39292)]

if len(sys.argv) > 1 and sys.argv[1] == "--leg":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import pim_embedding_lookup_amd as pel
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    for dim in DIMS:
        eng = pel.EmbeddingEngine(device=0, max_tables=32)
        for t in range(26):
            eng.load_table(t, torch.rand((1 << 20, dim), device=dev))
        for T, B in SHAPES:
            plans = []
            for _ in range(2):
                idx = [torch.from_numpy(rng.integers(0, 1 << 20, size=B).astype(np.uint32).view(np.int32)).to(dev) for _ in range(T)]
                off = torch.arange(B, dtype=torch.int32, device=dev)
                plans.append(eng.plan(list(range(T)), idx, [off] * T))
            eng.reset_stats()
            us = float(np.mean([p.time_us(5, 60) for p in plans]))
            kinds = eng.stats()["n_launches_by_kind"]
            print("%d %d %d %.2f %s" % (dim, T, B, us, "".join(str(int(k > 0)) for k in kinds)), flush=True)
            for p in plans:
                p.destroy()
        eng.close()
        torch.cuda.empty_cache()
    sys.exit(0)

res = {}
for kind in ("group", "wavebatch", ""):
    env = dict(os.environ)
    env.pop("PIMEMB_FORCE_KERNEL", None)
    if kind:
        env["PIMEMB_FORCE_KERNEL"] = kind
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg"], env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        print(out.stderr[-2000:])
        sys.exit(1)
    for line in out.stdout.splitlines():
        f = line.split()
        if len(f) == 5 and f[0].isdigit():
            res[(int(f[0]), int(f[1]), int(f[2]), kind or "chosen")] = (float(f[3]), f[4])
print("| dim | tables x bags | wave batches of 64 | lane-group us | wave-batch us | chosen today us (kinds) | faster |")
print("|---|---|---|---|---|---|---|")
for dim in DIMS:
    for T, B in SHAPES:
        g, w, c = res[(dim, T, B, "group")], res[(dim, T, B, "wavebatch")], res[(dim, T, B, "chosen")]
        print("| %d | %d x %d | %d | %.2f | %.2f | %.2f (%s) | %s %.2fx |" % (dim, T, B, T * B // 64, g[0], w[0], c[0], c[1],
                                                                                "wave-batch" if w[0] < g[0] else "lane-group", max(g[0], w[0]) / min(g[0], w[0])))
