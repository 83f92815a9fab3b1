#!/usr/bin/env python3
"""profiles/<round>/dist_world1/*.json (+ host_us.txt, kernel stats) -> profiles/<round>/dist_world1.md.
usage: summarize_dist_world1.py r03        (after copying gpurun_out/profiles_<round>/dist_world1 into profiles/<round>/)"""
import csv
import json
import os
import re
import sys

rnd = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
d = os.path.join(here, rnd, "dist_world1") + "/"
rows = [("c2_auto_k20", "`--steps 20 --warmup 5` (the driver's command shape: replica leg + `whole` exchange leg)"),
        ("c2_auto", "`--steps 2000 --warmup 200`"),
        ("c2_whole", "`--shard-mode whole --replicate-mb 64`"),
        ("c2_rows", "`--shard-mode rows --replicate-mb 64`"),
        ("c4_rows_L1", "`--workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64`"),
        ("c4_rows_L32", "`--workload c4 --rows-scale 0.125 --pooling 32 --replicate-mb 64`"),
        ("c4_rows_L32_zipf", "`... --pooling 32 --index-dist zipf`")]
out = ["# N > 1 legs rehearsed with ONE RCCL rank (%s; `PIMEMB_FORCE_DIST=1 python bench.py --gpus 1 ...`, self exchange)" % rnd, "",
       "Collected by `profiles/collect_dist_world1.sh %s` on the one-GPU box, summarised by `profiles/summarize_dist_world1.py`." % rnd,
       "`ms / step` is the sync clock (the primary one on every line), the event clock beside it; `step_frac` = algorithmic bytes",
       "of the rank's own bags ÷ step time ÷ 8 TB/s; bytes out are 0 with one rank (the exchange is a self copy).  Host µs per",
       "call include blocking on the GPU; they show where the host waits.", "",
       "| run | flags | primary leg | ms / step (sync / event) | lookup kernels µs | `roofline.exchange.step_frac` | exchange leg ms / step | host µs / step by call |",
       "|---|---|---|---|---|---|---|---|"]
for k, flags in rows:
    j = json.load(open(d + k + ".json"))
    host = open(d + k + ".host_us.txt").read().strip()
    host = host.split("by call:")[1].strip() if "by call:" in host else "—"
    ex, cx = j["roofline"].get("exchange", {}), j["config"].get("exchange", {})
    leg = "replica (data-parallel)" if "replicated on every rank" in j["config"]["parallelism"] else cx.get("mode", "")
    sec = "%.4f (%s)" % (cx["ms_per_step"], cx["mode"]) if leg.startswith("replica") else "—"
    out.append("| %s | %s | %s | %.4f / %.4f | %.1f | %.3f | %s | %s |" % (
        k, flags, leg, j["ms_per_step"], j["ms_per_step_event"], j["roofline"]["kernel_us"], ex.get("step_frac", 0), sec, host))
out += ["", "All runs `verified: true` (every table on the rank bit for bit: the first rotation, two pipelined steps, the last timed step).", "",
        "## Kernels of the C4 row-range step, L = 1 (`dist_world1/c4_rows_L1_kernel_stats.csv`, rocprofv3 `--kernel-trace --stats`, 200 timed steps)", "",
        "| kernel | calls | average µs |", "|---|---|---|"]
gpu = 0.0
for r in csv.DictReader(open(d + "c4_rows_L1_kernel_stats.csv")):
    n = r["Name"]
    if "pimemb" in n or "rccl" in n:
        m = re.search(r"(rcclGenericKernel|bag_sum_\w+|unroute_bags_kernel|route_\w+_kernel|validate_\w+)", n)
        us = float(r["AverageNs"]) / 1e3
        gpu += us * (3 if "rccl" in n else 1)
        out.append("| `%s` | %s | %.1f |" % (m.group(1) if m else n[:50], r["Calls"], us))
out += ["", "One step = replicated-table lookup (wave-batch, 18 tables) + served lookup (the request pieces received) + router (two kernels)",
        "+ un-router + three self-collectives (`rcclGenericKernel`: counts, requests, partial rows): ≈ %.0f µs of GPU work per step." % gpu]
open(os.path.join(here, rnd, "dist_world1.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
