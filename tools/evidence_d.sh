#!/bin/bash
# Round evidence: every single-GPU bench line, the world-1 sharded legs, two processes on the one GPU (peer vs RCCL), the default
# and the driver-shaped run, the C programs over the ABI.   usage: bash tools/evidence_d.sh r05
round=${1:-r05}
mkdir -p gpurun_out/profiles_$round
bash profiles/collect_round.sh $round d 2>&1 | tail -13
bash profiles/collect_dist_world1.sh $round 2>&1 | tail -26
bash profiles/collect_peer_vs_rccl.sh $round 2>&1 | tail -13
bash tools/r05_checked_trace.sh $round 2>&1 | tail -3
python3 bench.py > gpurun_out/profiles_$round/bench_default.json 2> gpurun_out/profiles_$round/bench_default.err; echo "default bench rc=$?"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/profiles_$round/bench_short_w5_k20.json 2>/dev/null; echo "k20 bench rc=$?"
for m in 8 1; do echo "PIMEMB_RING_POOL=$m"; PIMEMB_RING_POOL=$m pim-embedding-lookup_amd/lib/emb_threads_bench 26 16 100000 2048 4000; PIMEMB_RING_POOL=$m pim-embedding-lookup_amd/lib/emb_threads_bench 26 16 100000 64 8000; done > gpurun_out/profiles_$round/threads_scaling.log 2>&1; tail -5 gpurun_out/profiles_$round/threads_scaling.log
{ NR_TABLES=9 NR_COLS=64 MAX_NR_BATCHES=64 MAX_INDICES_PER_BATCH=32 pim-embedding-lookup_amd/lib/emb_host_bench; pim-embedding-lookup_amd/lib/emb_host_bench 26 16 100000 512 1 100; pim-embedding-lookup_amd/lib/emb_host_bench 32 64 125000 64 120 100; } > gpurun_out/profiles_$round/emb_host_bench_presets.log 2>&1; grep "lookup():" gpurun_out/profiles_$round/emb_host_bench_presets.log
# the HBM pre-flight on a layout that does NOT fit: all 512 C5 tables on ONE rank (1.97 TB as configured) -- the rows are shrunk
# to what the budget allows, the line says by how much, and the measured use sits next to the estimate
PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 python3 bench.py --gpus 1 --workload c5 --steps 10 --warmup 3 --nbatch 6 --no-cpu-baseline > gpurun_out/profiles_$round/c5_world1_fit_to_hbm.json 2> gpurun_out/profiles_$round/c5_world1_fit_to_hbm.err; echo "c5 world-1 fit rc=$?"
python3 - gpurun_out/profiles_$round <<'PY'
import glob, json, os, sys
print("| run | rows scale to fit | HBM budget GB (tables / batches / staging / arena / runtime / total) | measured in use GB | HBM GB |")
print("|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "dist_world1", "*.json"))) + [os.path.join(sys.argv[1], "c5_world1_fit_to_hbm.json")]:
    try:
        c = json.load(open(f))["config"]
    except Exception:
        continue
    b = c.get("hbm_budget_GB")
    if b:
        print("| %s | %.3f | %.1f / %.1f / %.1f / %.1f / %.1f / %.1f | %s | %s |" % (os.path.basename(f)[:-5], c.get("rows_scale_to_fit", 1.0), b["tables"], b["batches"], b["staging"], b["arena"], b.get("runtime", 0), b["total"], c.get("hbm_in_use_GB_measured"), c.get("hbm_total_GB")))
PY
