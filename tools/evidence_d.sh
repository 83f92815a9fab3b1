#!/bin/bash
# Round evidence: every single-GPU bench line, the world-1 sharded legs, two processes on the one GPU (peer vs RCCL), the default
# and the driver-shaped run, the C programs over the ABI.   usage: bash tools/evidence_d.sh r05
round=${1:-r05}
mkdir -p gpurun_out/profiles_$round
bash profiles/collect_round.sh $round d 2>&1 | tail -13
bash profiles/collect_dist_world1.sh $round 2>&1 | tail -26
bash profiles/collect_peer_vs_rccl.sh $round 2>&1 | tail -13
python3 bench.py > gpurun_out/profiles_$round/bench_default.json 2> gpurun_out/profiles_$round/bench_default.err; echo "default bench rc=$?"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/profiles_$round/bench_short_w5_k20.json 2>/dev/null; echo "k20 bench rc=$?"
for m in 8 1; do echo "PIMEMB_RING_POOL=$m"; PIMEMB_RING_POOL=$m pim-embedding-lookup_amd/lib/emb_threads_bench 26 16 100000 2048 4000; PIMEMB_RING_POOL=$m pim-embedding-lookup_amd/lib/emb_threads_bench 26 16 100000 64 8000; done > gpurun_out/profiles_$round/threads_scaling.log 2>&1; tail -5 gpurun_out/profiles_$round/threads_scaling.log
{ NR_TABLES=9 NR_COLS=64 MAX_NR_BATCHES=64 MAX_INDICES_PER_BATCH=32 pim-embedding-lookup_amd/lib/emb_host_bench; pim-embedding-lookup_amd/lib/emb_host_bench 26 16 100000 512 1 100; pim-embedding-lookup_amd/lib/emb_host_bench 32 64 125000 64 120 100; } > gpurun_out/profiles_$round/emb_host_bench_presets.log 2>&1; grep "lookup():" gpurun_out/profiles_$round/emb_host_bench_presets.log
