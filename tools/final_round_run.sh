mkdir -p gpurun_out/profiles_r04
bash profiles/collect_round.sh r04 d 2>&1 | tail -13
bash profiles/collect_dist_world1.sh r04 2>&1 | tail -15
bash profiles/collect_peer_vs_rccl.sh r04 2>&1 | tail -13
python bench.py > gpurun_out/profiles_r04/bench_default.json 2> gpurun_out/profiles_r04/bench_default.err; echo "default bench rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_r04/bench_short_w5_k20.json 2>/dev/null; echo "k20 bench rc=$?"
for m in 8 1; do echo "PIMEMB_RING_POOL=$m"; PIMEMB_RING_POOL=$m pim-embedding-lookup_amd/lib/emb_threads_bench 26 16 100000 2048 4000; PIMEMB_RING_POOL=$m pim-embedding-lookup_amd/lib/emb_threads_bench 26 16 100000 64 8000; done > gpurun_out/profiles_r04/threads_scaling.log 2>&1; tail -5 gpurun_out/profiles_r04/threads_scaling.log
{ NR_TABLES=9 NR_COLS=64 MAX_NR_BATCHES=64 MAX_INDICES_PER_BATCH=32 pim-embedding-lookup_amd/lib/emb_host_bench; pim-embedding-lookup_amd/lib/emb_host_bench 26 16 100000 512 1 100; pim-embedding-lookup_amd/lib/emb_host_bench 32 64 125000 64 120 100; } > gpurun_out/profiles_r04/emb_host_bench_presets.log 2>&1; grep "lookup():" gpurun_out/profiles_r04/emb_host_bench_presets.log
