#!/bin/bash
# The N > 1 legs rehearsed with ONE RCCL rank (self exchange) on the one-GPU box: JSON lines + per-call host times, and a
# kernel trace of the C4 row-range step (which kernels a step is made of).  usage: bash profiles/collect_dist_world1.sh r03
round=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_${round}/dist_world1
mkdir -p "$out"
export PIMEMB_FORCE_DIST=1 PIMEMB_DIST_PROFILE=1 MASTER_ADDR=127.0.0.1
run() { key=$1; shift; python3 "$root/bench.py" --gpus 1 "$@" > "$out/$key.log" 2>&1; grep '^{' "$out/$key.log" | tail -1 > "$out/$key.json"; grep 'host microseconds' "$out/$key.log" | tail -1 > "$out/$key.host_us.txt"; echo "$key: $(python3 -c "import json,sys; d=json.load(open('$out/$key.json')); print(d['ms_per_step'], d.get('ms_per_step_event'), d['roofline'].get('exchange'), d['config'].get('exchange'))")"; }
run c2_auto_k20 --steps 20 --warmup 5
run c2_auto --steps 2000 --warmup 200
run c2_rows --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
run c2_whole --shard-mode whole --replicate-mb 64 --steps 400 --warmup 40
run c4_rows_L1 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
run c4_rows_L32 --workload c4 --rows-scale 0.125 --pooling 32 --replicate-mb 64 --steps 200 --warmup 20
run c4_rows_L32_zipf --workload c4 --rows-scale 0.125 --pooling 32 --index-dist zipf --replicate-mb 64 --steps 200 --warmup 20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c4_rows_L1" -- python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 \
    --pooling 1 --replicate-mb 64 --steps 200 --warmup 20 > "$out/trace_c4_rows_L1.log" 2>&1
cp "$(find "$out/trace_c4_rows_L1" -name '*kernel_stats.csv' | head -1)" "$out/c4_rows_L1_kernel_stats.csv" 2>/dev/null
rm -rf "$out/trace_c4_rows_L1"
head -14 "$out/c4_rows_L1_kernel_stats.csv" | cut -c1-200
