#!/bin/bash
# The sharded legs with ONE rank (self pieces served in place) on the one-GPU box: JSON lines, compared with round 3's
# lines of the same commands (time per step, and the digest of rank 0's row-split outputs of the last timed step: the library
# call must leave the bits round 3's bench-side pipeline left).  usage: bash profiles/collect_dist_world1.sh r04
round=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_${round}/dist_world1
mkdir -p "$out"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
run() { key=$1; shift; python3 "$root/bench.py" --gpus 1 --no-cpu-baseline "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -5 "$out/$key.err"; exit 1; }; }
run c2_auto_k20 --steps 20 --warmup 5
run c2_auto --steps 2000 --warmup 200
run c2_whole --shard-mode whole --replicate-mb 64 --steps 400 --warmup 40
export PIMEMB_SHARD_DIRECT=0          # the ROUTED path (what round 3 ran, and what a rank behind RCCL runs)
run c2_rows --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
run c4_rows_L1 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
run c4_rows_L32 --workload c4 --rows-scale 0.125 --pooling 32 --replicate-mb 64 --steps 200 --warmup 20
run c4_rows_L32_zipf --workload c4 --rows-scale 0.125 --pooling 32 --index-dist zipf --replicate-mb 64 --steps 200 --warmup 20
run c4_plan_L32 --workload c4 --rows-scale 0.125 --pooling 32 --shard-mode plan --replicate-mb 64 --steps 200 --warmup 20
unset PIMEMB_SHARD_DIRECT             # the DIRECT one-hot path (default when no peer sits behind RCCL)
run c2_rows_direct --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
run c4_rows_L1_direct --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
PIMEMB_SHARD_DIRECT=0 PIMEMB_SHARD_SELF_VIA_COMM=1 run c4_rows_L1_self_via_rccl --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
run c4_rows_L1_peer --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --exchange peer
# ... and with depth 0 (every stage of a batch inside its own call: what the synchronous forward() / apply_emb harness runs)
PIMEMB_SHARD_DEPTH=0 run c2_rows_direct_depth0 --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
PIMEMB_SHARD_DEPTH=0 run c4_rows_L1_direct_depth0 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
PIMEMB_SHARD_DEPTH=0 PIMEMB_SHARD_DIRECT=0 run c4_rows_L1_depth0 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
# CHECKED shards (EMB_SHARD_CHECK_SERVED, round 5): one-index batches keep the direct path and count what every launch serves
run c2_rows_direct_checked --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40 --checked
run c4_rows_L1_direct_checked --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --checked
PIMEMB_SHARD_DEPTH=0 run c4_rows_L1_direct_checked_depth0 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --checked
PIMEMB_SHARD_DIRECT=0 run c4_rows_L1_routed_checked --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --checked
run c4_rows_L1_peer_checked --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --exchange peer --checked
# round 6: (a) the comparison INSIDE the completing call, as round 5 had it (PIMEMB_BENCH_CHECK=sync), next to the deferred report the
# --checked legs above now run; (b) DLRM's int64 ids handed over in place (--ids int64): direct, routed, checked
PIMEMB_BENCH_CHECK=sync run c4_rows_L1_direct_checked_sync --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --checked
PIMEMB_BENCH_CHECK=sync PIMEMB_SHARD_DEPTH=0 run c4_rows_L1_direct_checked_sync_depth0 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --checked
run c4_rows_L1_direct_int64 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --ids int64
run c4_rows_L1_direct_int64_checked --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --ids int64 --checked
PIMEMB_SHARD_DIRECT=0 run c4_rows_L1_routed_int64 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --ids int64
PIMEMB_SHARD_DIRECT=0 run c4_rows_L32_int64 --workload c4 --rows-scale 0.125 --pooling 32 --replicate-mb 64 --steps 200 --warmup 20 --ids int64
# the same two direct legs with the loop's consumer on a SECOND stream (emb_shard_wait then records an event between two
# kernels of the caller's stream every step: the cost of that hand-over, whatever the sharding does)
PIMEMB_BENCH_CONSUMER=other run c2_rows_direct_other_stream --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
PIMEMB_BENCH_CONSUMER=other run c4_rows_L1_direct_other_stream --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40
python3 "$root/profiles/summarize_dist_world1.py" "$out" "$root/profiles/r05/dist_world1" > "$root/gpurun_out/profiles_${round}/dist_world1.md"
cat "$root/gpurun_out/profiles_${round}/dist_world1.md"
