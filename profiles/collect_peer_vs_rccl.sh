#!/bin/bash
# Two PROCESSES on the one GPU: the sharded legs with the collective-free exchange (--exchange peer: HIP IPC mappings, the owner
# gathers in place and stores into the requester's HBM) against the same commands over RCCL (--exchange rccl: every rank claims
# a host of its own, RCCL talks through sockets over loopback).  Both verify all tables on every rank; the digest over rank 0's
# sharded outputs of the last timed step must agree.  NOT link numbers: both ranks' kernels share one device.
#   usage: bash profiles/collect_peer_vs_rccl.sh r04
round=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_${round}/peer_vs_rccl
mkdir -p "$out"
export PIMEMB_RCCL_ONE_GPU=1 PIMEMB_SHARD_TIMEOUT_S=60
export PIMEMB_RUN_TIMEOUT=150 PIMEMB_LAUNCH_TIMEOUT=170
run() { key=$1; shift; echo "running $key"; timeout -k 10 200 python3 "$root/bench.py" --no-cpu-baseline "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -20 "$out/$key.err"; exit 1; }; }
for ex in rccl peer; do
  run c2_whole_2r_$ex --gpus 2 --shard-mode whole --replicate-mb 64 --steps 100 --warmup 10 --exchange $ex || exit 1
  run c2_rows_2r_$ex --gpus 2 --shard-mode rows --replicate-mb 64 --steps 100 --warmup 10 --exchange $ex || exit 1
  run c2_rows_l5_zipf_2r_$ex --gpus 2 --shard-mode rows --replicate-mb 64 --pooling 5 --index-dist zipf --batch 8192 --steps 50 --warmup 10 --exchange $ex || exit 1
  run c5_2r_$ex --gpus 2 --workload c5 --rows-scale 0.000244140625 --replicate-mb 0 --batch 257 --steps 20 --warmup 4 --exchange $ex || exit 1
  run c4_l1_2r_$ex --gpus 2 --workload c4 --rows-scale 0.00390625 --replicate-mb 8 --batch 16384 --steps 50 --warmup 10 --exchange $ex || exit 1
  run c4_l32_2r_$ex --gpus 2 --workload c4 --rows-scale 0.00390625 --replicate-mb 8 --batch 2051 --pooling 32 --steps 20 --warmup 4 --exchange $ex || exit 1
done
PIMEMB_SHARD_DIRECT=0 run c2_rows_2r_peer_routed --gpus 2 --shard-mode rows --replicate-mb 64 --steps 100 --warmup 10 --exchange peer
run c2_rows_3r_peer --gpus 3 --shard-mode rows --replicate-mb 64 --steps 100 --warmup 10 --exchange peer
python3 - "$out" > "$root/gpurun_out/profiles_${round}/peer_vs_rccl.md" <<'PY'
import json, sys, os
out = sys.argv[1]
print("# The collective-free exchange (`bench.py --exchange peer`) against RCCL (`--exchange rccl`): two PROCESSES on the one GPU\n")
print("Collected by `profiles/collect_peer_vs_rccl.sh`.  Both ranks' kernels share one device and RCCL talks through sockets over")
print("loopback (`PIMEMB_RCCL_ONE_GPU=1`), so these are NOT link numbers: what they show is that the two transports leave the same bits")
print("and what a step costs when nothing but the GPU and the host are involved.  `host us` = time inside `emb_shard_submit` per step")
print("(counts / served: of it polling the peers' mailbox words).\n")
print("| leg | rccl ms / step | peer ms / step | peer host us / step (counts / served) | direct one-hot path | same bits |")
print("|---|---|---|---|---|---|")
for k in ("c2_whole_2r", "c2_rows_2r", "c2_rows_l5_zipf_2r", "c5_2r", "c4_l1_2r", "c4_l32_2r"):
    a, b = json.load(open(os.path.join(out, k + "_rccl.json"))), json.load(open(os.path.join(out, k + "_peer.json")))
    x = b["roofline"]["exchange"]
    print("| %s | %.4f | %.4f | %.1f (%.1f / %.1f) | %s | %s |" % (k, a["ms_per_step"], b["ms_per_step"], x["host_us_per_step"], x["host_wait_counts_us_per_step"],
          x["host_wait_served_us_per_step"], b["config"]["direct_one_hot_path"],
          "yes" if a["config"]["last_step_sharded_outputs_sha1"] == b["config"]["last_step_sharded_outputs_sha1"] else "NO"))
for k, what in (("c2_rows_2r_peer_routed", "c2 rows, two ranks, peer stores, ROUTED (PIMEMB_SHARD_DIRECT=0)"), ("c2_rows_3r_peer", "c2 rows, THREE ranks, peer stores")):
    b = json.load(open(os.path.join(out, k + ".json")))
    x = b["roofline"]["exchange"]
    print("| %s | | %.4f | %.1f (%.1f / %.1f) | %s | verified |" % (what, b["ms_per_step"], x["host_us_per_step"], x["host_wait_counts_us_per_step"],
          x["host_wait_served_us_per_step"], b["config"]["direct_one_hot_path"]))
print("\n`exchange_transport` of the peer runs:", b["config"]["exchange_transport"])
PY
cat "$root/gpurun_out/profiles_${round}/peer_vs_rccl.md"
