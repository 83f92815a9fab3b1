#!/bin/bash
# The sharded step (ONE library call per batch) with verification INSIDE the timed loop (PIMEMB_VERIFY_EVERY: every n-th step all
# tables bit for bit on every rank; times mean nothing in this mode).  usage: bash profiles/soak_exchange.sh <out-file>
out=${1:-gpurun_out/soak_exchange.log}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export PIMEMB_RUN_TIMEOUT=400 PIMEMB_LAUNCH_TIMEOUT=420 PIMEMB_SHARD_TIMEOUT_S=60
run() { what=$1; shift; echo "soak leg $what"; if timeout -k 10 450 "$@" > /tmp/soak_ex.out 2> /tmp/soak_ex.err; then python3 -c "
import json,sys
d=json.loads([l for l in open('/tmp/soak_ex.out') if l.startswith('{')][-1]); print('  rc 0, verified:', d['verified'], ' steps', d['steps'], ' ms/step', round(d['ms_per_step'],4), ' direct path:', d['config'].get('direct_one_hot_path'), ' transport:', d['config'].get('exchange_transport'))" >> "$out"; else echo "  FAILED rc $?" >> "$out"; tail -5 /tmp/soak_ex.err >> "$out"; fi; }
: > "$out"
echo "one rank, C4 shape at 1/8 of the rows, one index per bag, ROUTED (PIMEMB_SHARD_DIRECT=0: two-kernel router + un-router), verify every 37th step:" >> "$out"
PIMEMB_FORCE_DIST=1 PIMEMB_SHARD_DIRECT=0 PIMEMB_VERIFY_EVERY=37 run a python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 --pooling 1 --steps 3000 --nbatch 7 --replicate-mb 64 --no-cpu-baseline
echo "one rank, the same on the DIRECT one-hot path (no router, no un-router):" >> "$out"
PIMEMB_FORCE_DIST=1 PIMEMB_VERIFY_EVERY=37 run b python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 --pooling 1 --steps 3000 --nbatch 7 --replicate-mb 64 --no-cpu-baseline
echo "one rank, pooling 8, Zipf (a hot shard), verify every 37th step:" >> "$out"
PIMEMB_FORCE_DIST=1 PIMEMB_VERIFY_EVERY=37 run c python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 --pooling 8 --index-dist zipf --steps 3000 --nbatch 7 --replicate-mb 64 --no-cpu-baseline
echo "three RCCL ranks on the one GPU (PIMEMB_RCCL_ONE_GPU=1: sockets over loopback), Kaggle tables, one index per bag, verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run d python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 600 --nbatch 7 --exchange rccl --no-cpu-baseline
echo "three RCCL ranks, pooling 5, Zipf, verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run e python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --pooling 5 --index-dist zipf --batch 2003 --steps 600 --nbatch 7 --exchange rccl --no-cpu-baseline
echo "three ranks, peer stores (--exchange peer: no RCCL in the data path), one index per bag (direct path), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run f python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 3000 --nbatch 7 --exchange peer --no-cpu-baseline
echo "three ranks, peer stores, pooling 5, Zipf (routed), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run g python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --pooling 5 --index-dist zipf --batch 2003 --steps 3000 --nbatch 7 --exchange peer --no-cpu-baseline
echo "four ranks, peer stores, whole tables (one rank serves nothing), pooling 3, verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run h python3 "$root/bench.py" --gpus 4 --shard-mode whole --replicate-mb 400 --pooling 3 --batch 2003 --steps 3000 --nbatch 7 --exchange peer --no-cpu-baseline
echo "one rank, CHECKED shard (EMB_SHARD_CHECK_SERVED) on the direct path: every launch counts what it serves, the requester compares; verify every 37th step:" >> "$out"
PIMEMB_FORCE_DIST=1 PIMEMB_VERIFY_EVERY=37 run i python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 --pooling 1 --steps 3000 --nbatch 7 --replicate-mb 64 --checked --no-cpu-baseline
echo "three ranks, peer stores, CHECKED, one index per bag (counts return through the mailboxes), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run j python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 3000 --nbatch 7 --exchange peer --checked --no-cpu-baseline
echo "two ranks, the driver's default flags (--exchange both: the RCCL leg, then the peer-store leg of the same run), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 PIMEMB_PEER_LEG_TIMEOUT=400 run k python3 "$root/bench.py" --gpus 2 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 600 --nbatch 7 --no-cpu-baseline
echo "round 6 -- int64 ids in place: three RCCL ranks, pooling 5, Zipf (the router reads int64, pieces are uint32: two launches per served batch), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run l python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --pooling 5 --index-dist zipf --batch 2003 --steps 600 --nbatch 7 --exchange rccl --ids int64 --no-cpu-baseline
echo "round 6 -- int64 ids, three ranks, peer stores, CHECKED with the deferred report, one index per bag (int64 ranged launches through the peers' arenas), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run m python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 3000 --nbatch 7 --exchange peer --checked --ids int64 --no-cpu-baseline
echo "round 6 -- int64 ids, whole tables over RCCL (index arrays travel at 8 bytes per id), four ranks, pooling 3, verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run n python3 "$root/bench.py" --gpus 4 --shard-mode whole --replicate-mb 400 --pooling 3 --batch 2003 --steps 300 --nbatch 7 --exchange rccl --ids int64 --no-cpu-baseline
echo "round 6 -- int64 ids, three RCCL ranks, ONE index per bag (routed behind RCCL; the uint32 pieces are widened into the one int64 launch), verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run o python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 600 --nbatch 7 --exchange rccl --ids int64 --no-cpu-baseline
echo "round 6 -- the same, CHECKED (the serving rank validates the one widened launch):" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run p python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 600 --nbatch 7 --exchange rccl --ids int64 --checked --no-cpu-baseline
cat "$out"
