#!/bin/bash
# Pipelined row-range exchange with verification INSIDE the timed loop (PIMEMB_VERIFY_EVERY: every n-th step all tables bit
# for bit on every rank; times mean nothing in this mode).  usage: bash profiles/soak_exchange.sh <out-file>
out=${1:-gpurun_out/soak_exchange.log}
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() { what=$1; shift; if "$@" > /tmp/soak_ex.out 2> /tmp/soak_ex.err; then python3 -c "
import json,sys
d=json.loads([l for l in open('/tmp/soak_ex.out') if l.startswith('{')][-1]); print('  rc 0, verified:', d['verified'], ' steps', d['steps'], ' ms/step', round(d['ms_per_step'],4))" >> "$out"; else echo "  FAILED rc $?" >> "$out"; tail -5 /tmp/soak_ex.err >> "$out"; fi; }
: > "$out"
echo "one RCCL rank, C4 shape at 1/8 of the rows, one index per bag (the two-kernel router), verify every 37th step:" >> "$out"
PIMEMB_FORCE_DIST=1 PIMEMB_VERIFY_EVERY=37 run a python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 --pooling 1 --steps 3000 --nbatch 5 --replicate-mb 64
echo "one RCCL rank, pooling 8, Zipf (a hot shard), verify every 37th step:" >> "$out"
PIMEMB_FORCE_DIST=1 PIMEMB_VERIFY_EVERY=37 run b python3 "$root/bench.py" --gpus 1 --workload c4 --rows-scale 0.125 --pooling 8 --index-dist zipf --steps 3000 --nbatch 5 --replicate-mb 64
echo "three gloo ranks on the one GPU, Kaggle tables, one index per bag, verify every 37th step:" >> "$out"
PIMEMB_DIST_BACKEND=gloo PIMEMB_VERIFY_EVERY=37 run c python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 1500 --nbatch 4
echo "three gloo ranks, pooling 5, Zipf, verify every 37th step:" >> "$out"
PIMEMB_DIST_BACKEND=gloo PIMEMB_VERIFY_EVERY=37 run d python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --pooling 5 --index-dist zipf --batch 2003 --steps 1500 --nbatch 4
echo "three RCCL ranks on the one GPU (PIMEMB_RCCL_ONE_GPU=1: sockets over loopback), Kaggle tables, one index per bag, verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run e python3 "$root/bench.py" --gpus 3 --shard-mode rows --replicate-mb 64 --batch 2003 --steps 600 --nbatch 4
echo "three RCCL ranks on the one GPU, the library's own grouped send / receive (--collective native), pooling 5, Zipf, verify every 37th step:" >> "$out"
PIMEMB_RCCL_ONE_GPU=1 PIMEMB_VERIFY_EVERY=37 run f python3 "$root/bench.py" --gpus 3 --shard-mode rows --collective native --replicate-mb 64 --pooling 5 --index-dist zipf --batch 2003 --steps 600 --nbatch 4
cat "$out"
