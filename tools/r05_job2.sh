#!/bin/bash
# Round-5 GPU job 2: the checked direct path with lane-spread counters (step times), shard tests again.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05_job2; mkdir -p "$out"
cd "$root"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
run() { key=$1; shift; python3 "$root/bench.py" --gpus 1 --no-cpu-baseline "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -5 "$out/$key.err"; exit 1; }; python3 -c "
import json,sys
d=json.load(open('$out/$key.json')); c=d['config']; r=d['roofline']
print('$key', 'us/step %.1f' % (d['ms_per_step']*1e3), 'direct', c.get('direct_one_hot_path'), 'checked', c.get('checked'), 'host_us %.1f' % r['exchange']['host_us_per_step'], 'wait_served %.1f' % r['exchange'].get('host_wait_served_us_per_step',0), 'kernels', {k: round(v,1) for k,v in r['kernels'].items() if k.endswith('_us')})"; }
C4="--workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40"
run c4_direct $C4
run c4_direct_checked $C4 --checked
PIMEMB_SHARD_DEPTH=0 run c4_direct_checked_depth0 $C4 --checked
run c2_direct --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
run c2_direct_checked --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40 --checked
PIMEMB_SHARD_DEPTH=0 run c2_direct_checked_depth0 --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40 --checked
run c4_peer_checked $C4 --checked --exchange peer
run c2_whole_checked --shard-mode whole --replicate-mb 64 --steps 400 --warmup 40 --checked
unset PIMEMB_FORCE_DIST
python3 -m pytest tests/test_gpu_shard.py -x -q > "$out/pytest_shard.log" 2>&1 || { tail -40 "$out/pytest_shard.log"; exit 1; }
tail -3 "$out/pytest_shard.log"
