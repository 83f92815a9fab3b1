#!/bin/bash
# Round-5 GPU job 4: the publish kernel with self-describing entries (no fence / ticket / flag on the counts' path).
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05_job4; mkdir -p "$out"
cd "$root"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
run() { key=$1; shift; python3 "$root/bench.py" --gpus 1 --no-cpu-baseline "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -5 "$out/$key.err"; exit 1; }; python3 -c "
import json,sys
d=json.load(open('$out/$key.json')); c=d['config']; r=d['roofline']
print('$key', 'us/step %.1f' % (d['ms_per_step']*1e3), 'direct', c.get('direct_one_hot_path'), 'checked', c.get('checked'), 'host_us %.1f' % r['exchange']['host_us_per_step'], 'wait_served %.1f' % r['exchange'].get('host_wait_served_us_per_step',0))"; }
C4="--workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40"
run c4_direct $C4
run c4_direct_checked $C4 --checked
run c4_direct_b $C4
run c4_direct_checked_b $C4 --checked
PIMEMB_SHARD_DEPTH=0 run c4_direct_checked_depth0 $C4 --checked
run c2_direct --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
run c2_direct_checked --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40 --checked
run c4_peer_checked $C4 --checked --exchange peer
unset PIMEMB_FORCE_DIST
bash tools/r05_checked_trace.sh 2>&1 | grep -v "^\"Name\|copyBuffer\|elementwise" | cut -c1-70
python3 - <<'PY'
import csv
for k in ('c4','c2'):
    for r in csv.reader(open(f'gpurun_out/profiles_r05/dist-{k}-rows-l1-direct-checked_kernel_stats.csv')):
        if 'served_counts' in r[0]: print(k, 'served_counts_kernel calls', r[1], 'avg ns', r[3], 'min', r[5], 'max', r[6])
PY
python3 -m pytest tests/test_gpu_shard.py -x -q > "$out/pytest_shard.log" 2>&1 || { tail -40 "$out/pytest_shard.log"; exit 1; }
tail -3 "$out/pytest_shard.log"
