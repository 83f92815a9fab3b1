#!/bin/bash
# Round-5 GPU job 1: the checked direct path (tests + step times) and the ideal-bucketing experiment (sorted indices).
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05_job1; mkdir -p "$out"
cd "$root"
python3 -m pytest tests/test_gpu_shard.py -x -q > "$out/pytest_shard.log" 2>&1 || { tail -40 "$out/pytest_shard.log"; exit 1; }
tail -3 "$out/pytest_shard.log"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
run() { key=$1; shift; python3 "$root/bench.py" --gpus 1 --no-cpu-baseline "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -5 "$out/$key.err"; exit 1; }; python3 -c "
import json,sys
d=json.load(open('$out/$key.json')); c=d['config']; r=d['roofline']
print('$key', 'us/step %.1f' % (d['ms_per_step']*1e3), 'direct', c.get('direct_one_hot_path'), 'checked', c.get('checked'), 'host_us %.1f' % r['exchange']['host_us_per_step'], 'wait_served %.1f' % r['exchange'].get('host_wait_served_us_per_step',0))"; }
C4="--workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40"
run c4_direct $C4
run c4_direct_checked $C4 --checked
PIMEMB_SHARD_DEPTH=0 run c4_direct_depth0 $C4
PIMEMB_SHARD_DEPTH=0 run c4_direct_checked_depth0 $C4 --checked
PIMEMB_SHARD_DIRECT=0 run c4_routed_checked $C4 --checked
run c2_direct --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40
run c2_direct_checked --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40 --checked
run c4_peer_checked $C4 --checked --exchange peer
unset PIMEMB_FORCE_DIST
for k in "c2-zipf --index-dist zipf" "c2-zipf-sorted --index-dist zipf --index-order sorted" "c2 " "c2-sorted --index-order sorted"; do
  set -- $k; key=$1; shift
  python3 bench.py --no-cpu-baseline "$@" > "$out/bench_$key.json" 2> "$out/bench_$key.err" || { echo "FAILED bench $key"; tail -5 "$out/bench_$key.err"; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/bench_$key.json')); r=d['roofline']
print('$key', 'us/step %.2f' % (d['ms_per_step']*1e3), 'kernel_us %.2f' % r['kernel_us'], 'uniq_rows', r.get('unique_row_bytes'), 'uniq_lines', r.get('unique_line_bytes'))"
done
bash profiles/collect.sh r05x c2-zipf --index-dist zipf > "$out/collect_c2-zipf.log" 2>&1 || { tail -20 "$out/collect_c2-zipf.log"; exit 1; }
bash profiles/collect.sh r05x c2-zipf-sorted --index-dist zipf --index-order sorted > "$out/collect_c2-zipf-sorted.log" 2>&1 || { tail -20 "$out/collect_c2-zipf-sorted.log"; exit 1; }
tail -9 "$out/collect_c2-zipf.log"; tail -9 "$out/collect_c2-zipf-sorted.log"
