#!/bin/bash
# A/B on ONE box: the routed world-1 legs of round 4's tree (worktree _r04, commit 366d50f) against this tree's.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ab_routed; mkdir -p "$out"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 PIMEMB_SHARD_DIRECT=0 PIMEMB_SHARD_PROFILE=1
for rep in 1 2; do
for tree in r04 r05; do
  dir=$root; [ $tree = r04 ] && dir=$root/_r04
  for leg in "c2_rows --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40" "c4_rows --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40"; do
    set -- $leg; key=$1; shift
    (cd $dir && python3 bench.py --gpus 1 --no-cpu-baseline "$@" > "$out/${tree}_${key}_$rep.json" 2> "$out/${tree}_${key}_$rep.err")
    python3 -c "
import json
d=json.load(open('$out/${tree}_${key}_$rep.json')); x=d['roofline']['exchange']
print('$tree $key rep$rep: us/step %.1f host %.1f wait_counts %.1f busy %.1f' % (d['ms_per_step']*1e3, x['host_us_per_step'], x['host_wait_counts_us_per_step'], x['host_us_per_step']-x['host_wait_counts_us_per_step']))"
    grep "host profile" "$out/${tree}_${key}_$rep.err" | tail -1 | cut -c1-400
  done
done
done
