#!/bin/bash
# Kernel trace of the CHECKED direct path (world 1): what the publish kernel behind the counted launch costs by itself.
root=${GRAFT_REPO_ROOT:-$(pwd)}
round=${1:-r05}
out=$root/gpurun_out/profiles_$round; mkdir -p "$out"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
cd /tmp && export TMPDIR=/tmp
for key in c4 c2; do
  args="--workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64"; [ $key = c2 ] && args="--shard-mode rows --replicate-mb 64"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/chk_$key -- python3 "$root/bench.py" --gpus 1 --no-cpu-baseline --steps 400 --warmup 40 --checked $args > /tmp/chk_$key.log 2>&1 || { tail -5 /tmp/chk_$key.log; exit 1; }
  cp "$(find /tmp/chk_$key -name '*kernel_stats.csv' | head -1)" "$out/dist-$key-rows-l1-direct-checked_kernel_stats.csv"
  head -4 "$out/dist-$key-rows-l1-direct-checked_kernel_stats.csv" | cut -c1-60,150-260
done
