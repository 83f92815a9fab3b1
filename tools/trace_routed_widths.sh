#!/bin/bash
root=$(pwd)
out=$root/gpurun_out/r06j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 PIMEMB_SHARD_DIRECT=0
for ids in uint32 int64; do
  rm -rf /tmp/tr_$ids
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$ids -- python3 $root/bench.py --gpus 1 --no-cpu-baseline --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --ids $ids > $out/routed_$ids.json 2> $out/routed_$ids.err
  f=$(find /tmp/tr_$ids -name "*kernel_stats.csv" | sort | sed -n 1p)
  cp "$f" $out/routed_${ids}_kernel_stats.csv
  echo "== $ids"; cut -d, -f1-4 "$f" | sed -n 1,8p | cut -c1-200
done
