set -x
export PIMEMB_FORCE_DIST=1 PIMEMB_SHARD_PROFILE=1
for k in "c2_whole --replicate-mb 64 --shard-mode whole" "c2_rows --replicate-mb 64 --shard-mode rows" "c4_l1 --workload c4 --rows-scale 0.125 --replicate-mb 64" "c4_l32 --workload c4 --rows-scale 0.125 --replicate-mb 64 --pooling 32"; do
  set -- $k; key=$1; shift
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/r4c_w1_$key.json 2> gpurun_out/r4c_w1_$key.err || exit 1
  grep "host profile" gpurun_out/r4c_w1_$key.err
done
