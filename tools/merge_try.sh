set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/merge
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ranged or onehot or golden or plan" > gpurun_out/merge/t1.log 2>&1 || { tail -30 gpurun_out/merge/t1.log; exit 1; }
tail -2 gpurun_out/merge/t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_sharding.py -m gpu -x -q > gpurun_out/merge/t2.log 2>&1 || { tail -40 gpurun_out/merge/t2.log; exit 1; }
tail -2 gpurun_out/merge/t2.log
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
for m in 1 0; do
 for k in c2 c4; do
  if [ $k = c2 ]; then a="--shard-mode rows --replicate-mb 64 --steps 400 --warmup 40"; else a="--workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40"; fi
  PIMEMB_SHARD_DIRECT_MERGE=$m timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline $a > gpurun_out/merge/${k}_merge$m.json 2> gpurun_out/merge/${k}_merge$m.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/merge/${k}_merge$m.json"))
e=d["config"]["exchange"]
print("$k merge=$m", "ms/step", d.get("ms_per_step_exchange"), "sha", e.get("last_step_outputs_sha1"), "kernels", e.get("kernels_us") or e.get("kernels"), "verified", e.get("verified"))
PY
 done
done
