export PIMEMB_FORCE_DIST=1
mkdir -p gpurun_out/r4e
run() { key=$1; shift; python3 bench.py --gpus 1 "$@" > gpurun_out/r4e/$key.json 2> gpurun_out/r4e/$key.err || { echo "FAILED $key"; tail -20 gpurun_out/r4e/$key.err; exit 1; }; }
run c2_rows --shard-mode rows --replicate-mb 64 --steps 400 --warmup 40 --no-cpu-baseline && \
run c4_rows_L1 --workload c4 --rows-scale 0.125 --pooling 1 --replicate-mb 64 --steps 400 --warmup 40 --no-cpu-baseline && \
run c4_rows_L32 --workload c4 --rows-scale 0.125 --pooling 32 --replicate-mb 64 --steps 200 --warmup 20 --no-cpu-baseline && \
run c4_rows_L32_zipf --workload c4 --rows-scale 0.125 --pooling 32 --index-dist zipf --replicate-mb 64 --steps 200 --warmup 20 --no-cpu-baseline
python3 - <<'PY'
import json
for k in ("c2_rows","c4_rows_L1","c4_rows_L32","c4_rows_L32_zipf"):
    new=json.load(open("gpurun_out/r4e/%s.json"%k)); old=json.load(open("profiles/r03/dist_world1/%s.json"%k))
    print(k, "ms/step r03 %.4f -> r04 %.4f"%(old["ms_per_step"], new["ms_per_step"]), "sha1 same:", old["config"]["last_step_outputs_sha1"]==new["config"]["last_step_outputs_sha1"], new["config"]["last_step_outputs_sha1"])
PY
