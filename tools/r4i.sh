mkdir -p gpurun_out/r4i
export PIMEMB_RCCL_ONE_GPU=1 PIMEMB_SHARD_TIMEOUT_S=60
run() { key=$1; shift; python3 bench.py "$@" > gpurun_out/r4i/$key.json 2> gpurun_out/r4i/$key.err || { echo "FAILED $key"; tail -20 gpurun_out/r4i/$key.err; exit 1; }; }
for ex in rccl peer; do
  run c2_whole_2r_$ex --gpus 2 --shard-mode whole --replicate-mb 64 --steps 100 --warmup 10 --no-cpu-baseline --exchange $ex || exit 1
  run c2_rows_2r_$ex --gpus 2 --shard-mode rows --replicate-mb 64 --steps 100 --warmup 10 --no-cpu-baseline --exchange $ex || exit 1
  run c5_2r_$ex --gpus 2 --workload c5 --rows-scale 0.000244140625 --replicate-mb 0 --batch 257 --steps 20 --warmup 4 --no-cpu-baseline --exchange $ex || exit 1
  run c4_l32_2r_$ex --gpus 2 --workload c4 --rows-scale 0.00390625 --replicate-mb 8 --batch 2051 --pooling 32 --steps 20 --warmup 4 --no-cpu-baseline --exchange $ex || exit 1
done
PIMEMB_FORCE_DIST=1 python3 bench.py --gpus 1 --workload c4 --rows-scale 0.125 --replicate-mb 64 --steps 200 --warmup 20 --no-cpu-baseline --exchange peer > gpurun_out/r4i/c4_l1_w1_peer.json 2> gpurun_out/r4i/c4_l1_w1_peer.err || { echo FAILED w1 peer; tail gpurun_out/r4i/c4_l1_w1_peer.err; }
python3 - <<'PY'
import json
for k in ("c2_whole_2r","c2_rows_2r","c5_2r","c4_l32_2r"):
    a=json.load(open("gpurun_out/r4i/%s_rccl.json"%k)); b=json.load(open("gpurun_out/r4i/%s_peer.json"%k))
    print("%-12s rccl %.4f ms/step (host %.1f us)  peer %.4f ms/step (host %.1f us, wait counts %.1f served %.1f)  same bits: %s  %s"%(k, a["ms_per_step"], a["roofline"]["exchange"]["host_us_per_step"], b["ms_per_step"], b["roofline"]["exchange"]["host_us_per_step"], b["roofline"]["exchange"]["host_wait_counts_us_per_step"], b["roofline"]["exchange"]["host_wait_served_us_per_step"], a["config"]["last_step_sharded_outputs_sha1"]==b["config"]["last_step_sharded_outputs_sha1"], b["config"]["exchange_transport"]))
d=json.load(open("gpurun_out/r4i/c4_l1_w1_peer.json")); print("c4 L1 world 1 peer: %.4f ms/step"%d["ms_per_step"], d["config"]["exchange_transport"])
PY
