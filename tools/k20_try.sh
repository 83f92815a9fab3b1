cd $GRAFT_REPO_ROOT
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out/k20
timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/k20/w1.json 2> gpurun_out/k20/w1.err || { tail -5 gpurun_out/k20/w1.err; exit 1; }
timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 5 --shard-mode rows --replicate-mb 64 > gpurun_out/k20/w1rows.json 2> gpurun_out/k20/w1rows.err || { tail -5 gpurun_out/k20/w1rows.err; exit 1; }
unset PIMEMB_FORCE_DIST
PIMEMB_RCCL_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/k20/w2.json 2> gpurun_out/k20/w2.err || { tail -5 gpurun_out/k20/w2.err; exit 1; }
PIMEMB_RCCL_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu-baseline --steps 20 --warmup 5 --exchange peer > gpurun_out/k20/w2peer.json 2> gpurun_out/k20/w2peer.err || { tail -5 gpurun_out/k20/w2peer.err; exit 1; }
python3 - <<PY
import json
for k in ("w1","w1rows","w2","w2peer"):
    d=json.load(open("gpurun_out/k20/%s.json"%k))
    print(k, "ms/step", round(d["ms_per_step"],5), "exchange", d.get("exchange_mode"), d.get("ms_per_step_exchange"), "prewarm_steps", (d["config"].get("exchange") or {}).get("prewarm_steps", d["config"].get("prewarm_steps")))
PY
