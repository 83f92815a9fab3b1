mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_parity.py -x -q -k "request_queue or checked_engine" > gpurun_out/r4g/tests.log 2>&1; tail -5 gpurun_out/r4g/tests.log
for R in 1 2 4 8 16 32; do
  python bench.py --workload c1 --coalesce $R --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r4g/c1_R$R.json 2> gpurun_out/r4g/c1_R$R.err || { echo FAIL c1 $R; tail -5 gpurun_out/r4g/c1_R$R.err; }
  python bench.py --workload c2 --batch 32 --coalesce $R --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r4g/c2b32_R$R.json 2> gpurun_out/r4g/c2b32_R$R.err || echo FAIL b32 $R
  python bench.py --workload c2 --batch 512 --coalesce $R --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r4g/c2b512_R$R.json 2> gpurun_out/r4g/c2b512_R$R.err || echo FAIL b512 $R
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4g/*_R*.json"), key=lambda x:(x.split('_R')[0], int(x.split('_R')[1][:-5]))):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable"); continue
    c=d["coalesce"]; print("%-34s R=%2d  queued %.2f us/req  one-by-one %.2f us/req  speedup %.2fx  value %.3e  dev us/flush %.1f"%(f.split('/')[-1], c["requests_per_flush"], c["us_per_request_queued"], c["us_per_request_one_by_one"], c["speedup"], d["value"], c["device_us_per_flush"]))
PY
