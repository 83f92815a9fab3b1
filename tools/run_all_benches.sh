#!/bin/bash
# Every single-GPU bench line of DESIGN.md in one go; JSON lines kept per workload key (profiles/traffic.json keys).
# usage: run_all_benches.sh <out-dir>      (run from the repository root on a GPU box)
out=${1:-gpurun_out/benches}
mkdir -p "$out"
run() { key=$1; shift; python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 > "$out/bench_$key.json"; }
run c2
run c2-zipf --index-dist zipf
run c1 --workload c1
run c2-b2048 --batch 2048
run c3 --workload c3 --steps 100 --warmup 10
run c3-hot32 --workload c3 --steps 100 --warmup 10 --hot-rows 32
run c3-uniform --workload c3 --steps 50 --warmup 5 --index-dist uniform
run c4 --workload c4 --steps 300 --warmup 30
run c4-l32 --workload c4 --pooling 32 --steps 100 --warmup 10
run c5 --workload c5 --steps 100 --warmup 10
python3 - "$out" <<'PY'
import glob, json, os, sys
print("| key | ms / step (sync clock) | event clock | pooled lookups / s | algorithmic GB/s | frac (algorithmic) | measured HBM-side GB/s | frac_measured | l2_frac | L2 hit | TA busy | binding roof | frac_binding | read / unique rows | cold us/step | verified |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    d = json.load(open(f)); r = d["roofline"]
    g = lambda k, fmt="%.3f": (fmt % r[k]) if r.get(k) is not None else "—"
    cold = d["config"].get("timed_without_prewarm_us")
    print("| %s | %.4f | %.4f | %.3e | %.0f | %.3f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        os.path.basename(f)[6:-5], d["ms_per_step"], d.get("ms_per_step_event", float("nan")), d["value"], r["achieved"], r["frac"],
        g("achieved_measured", "%.0f"), g("frac_measured"), g("l2_frac"), g("l2_hit_rate"), g("ta_busy"), r.get("binding") or "—", g("frac_binding"),
        g("read_over_unique_rows", "%.2f"),
        "—" if cold is None else "%.1f" % cold, d["verified"]))
PY
