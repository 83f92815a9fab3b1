#!/bin/bash
# Every single-GPU bench line of DESIGN.md section 5 in one go (final-build consolidation).
# usage: run_all_benches.sh <out.md>      (run from the repository root on a GPU box)
out=${1:-gpurun_out/summary.md}
line() { python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print("| %s | %.4f | %.3e | %.0f | %.3f |" % (sys.argv[1], d["ms_per_step"], d["value"], r["achieved"], r["frac"]))' "$1"; }
{
echo "| workload (bench.py flags) | ms / step | pooled lookups / s | algorithmic GB/s | frac of 8 TB/s |"
echo "|---|---|---|---|---|"
python3 bench.py --no-cpu-baseline 2>/dev/null | line "c2 (default)"
python3 bench.py --no-cpu-baseline --index-dist zipf 2>/dev/null | line "c2 --index-dist zipf"
python3 bench.py --no-cpu-baseline --workload c1 2>/dev/null | line "c1 (mini-batch 1)"
python3 bench.py --no-cpu-baseline --batch 2048 2>/dev/null | line "c2 --batch 2048"
python3 bench.py --no-cpu-baseline --batch 16384 2>/dev/null | line "c2 --batch 16384"
python3 bench.py --no-cpu-baseline --workload c3 --steps 100 --warmup 10 2>/dev/null | line "c3 (48 x 10M x 128, L=32, Zipf)"
python3 bench.py --no-cpu-baseline --workload c3 --steps 100 --warmup 10 --hot-rows 100 2>/dev/null | line "c3 --hot-rows 100"
python3 bench.py --no-cpu-baseline --workload c3 --steps 50 --warmup 5 --index-dist uniform 2>/dev/null | line "c3 --index-dist uniform"
python3 bench.py --no-cpu-baseline --workload c4 --steps 300 --warmup 30 2>/dev/null | line "c4 one-rank share, L=1"
python3 bench.py --no-cpu-baseline --workload c4 --pooling 32 --steps 100 --warmup 10 2>/dev/null | line "c4 one-rank share, L=32"
python3 bench.py --no-cpu-baseline --workload c5 --steps 100 --warmup 10 2>/dev/null | line "c5 one-GPU share (fp16)"
} > "$out"
cat "$out"
