#!/bin/bash
# After profiles/traffic.json has been rebuilt from the round's PMC passes: every single-GPU bench line again, now priced with the
# measured HBM-side bytes of THIS build (frac_measured), the summary table, the default run and the driver-shaped run.
#   usage: bash tools/evidence_d2.sh r05
round=${1:-r05}
mkdir -p gpurun_out/profiles_$round
bash tools/run_all_benches.sh gpurun_out/profiles_$round/benches > gpurun_out/profiles_$round/SUMMARY_single_gpu_benches.md 2>&1
cat gpurun_out/profiles_$round/SUMMARY_single_gpu_benches.md
python3 bench.py > gpurun_out/profiles_$round/bench_default.json 2> gpurun_out/profiles_$round/bench_default.err; echo "default bench rc=$?"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/profiles_$round/bench_short_w5_k20.json 2>/dev/null; echo "k20 bench rc=$?"
