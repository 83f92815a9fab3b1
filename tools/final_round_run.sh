mkdir -p gpurun_out/profiles_r04
python bench.py > gpurun_out/profiles_r04/bench_default.json 2> gpurun_out/profiles_r04/bench_default.err; echo "default bench rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_r04/bench_short_w5_k20.json 2>/dev/null; echo "k20 bench rc=$?"
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/profiles_r04/gpu_tests_final.log | tail -15
bash profiles/soak_exchange.sh gpurun_out/profiles_r04/soak_exchange.log 2>&1 | tail -30
