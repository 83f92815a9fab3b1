mkdir -p gpurun_out/profiles_r04
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/profiles_r04/gpu_tests_final.log | tail -8
bash profiles/soak_exchange.sh gpurun_out/profiles_r04/soak_exchange.log 2>&1 | grep -E "rc 0|FAILED"
