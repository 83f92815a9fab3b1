#!/bin/bash
# Round evidence: the GPU suite and the exchange soak on the frozen build.   usage: bash tools/evidence_tests.sh r05
round=${1:-r05}
mkdir -p gpurun_out/profiles_$round
python3 -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/profiles_$round/gpu_tests_final.log | tail -8
bash profiles/soak_exchange.sh gpurun_out/profiles_$round/soak_exchange.log 2>&1 | grep -E "rc 0|FAILED"
