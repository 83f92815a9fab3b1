#!/bin/bash
# Round evidence, part 1 of the GPU-box calls: kernel traces + PMC passes of the small single-GPU workloads and the world-1
# sharded legs (profiles/collect_round.sh <round> a).   usage: bash tools/evidence_a.sh r05
round=${1:-r05}
bash profiles/collect_round.sh $round a 2>&1 | grep -E "^==|failed|FAILED|mean=" | cut -c1-200
