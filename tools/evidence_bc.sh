#!/bin/bash
# Round evidence: kernel traces + PMC passes of the big single-GPU workloads (246 GB of tables each).   usage: bash tools/evidence_bc.sh r05 b|c
round=${1:-r05}; part=${2:-b}
bash profiles/collect_round.sh $round $part 2>&1 | grep -E "^==|failed|FAILED|mean=" | cut -c1-200
