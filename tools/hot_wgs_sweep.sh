#!/bin/bash
# c3 with the engine-picked hot rows in LDS: how many persistent workgroups the hot-row kernel gets in all (PIMEMB_HOT_WGS_TOTAL, split
# evenly over the 48 tables) x how many rows are staged.  Round 6 question: the profile of `c3 --hot-rows 32` shows MORE HBM-side reads
# than the plain kernel (3.14 against 2.77 GB: L2 hit rate 49 % against 62 %) -- with 170 workgroups per table three tables are in flight
# at once and share the L2s; more workgroups per table = fewer tables in flight, but one LDS fill per workgroup.
#   usage: bash tools/hot_wgs_sweep.sh [out-file]
out=${1:-gpurun_out/hot_wgs_sweep.log}
: > "$out"
for hot in 32 100; do
  for total in 4096 8192 16384 24576 49152; do
    line=$(PIMEMB_HOT_WGS_TOTAL=$total python3 bench.py --workload c3 --hot-rows $hot --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
    python3 -c "
import json, sys
d = json.loads(sys.argv[1])
print('hot rows %3d, workgroups in all %6d: %.4f ms / step, kernel %.1f us, verified %s' % ($hot, $total, d['ms_per_step'], d['roofline']['kernel_us'], d['verified']))" "$line" | tee -a "$out"
  done
done
line=$(python3 bench.py --workload c3 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
python3 -c "
import json, sys
d = json.loads(sys.argv[1])
print('no hot rows (the lane-group kernel): %.4f ms / step, kernel %.1f us' % (d['ms_per_step'], d['roofline']['kernel_us']))" "$line" | tee -a "$out"
