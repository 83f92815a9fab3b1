#!/bin/bash
# How many wavefronts of the one-hot wave-batch kernel should a CU hold at once?  The C4 one-of-8 share (8 032 one-wave workgroups:
# the whole grid fits the chip's 8 192 wave slots in ONE generation) with occupancy capped through unused dynamic LDS
# (PIMEMB_WAVEBATCH_LDS_PAD bytes per workgroup; 160 KB per CU, four SIMDs).  usage: bash tools/onehot_occupancy_sweep.sh <out file>
root=$(cd "$(dirname "$0")/.." && pwd)
out=$1; : > "$out"
for rep in 1 2; do
  for pad in 0 5120 6400 8192 10240 13312 20480 40960; do
    PIMEMB_PROBE_TAG="LDS pad $pad B ($([ $pad = 0 ] && echo 8 || echo $((163840 / pad / 4)).$(( (163840 / pad % 4) * 25 )) ) waves per SIMD)" PIMEMB_WAVEBATCH_LDS_PAD=$pad \
      timeout -k 10 300 python3 "$root/tools/onehot_inflight_probe.py" 2>/dev/null | tee -a "$out"
  done
done
