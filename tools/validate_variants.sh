#!/bin/bash
# Builds (here, "build") or times (on the GPU box, "run <out dir>") variants of the validation kernel: 16-byte loads in flight per
# lane, number of sub-ticket counters, bytes between two counters.  The variants live under pim-embedding-lookup_amd/lib/variants/
# (ignored by git, travel to the GPU box).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
vdir=$root/pim-embedding-lookup_amd/lib/variants
variants=${VARIANTS:-"v2_l32_s1024_x0 v2_l32_s1024_x1 v2_l32_s128_x0 v4_l32_s1024_x0"}      # _x1: the timing floor (no tickets; WRONG verdicts)
if [ "$1" = build ]; then
    for v in $variants; do
        IFS=_ read -r a b c d <<< "$v"
        mkdir -p "$vdir/$v"
        make -s -C "$root/pim-embedding-lookup_amd/csrc" -j8 OUT="$vdir/$v/libpimemb.so" OBJDIR="$vdir/$v/obj" \
            CXXFLAGS="-O3 -std=c++17 -fPIC -w -I../../include -I. -DPIMEMB_VALIDATE_VECS=${a#v} -DPIMEMB_VALIDATE_LANES=${b#l} -DPIMEMB_VALIDATE_STRIDE=${c#s} -DPIMEMB_VALIDATE_EXPERIMENT=${d#x}"
        rm -rf "$vdir/$v/obj"
    done
    exit 0
fi
out=$2; mkdir -p "$out"; out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
for v in $variants; do
    for B in 39292 2048; do
        rm -rf /tmp/vv
        PIMEMB_PROBE_LIB="$vdir/$v/libpimemb.so" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vv -- python3 "$root/tools/checked_call_probe.py" $B deferred > "$out/$v-$B.log" 2>&1
        f=$(find /tmp/vv -name "*kernel_stats.csv" | sort | sed -n 1p)
        echo "$v B=$B: $(grep 'per call' "$out/$v-$B.log" | sed 's/.*: //') | validate_kernel avg ns: $(grep validate_kernel "$f" | awk -F'",' '{print $2}' | cut -d, -f3)" | tee -a "$out/summary.txt"
    done
done
