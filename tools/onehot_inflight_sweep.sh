#!/bin/bash
# Rounds in flight of the one-hot wave-batch kernel for 512-byte rows (LPR = 32): builds (here: "build") or times (GPU box:
# "run <out file>") libraries with -DPIMEMB_LPR32_ONEHOT_INFLIGHT=N on the C4 one-of-8 share (tools/onehot_inflight_probe.py).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
vdir=$root/pim-embedding-lookup_amd/lib/variants
variants=${VARIANTS:-"1 2 4 8 16"}
if [ "$1" = build ]; then
    for v in $variants; do
        mkdir -p "$vdir/inflight$v"
        make -s -C "$root/pim-embedding-lookup_amd/csrc" -j8 OUT="$vdir/inflight$v/libpimemb.so" OBJDIR="$vdir/inflight$v/obj" \
            CXXFLAGS="-O3 -std=c++17 -fPIC -w -I../../include -I. -DPIMEMB_LPR32_ONEHOT_INFLIGHT=$v"
        rm -rf "$vdir/inflight$v/obj"
    done
    exit 0
fi
out=$2; : > "$out"
for rep in 1 2; do
    for v in $variants; do
        PIMEMB_PROBE_TAG="in flight $v" PIMEMB_PROBE_LIB="$vdir/inflight$v/libpimemb.so" timeout -k 10 300 python3 "$root/tools/onehot_inflight_probe.py" 2>/dev/null | tee -a "$out"
    done
done
