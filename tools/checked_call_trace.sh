#!/bin/bash
# Kernel trace of a checked 26-table call (tools/checked_call_probe.py) with the verdict waited for inside the call and deferred:
# what validate_kernel / validate_publish_kernel cost next to the lookup kernel.  usage: tools/checked_call_trace.sh <out dir> [bags]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$1; B=${2:-39292}
mkdir -p "$out"
out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
for m in off sync deferred; do
    rm -rf /tmp/cct_$m
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cct_$m -- python3 "$root/tools/checked_call_probe.py" $B $m > "$out/checked_call_${B}_$m.log" 2>&1
    f=$(find /tmp/cct_$m -name "*kernel_stats.csv" | sort | sed -n 1p)
    cp "$f" "$out/checked_call_${B}_${m}_kernel_stats.csv"
    grep "per call" "$out/checked_call_${B}_$m.log"
    cut -d, -f1-4 "$out/checked_call_${B}_${m}_kernel_stats.csv" | sed -n 1,6p
done
