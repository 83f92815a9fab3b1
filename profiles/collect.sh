#!/bin/bash
# Profile ONE bench.py workload on the GPU box: a `--kernel-trace --stats` pass plus the four PMC passes
# MI355X_MICROARCH.md prescribes for HBM-side bytes (one counter group per pass; FETCH_SIZE and WRITE_SIZE never
# share a pass).  Summaries land in gpurun_out/profiles_<round>/ (copied into profiles/<round>/ afterwards).
#   usage (from the repository root on the GPU box):  bash profiles/collect.sh r02 c3 --workload c3
round=$1; key=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
scratch=$root/gpurun_out/prof_${round}/$key
out=$root/gpurun_out/profiles_${round}
mkdir -p "$scratch" "$out"
cd /tmp && export TMPDIR=/tmp
steps_stats=${STEPS_STATS:-200}
echo "[collect $key] kernel-trace pass"
rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/stats" -- python3 "$root/bench.py" --no-cpu-baseline \
    --steps "$steps_stats" --warmup 20 "$@" > "$scratch/bench_under_stats.log" 2>&1 || { echo "stats pass failed for $key"; tail -5 "$scratch/bench_under_stats.log"; exit 1; }
cp "$(find "$scratch/stats" -name '*kernel_stats.csv' | head -1)" "$out/${key}_kernel_stats.csv"
grep '^{' "$scratch/bench_under_stats.log" | tail -1 > "$out/${key}_bench_under_stats.json"
pass=0
# (fifth pass, round 6: how busy the texture-address / vector-L1 path is -- the roof that binds the cache-served pooled launches)
for counters in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    pass=$((pass + 1))
    echo "[collect $key] PMC pass $pass: $counters"
    timeout -k 10 600 rocprofv3 --pmc $counters --output-format csv -d "$scratch/pmc$pass" -- python3 "$root/bench.py" --no-cpu-baseline \
        --steps 24 --warmup 8 --prewarm-ms 0 "$@" > "$scratch/pmc$pass.log" 2>&1 || { echo "pmc pass $pass failed for $key"; tail -5 "$scratch/pmc$pass.log"; exit 1; }
done
python3 "$root/profiles/pmc_summary.py" "$scratch" bag_sum > "$out/${key}_pmc_summary.txt"
# what the entry is tied to: the code of the kernel and the signature of the launch, as the bench line under the stats pass printed
# them (roofline.kernel_sha256 / launch_signature) -- the library / source hashes are recorded beside them
(cd "$root" && python3 -c "import json, sys, bench; d = json.load(open(sys.argv[1]))['roofline']; print(json.dumps(dict(bench.library_identity(), **{k: d.get(k) for k in ('kernel_symbol', 'kernel_sha256', 'launch_signature', 'device_code_sha256')})))" "$out/${key}_bench_under_stats.json") > "$out/${key}_identity.json"
rm -rf "$scratch"        # (counter CSVs of the big workloads exceed what gpurun copies back)
echo "== $key"; head -3 "$out/${key}_kernel_stats.csv" | cut -c1-260; cat "$out/${key}_pmc_summary.txt"
