#!/bin/bash
# Price the kernels of the SHARDED step (world 1 on the one-GPU box: every piece is served in place, so a step is exactly
# router + ONE fused lookup + un-router): a `--kernel-trace --stats` pass plus the four PMC passes MI355X_MICROARCH.md
# prescribes for HBM-side bytes, summarised per kernel family.  PIMEMB_FORCE_DIST is exported in the shell and the program
# sits directly behind `--` (no env / bash hop under rocprofv3).
#   usage (repository root, GPU box):  bash profiles/collect_dist_pmc.sh r04 dist-c4-rows-l1 --workload c4 --rows-scale 0.125 --replicate-mb 64
round=$1; key=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
scratch=$root/gpurun_out/prof_${round}/$key
out=$root/gpurun_out/profiles_${round}
mkdir -p "$scratch" "$out"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
# default: the ROUTED step (router + fused lookup + un-router); DIST_PMC_DIRECT=1: whatever the library picks -- the direct
# one-hot path for one index per bag (ONE ranged launch per step), or a `whole` placement (ONE fused launch per step)
if [ -z "$DIST_PMC_DIRECT" ]; then export PIMEMB_SHARD_DIRECT=0; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/stats" -- python3 "$root/bench.py" --gpus 1 --no-cpu-baseline \
    --steps 200 --warmup 20 "$@" > "$scratch/bench_under_stats.log" 2>&1 || { echo "stats pass failed for $key"; tail -5 "$scratch/bench_under_stats.log"; exit 1; }
cp "$(find "$scratch/stats" -name '*kernel_stats.csv' | head -1)" "$out/${key}_kernel_stats.csv"
grep '^{' "$scratch/bench_under_stats.log" | tail -1 > "$out/${key}_bench_under_stats.json"
pass=0
for counters in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
    pass=$((pass + 1))
    timeout -k 10 600 rocprofv3 --pmc $counters --output-format csv -d "$scratch/pmc$pass" -- python3 "$root/bench.py" --gpus 1 --no-cpu-baseline \
        --steps 48 --warmup 8 "$@" > "$scratch/pmc$pass.log" 2>&1 || { echo "pmc pass $pass failed for $key"; tail -5 "$scratch/pmc$pass.log"; exit 1; }
    grep '^{' "$scratch/pmc$pass.log" | tail -1 > "$scratch/pmc$pass.json"
done
cp "$scratch/pmc1.json" "$out/${key}_bench_under_pmc.json"
python3 "$root/profiles/pmc_by_kernel.py" "$scratch" lookup=bag_sum "router=route_bags_,route_onehot_,!unroute" unrouter=unroute_bags > "$out/${key}_pmc_summary.txt"
# what the entry is tied to: the code of the library's kernels the kernel trace of THIS step lists + pimemb_shard.cpp
(cd "$root" && python3 -c "
import csv, json, sys, bench
names = sorted({r['Name'] for r in csv.DictReader(open(sys.argv[1])) if 'pimemb::' in r['Name']})
print(json.dumps(dict(bench.library_identity(), **bench.shard_identity(names), **bench.launch_identity(), shard_kernel_names=names)))" "$out/${key}_kernel_stats.csv") > "$out/${key}_identity.json"
rm -rf "$scratch"
echo "== $key"; grep -E "bag_sum|route" "$out/${key}_kernel_stats.csv" | cut -c1-160 | head -8; true || head -8 "$out/${key}_kernel_stats.csv" | cut -c1-200; cat "$out/${key}_pmc_summary.txt"
