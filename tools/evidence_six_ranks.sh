#!/bin/bash
# The driver's command with SIX processes on the one GPU (the box admits six; the self-launching parent touches no GPU): the
# largest world this pool can rehearse end to end -- RCCL over loopback sockets, then the peer-store leg of the same run.
#   usage: bash tools/evidence_six_ranks.sh r05
round=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$round/six_ranks; mkdir -p "$out"
export PIMEMB_RCCL_ONE_GPU=1 PIMEMB_SHARD_TIMEOUT_S=90 PIMEMB_RUN_TIMEOUT=500 PIMEMB_LAUNCH_TIMEOUT=520 PIMEMB_PEER_LEG_TIMEOUT=300 PIMEMB_EXCHANGE_TIMEOUT=300
run() { key=$1; shift; echo "running $key"; timeout -k 10 560 python3 "$root/bench.py" --no-cpu-baseline "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -20 "$out/$key.err"; return 1; }; }
run driver_default --gpus 6 --steps 20 --warmup 5 || exit 1
run c4_l1 --gpus 6 --workload c4 --rows-scale 0.00390625 --replicate-mb 8 --batch 8192 --steps 20 --warmup 5 || exit 1
run c4_l32 --gpus 6 --workload c4 --rows-scale 0.00390625 --replicate-mb 8 --batch 2051 --pooling 32 --steps 10 --warmup 3 || exit 1
run c5 --gpus 6 --workload c5 --rows-scale 0.000244140625 --replicate-mb 0 --batch 257 --steps 10 --warmup 3 || exit 1
run c2_rows_checked --gpus 6 --shard-mode rows --replicate-mb 64 --batch 8192 --steps 20 --warmup 5 --checked || exit 1
python3 - "$out" > "$root/gpurun_out/profiles_$round/six_ranks.md" <<'PY'
import glob, json, os, sys
print("# Six processes on the one GPU: `PIMEMB_RCCL_ONE_GPU=1 python3 bench.py --gpus 6 ...` (default `--exchange both`)\n")
print("RCCL connects the ranks through sockets over loopback and all six ranks' kernels share one device: NOT link numbers. What the")
print("lines show: the driver's command shape comes through with six ranks (communicator start-up, counts-first exchange, teardown),")
print("both transports verify every table on every rank and leave the same bits.\n")
print("| run | placement | RCCL leg ms / step | peer-store leg ms / step | same bits | verified | HBM budget GB (fullest rank) | measured in use GB (all six ranks) |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.load(open(f)); c = d["config"]
    pl = c.get("placement") or {}
    xp = d.get("exchange_peer") or {}
    print("| %s | %s | %.3f | %s | %s | %s | %s | %s |" % (
        os.path.basename(f)[:-5], ("%d repl / %d whole / %d split" % (pl.get("replicated", 0), pl.get("whole", 0), pl.get("row_split", 0))) if pl else "replica leg + secondary sharded leg",
        d.get("ms_per_step_exchange", d["ms_per_step"]), ("%.3f" % xp["ms_per_step"]) if "ms_per_step" in xp else ("skipped: " + xp.get("skipped", "?")[:60]),
        d.get("exchange_same_bits"), d["verified"], (c.get("hbm_budget_GB") or {}).get("total", "—"), c.get("hbm_in_use_GB_measured", "—")))
PY
cat "$root/gpurun_out/profiles_$round/six_ranks.md"
