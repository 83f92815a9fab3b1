#!/bin/bash
# An int64 job's ROUTED step with the request pieces widened into the one int64 launch (default) against two launches
# (PIMEMB_SHARD_WIDEN=0), world 1, C4's one-of-8 share, one and 32 indices per bag; the uint32 job beside them.
# usage: bash tools/ab_widen_pieces.sh <out dir>
root=$(cd "$(dirname "$0")/.." && pwd)
out=$1; mkdir -p "$out"
export PIMEMB_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 PIMEMB_SHARD_DIRECT=0
run() { key=$1; shift; python3 "$root/bench.py" --gpus 1 --no-cpu-baseline --workload c4 --rows-scale 0.125 --replicate-mb 64 "$@" > "$out/$key.json" 2> "$out/$key.err" || { echo "FAILED $key"; tail -5 "$out/$key.err"; exit 1; }
  python3 - "$out/$key.json" "$key" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]
print("%-28s %.1f us per step  (router %.1f / lookups %.1f / un-router %.1f)  verified %s  sha1 %s" % (sys.argv[2], d["ms_per_step"] * 1e3, k["router_us"], k["lookup_us"], k["unrouter_us"], d.get("verified"), str(d["config"].get("last_step_outputs_sha1"))[:12]))
PY
}
for L in 1 32; do
  st=$([ $L = 1 ] && echo 400 || echo 200)
  run u32_L$L --pooling $L --steps $st --warmup 40
  run i64_widened_L$L --pooling $L --steps $st --warmup 40 --ids int64
  PIMEMB_SHARD_WIDEN=0 run i64_two_launches_L$L --pooling $L --steps $st --warmup 40 --ids int64
  run i64_widened_checked_L$L --pooling $L --steps $st --warmup 40 --ids int64 --checked
  PIMEMB_SHARD_WIDEN=0 run i64_two_launches_checked_L$L --pooling $L --steps $st --warmup 40 --ids int64 --checked
done
