#!/usr/bin/env python3
"""Table of the world-1 sharded legs (collect_dist_world1.sh) next to the previous round's lines of the same commands.
usage: summarize_dist_world1.py <dir with this round's *.json> [<dir with last round's *.json>]"""
import glob
import json
import os
import sys

cur, old = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
print("# N > 1 legs with ONE rank on the one-GPU box (`PIMEMB_FORCE_DIST=1 python bench.py --gpus 1 ...`, self pieces served in place)\n")
print("Collected by `profiles/collect_dist_world1.sh`.  `ms / step` is the sync clock.  `host us` = time inside `emb_shard_submit` per step")
print("(of it waiting for the counts: when the GPU is the bottleneck the host waits there).  Kernel times from HIP events inside the")
print("library over 16 extra steps (`emb_shard_set_kernel_timing`).  `same bits` = the digest of rank 0's row-split outputs of the last")
print("timed step equals the previous round's for the same command (the previous round ran the step as a pipeline inside the bench).")
print("The loop consumes a finished batch on the CALLER's stream (stream order is the hand-over: what DLRM's interaction layer does")
print("with apply_emb's outputs, and what the previous round's loop did); `*_other_stream` legs hand it to a second stream instead --")
print("`emb_shard_wait` then records an event between two kernels of the caller's stream every step, 4-5 us of GPU time on this")
print("runtime whatever the sharding does.\n")
print("| run | placement | ms / step | previous round ms / step | same bits | host us / step (waiting for counts) | kernels us: router / lookups (direct part) / un-router | step_frac |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(cur, "*.json"))):
    key = os.path.basename(f)[:-5]
    try:
        d = json.load(open(f))
    except ValueError:
        print(f"| {key} | unreadable | | | | | | |")
        continue
    prev = same = ""
    base = key        # the previous round's line of the SAME command if it has one, else of the command it is a variant of
    if not (old and os.path.exists(os.path.join(old, base + ".json"))):
        base = key.replace("_sync", "").replace("_int64", "").replace("_other_stream", "").replace("_depth0", "").replace("_checked", "").replace("_routed", "").replace("_direct", "").replace("_self_via_rccl", "").replace("_peer", "")
    if old and os.path.exists(os.path.join(old, base + ".json")):
        o = json.load(open(os.path.join(old, base + ".json")))
        prev = "%.4f" % o["ms_per_step"]
        a, b = o["config"].get("last_step_outputs_sha1"), d["config"].get("last_step_outputs_sha1")
        same = "yes" if a and a == b else ("n/a" if not a or not b else "NO")
    r = d["roofline"]
    k, x = r.get("kernels"), r.get("exchange", {})
    if k is None:       # replica leg: the exchange leg sits beside it
        print("| %s | replica (+ exchange leg %s: %.4f ms) | %.4f | %s | %s | | %.1f | %.3f |" % (
            key, d.get("exchange_mode"), d.get("ms_per_step_exchange", 0), d["ms_per_step"], prev, same, r["kernel_us"], x.get("step_frac", 0)))
        continue
    pl = d["config"]["placement"]
    print("| %s | %d repl / %d whole / %d split%s | %.4f | %s | %s | %.1f (%.1f) | %.1f / %.1f (%.1f) / %.1f | %.3f |" % (
        key, pl["replicated"], pl["whole"], pl["row_split"], (", direct one-hot path" if d["config"].get("direct_one_hot_path") else "") + (", consumer on a second stream" if d["config"].get("consumer_stream") == "other" else "") + (", depth 0" if d["config"].get("pipeline_depth") == 0 else "") + (", CHECKED" if d["config"].get("checked") else "") + (", int64 ids" if d["config"].get("index_type") == "int64" else "") + (" (headline of the auto run; replica leg %.4f ms)" % d["ms_per_step_replica"] if "ms_per_step_replica" in d else ""),
        d["ms_per_step"], prev, same, x.get("host_us_per_step", 0), x.get("host_wait_counts_us_per_step", 0),
        k["router_us"], k["lookup_us"], k.get("direct_lookup_us", 0), k["unrouter_us"], x.get("step_frac", 0)))
