#!/usr/bin/env python3
"""profiles/<round>/dist-*_pmc_summary.txt (per kernel family of the sharded step, written by collect_dist_pmc.sh) ->
profiles/traffic.json entries.  The entry itself prices the step's ONE fused lookup launch (family `lookup`); `router` and
`unrouter` are sub-entries, per STEP (a family's kernels summed).  Corrections per MI355X_MICROARCH.md "HBM" for gfx950 as in
make_traffic.py: read bytes = TCC_EA0_RDREQ_128B x 128 + TCC_EA0_RDREQ_64B x 64, write bytes = WRITE_SIZE x 1024.
usage: make_traffic_dist.py <round> [key ...]"""
import csv
import glob
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
rnd = sys.argv[1]
keys = sys.argv[2:] or sorted(os.path.basename(f)[:-len("_pmc_summary.txt")]
                              for f in glob.glob(os.path.join(here, rnd, "dist-*_pmc_summary.txt")))
path = os.path.join(here, "traffic.json")
data = json.load(open(path))
FAMILY_KERNELS = {"lookup": ["bag_sum"], "router": ["route_bags_", "route_onehot_"], "unrouter": ["unroute_bags"]}
for key in keys:
    src = os.path.join(rnd, f"{key}_pmc_summary.txt")
    fam = {}
    for line in open(os.path.join(here, src)):
        parts = line.split()
        if len(parts) >= 3 and parts[2].startswith("per_step="):
            v = parts[2][len("per_step="):] or parts[3]
            fam.setdefault(parts[0], {})[parts[1]] = float(v)
        elif len(parts) >= 4 and parts[2] == "per_step=":
            fam.setdefault(parts[0], {})[parts[1]] = float(parts[3])
    stats = {}
    sfile = os.path.join(here, rnd, f"{key}_kernel_stats.csv")
    if os.path.exists(sfile):
        rows = list(csv.DictReader(open(sfile)))
        for f, subs in FAMILY_KERNELS.items():
            mine = [r for r in rows if any(s in r["Name"] for s in subs) and not (f == "router" and "unroute" in r["Name"])]
            if mine:
                most = max(int(r["Calls"]) for r in mine)
                every_step = [r for r in mine if 2 * int(r["Calls"]) >= most]          # (fill / drain variants left out)
                stats[f] = {"kernel_avg_ns": sum(float(r["AverageNs"]) for r in every_step),
                            "kernel": max(every_step, key=lambda r: float(r["TotalDurationNs"]))["Name"][:200],
                            "kernel_calls": most}
    ident_file = os.path.join(here, rnd, f"{key}_identity.json")
    ident = json.load(open(ident_file)) if os.path.exists(ident_file) else {}

    def entry(f):
        c = fam.get(f, {})
        if "TCC_EA0_RDREQ_128B_sum" not in c or "WRITE_SIZE" not in c:
            return None
        rd = int(c["TCC_EA0_RDREQ_128B_sum"] * 128 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64)
        wr = int(c["WRITE_SIZE"] * 1024)
        e = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr}
        if "FETCH_SIZE" in c:
            e["fetch_size_x2_bytes"] = int(c["FETCH_SIZE"] * 1024 * 2)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["tcc_hit"], e["tcc_miss"] = int(c["TCC_HIT_sum"]), int(c["TCC_MISS_sum"])
        e.update(stats.get(f, {}))
        return e

    e = entry("lookup")
    if e is None:
        print(f"{key}: no lookup counters in {src}", file=sys.stderr)
        continue
    for k in ("shard_kernels_sha256", "shard_src_sha256", "device_code_sha256", "shard_kernel_names", "shard_kernels_basis"):       # round 6: what a dist-* entry is tied to
        if ident.get(k):
            e[k] = ident[k]
    e.update({"round": rnd, "source": f"profiles/{src}", "lib_sha256": ident.get("lib_sha256"), "src_sha256": ident.get("src_sha256"),
              "what": "world-1 sharded step (every piece served in place): the entry is the step's ONE fused lookup launch; "
                      "router / unrouter = that family's kernels summed, per step"})
    for f in ("router", "unrouter"):
        sub = entry(f)
        if sub:
            e[f] = sub
    bj = os.path.join(here, rnd, f"{key}_bench_under_pmc.json")
    if os.path.exists(bj):
        try:
            uniq = json.load(open(bj))["roofline"].get("unique_row_bytes")
            if uniq:
                e["unique_row_bytes"] = uniq
                e["read_over_unique_rows"] = max(e["read_bytes"] - json.load(open(bj))["roofline"].get("index_bytes", 0), 0) / uniq
        except (ValueError, KeyError):
            pass
    data[key] = e
    print(key, json.dumps(e)[:600])
json.dump(data, open(path, "w"), indent=2)
