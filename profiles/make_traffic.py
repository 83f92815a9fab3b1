#!/usr/bin/env python3
"""profiles/<round>/<key>_pmc_summary.txt (mean counters per dispatch of the bag kernel, written by collect.sh)
-> profiles/traffic.json entries.  Corrections per MI355X_MICROARCH.md "HBM" for gfx950: FETCH_SIZE tallies a
128-B request at 64 B, so read bytes = TCC_EA0_RDREQ_128B x 128 + TCC_EA0_RDREQ_64B x 64 (cross-checked against
2 x FETCH_SIZE x 1024 in the entry); write bytes = WRITE_SIZE x 1024.
usage: make_traffic.py <round> [key ...]      (no keys: every *_pmc_summary.txt of the round)"""
import glob
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import bench  # noqa: E402  (library_identity: the hashes an entry is tied to)
rnd = sys.argv[1]
keys = sys.argv[2:] or sorted(os.path.basename(f)[:-len("_pmc_summary.txt")]
                              for f in glob.glob(os.path.join(here, rnd, "*_pmc_summary.txt")))
path = os.path.join(here, "traffic.json")
data = json.load(open(path))
for key in keys:
    src = os.path.join(rnd, f"{key}_pmc_summary.txt")
    c = {}
    for line in open(os.path.join(here, src)):
        parts = line.replace("=", "= ").split()
        if "mean=" in parts:
            c[parts[0]] = float(parts[parts.index("mean=") + 1])
    need = ("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "WRITE_SIZE")
    if any(k not in c for k in need):
        print(f"{key}: counters missing in {src}: {sorted(c)}", file=sys.stderr)
        continue
    rd = int(c["TCC_EA0_RDREQ_128B_sum"] * 128 + c["TCC_EA0_RDREQ_64B_sum"] * 64)
    wr = int(c["WRITE_SIZE"] * 1024)
    e = {"round": rnd, "source": f"profiles/{src}", "read_bytes": rd, "write_bytes": wr,
         "traffic_bytes_per_launch": rd + wr}
    ident_file = os.path.join(here, rnd, f"{key}_identity.json")     # written on the GPU box next to the counters (collect.sh)
    ident = json.load(open(ident_file)) if os.path.exists(ident_file) else bench.library_identity()
    e["lib_sha256"], e["src_sha256"] = ident.get("lib_sha256"), ident.get("src_sha256")
    for k in ("kernel_symbol", "kernel_sha256", "launch_signature", "device_code_sha256"):     # round 6: the entry is tied to THESE
        if ident.get(k):
            e[k] = ident[k]
    if "TA_BUSY_avr" in c and c.get("GRBM_GUI_ACTIVE"):     # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = kernel cycles
        e["ta_busy_frac"] = c["TA_BUSY_avr"] / (c["GRBM_GUI_ACTIVE"] / 8.0)
    if "FETCH_SIZE" in c:
        e["fetch_size_x2_bytes"] = int(c["FETCH_SIZE"] * 1024 * 2)     # cross-check of read_bytes
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        e["tcc_hit"], e["tcc_miss"] = int(c["TCC_HIT_sum"]), int(c["TCC_MISS_sum"])
    stats = os.path.join(here, rnd, f"{key}_kernel_stats.csv")
    if os.path.exists(stats):
        import csv
        rows = [r for r in csv.DictReader(open(stats)) if "bag_sum" in r["Name"]]
        if rows:
            top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
            e["kernel_avg_ns"], e["kernel_calls"] = float(top["AverageNs"]), int(top["Calls"])
            e["kernel"] = top["Name"]
    data[key] = e
    print(key, e)
json.dump(data, open(path, "w"), indent=2)
