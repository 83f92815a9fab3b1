#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of the
bag kernel.  usage: tools_pmc_summary.py <dir-with-pass-subdirs> [kernel-substring]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "bag_sum"
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v)//4:]   # drop warm-up dispatches
        print(f"{k:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
