#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files PER KERNEL FAMILY of a sharded step: for every kernel name that
contains one of the family's substrings, the mean counter value over the dispatches of its most common grid size (the
steady-state launches: pipeline fill / drain launches have other grids), summed over the family's kernels that run every
step (kernels with fewer than half as many such dispatches as the family's busiest one -- fill / drain variants -- are
left out) -- i.e. counter value per STEP.
usage: pmc_by_kernel.py <dir-with-pass-subdirs> family=substr[,substr...][,!excluded-substr] [family=...]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
families = collections.OrderedDict()
for a in sys.argv[2:]:
    name, subs = a.split("=", 1)
    families[name] = subs.split(",")
# counter -> kernel name -> grid -> [values]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row["Counter_Name"]][row["Kernel_Name"]][row.get("Grid_Size", "")].append(float(row["Counter_Value"]))
for fam, subs in families.items():
    for counter in sorted(acc):
        total, parts, picked = 0.0, [], []
        pos, neg = [x for x in subs if not x.startswith("!")], [x[1:] for x in subs if x.startswith("!")]
        for kname, grids in acc[counter].items():
            if not any(x in kname for x in pos) or any(x in kname for x in neg):
                continue
            grid, vals = max(grids.items(), key=lambda kv: len(kv[1]))
            picked.append((kname, grid, vals))
        most = max((len(v) for _, _, v in picked), default=0)
        for kname, grid, vals in picked:
            if 2 * len(vals) < most:
                continue
            vals = vals[len(vals) // 4:]            # drop warm-up dispatches
            mean = sum(vals) / len(vals)
            total += mean
            short = kname.split("<")[0].split("::")[-1].split(" ")[-1]
            parts.append("%s[grid %s n=%d]=%.1f" % (short, grid, len(vals), mean))
        if parts:
            print(f"{fam:10s} {counter:28s} per_step={total:16.1f}   " + "  ".join(parts))
