#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes: per counter, mean over the dispatches of kernels matching a pattern."""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "pet_"
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f"{d}/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            agg[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    v = v[1:] if len(v) > 1 else v      # drop the first (cold) dispatch
    print(f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
