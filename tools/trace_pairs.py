#!/usr/bin/env python3
"""Timeline of consecutive (wgrad kernel, finalize) dispatch pairs in a rocprofv3 kernel-trace CSV: per group of `n` pairs
(one group per timed mode of tools/wgbench.py), mean kernel durations, the gap between the two kernels, and the period."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pairs = []
for a, b in zip(rows, rows[1:]):
    if "finalize" not in a["Kernel_Name"] and "finalize" in b["Kernel_Name"]:
        pairs.append((int(a["Start_Timestamp"]), int(a["End_Timestamp"]), int(b["Start_Timestamp"]), int(b["End_Timestamp"])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pairs = pairs[skip:]
for g in range(0, len(pairs) - n + 1, n):
    p = pairs[g + 5:g + n]          # drop the warm-up launches of the group
    k1 = sum(e - s for s, e, _, _ in p) / len(p) / 1e3
    gap = sum(s2 - e for _, e, s2, _ in p) / len(p) / 1e3
    k2 = sum(e2 - s2 for _, _, s2, e2 in p) / len(p) / 1e3
    per = (p[-1][0] - p[0][0]) / (len(p) - 1) / 1e3
    print(f"group {g // n:2d}: wgrad {k1:6.1f} us  gap {gap:5.1f}  finalize {k2:6.1f}  period {per:6.1f}")
