#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to the top kernels with shortened names (the torch
elementwise template names are kilobytes long).  usage: trim_kernel_stats.py in.csv out.csv [N]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:top]:
        name = r["Name"]
        if len(name) > 110:
            name = name[:107] + "..."
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
