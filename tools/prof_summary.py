#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel stats and per-launch PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


f = find("stats", "*kernel_stats.csv")
if f:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(f)):
        print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:90], row.get("Calls"),
              row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mfma"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== %s: per-launch mean of each counter ==" % sub)
    for k, cs in acc.items():
        if not any(t in k for t in ("scan", "seq", "fir", "pipe")):
            continue
        print(k[:100])
        for c, vals in sorted(cs.items()):
            print("   %-24s n=%-4d mean=%.6g" % (c, len(vals), sum(vals) / len(vals)))
