#!/usr/bin/env python3
"""Diagnostic: from a rocprofv3 --kernel-trace CSV, the duration of each launch of a kernel and the idle gap
between consecutive launches (end of launch k -> start of launch k+1).  usage: kernel_gaps.py <dir> [substring]"""
import csv, glob, sys
import numpy as np
pat = sys.argv[2] if len(sys.argv) > 2 else "iir_pipe_kernel"
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur, gap = [], []
for i, (s, e, name) in enumerate(rows):
    if pat not in name:
        continue
    dur.append(e - s)
    if i + 1 < len(rows) and pat in rows[i + 1][2]:
        gap.append(rows[i + 1][0] - e)
dur, gap = np.array(dur) / 1e3, np.array(gap) / 1e3
q = lambda v: "min %.1f  p50 %.1f  mean %.1f  p90 %.1f  max %.1f" % (v.min(), np.percentile(v, 50), v.mean(), np.percentile(v, 90), v.max())
print("%s: %d launches, duration us: %s" % (pat, len(dur), q(dur)))
print("   gap to the next launch of the same kernel us (%d): %s" % (len(gap), q(gap)))
