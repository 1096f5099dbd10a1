#!/usr/bin/env python3
"""Print the top-kernel table of a rocprofv3 (rocpd sqlite) result: python tools/prof_top.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("%-100s %6s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-100s %6d %12.1f %12.1f %6.2f" % (name[:100], calls, total, avg, pct))
