#!/usr/bin/env python3
"""Dump the kernel-stats view of a rocprofv3 (--kernel-trace --stats) rocpd database as text.
usage: summarize_rocpd.py results.db > summary.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print(f"# source: {sys.argv[1]}")
print(f"{'kernel':90s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:90]:90s} {calls:7d} {total / 1e0:12.1f} {avg:10.2f} {pct:7.2f}")
