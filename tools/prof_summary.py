#!/usr/bin/env python3
"""Prints the per-kernel summary (calls, total, average duration) of a rocprofv3 rocpd
database, as `rocprofv3 --kernel-trace --stats` records it.  usage: prof_summary.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("%-64s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-64s %8d %14.1f %12.2f %7.2f" % (name[:64], calls, total / 1e3 if total > 1e6 else total, avg / 1e3 if total > 1e6 else avg, pct))
