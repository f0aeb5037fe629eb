#!/usr/bin/env python3
"""Prints the kernels of the last steps of a rocprofv3 --kernel-trace run (csv) on one time axis:
start and end in microseconds relative to the first listed kernel, and the queue they ran on.
Usage: python3 tools/timeline.py <dir with *kernel_trace.csv> [number of kernels, default 40
       [kernels to leave out at the end, default 0]]"""
import csv, glob, os, sys
path = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)  # the newest run
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[max(0, len(rows) - skip - n):len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void psa::", "")
    print("%9.1f %9.1f  %7.1f us  q%-3s %s" % (a / 1e3, b / 1e3, (b - a) / 1e3, r.get("Queue_Id", "?"), name[:60]))
