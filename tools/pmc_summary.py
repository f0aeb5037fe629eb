#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc passes (csv output directories) into one JSON: per kernel, the mean
of every counter over its dispatches.  usage: pmc_summary.py [--command TEXT] [--workload TEXT] dir [dir ...] > summary.json"""
import collections
import csv
import glob
import json
import sys

args = sys.argv[1:]
command = workload = None
while args and args[0].startswith("--"):
    if args[0] == "--command":
        command = args[1]
    elif args[0] == "--workload":
        workload = args[1]
    args = args[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {
    "command": command or "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py ... (one run per counter list)",
    "units": "mean per dispatch; FETCH_SIZE and WRITE_SIZE in KB as rocprofv3 reports them (on gfx950 FETCH_SIZE "
             "counts half the bytes of wide reads); GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_ACTIVE_INST_*, "
             "SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_BUSY_CYCLES in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles",
    "workload": workload or "",
    "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())
                if k.startswith("psa::") or "psa::" in k},
}
json.dump(out, sys.stdout, indent=1)
print()
