#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc passes (csv output directories) into one JSON: per kernel, the mean
of every counter over its dispatches.  usage: pmc_summary.py dir [dir ...] > summary.json"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --no-cpu-baseline --sync-steps --steps 3 --warmup 1 "
               "(one pass each: FETCH_SIZE; WRITE_SIZE; GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU "
               "SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES)",
    "units": "mean per dispatch; FETCH_SIZE and WRITE_SIZE in KB as rocprofv3 reports them (on gfx950 FETCH_SIZE "
             "counts half the bytes of wide reads); GRBM_GUI_ACTIVE summed over the 8 XCDs",
    "workload": "C2: Seq1=3000, 4 x Seq2=2000",
    "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())
                if k.startswith("psa::") or "psa::" in k},
}
json.dump(out, sys.stdout, indent=1)
print()
