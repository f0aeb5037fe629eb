"""Manual GPU diagnostic (not collected by pytest): runs golden cases one by one and
prints each result; with PSA_DEBUG_SYNC=1 the library names every finished kernel.

usage: python tests/gpu_debug_case.py [first_case [last_case]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import numpy as np  # noqa: E402
import psa_amd  # noqa: E402

cases = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else len(cases)
ctx = psa_amd.Context(0)
bad = 0
for ci in range(lo, min(hi, len(cases))):
    case = cases[ci]
    w = np.array([np.float32(t) for t in case["weights"]], dtype=np.float32)
    ctx.set_seq1(case["seq1"])
    ctx.set_weights(w)
    print("case", ci, case["name"], "L1", len(case["seq1"]), "L2", [len(s) for s in case["seq2"]], flush=True)
    res = ctx.score_batch(case["seq2"])
    for r, e in zip(res, case["expected"]):
        ok = r.key() == (e["n"], e["k"], e["offset"], e["bits"])
        bad += not ok
        print("   ", "ok " if ok else "BAD", r.key(), (e["n"], e["k"], e["offset"], e["bits"]), "enum", r.enum_key(),
              flush=True)
print("mismatches:", bad)
