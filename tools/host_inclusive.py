#!/usr/bin/env python3
"""Times the host-buffer entry points (upload + work lists + kernels + download per call) on the
benchmark workload: the PCIe-inclusive figure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import bench  # noqa: E402
import psa_amd  # noqa: E402

seq1, seq2s = bench.synth(2, 3000, [2000] * 4)
ctx = psa_amd.Context(0)
ctx.set_seq1(seq1)
ctx.set_weights(bench.WEIGHTS)
cand = sum((len(seq1) - len(s) + 3) * (len(s) * (len(s) - 1) // 2) for s in seq2s)
for stages, name in ((psa_amd.STAGE_BOTH, "psa_score_batch, both stages"), (psa_amd.STAGE_RESOLVE, "psa_score_batch, resolve only")):
    ctx.score_batch(seq2s, stages)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ctx.score_batch(seq2s, stages)
    dt = (time.perf_counter() - t0) / n
    print("%-34s %.3f ms per call  %.3e candidates/s" % (name, dt * 1e3, cand / dt))
table = psa_amd.createCompMatrix(bench.WEIGHTS)
os.environ["PSA_SHIM_QUIET"] = "1"
psa_amd.calc_best_score(seq1, seq2s[0], table)
t0 = time.perf_counter()
for s in seq2s * 5:
    psa_amd.calc_best_score(seq1, s, table)
dt = (time.perf_counter() - t0) / 20
print("%-34s %.3f ms per Seq2" % ("psa_calc_best_score (drop-in shim)", dt * 1e3))
