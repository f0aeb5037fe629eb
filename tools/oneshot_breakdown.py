#!/usr/bin/env python3
"""Where the time of ONE call on a fresh batch goes (4 x Seq2 = 2000 against Seq1 = 3000): host time of
psa_batch_create / enqueue / wait / results / destroy, and the GPU's own span from the first kernel's start to
the last one's end (psa_batch_timing).  Usage: python3 tools/oneshot_breakdown.py"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import bench, psa_amd
ctx = psa_amd.Context(0)
ctx.set_weights(bench.WEIGHTS)
seq1, seq2s = bench.synth(7, 3000, [2000] * 4)
ctx.set_seq1(seq1)
names = ["create", "enqueue", "wait", "results+timing", "destroy", "whole call", "candidate loop", "gpu span"]
for stages, label in ((psa_amd.STAGE_RESOLVE, "resolve"), (psa_amd.STAGE_ENUMERATE, "enumerate"), (psa_amd.STAGE_BOTH, "both")):
    rows = []
    for i in range(120):
        t0 = time.perf_counter(); b = ctx.batch(seq2s)
        t1 = time.perf_counter(); b.enqueue(stages)
        t2 = time.perf_counter(); b.wait()
        t3 = time.perf_counter(); b.results(); tm = b.timing()
        t4 = time.perf_counter(); b.close()
        t5 = time.perf_counter()
        if i >= 20:
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0, tm[0] / 1e3, tm[1] / 1e3))
        time.sleep(0.0005)
    print("stages = %s, medians of 100 calls, microseconds:" % label)
    print("   " + "   ".join("%s %.1f" % (n, statistics.median(x[k] for x in rows) * 1e6) for k, n in enumerate(names)))
