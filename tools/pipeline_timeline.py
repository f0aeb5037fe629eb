#!/usr/bin/env python3
"""Unprofiled timeline of a pipeline of runs: K batches of one shape (default: 4 x Seq2 = 2000 against Seq1 = 3000)
are enqueued one after the other on one context, and after the wait every batch reports where its run sat on the
GPU's time axis (psa_batch_event_times).  Usage: python3 tools/pipeline_timeline.py [both|enumerate|resolve [K]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import bench, psa_amd
stages = psa_amd.parse_stages(sys.argv[1] if len(sys.argv) > 1 else "both")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = psa_amd.Context(0)
ctx.set_weights(bench.WEIGHTS)
seq1, seq2s = bench.synth(7, 3000, [2000] * 4)
ctx.set_seq1(seq1)
batches = [ctx.batch(seq2s) for _ in range(K)]
for _ in range(30):  # every batch has run before (no set-up kernels in the timed pass), the clocks are up
    for b in batches:
        b.enqueue(stages)
for b in batches:
    b.wait()
for b in batches:
    b.enqueue(stages)
for b in batches:
    b.wait()
names = ["pre-pass starts", "loop starts", "loop ends", "enum records", "resolve starts", "resolve records"]
print("microseconds after the first batch's first kernel; one line per run, in the order they were enqueued")
print("      " + "".join("%17s" % n for n in names))
prev_end = None
for i, b in enumerate(batches):
    t = b.event_times(batches[0])
    note = ""
    if t[1] >= 0 and prev_end is not None:
        note = "   loop starts %+.1f us after the loop before it ended" % (1e3 * (t[1] - prev_end))
    if t[2] >= 0:
        prev_end = t[2]
    print("%4d  " % i + "".join("%17.1f" % (1e3 * x) if x >= 0 else "%17s" % "-" for x in t) + note)
last = batches[-1].event_times(batches[0])
print("period: %.1f us per run" % (1e3 * (max(last) - max(batches[0].event_times(batches[0]))) / (K - 1)))
