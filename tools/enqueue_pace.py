#!/usr/bin/env python3
"""Host time of each psa_batch_enqueue call on the default bench workload (is the host ahead of
the GPU when runs are enqueued back to back?).  Usage: python3 tools/enqueue_pace.py [steps [c1 | c2]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import bench, psa_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
seq1, seq2s = bench.synth(2, 3000, [2000] * 4) if cfg == 'c2' else bench.synth(3, 1489, [138, 1202, 466, 377])
ctx = psa_amd.Context(0)
ctx.set_seq1(seq1)
ctx.set_weights(bench.WEIGHTS)
batch = ctx.batch(seq2s)
for _ in range(3):
    batch.launch(psa_amd.STAGE_BOTH)
t = [time.perf_counter()]
for _ in range(steps):
    batch.enqueue(psa_amd.STAGE_BOTH)
    t.append(time.perf_counter())
batch.wait()
tw = time.perf_counter()
print("enqueue calls, us each:", " ".join("%.0f" % (1e6 * (b - a)) for a, b in zip(t, t[1:])))
print("all enqueued after %.0f us, finished after %.0f us" % (1e6 * (t[-1] - t[0]), 1e6 * (tw - t[0])))
