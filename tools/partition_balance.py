#!/usr/bin/env python3
"""How well psa_partition balances BASELINE's list of 4096 Seq2 (500..2000 characters, Seq1 = 10000)
over 8 workers: every share is scored alone on this GPU (both stages, 2 runs, the second timed), and
the measured time per share is printed beside the predicted cost and the raw candidate count.
Run on the GPU box from the repository root: python3 tools/partition_balance.py [parts]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import bench  # noqa: E402
import psa_amd  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seq1, seq2s = bench.c3_list(4096)
lens = [len(s) for s in seq2s]
ctx = psa_amd.Context(0)
ctx.set_seq1(seq1)
ctx.set_weights(bench.WEIGHTS)
cand = np.array([(len(seq1) - L + 3) * (L * (L - 1) // 2) for L in lens], dtype=np.float64)
cost = np.array([psa_amd.cost_estimate(L, len(seq1), 0) for L in lens])
for label, owner in (("psa_partition (tile-count model)", psa_amd.partition(lens, len(seq1), parts)),
                     ("split by raw candidate counts", None)):
    if owner is None:  # longest-processing-time on the candidate counts, what the first version did
        order = np.argsort(-cand, kind="stable")
        load = np.zeros(parts)
        owner = np.zeros(len(lens), dtype=np.int32)
        for i in order:
            p = int(np.argmin(load))
            owner[i] = p
            load[p] += cand[i]
    ms = []
    for p in range(parts):
        mine = [seq2s[i] for i in range(len(lens)) if owner[i] == p]
        b = ctx.batch(mine)
        b.launch()
        t0 = time.perf_counter()
        b.launch()
        ms.append(1e3 * (time.perf_counter() - t0))
        b.close()
    ms = np.array(ms)
    pc = np.array([cost[owner == p].sum() for p in range(parts)])
    pn = np.array([cand[owner == p].sum() for p in range(parts)])
    print("%s, %d parts:" % (label, parts))
    print("  measured ms per part   ", " ".join("%7.1f" % x for x in ms), "  max/mean %.4f" % (ms.max() / ms.mean()))
    print("  predicted cost (rel.)  ", " ".join("%7.4f" % x for x in pc / pc.mean()), "  max/mean %.4f" % (pc.max() / pc.mean()))
    print("  candidates (rel.)      ", " ".join("%7.4f" % x for x in pn / pn.mean()), "  max/mean %.4f" % (pn.max() / pn.mean()))
ctx.close()
