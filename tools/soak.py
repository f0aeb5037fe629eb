#!/usr/bin/env python3
"""Soak run of the run pipeline: several resident batches of one context, enqueued in random
order with random stages and random waits, every result compared with the batch's own first
result.  Usage: python3 tools/soak.py [seconds, default 20] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import bench, psa_amd
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
seq1, seq2s = bench.synth(5, 2600, [2000, 64, 700, 1999, 333, 1200, 3, 900, 1500, 257])
ctx = psa_amd.Context(0)
ctx.set_seq1(seq1)
ctx.set_weights(bench.WEIGHTS)
groups = [seq2s[0:4], seq2s[4:5], seq2s[5:10], seq2s[0:1], seq2s[6:8]]
batches = [ctx.batch(g) for g in groups]
full = lambda b: [r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) for r in b.results()]
want = [full(b) if b.run() is not None else None for b in batches]
last = [None] * len(batches)   # stages of the run whose records a wait will show
hist = [[] for _ in batches]
t0, runs, checks = time.time(), 0, 0
STAGES = [psa_amd.STAGE_BOTH, psa_amd.STAGE_BOTH, psa_amd.STAGE_BOTH, psa_amd.STAGE_ENUMERATE, psa_amd.STAGE_RESOLVE]
while time.time() - t0 < secs:
    for _ in range(rnd.randint(1, 12)):
        i = rnd.randrange(len(batches))
        st = rnd.choice(STAGES)
        batches[i].enqueue(st)
        last[i] = st
        hist[i].append(st)
        runs += 1
    for i in rnd.sample(range(len(batches)), rnd.randint(1, len(batches))):
        if last[i] is None:
            continue
        batches[i].wait()
        hist[i].append('w')
        got, st = full(batches[i]), last[i]
        last[i] = None
        short = batches[i].shortlist()
        for g, w, (_, sstate) in zip(got, want[i], short):
            exp = list(w)
            # a run of the enumeration stage alone reports the float32 answer from its replayed
            # shortlist when that list is complete, and leaves the float32 half empty otherwise
            if not st & psa_amd.STAGE_RESOLVE and not (st & psa_amd.STAGE_ENUMERATE and sstate == 1):
                exp[0:4] = [0, 0, 0, 0]; exp[8:12] = [0, 0, 0, 0]
            if not st & psa_amd.STAGE_ENUMERATE:
                exp[4:8] = [0, 0, 0, 0]; exp[12:16] = [0, 0, 0, 0]
            assert list(g) == exp, (i, st, g, exp, hist[i][-12:])
        checks += 1
    if rnd.random() < 0.05:  # replace a batch (destroyed perhaps with a run pending)
        i = rnd.randrange(len(batches))
        batches[i].close()
        batches[i] = ctx.batch(groups[i])
        last[i] = None
        hist[i].append('new')
for i, b in enumerate(batches):
    if last[i] is not None:
        b.wait()
    b.close()
ctx.close()
print("soak ok: %d runs enqueued, %d waits checked in %.1f s" % (runs, checks, time.time() - t0))
