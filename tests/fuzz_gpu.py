#!/usr/bin/env python3
"""Time-bounded random comparison of libpsa_hip with the CPU oracle (not collected by pytest; run
it on a GPU box: python3 tests/fuzz_gpu.py [seconds, default 60] [seed]).  Wider than
test_fuzz_small: Seq1 up to 9000, Seq2 up to 8191, batches of 1 to 12 Seq2 of mixed lengths,
alphabets of 1 to 26 letters, dyadic and non-dyadic weights, arbitrary float tables."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "parallel-sequence-alignment_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle_py as oracle  # noqa: E402
import psa_amd  # noqa: E402
from test_gpu_parity import check_against_oracle  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
oracle.build()
ctx = psa_amd.Context(0)
WEIGHTS = ([2, 1.5, 1.1, 1.3], [1, 1, 1, 1], [3, 0.7, 0.2, 5], [1.1, 1.1, 1.3, 1.3], [2, 1.5, 1, 1.25], [0, 0, 0, 1])
t0, cases, seqs = time.time(), 0, 0
alone_ok = alone_over = 0
while time.time() - t0 < secs:
    kind = rng.random()
    if kind < 0.08:
        L1, maxl2 = int(rng.integers(6000, 9001)), 8191
    elif kind < 0.5:
        L1, maxl2 = int(rng.integers(2, 600)), 8191
    else:
        L1, maxl2 = int(rng.integers(300, 4001)), 8191
    nalpha = int(rng.integers(1, 27))
    alpha = "".join(rng.permutation(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ"))[:nalpha])
    pick = lambda n: "".join(alpha[i] for i in rng.integers(0, len(alpha), size=n))
    seq1 = pick(L1)
    lens = []
    for _ in range(int(rng.integers(1, 13))):
        hi = min(L1 + 2, maxl2)
        if hi < 2:
            continue
        r = rng.random()
        if r < 0.3:  # near a 16 / 64 / 256 boundary
            base = int(rng.choice([16, 64, 256])) * int(rng.integers(1, max(2, hi // 16)))
            L2 = int(np.clip(base + int(rng.integers(-2, 3)), 2, hi))
        elif r < 0.4:
            L2 = int(rng.integers(max(2, hi - 3), hi + 1))
        else:
            L2 = int(rng.integers(2, hi + 1))
        lens.append(L2)
    if not lens:
        continue
    if L1 > 5000:
        lens = lens[:3]
    seq2s = []
    for L2 in lens:
        if rng.random() < 0.5 and L2 <= L1:
            st = int(rng.integers(0, L1 - L2 + 1))
            s2 = list(seq1[st:st + L2])
            for _ in range(max(1, L2 // 10)):
                s2[int(rng.integers(0, L2))] = alpha[int(rng.integers(0, len(alpha)))]
            seq2s.append("".join(s2))
        else:
            seq2s.append(pick(L2))
    ctx.set_seq1(seq1)
    if rng.random() < 0.15:
        table = rng.normal(0, 3, size=676).astype(np.float32)
        ctx.set_table(table)
    else:
        w = np.array(WEIGHTS[int(rng.integers(0, len(WEIGHTS)))], dtype=np.float32)
        table = oracle.comp_matrix(w)
        ctx.set_weights(w)
    try:
        check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))
        if rng.random() < 0.25:
            # the enumeration stage alone: where its shortlist is complete, the replayed winner is the
            # reference's answer; where it overflowed, the float32 half stays empty
            b = ctx.batch(seq2s)
            b.run(psa_amd.STAGE_ENUMERATE)
            for s2, r, (cnt, state) in zip(seq2s, b.results(), b.shortlist()):
                if r.enum_frac_bits < 0:
                    continue
                if state == 1:
                    o = oracle.fast(seq1, s2, table)
                    assert r.key() == o.as_tuple(), ("enumeration alone", s2[:40], r.key(), o.as_tuple())
                    assert tuple(r.counts) == oracle.counts(seq1, s2, r.offset, r.n, r.k)
                    alone_ok += 1
                else:
                    assert state == 2 and r.n == 0 and r.score_bits == 0, (state, r.key())
                    alone_over += 1
            b.close()
    except AssertionError:
        print("MISMATCH seed=%d case=%d L1=%d lens=%r alpha=%r" % (seed, cases, L1, lens, alpha))
        raise
    cases += 1
    seqs += len(seq2s)
ctx.close()
print("fuzz ok: %d batches, %d Seq2 in %.1f s (seed %d); enumeration stage alone: %d answers from the shortlist, %d lists overflowed"
      % (cases, seqs, time.time() - t0, seed, alone_ok, alone_over))
