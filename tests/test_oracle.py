"""Pins the CPU oracle (oracle/psa_oracle.c) to the reference.

The expected values in tests/golden/golden.json were printed by the reference's
own Sequential Solution compiled from /root/reference (see make_golden.py)."""
import hashlib

import numpy as np
import pytest

from conftest import parse_weights, synth_problem

# class table and counts as listed for the reference's samples: (file, index) -> counts
SAMPLE_COUNTS = {
    ("ref_s_input1.txt", 0): (4, 0, 0, 0), ("ref_s_input1.txt", 1): (4, 0, 0, 0),
    ("ref_p_input1.txt", 0): (2, 0, 0, 0), ("ref_p_input1.txt", 1): (4, 0, 0, 0),
    ("ref_s_input2.txt", 0): (43, 0, 0, 0), ("ref_s_input2.txt", 1): (23, 3, 2, 1),
    ("ref_s_input3.txt", 0): (37, 2, 3, 16), ("ref_s_input3.txt", 1): (21, 4, 11, 21),
    ("ref_s_input3.txt", 2): (39, 4, 3, 10), ("ref_s_input3.txt", 3): (25, 3, 5, 21),
    ("ref_s_input3.txt", 4): (22, 5, 7, 19), ("ref_s_input3.txt", 5): (19, 4, 6, 23),
    ("ref_input.txt", 0): (136, 0, 0, 0), ("ref_input.txt", 1): (1195, 1, 1, 3),
    ("ref_input.txt", 2): (287, 34, 31, 112), ("ref_input.txt", 3): (375, 0, 0, 0),
}

# recorded by the reference's authors: P/output.txt:1-4 for P/input.txt
AUTHORS_OUTPUT = [(84, 1, 86, "272.00", 364), (176, 1, 178, "2383.50", 129),
                  (207, 1, 209, "343.30", 858), (208, 1, 210, "750.00", 976)]


def test_class_table_digest(oracle):
    t = oracle.class_table()
    digits = "".join(str(int(x)) for x in t)
    assert hashlib.sha256(digits.encode()).hexdigest() == \
        "72dc2f538a51c5ae22d54a488587915ca359e2fd55ad8f5bf7f369bdf654f1cd"
    assert [int((t == c).sum()) for c in range(4)] == [26, 62, 50, 538]
    assert (t.reshape(26, 26) == t.reshape(26, 26).T).all()


def test_comp_matrix(oracle):
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    m = oracle.comp_matrix(w).reshape(26, 26)
    A = lambda c: ord(c) - 65
    assert m[A('A'), A('A')] == w[0]
    assert m[A('N'), A('D')] == -w[1]
    assert m[A('S'), A('G')] == -w[2]
    assert m[A('A'), A('B')] == -w[3]


def test_mutant_numbering(oracle):
    for L2 in (2, 3, 4, 7, 138):
        t = 0
        for n in range(1, L2):
            for k in range(n + 1, L2 + 1):
                t += 1
                assert oracle.get_nk(t, L2) == (n, k)
                assert oracle.mutant_number(n, k, L2) == t
    # P/output.txt:1  "mutant num: 84, MS(1,86)" for L2 = 138
    assert oracle.mutant_number(1, 86, 138) - 1 == 84


def test_brute_matches_reference_goldens(oracle, golden):
    n_checked = 0
    for case in golden:
        if case["name"] == "ref_input.txt":
            continue  # brute force needs ~15 min there; covered by the fast path below
        table = oracle.comp_matrix(parse_weights(case["weights"]))
        for seq2, exp in zip(case["seq2"], case["expected"]):
            if len(seq2) > 80 and len(case["seq1"]) > 1000:
                continue
            r = oracle.brute(case["seq1"], seq2, table)
            assert (r.n, r.k, r.offset, r.score_bits) == (exp["n"], exp["k"], exp["offset"], exp["bits"]), \
                (case["name"], seq2)
            n_checked += 1
    assert n_checked > 400


def test_fast_matches_reference_goldens(oracle, golden):
    n_checked = 0
    for case in golden:
        table = oracle.comp_matrix(parse_weights(case["weights"]))
        for i, (seq2, exp) in enumerate(zip(case["seq2"], case["expected"])):
            r = oracle.fast(case["seq1"], seq2, table)
            assert (r.n, r.k, r.offset, r.score_bits) == (exp["n"], exp["k"], exp["offset"], exp["bits"]), \
                (case["name"], i)
            assert r.mutant_num == oracle.mutant_number(r.n, r.k, len(seq2)) - 1
            n_checked += 1
    assert n_checked >= 440


def test_brute_on_input3_sample(oracle, golden):
    case = [c for c in golden if c["name"] == "ref_s_input3.txt"][0]
    table = oracle.comp_matrix(parse_weights(case["weights"]))
    for seq2, exp in zip(case["seq2"][:2], case["expected"][:2]):
        r = oracle.brute(case["seq1"], seq2, table, threads=4)
        assert (r.n, r.k, r.offset, r.score_bits) == (exp["n"], exp["k"], exp["offset"], exp["bits"])


def test_sample_counts(oracle, golden):
    seen = 0
    for case in golden:
        for i, (seq2, exp) in enumerate(zip(case["seq2"], case["expected"])):
            key = (case["name"], i)
            if key in SAMPLE_COUNTS:
                assert oracle.counts(case["seq1"], seq2, exp["offset"], exp["n"], exp["k"]) == SAMPLE_COUNTS[key]
                seen += 1
    assert seen >= 12


def test_authors_recorded_output(oracle, golden):
    """P/output.txt holds the CUDA program's own result lines for P/input.txt."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "inputs", "ref_input.txt")
    tok = open(path).read().split()
    table = oracle.comp_matrix(parse_weights(tok[0:4]))
    seq1, seq2s = tok[4], tok[6:6 + int(tok[5])]
    for seq2, (mnum, n, k, score, off) in zip(seq2s, AUTHORS_OUTPUT):
        r = oracle.fast(seq1, seq2, table)
        assert (r.mutant_num, r.n, r.k, "%1.2f" % r.score, r.offset) == (mnum, n, k, score, off)


def test_fast_equals_brute_random(oracle):
    rng = np.random.default_rng(7)
    alphabets = ["AB", "NDEQ", "STAG", "HY", "ACDEFGHIKLMNPQRSTVWY", "A"]
    weights = [(2, 1.5, 1.1, 1.3), (1, 1, 1, 1), (3, 0, 0, 0), (1.1, 1.1, 1.3, 1.3),
               (0.1, 0.2, 0.3, 0.7), (1000.1, 0.3, 0.7, 0.001), (16777216.0, 1.0, 0.5, 0.25)]
    for it in range(1500):
        alpha = alphabets[it % len(alphabets)]
        w = np.array(weights[it % len(weights)], dtype=np.float32)
        table = oracle.comp_matrix(w)
        L2 = int(rng.integers(2, 11))
        L1 = int(rng.integers(max(1, L2 - 2), L2 + 13))
        seq1 = "".join(alpha[i] for i in rng.integers(0, len(alpha), L1))
        seq2 = "".join(alpha[i] for i in rng.integers(0, len(alpha), L2))
        b = oracle.brute(seq1, seq2, table)
        f = oracle.fast(seq1, seq2, table)
        assert b.as_tuple() == f.as_tuple(), (seq1, seq2, w)


def test_fast_offset_subranges(oracle):
    seq1, seq2s = synth_problem(3, 300, [40, 41])
    table = oracle.comp_matrix(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
    for seq2 in seq2s:
        S = len(seq1) - len(seq2) + 3
        whole = oracle.fast(seq1, seq2, table)
        parts = [oracle.fast(seq1, seq2, table, a, b) for a, b in ((0, 100), (100, 200), (200, S))]
        best = max(p.score for p in parts)
        cands = [(p.n, p.k, p.offset) for p in parts if p.score == best]
        assert whole.score == best and (whole.n, whole.k, whole.offset) == min(cands)
        for (a, b), p in zip(((0, 100), (100, 200), (200, S)), parts):
            bb = oracle.brute(seq1, seq2, table, s_begin=a, s_end=b)
            assert bb.as_tuple() == p.as_tuple()


def test_fixed_table(oracle):
    t = oracle.comp_matrix(np.array([2, 1.5, 1, 1.25], dtype=np.float32))
    for bits in (oracle.SUM_BITS_F32, oracle.SUM_BITS_I32):
        r, it, exact = oracle.fixed_table(t, 2000, bits)
        assert exact == 2 and r >= 2 and int(np.abs(it).max()) * 2000 <= 2 ** bits - 1
    t = oracle.comp_matrix(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
    r, it, exact = oracle.fixed_table(t, 2000, oracle.SUM_BITS_I32)
    assert not exact and r == 17
    r, it, exact = oracle.fixed_table(t, 2000)
    assert not exact and r == 11 and int(np.abs(it).max()) == 4096
    r, it, exact = oracle.fixed_table(t, 8000)
    assert not exact and r == 9
    r, it, exact = oracle.fixed_table(t * np.float32(1e9), 2000)
    assert r == -1


def test_enum_fast_equals_enum_brute(oracle):
    rng = np.random.default_rng(11)
    alphabets = ["AB", "NDEQ", "ACDEFGHIKLMNPQRSTVWY", "A"]
    weights = [(2, 1.5, 1.1, 1.3), (1, 1, 1, 1), (2, 1.5, 1, 1.25), (1.1, 1.1, 1.3, 1.3)]
    for it in range(600):
        alpha = alphabets[it % len(alphabets)]
        w = np.array(weights[it % len(weights)], dtype=np.float32)
        table = oracle.comp_matrix(w)
        L2 = int(rng.integers(2, 12))
        L1 = int(rng.integers(max(1, L2 - 2), L2 + 10))
        seq1 = "".join(alpha[i] for i in rng.integers(0, len(alpha), L1))
        seq2 = "".join(alpha[i] for i in rng.integers(0, len(alpha), L2))
        r, itab, exact = oracle.fixed_table(table, L2)
        a = oracle.enum_best(seq1, seq2, itab, brute_force=True)
        b = oracle.enum_best(seq1, seq2, itab)
        assert (a.score, a.n, a.k, a.offset) == (b.score, b.n, b.k, b.offset), (seq1, seq2)
        assert oracle.enum_score(seq1, seq2, itab, a.offset, a.n, a.k) == a.score
        if exact == 2:
            # every partial sum is representable: the float path must agree exactly
            f = oracle.fast(seq1, seq2, table)
            assert (f.n, f.k, f.offset) == (a.n, a.k, a.offset)
            assert float(f.score) == a.score / 2.0 ** r
