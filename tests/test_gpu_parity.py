"""GPU parity: libpsa_hip (through its C ABI) against the CPU oracle and the golden
vectors the compiled reference produced.  Bit-exact bar: n, k, offset and the
float32 score bits; class counts; and for the enumeration stage the fixed-point
score and its first maximiser."""
import os

import numpy as np
import pytest

from conftest import parse_weights, synth_problem, synth_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psa():
    import psa_amd
    psa_amd.lib()
    return psa_amd


@pytest.fixture(scope="module")
def ctx(psa):
    c = psa.Context(0)
    yield c
    c.close()


def check_against_oracle(oracle, seq1, seq2s, table, results, expect=None, enum=True, sum_bits=None):
    for i, (seq2, r) in enumerate(zip(seq2s, results)):
        assert r.status == 0, (i, r.status)
        o = oracle.fast(seq1, seq2, table)
        assert r.key() == o.as_tuple(), ("float path", i, seq2[:40], r.key(), o.as_tuple())
        if expect is not None:
            e = expect[i]
            assert r.key() == (e["n"], e["k"], e["offset"], e["bits"]), ("golden", i)
        assert r.mutant_num == o.mutant_num
        assert tuple(r.counts) == oracle.counts(seq1, seq2, r.offset, r.n, r.k)
        L2 = len(seq2)
        assert r.candidates == (len(seq1) - L2 + 3) * (L2 * (L2 - 1) // 2)
        if enum:
            fr, itab, exact = oracle.fixed_table(table, L2, oracle.SUM_BITS_F32 if sum_bits is None else sum_bits)
            assert r.enum_frac_bits == fr and r.enum_exact == exact
            if fr >= 0:
                e = oracle.enum_best(seq1, seq2, itab)
                assert r.enum_key() == (e.n, e.k, e.offset, e.score), ("enum", i, seq2[:40], r.enum_key(),
                                                                     (e.n, e.k, e.offset, e.score))
                assert tuple(r.enum_counts) == oracle.counts(seq1, seq2, e.offset, e.n, e.k)
                if exact == 2:
                    assert (r.enum_n, r.enum_k, r.enum_offset) == (r.n, r.k, r.offset)
                    assert r.enum_score == float(r.score) * 2.0 ** fr
                # The enumerated maximum and the float32 answer are the same quantity up to rounding.
                # For one candidate of m table entries x_i, the float32 left-to-right sum differs from
                # the exact sum by at most gamma_{m-1} sum|x_i| (gamma_k = k u / (1 - k u), u = 2^-24) and
                # the fixed-point score times 2^-r by at most m 2^-(r+1) (entries rounded to nearest):
                # b bounds both together.  Then the two maxima differ by at most b, and the float32
                # winner's own fixed-point score lies within 2b of the enumerated maximum.
                m, u = L2 - 2, 2.0 ** -24
                b = (m - 1) * u / (1 - (m - 1) * u) * m * float(np.abs(table).max()) + m * 2.0 ** -(fr + 1)
                assert abs(float(r.score) - r.enum_score * 2.0 ** -fr) <= b, ("maxima apart", i, r.score, r.enum_score, fr, b)
                # 2: the float32 replay of the shortlisted candidates gave this very record; 1: the list overflowed
                assert r.enum_check in (1, 2), ("the library's own cross-check", i, r.enum_check)
                ef = oracle.enum_score(seq1, seq2, itab, r.offset, r.n, r.k)
                assert ef <= r.enum_score and (r.enum_score - ef) * 2.0 ** -fr <= 2 * b, ("winner apart", i, ef, r.enum_score)


def test_golden_vectors(psa, ctx, oracle, golden):
    total = 0
    for case in golden:
        w = parse_weights(case["weights"])
        table = oracle.comp_matrix(w)
        ctx.set_seq1(case["seq1"])
        ctx.set_weights(w)
        res = ctx.score_batch(case["seq2"])
        check_against_oracle(oracle, case["seq1"], case["seq2"], table, res, expect=case["expected"])
        total += len(res)
    assert total >= 440


def test_table_builder_matches_oracle(psa, oracle):
    for w in ([2, 1.5, 1.1, 1.3], [1, 1, 1, 1], [0.1, 0.2, 0.3, 0.7]):
        w = np.array(w, dtype=np.float32)
        assert (psa.createCompMatrix(w) == oracle.comp_matrix(w)).all()


def test_random_small_tie_heavy(psa, ctx, oracle):
    rng = np.random.default_rng(123)
    alphabets = ["AB", "NDEQ", "STAG", "HY", "ACDEFGHIKLMNPQRSTVWY", "A"]
    weights = [(2, 1.5, 1.1, 1.3), (1, 1, 1, 1), (3, 0, 0, 0), (1.1, 1.1, 1.3, 1.3), (0.1, 0.2, 0.3, 0.7),
               (1000.1, 0.3, 0.7, 0.001), (2, 1.5, 1, 1.25)]
    for it in range(60):
        alpha = alphabets[it % len(alphabets)]
        w = np.array(weights[it % len(weights)], dtype=np.float32)
        table = oracle.comp_matrix(w)
        L1 = int(rng.integers(1, 90))
        seq1 = synth_seq(rng, L1, alpha)
        seq2s = [synth_seq(rng, int(rng.integers(2, min(L1 + 2, 40) + 1)), alpha) for _ in range(8)]
        ctx.set_seq1(seq1)
        ctx.set_weights(w)
        check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))


@pytest.mark.parametrize("L1,L2s", [(700, [130, 255, 256, 257, 300]), (1500, [511, 512, 513, 700]),
                                    (2500, [1023, 1024, 1025, 1300]), (3000, [2000, 2047, 2048])])
def test_register_class_boundaries(psa, ctx, oracle, L1, L2s):
    """Seq2 lengths on both sides of every (threads x registers) kernel variant."""
    seq1, seq2s = synth_problem(L1 + 1, L1, L2s)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))


def test_config2_shape(psa, ctx, oracle):
    """Seq1 = 3000, 4 x Seq2 = 2000 (the benchmark workload)."""
    seq1, seq2s = synth_problem(2, 3000, [2000] * 4)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    b = ctx.batch(seq2s)
    res = b.run()
    check_against_oracle(oracle, seq1, seq2s, table, res)
    again = b.run()  # same resident batch, second launch: identical records
    assert [r.key() + r.enum_key() for r in again] == [r.key() + r.enum_key() for r in res]
    em, tm = b.timing()
    assert em > 0 and tm >= em
    # three runs enqueued back to back, one wait: the records of the last one
    with pytest.raises(psa.PsaError):
        b.wait()  # nothing enqueued
    for _ in range(3):
        b.enqueue()
    b.wait()
    assert [r.key() + r.enum_key() for r in b.results()] == [r.key() + r.enum_key() for r in res]
    b.enqueue(psa.STAGE_RESOLVE)
    b.wait()
    assert [r.key() for r in b.results()] == [r.key() for r in res]
    assert all(r.enum_n == 0 for r in b.results())
    b.close()


def test_exact_regime_weights(psa, ctx, oracle):
    seq1, seq2s = synth_problem(5, 1200, [300, 301, 640])
    for w in ([1, 1, 1, 1], [2, 1.5, 1, 1.25]):
        w = np.array(w, dtype=np.float32)
        table = oracle.comp_matrix(w)
        ctx.set_seq1(seq1)
        ctx.set_weights(w)
        res = ctx.score_batch(seq2s)
        assert all(r.enum_exact == 2 for r in res)
        check_against_oracle(oracle, seq1, seq2s, table, res)


def test_enum_float32_limit(psa, ctx, oracle):
    """The enumeration adds and compares in float32, which is exact only while every value is an
    integer of at most 24 bits: weights with many fraction bits and tables whose scaled entries
    times the Seq2 length come close to 2^23, against the oracle's integer enumeration."""
    rng = np.random.default_rng(77)
    seq1, seq2s = synth_problem(6, 1500, [255, 700, 1023, 1400])
    tables = [oracle.comp_matrix(np.array(w, dtype=np.float32))
              for w in ([1000.1, 0.3, 0.7, 0.001], [123.456, 77.7, 3.3, 0.9], [4095, 4094.5, 1, 3], [4, 0.25, 4, 4])]
    tables.append(rng.uniform(-1, 1, 676).astype(np.float32))
    tables.append((rng.integers(-2047, 2048, 676) * 2.0 ** -3).astype(np.float32))
    ctx.set_seq1(seq1)
    near = 0
    for table in tables:
        ctx.set_table(table)
        res = ctx.score_batch(seq2s)
        for seq2 in seq2s:
            fr, itab, _ = oracle.fixed_table(table, len(seq2))
            bound = int(np.abs(itab).max()) * len(seq2)
            assert fr >= 0 and bound <= 2 ** 23 - 1
            near += bound >= 2 ** 22
        check_against_oracle(oracle, seq1, seq2s, table, res)
    assert near >= 8  # most cases sit within a factor of two of the limit


def test_enum_int32_arithmetic(psa, oracle, monkeypatch):
    """PSA_ENUM_I32=1: the enumeration in int32 with 30-bit values (6 more fraction bits)."""
    seq1, seq2s = synth_problem(8, 1800, [63, 64, 65, 300, 1025, 1700])
    monkeypatch.setenv("PSA_ENUM_I32", "1")
    c = psa.Context(0)
    monkeypatch.delenv("PSA_ENUM_I32")
    c.set_seq1(seq1)
    for w in ([2, 1.5, 1.1, 1.3], [2, 1.5, 1, 1.25]):
        w = np.array(w, dtype=np.float32)
        c.set_weights(w)
        res = c.score_batch(seq2s)
        assert res[-1].enum_frac_bits == oracle.fixed_table(oracle.comp_matrix(w), 1700, oracle.SUM_BITS_I32)[0]
        check_against_oracle(oracle, seq1, seq2s, oracle.comp_matrix(w), res, sum_bits=oracle.SUM_BITS_I32)
    c.close()


def test_enum_vector_alu_form(psa, oracle, monkeypatch):
    """PSA_ENUM_MFMA=0 when a batch is created: its candidate loop runs on the vector ALU (the form
    the int32 arithmetic uses) and returns the records of the matrix-core loop."""
    seq1, seq2s = synth_problem(9, 2300, [64, 65, 300, 511, 1025, 2000])
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(w)
    matrix = c.score_batch(seq2s)
    check_against_oracle(oracle, seq1, seq2s, oracle.comp_matrix(w), matrix)
    monkeypatch.setenv("PSA_ENUM_MFMA", "0")
    vector = c.score_batch(seq2s)
    monkeypatch.delenv("PSA_ENUM_MFMA")
    assert [bytes(r) for r in vector] == [bytes(r) for r in matrix]
    c.close()


def test_runs_back_to_back_alternate(psa, oracle, monkeypatch):
    """Seven runs of one resident batch enqueued at once (consecutive candidate loops alternate
    between two streams, two slabs and two copies of the accumulators and overlap), for a one-pass
    and a many-pass batch, and the same with a single candidate-loop stream."""
    seq1, seq2s = synth_problem(12, 2500, [1999, 700, 1200, 64, 2000])
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    want = None
    for slab_mb, streams in ((None, None), ("5", None), (None, "1"), ("5", "1")):
        if slab_mb:
            monkeypatch.setenv("PSA_SLAB_MB", slab_mb)
        if streams:
            monkeypatch.setenv("PSA_ENUM_STREAMS", streams)
        c = psa.Context(0)
        c.set_seq1(seq1)
        c.set_weights(w)
        b = c.batch(seq2s)
        monkeypatch.delenv("PSA_SLAB_MB", raising=False)
        monkeypatch.delenv("PSA_ENUM_STREAMS", raising=False)
        for _ in range(7):
            b.enqueue()
        b.wait()
        got = [r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) for r in b.results()]
        if want is None:
            check_against_oracle(oracle, seq1, seq2s, oracle.comp_matrix(w), b.results())
            want = got
        assert got == want
        b.enqueue(psa.STAGE_ENUMERATE)
        b.enqueue()
        b.enqueue(psa.STAGE_RESOLVE)
        b.enqueue()
        b.wait()
        assert [r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) for r in b.results()] == want
        # two runs without the resolve stage in a row end on different streams, and each one's record
        # kernel clears the float32 half: neither may land behind the record kernels of the run after them
        for _ in range(6):
            b.enqueue()
            b.enqueue(psa.STAGE_ENUMERATE)
            b.enqueue(psa.STAGE_ENUMERATE)
            b.enqueue()
            b.wait()
            assert [r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) for r in b.results()] == want
        # mixed stages back to back: the records are those of the last run enqueued
        b.enqueue()
        b.enqueue(psa.STAGE_RESOLVE)
        b.wait()
        assert [r.key() + tuple(r.counts) for r in b.results()] == [w[:4] + w[8:12] for w in want]
        assert all(r.enum_n == 0 and r.enum_score == 0 for r in b.results())
        b.enqueue(psa.STAGE_RESOLVE)
        b.enqueue(psa.STAGE_ENUMERATE)
        b.wait()
        assert [r.enum_key() + tuple(r.enum_counts) for r in b.results()] == [w[4:8] + w[12:16] for w in want]
        # the float32 half of an enumeration-only run comes from the replayed shortlist, or stays empty
        for r, (cnt, state), w_ in zip(b.results(), b.shortlist(), want):
            assert (state == 1 and cnt >= 1 and r.key() + tuple(r.counts) == w_[:4] + w_[8:12]) or \
                   (state == 2 and r.n == 0 and r.score_bits == 0), (state, cnt, r.key())
        b.close()
        c.close()


def test_shortlist_replay(psa, oracle, golden, monkeypatch):
    """The enumeration stage's shortlist: every candidate within the rounding window of the enumerated
    maximum, replayed in float32.  On inputs with few near-maximal candidates the replayed winner is the
    reference's record (enum_check = 2) and an enumeration-only run returns it; a list that overflows
    is given up (enum_check = 1, enumeration-only run without a float32 record)."""
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    seq1, seq2s = synth_problem(31, 3000, [2000, 1999, 700, 301, 64, 5, 2])
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(w)
    b = c.batch(seq2s)
    both = b.run()
    check_against_oracle(oracle, seq1, seq2s, table, both)
    sl = b.shortlist()
    # (the Seq2 of two characters has one mutant, the empty one, which scores 0 at every offset: 3001 ties)
    assert all(r.enum_check == 2 for r in both[:-1]) and both[-1].enum_check == 1, [r.enum_check for r in both]
    assert all(state == 1 and cnt >= 1 for cnt, state in sl[:-1]) and sl[-1][1] == 2, sl
    only = b.run(psa.STAGE_ENUMERATE)
    for r, r2 in zip(only[:-1], both):
        assert r.key() == r2.key() and tuple(r.counts) == tuple(r2.counts) and r.mutant_num == r2.mutant_num
        assert r.enum_key() == r2.enum_key() and r.enum_check == 0
    assert only[-1].n == 0 and only[-1].enum_key() == both[-1].enum_key()
    assert b.shortlist() == sl
    res_only = b.run(psa.STAGE_RESOLVE)
    assert [r.key() for r in res_only] == [r.key() for r in both]
    assert all(state == 0 for _, state in b.shortlist())
    b.close()
    # the reference's own sample input
    case = golden[0]
    c.set_seq1(case["seq1"])
    c.set_weights(parse_weights(case["weights"]))
    res = c.score_batch(case["seq2"])
    check_against_oracle(oracle, case["seq1"], case["seq2"], oracle.comp_matrix(parse_weights(case["weights"])), res,
                         expect=case["expected"])
    assert all(r.enum_check == 2 for r in res), [r.enum_check for r in res]
    # every row inside the window (all-same sequences): more rows than a Seq2's quota, the list is given up
    rng = np.random.default_rng(5)
    seq1 = "AB" * 200
    seq2s = ["AB" * 45, "A" * 200, synth_seq(rng, 120, "ACDEFGHIKLMNPQRSTVWY")]
    c.set_seq1(seq1)
    c.set_weights(w)
    b = c.batch(seq2s)
    res = b.run()
    check_against_oracle(oracle, seq1, seq2s, table, res)
    sl = b.shortlist()
    assert sl[1][1] == 2 and res[1].enum_check == 1, (sl, [r.enum_check for r in res])
    only = b.run(psa.STAGE_ENUMERATE)
    for r, o, (cnt, state) in zip(res, only, sl):
        assert o.enum_key() == r.enum_key()
        if state == 1:
            assert r.enum_check == 2 and o.key() == r.key()
        else:
            assert state == 2 and r.enum_check == 1 and o.n == 0 and o.score_bits == 0
    b.close()
    # a small quota (3 rows per Seq2): the Seq2 with more rows are given up -- the same ones in every run and
    # in whatever company they are scored (the list holds every Seq2's full quota, so nothing depends on the
    # order in which blocks append)
    seq1b, seq2b = synth_problem(32, 1200, [300, 280, 250, 200, 310, 270])
    c.set_seq1(seq1b)
    full = c.batch(seq2b)
    full.run()
    rows = [cnt for cnt, _ in full.shortlist()]  # with the default quota: every list complete
    assert all(state == 1 for _, state in full.shortlist()) and max(rows) > 3, rows
    full.close()
    monkeypatch.setenv("PSA_SHORT_ROWS", "3")
    b = c.batch(seq2b)
    b1 = c.batch(seq2b[::-1][:4])
    monkeypatch.delenv("PSA_SHORT_ROWS")
    res = b.run()
    check_against_oracle(oracle, seq1b, seq2b, table, res)
    sl = b.shortlist()
    assert [state for _, state in sl] == [1 if n <= 3 else 2 for n in rows], (sl, rows)
    assert all((r.enum_check == 2) == (state == 1) for r, (_, state) in zip(res, sl)), sl
    for _ in range(3):  # the counters were left reset; the same records again
        assert [bytes(r) for r in b.run()] == [bytes(r) for r in res] and b.shortlist() == sl
    assert [bytes(r) for r in b1.run()] == [bytes(r) for r in res[::-1][:4]] and b1.shortlist() == sl[::-1][:4]
    b.close()
    b1.close()
    c.close()
    # switched off
    monkeypatch.setenv("PSA_SHORTLIST", "0")
    c = psa.Context(0)
    monkeypatch.delenv("PSA_SHORTLIST")
    c.set_seq1(seq1)
    c.set_weights(w)
    b = c.batch(seq2s)
    res = b.run()
    check_against_oracle(oracle, seq1, seq2s, table, res)
    assert all(r.enum_check == 1 for r in res) and all(state == 0 for _, state in b.shortlist())
    b.close()
    c.close()


def test_edge_stress(psa, ctx, oracle):
    """L2 = L1-1, L2 = L1+2, period-2/3 repeats, all-same, tie-forcing weights."""
    cases = []
    for unit, L1 in (("AB", 300), ("ABC", 301), ("NDE", 257), ("A", 200), ("HYH", 130)):
        seq1 = (unit * 400)[:L1]
        cases.append((seq1, [(unit * 400)[1:L1], (unit * 400)[:L1 + 2], (unit[::-1] * 400)[:L1 - 3], seq1,
                             (unit * 400)[:2], (unit * 400)[:3]]))
    for w in ([1, 1, 1, 1], [2, 1.5, 1, 1.25], [1.1, 1.1, 1.3, 1.3], [0, 0, 0, 0]):
        w = np.array(w, dtype=np.float32)
        table = oracle.comp_matrix(w)
        for seq1, seq2s in cases:
            ctx.set_seq1(seq1)
            ctx.set_weights(w)
            check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))


def test_fuzz_small(psa, ctx, oracle):
    """Seeded random small problems: lengths on both sides of the 16-, 64- and 256-position steps of
    the candidate loop, small alphabets (many ties), weights with and without exact float sums."""
    rng = np.random.default_rng(2024)
    edge = [2, 3, 4, 15, 16, 17, 18, 31, 32, 33, 47, 48, 49, 63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193,
            255, 256, 257, 258, 319, 320, 321]
    weights = ([2, 1.5, 1.1, 1.3], [1, 1, 1, 1], [3, 0.7, 0.2, 5], [1.1, 1.1, 1.3, 1.3])
    for it in range(12):
        alpha = "ACDEFGHIKLMNPQRSTVWY"[:int(rng.integers(1, 21))] if it % 3 else "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
        pick = lambda n: "".join(alpha[i] for i in rng.integers(0, len(alpha), size=n))
        L1 = int(rng.integers(330, 420))
        seq1 = pick(L1)
        lens = [int(x) for x in rng.choice(edge, size=6, replace=False)] + [int(rng.integers(2, L1 + 3)) for _ in range(4)]
        seq2s = []
        for L2 in lens:
            if rng.random() < 0.5 and L2 <= L1:  # a mutated piece of Seq1: a clear winner
                st = int(rng.integers(0, L1 - L2 + 1))
                s2 = list(seq1[st:st + L2])
                for _ in range(max(1, L2 // 10)):
                    s2[int(rng.integers(0, L2))] = alpha[int(rng.integers(0, len(alpha)))]
                seq2s.append("".join(s2))
            else:
                seq2s.append(pick(L2))
        w = np.array(weights[it % len(weights)], dtype=np.float32)
        table = oracle.comp_matrix(w)
        ctx.set_seq1(seq1)
        ctx.set_weights(w)
        check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))


def test_arbitrary_float_table(psa, ctx, oracle):
    rng = np.random.default_rng(9)
    table = rng.normal(0, 3, size=676).astype(np.float32)
    seq1, seq2s = synth_problem(8, 500, [40, 100, 260])
    ctx.set_seq1(seq1)
    ctx.set_table(table)
    check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))


def test_invalid_inputs(psa, ctx):
    ctx.set_seq1("ABCDEFGH")
    ctx.set_weights(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
    res = ctx.score_batch(["A", "ABCDEFGHIJK", "AB1D", "ABCD"])
    assert [r.status for r in res] == [2, 3, 4, 0]
    assert res[3].n >= 1
    with pytest.raises(psa.PsaError):
        ctx.set_seq1("abc")


def test_context_closed_before_its_batches(psa):
    """Closing a context closes the batches still open on it (a batch holds a pointer to its
    context); a closed context takes no new batch."""
    c = psa.Context(0)
    c.set_seq1("ABCDEFGHABCDEFGH")
    c.set_weights(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
    b = c.batch(["ABCDEF", "HGFEDCBA"])
    b.enqueue()  # a run nobody waits for
    c.close()
    b.close()  # already closed with the context: nothing left to do
    with pytest.raises(ValueError):
        c.batch(["ABCD"])


def test_offset_range_shards_combine(psa, ctx, oracle):
    """Long-pair mode: shards of the offset range combine to the whole-range answer."""
    seq1, seq2s = synth_problem(21, 5000, [600])
    seq2 = seq2s[0]
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    whole = ctx.score_batch([seq2])[0]
    S = len(seq1) - len(seq2) + 3
    cuts = [0, S // 8, S // 3, S // 2 + 1, S]
    parts = [ctx.score_range(seq2, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    for (a, b), p in zip(zip(cuts[:-1], cuts[1:]), parts):
        o = oracle.fast(seq1, seq2, table, a, b)
        assert p.key() == o.as_tuple()
    comb = psa.combine_results(parts)
    assert comb.key() == whole.key() and comb.enum_key() == whole.enum_key()
    assert comb.candidates == whole.candidates
    # all-equal scores: the tie must go to the smallest (n, k, offset), not to a shard
    ctx.set_seq1("A" * 400)
    parts = [ctx.score_range("A" * 50, a, b) for a, b in ((0, 100), (100, 250), (250, 353))]
    comb = psa.combine_results(parts[::-1])
    assert (comb.n, comb.k, comb.offset) == (1, 2, 0)


def test_offset_shards_more_than_offsets(psa, ctx, oracle):
    """More shards than offsets (Seq2 two characters longer than Seq1: one offset): the shards
    without an offset return a valid record without a candidate, and the combine passes over them."""
    from psa_amd import dist as pdist
    seq1, seq2s = synth_problem(22, 60, [62])
    seq2 = seq2s[0]
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    whole = ctx.score_batch([seq2])[0]
    assert whole.status == 0 and whole.candidates == 62 * 61 // 2
    parts = [ctx.score_range(seq2, *pdist.offset_shard(1, r, 4)) for r in range(4)]
    assert [p.status for p in parts] == [0, 0, 0, 0]
    assert sorted(p.candidates for p in parts) == [0, 0, 0, whole.candidates]
    assert all(p.n == 0 and p.enum_n == 0 for p in parts if p.candidates == 0)
    comb = psa.combine_results(parts)
    assert comb.key() == whole.key() and comb.enum_key() == whole.enum_key() and comb.candidates == whole.candidates
    assert psa.combine_results(parts[:3]).n == 0  # only empty shards: a record without a candidate


class _ThreadRanks:
    """nranks threads of one process as the ranks of a job: every rank has its own context on GPU 0, and the
    all-gather callback meets the other ranks at a barrier (the library calls release the interpreter lock)."""

    def __init__(self, nranks):
        import threading
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.slots = [None] * nranks
        self.calls = [[] for _ in range(nranks)]

    def gather(self, rank):
        def fn(send):
            self.calls[rank].append(len(send))
            self.slots[rank] = bytes(send)
            self.barrier.wait(timeout=120)
            out = b"".join(self.slots)
            self.barrier.wait(timeout=120)
            return out
        return fn

    def run(self, work):
        """work(rank, gather callback) on every rank; returns the list of results (exceptions re-raised)."""
        import threading
        out, err = [None] * self.n, [None] * self.n

        def body(r):
            try:
                out[r] = work(r, self.gather(r))
            except BaseException as e:  # noqa: B902 -- reported by the main thread
                err[r] = e
                self.barrier.abort()
        th = [threading.Thread(target=body, args=(r,)) for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        for e in err:
            if e is not None:
                raise e
        return out


def test_long_pair_allgather_entry(psa, ctx, oracle):
    """psa_score_range_allgather / psa_batch_run_allgather with 1, 2 and 3 ranks (threads, one context each):
    an enumerating run exchanges the shards' enumerated maxima after the candidate loops, every shard
    shortlists against the maximum over all shards, and the replayed winners of all shards together give
    the combined float32 answer -- enum_check 2 at every rank count, as on one GPU."""
    from psa_amd import dist as pdist
    seq1, seq2s = synth_problem(23, 4000, [700])
    seq2 = seq2s[0]
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    whole = ctx.score_batch([seq2])[0]
    check_against_oracle(oracle, seq1, [seq2], table, [whole])
    assert whole.enum_check == 2
    S = len(seq1) - len(seq2) + 3
    full = lambda r: (r.key(), r.enum_key(), tuple(r.counts), tuple(r.enum_counts), r.candidates, r.mutant_num)
    got1 = ctx.score_range_allgather(seq2, 0, 1, None)
    assert full(got1) == full(whole) and got1.enum_check == 2
    for nranks in (2, 3):
        for stages, sizes in ((psa.STAGE_BOTH, [16, 136]), (psa.STAGE_RESOLVE, [136]), (psa.STAGE_ENUMERATE, [16, 136])):
            ranks = _ThreadRanks(nranks)

            def work(rank, gather, stages=stages, nranks=nranks):
                c = psa.Context(0)
                try:
                    c.set_seq1(seq1)
                    c.set_weights(w)
                    one_shot = c.score_range_allgather(seq2, rank, nranks, gather, stages)
                    b = c.batch_range(seq2, *pdist.offset_shard(S, rank, nranks))
                    resident = [b.run_allgather(nranks, gather, stages) for _ in range(2)]  # resident shard, run twice
                    b.close()
                    return [one_shot] + resident
                finally:
                    c.close()
            per_rank = ranks.run(work)
            assert all(c == sizes * 3 for c in ranks.calls), ranks.calls
            for recs in per_rank:
                for r in recs:
                    assert r.status == 0 and r.candidates == whole.candidates
                    assert r.key() == whole.key() and tuple(r.counts) == tuple(whole.counts) and r.mutant_num == whole.mutant_num
                    if stages & psa.STAGE_ENUMERATE:
                        assert r.enum_key() == whole.enum_key() and tuple(r.enum_counts) == tuple(whole.enum_counts)
                    else:
                        assert r.enum_n == 0
                    assert r.enum_check == (2 if stages == psa.STAGE_BOTH else 0), r.enum_check
            assert all(bytes(recs[0]) == bytes(per_rank[0][0]) for recs in per_rank)  # every rank the same record
    # the hit sits in one shard; the others must list nothing instead of overflowing on their own near-maxima:
    # a planted hit in a long random pair, 4 ranks
    seq1b, seq2b = synth_problem(4, 60000, [3000])
    ranks = _ThreadRanks(4)

    def work_b(rank, gather):
        c = psa.Context(0)
        try:
            c.set_seq1(seq1b)
            c.set_weights(w)
            return c.score_range_allgather(seq2b[0], rank, 4, gather, psa.STAGE_BOTH)
        finally:
            c.close()
    recs = ranks.run(work_b)
    ctx.set_seq1(seq1b)
    whole_b = ctx.score_batch([seq2b[0]])[0]
    check_against_oracle(oracle, seq1b, [seq2b[0]], table, [whole_b])
    assert all(full(r) == full(whole_b) and r.enum_check == 2 for r in recs), [r.enum_check for r in recs]
    # every shard overflows (all-same sequences): bound only, and an enumeration-only run has no float32 answer
    ranks = _ThreadRanks(2)

    def work_c(rank, gather):
        c = psa.Context(0)
        try:
            c.set_seq1("A" * 3000)
            c.set_weights(w)
            return [c.score_range_allgather("A" * 400, rank, 2, gather, st) for st in (psa.STAGE_BOTH, psa.STAGE_ENUMERATE)]
        finally:
            c.close()
    for both, only in ranks.run(work_c):
        assert both.key() == oracle.fast("A" * 3000, "A" * 400, table).as_tuple() and both.enum_check == 1
        assert only.n == 0 and only.score_bits == 0 and only.enum_key() == both.enum_key()
    # a rank whose context is not ready still takes part in every exchange, and all ranks return its status
    ranks = _ThreadRanks(2)

    def work_d(rank, gather):
        c = psa.Context(0)
        try:
            if rank == 0:
                c.set_seq1(seq1)
                c.set_weights(w)
            try:
                c.score_range_allgather(seq2, rank, 2, gather, psa.STAGE_BOTH)
            except psa.PsaError as e:
                return e.status
            return 0
        finally:
            c.close()
    assert ranks.run(work_d) == [7, 7] and ranks.calls == [[16, 136], [16, 136]]
    # a callback that fails is reported, not swallowed
    ctx.set_seq1(seq1)
    with pytest.raises(psa.PsaError) as ei:
        ctx.score_range_allgather(seq2, 0, 2, lambda send: b"")
    assert ei.value.status == 9
    # an invalid pair gives the same status on every rank
    with pytest.raises(psa.PsaError) as ei:
        ctx.score_range_allgather("A", 0, 1, None)
    assert ei.value.status == 2


def test_answer_path_switch(psa, ctx, oracle, golden):
    """psa_score_batch_answer: the answer path a host program chooses -- resolve, enumerate (candidate loop +
    float32 replay of its shortlist; the resolve stage only for the Seq2 whose list overflowed) or both --
    gives the reference's answer for every Seq2 either way; a table with a NaN or an infinity is refused."""
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    assert [psa.parse_stages(t) for t in (None, "", "resolve", "enumerate", "both")] == [2, 2, 2, 1, 3]
    with pytest.raises(psa.PsaError):
        psa.parse_stages("everything")
    seq1, seq2s = synth_problem(41, 2500, [2000, 700, 301, 64, 5, 2])
    seq2s += ["A" * 300, "AB" * 150]  # every row ties: the shortlist overflows
    seq1 = seq1[:1200] + "A" * 400 + seq1[1200:]
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    want = [oracle.fast(seq1, s, table) for s in seq2s]
    for stages in (psa.STAGE_RESOLVE, psa.STAGE_ENUMERATE, psa.STAGE_BOTH):
        res, fallbacks = ctx.score_batch_answer(seq2s, stages)
        for r, o, s in zip(res, want, seq2s):
            assert r.status == 0 and r.key() == o.as_tuple() and r.mutant_num == o.mutant_num, (stages, len(s), r.key(), o.as_tuple())
            assert tuple(r.counts) == oracle.counts(seq1, s, r.offset, r.n, r.k)
            assert (r.enum_n > 0) == bool(stages & psa.STAGE_ENUMERATE)
        assert (fallbacks >= 2) if stages == psa.STAGE_ENUMERATE else fallbacks == 0, fallbacks
    res, fallbacks = ctx.score_batch_answer(["ABCD", "A", "AB1D"], psa.STAGE_ENUMERATE)
    assert [r.status for r in res] == [0, 2, 4] and res[0].n >= 1
    # the reference's sample input through the enumeration stage alone: no fallback needed
    case = golden[0]
    ctx.set_seq1(case["seq1"])
    ctx.set_weights(parse_weights(case["weights"]))
    res, fallbacks = ctx.score_batch_answer(case["seq2"], psa.STAGE_ENUMERATE)
    assert fallbacks == 0
    assert [r.key() for r in res] == [(e["n"], e["k"], e["offset"], e["bits"]) for e in case["expected"]]
    # non-finite table entries
    for bad in (np.nan, np.inf, -np.inf):
        t = table.copy()
        t[77] = bad
        with pytest.raises(psa.PsaError) as ei:
            ctx.set_table(t)
        assert ei.value.status == 1
    ctx.set_table(table)
    assert ctx.score_batch(case["seq2"][:1])[0].status == 0  # the context still holds a valid table


def test_context_guards(psa):
    """Seq1 and the table cannot be replaced under live batches; a context shut down before its
    batches detaches them (C callers: the Python wrapper closes batches first by itself)."""
    import ctypes as C
    L = psa.lib()
    h = C.c_void_p()
    assert L.psa_init(0, C.byref(h)) == 0
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    assert L.psa_set_seq1(h, b"ABCDEFGHABCDEFGH", -1) == 0
    assert L.psa_set_weights(h, w.ctypes.data_as(C.POINTER(C.c_float))) == 0
    arr = (C.c_char_p * 2)(b"ABCDEF", b"HGFEDCBA")
    bh = C.c_void_p()
    assert L.psa_batch_create(h, 2, arr, None, C.byref(bh)) == 0
    assert L.psa_set_seq1(h, b"ABCD", -1) == 7  # PSA_ERR_STATE: a shorter Seq1 under a live batch
    assert L.psa_set_weights(h, w.ctypes.data_as(C.POINTER(C.c_float))) == 7
    assert L.psa_batch_run(bh, 0) == 0
    out = (psa.Result * 2)()
    assert L.psa_batch_results(bh, out) == 0 and out[0].n >= 1
    assert L.psa_batch_enqueue(bh, 0) == 0  # a run nobody waits for
    L.psa_shutdown(h)  # before the batch is destroyed
    assert L.psa_batch_run(bh, 0) == 7 and L.psa_batch_results(bh, out) == 7 and L.psa_batch_wait(bh) == 7
    L.psa_batch_destroy(bh)  # the only call still allowed on it
    # with the batches gone both can be replaced
    c = psa.Context(0)
    c.set_seq1("ABCDEFGH")
    c.set_weights(w)
    b = c.batch(["ABCD"])
    with pytest.raises(psa.PsaError):
        c.set_seq1("ABCDEFGHIJ")
    b.close()
    c.set_seq1("ABCDEFGHIJ")
    assert c.score_batch(["ABCDEFGHIJ"])[0].status == 0
    c.close()


def test_dropin_shim(psa, oracle, golden, capfd):
    """psa_calc_best_score has calc_best_score_CUDA's contract: returns and echoes the line."""
    case = [c for c in golden if c["name"] == "ref_s_input2.txt"][0]
    table = psa.createCompMatrix(parse_weights(case["weights"]))
    for seq2, e in zip(case["seq2"], case["expected"]):
        line = psa.calc_best_score(case["seq1"], seq2, table)
        score = np.array([e["bits"]], dtype=np.uint32).view(np.float32)[0]
        mnum = psa.mutant_number(e["n"], e["k"], len(seq2)) - 1
        assert line == "mutant num: %d, MS(%d,%d), score: %1.2f, offset: %d\n" % (mnum, e["n"], e["k"], score,
                                                                                   e["offset"])


def _expected_lines(psa, case):
    out = []
    for seq2, e in zip(case["seq2"], case["expected"]):
        score = np.array([e["bits"]], dtype=np.uint32).view(np.float32)[0]
        mnum = psa.mutant_number(e["n"], e["k"], len(seq2)) - 1
        out.append("mutant num: %d, MS(%d,%d), score: %1.2f, offset: %d\n" % (mnum, e["n"], e["k"], score, e["offset"]))
    return out


def test_host_program_files(psa, golden, tmp_path):
    """input.txt -> output.txt through the C++ driver (psa_host) and the Python one."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "parallel-sequence-alignment_amd", "psa_host")
    assert os.path.exists(exe), "psa_host not built"
    from psa_amd import host
    for name in ("ref_s_input3.txt", "ref_p_input1.txt", "ref_input.txt"):
        case = [c for c in golden if c["name"] == name][0]
        inp = os.path.join(root, "tests", "golden", "inputs", name)
        out = tmp_path / ("out_" + name)
        r = subprocess.run([exe, inp, str(out), "1"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        want = _expected_lines(psa, case)
        assert open(out).read().splitlines(keepends=True) == want
        assert all(line in r.stdout for line in want)
        out2 = tmp_path / ("pyout_" + name)
        assert host.run(inp, str(out2), echo=False) == want
        assert open(out2).read().splitlines(keepends=True) == want
    # one worker thread per GPU (as many as the box has, at most 2), Seq2 above 5000 characters: the
    # kernels' large dynamic-LDS requests on every device a worker runs on
    seq1, seq2s = synth_problem(55, 7000, [5200, 900, 6100, 300])
    inp = tmp_path / "long_input.txt"
    inp.write_text("2 1.5 1.1 1.3\n%s\n%d\n%s\n" % (seq1, len(seq2s), "\n".join(seq2s)))
    out = tmp_path / "long_output.txt"
    r = subprocess.run([exe, str(inp), str(out), "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    import oracle_py
    table = oracle_py.comp_matrix(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
    want = []
    for s2 in seq2s:
        o = oracle_py.fast(seq1, s2, table)
        want.append("mutant num: %d, MS(%d,%d), score: %1.2f, offset: %d\n" % (o.mutant_num, o.n, o.k, o.score, o.offset))
    assert open(out).read().splitlines(keepends=True) == want


def test_mpi_host_program(psa, golden, tmp_path):
    """input.txt -> output.txt through the MPI driver with two ranks (both on GPU 0 here)."""
    import os
    import subprocess
    from shutil import which
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "parallel-sequence-alignment_amd", "psa_mpi")
    mpiexec = which("/opt/conda/bin/mpiexec") or which("mpiexec")
    if mpiexec is None or not os.path.exists(exe):
        pytest.skip("no MPI installation: psa_mpi is not built")
    for name, ranks in (("ref_s_input3.txt", 2), ("ref_input.txt", 2), ("ref_p_input1.txt", 3)):
        case = [c for c in golden if c["name"] == name][0]
        inp = os.path.join(root, "tests", "golden", "inputs", name)
        out = tmp_path / ("mpi_out_" + name)
        r = subprocess.run([mpiexec, "-np", str(ranks), exe, inp, str(out)], capture_output=True, text=True,
                           timeout=120, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr + r.stdout
        want = _expected_lines(psa, case)
        assert open(out).read().splitlines(keepends=True) == want
        assert all(line in r.stdout for line in want)


def test_mpi_host_program_long_pair(psa, golden, tmp_path):
    """psa_mpi --long-pair: every Seq2 scored by all ranks together (offset shards, one MPI_Allgather
    of one record per rank inside psa_score_range_allgather); the output is that of the normal mode."""
    import os
    import subprocess
    from shutil import which
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "parallel-sequence-alignment_amd", "psa_mpi")
    mpiexec = which("/opt/conda/bin/mpiexec") or which("mpiexec")
    if mpiexec is None or not os.path.exists(exe):
        pytest.skip("no MPI installation: psa_mpi is not built")
    # (ref_s_input2: Seq1 = 46, Seq2 = 45 -> 4 offsets over 3 ranks; ref_p_input1: 5 and 3 offsets over 4 ranks)
    for name, ranks in (("ref_input.txt", 2), ("ref_s_input3.txt", 3), ("ref_s_input2.txt", 3), ("ref_p_input1.txt", 4)):
        case = [c for c in golden if c["name"] == name][0]
        inp = os.path.join(root, "tests", "golden", "inputs", name)
        out = tmp_path / ("mpi_lp_out_" + name)
        r = subprocess.run([mpiexec, "-np", str(ranks), exe, "--long-pair", inp, str(out)], capture_output=True,
                           text=True, timeout=120, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr + r.stdout
        want = _expected_lines(psa, case)
        assert open(out).read().splitlines(keepends=True) == want
        assert all(line in r.stdout for line in want)


def test_reference_host_program_bound_to_libpsa_hip(psa, golden, tmp_path):
    """The drop-in boundary end to end: the reference's own MPI host program (its main.c and
    cFunctions.c compiled unmodified by oracle/Makefile, target refhost, with calc_best_score_CUDA
    bound to psa_calc_best_score) run as the reference runs it -- mpiexec -np 2, input.txt in the
    working directory -- must write the output.txt the compiled Sequential reference implies."""
    import os
    import shutil
    import subprocess
    from shutil import which
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "refhost_psa")
    mpiexec = which("/opt/conda/bin/mpiexec") or which("mpiexec")
    if mpiexec is None or not os.path.exists(exe):
        pytest.skip("no MPI installation or oracle/_ref/refhost_psa not built (make -C oracle refhost)")
    # inputs within the reference's fixed buffers (Seq1 < 5000, Seq2 < 3000 characters)
    for name in ("ref_input.txt", "ref_s_input3.txt", "ref_p_input1.txt", "ref_s_input2.txt"):
        cases = [c for c in golden if c["name"] == name]
        if not cases:
            continue
        case = cases[0]
        work = tmp_path / ("refhost_" + name)
        work.mkdir()
        shutil.copy(os.path.join(root, "tests", "golden", "inputs", name), work / "input.txt")
        env = dict(os.environ, PSA_DEBUG_SYNC="1")  # also makes the shim say which GPU each rank took
        env.pop("PSA_DEVICE", None)
        r = subprocess.run([mpiexec, "-np", "2", exe], capture_output=True, text=True, timeout=300, cwd=str(work), env=env)
        assert r.returncode == 0, r.stderr + r.stdout
        want = _expected_lines(psa, case)
        assert open(work / "output.txt").read().splitlines(keepends=True) == want
        assert all(line in r.stdout for line in want)  # the echo of P/cudaFunctions.cu:321
        # one rank per GPU: each rank picks the GPU of its rank on the node (both GPU 0 on a one-GPU box)
        import re
        picks = re.findall(r"psa_calc_best_score: pid \d+ uses GPU (\d+) of (\d+)", r.stderr)
        assert len(picks) == 2, r.stderr[-2000:]
        ngpu = int(picks[0][1])
        assert sorted(int(p[0]) for p in picks) == ([0, 1] if ngpu >= 2 else [0, 0])


def test_stages_switch_through_the_host_programs(psa, golden, tmp_path):
    """PSA_STAGES / --stages on the boundary: the reference's own host program (main.c + cFunctions.c,
    unmodified, bound to psa_calc_best_score), psa_host, psa_mpi and psa_mpi --long-pair write the same
    output.txt whether the answer comes from the resolve stage, from the candidate loop's replayed shortlist
    (enumerate) or from both; psa_host --sequential-format prints the Sequential Solution's stdout."""
    import os
    import shutil
    import subprocess
    from shutil import which
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "parallel-sequence-alignment_amd")
    mpiexec = which("/opt/conda/bin/mpiexec") or which("mpiexec")
    refhost = os.path.join(root, "oracle", "_ref", "refhost_psa")
    names = ("ref_input.txt", "ref_s_input3.txt", "ref_p_input1.txt")
    for name in names:
        case = [c for c in golden if c["name"] == name][0]
        want = _expected_lines(psa, case)
        inp = os.path.join(root, "tests", "golden", "inputs", name)
        for stages in ("both", "enumerate"):
            tag = "%s_%s" % (stages, name)
            out = tmp_path / ("host_" + tag)
            r = subprocess.run([os.path.join(pkg, "psa_host"), "--stages=" + stages, inp, str(out), "1"],
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            assert open(out).read().splitlines(keepends=True) == want
            if mpiexec and os.path.exists(os.path.join(pkg, "psa_mpi")):
                for extra in ([], ["--long-pair"]):
                    out = tmp_path / ("mpi_%s_%s" % ("lp" if extra else "list", tag))
                    r = subprocess.run([mpiexec, "-np", "2", os.path.join(pkg, "psa_mpi"), "--stages", stages] + extra +
                                       [inp, str(out)], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
                    assert r.returncode == 0, r.stderr + r.stdout
                    assert open(out).read().splitlines(keepends=True) == want
            if mpiexec and os.path.exists(refhost):
                work = tmp_path / ("refhost_" + tag)
                work.mkdir()
                shutil.copy(inp, work / "input.txt")
                env = dict(os.environ, PSA_STAGES=stages)
                env.pop("PSA_DEVICE", None)
                r = subprocess.run([mpiexec, "-np", "2", refhost], capture_output=True, text=True, timeout=300,
                                   cwd=str(work), env=env)
                assert r.returncode == 0, r.stderr + r.stdout
                assert open(work / "output.txt").read().splitlines(keepends=True) == want
                assert all(line in r.stdout for line in want)  # the echo of P/cudaFunctions.cu:321
    # a misspelt switch is an error, not a silent default
    r = subprocess.run([os.path.join(pkg, "psa_host"), "--stages=all", inp, str(tmp_path / "x")], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 2 and "resolve, enumerate or both" in r.stderr
    # the Sequential Solution's stdout (S/main.c:82, :87) as the compiled reference printed it for its input.txt
    ref = [l for l in open(os.path.join(root, "tests", "golden", "ref_input_reference_stdout.txt")).read().splitlines(keepends=True)
           if not l.startswith("PSA_REF_BITS")]
    inp = os.path.join(root, "tests", "golden", "inputs", "ref_input.txt")
    for stages in ("resolve", "enumerate"):
        out = tmp_path / ("seqfmt_" + stages)
        r = subprocess.run([os.path.join(pkg, "psa_host"), "--sequential-format", "--stages=" + stages, inp, str(out)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.splitlines(keepends=True) == ref
        assert open(out).read() == ""  # S/main.c:28-32 opens output.txt and never writes it
    from psa_amd import host
    lines = host.run(inp, str(tmp_path / "py_seq"), echo=False, stages=psa.STAGE_ENUMERATE, sequential_format=True)
    assert lines == ref[:-1]


def test_config3_shape_small(psa, ctx, oracle):
    """Batch of Seq2 of mixed lengths (every kernel variant in one launch set) against Seq1 = 10000."""
    rng = np.random.default_rng(33)
    lens = [int(x) for x in rng.integers(500, 2001, size=24)] + [40, 300, 2048, 2100]
    seq1, seq2s = synth_problem(34, 10000, lens)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))


def test_slab_passes(psa, oracle, monkeypatch):
    """Enumeration in many slab passes (5 MB slabs, the smallest allowed; unit sub-runs of 1, 3 and 64
    offsets): pass boundaries inside a Seq2 and between kernel variants, pre-pass and candidate
    loop alternating over the two slabs."""
    lens = [40, 300, 511, 512, 700, 1100, 2047, 2048, 2100, 4100]
    seq1, seq2s = synth_problem(91, 4500, lens)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    monkeypatch.setenv("PSA_SLAB_MB", "5")
    want = None
    for ns in ("1", "3", "64", None):
        if ns is None:
            monkeypatch.delenv("PSA_ENUM_NS")
        else:
            monkeypatch.setenv("PSA_ENUM_NS", ns)
        c = psa.Context(0)
        c.set_seq1(seq1)
        c.set_weights(w)
        res = c.score_batch(seq2s)
        if want is None:
            check_against_oracle(oracle, seq1, seq2s, table, res)
            want = [r.key() + r.enum_key() + tuple(r.enum_counts) for r in res]
        else:
            assert [r.key() + r.enum_key() + tuple(r.enum_counts) for r in res] == want
        part = c.score_range(seq2s[-1], 100, 333)
        e = oracle.enum_best(seq1, seq2s[-1], oracle.fixed_table(table, lens[-1])[1], 100, 333)
        assert part.enum_key() == (e.n, e.k, e.offset, e.score)
        assert part.key() == oracle.fast(seq1, seq2s[-1], table, 100, 333).as_tuple()
        c.close()


def test_prepass_forms_and_unit_numbering(psa, oracle, monkeypatch):
    """The tuning switches that change how the enumeration is laid out, not what it returns: PSA_PREP_FORM
    (pre-pass by one workgroup or one wave per chunk, for every batch instead of by batch size) and
    PSA_ENUM_BAND (candidate-loop units numbered in bands of levels per XCD): same records as the defaults."""
    lens = [40, 300, 511, 512, 700, 1100, 2047, 2048, 2100, 4100]
    seq1, seq2s = synth_problem(93, 4300, lens)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    monkeypatch.setenv("PSA_SLAB_MB", "5")
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(w)
    res = c.score_batch(seq2s)
    check_against_oracle(oracle, seq1, seq2s, oracle.comp_matrix(w), res)
    want = [bytes(r) for r in res]
    for name, value in (("PSA_PREP_FORM", "block"), ("PSA_PREP_FORM", "wave"), ("PSA_ENUM_BAND", "2"),
                        ("PSA_ENUM_BAND", "3"), ("PSA_ENUM_BAND", "64")):
        monkeypatch.setenv(name, value)
        assert [bytes(r) for r in c.score_batch(seq2s)] == want, (name, value)
        monkeypatch.delenv(name)
    c.close()


def test_stream_of_batches(psa, oracle):
    """Several batches of one context enqueued back to back before any is waited for: they share
    the context's slabs, streams and recycled buffers, and each returns what it returns alone."""
    seq1, seq2s = synth_problem(92, 2600, [900, 1500, 64, 2000, 333, 1200, 777, 1999])
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(w)
    groups = [seq2s[0:3], seq2s[3:4], seq2s[4:8], seq2s[1:2]]
    alone = [[r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) for r in c.score_batch(g)] for g in groups]
    for rounds in range(2):
        batches = [c.batch(g) for g in groups]
        for rep in range(3):
            for b in batches:
                b.enqueue()
            for b in (batches if rep % 2 == 0 else batches[::-1]):
                b.wait()
            got = [[r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) for r in b.results()] for b in batches]
            assert got == alone
        batches[2].enqueue()  # destroyed while its run is still pending
        for b in batches:
            b.close()
    table = oracle.comp_matrix(w)
    check_against_oracle(oracle, seq1, groups[2], table, c.score_batch(groups[2]))
    # a new Seq1 while nothing is pending, then the same batches again
    seq1b, _ = synth_problem(93, 2600, [10])
    c.set_seq1(seq1b)
    check_against_oracle(oracle, seq1b, groups[0], table, c.score_batch(groups[0]))
    c.close()


def test_event_times_of_a_pipeline(psa, oracle):
    """psa_batch_event_times: four batches enqueued one after the other report where their runs sat on the GPU's
    time axis -- ordered, inside the span psa_batch_timing gives, and -1 for a stage the run did not have."""
    seq1, seq2s = synth_problem(17, 1500, [700, 300])
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
    bs = [c.batch(seq2s) for _ in range(4)]
    for b in bs:
        b.run()
    for b in bs:
        b.enqueue()
    for b in bs:
        b.wait()
    starts = []
    for b in bs:
        t = b.event_times(bs[0])
        own = b.event_times()
        assert own[0] == 0.0 and all(x >= 0.0 for x in t), (t, own)
        assert t[0] <= t[1] <= t[2] <= t[3] and t[4] <= t[5], t
        assert max(own) <= b.timing()[1] + 1e-3, (own, b.timing())
        starts.append(t[1])
    assert starts == sorted(starts), starts
    bs[1].run(psa.STAGE_RESOLVE)
    t = bs[1].event_times()
    assert t[:4] == [-1.0] * 4 and 0.0 <= t[4] <= t[5], t
    with pytest.raises(psa.PsaError):
        c.batch(seq2s).event_times()  # has not run
    c.close()


def test_clock_probe_variant(psa, oracle):
    """The stamped variant of the candidate loop returns the same records as the plain one, and the
    clock it reports is a shader clock this chip can hold."""
    if os.environ.get("PSA_ENUM_MFMA") == "0" or os.environ.get("PSA_ENUM_I32") == "1":
        pytest.skip("the probe exists in the matrix-core candidate loop only")
    seq1, seq2s = synth_problem(94, 3000, [2000, 700, 1300, 64])
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(w)
    b = c.batch(seq2s)
    b.run()
    plain = [r.key() + r.enum_key() + tuple(r.enum_counts) for r in b.results()]
    assert c.clock_probe(True) == (0.0, 0.0)       # nothing ran with the probe on yet
    for _ in range(20):
        b.run()
    stamped = [r.key() + r.enum_key() + tuple(r.enum_counts) for r in b.results()]
    mhz, cycles = c.clock_probe(False)
    assert stamped == plain
    assert 500.0 < mhz < 2500.0 and cycles > 0
    b.run()
    assert c.clock_probe(False) == (0.0, 0.0)      # probe off: the counters stay clear
    assert [r.key() + r.enum_key() + tuple(r.enum_counts) for r in b.results()] == plain
    b.close()
    c.close()


def test_long_seq2(psa, ctx, oracle):
    """Seq2 of 4095/4096/8000/8191 characters: the two largest kernel variants."""
    seq1, seq2s = synth_problem(77, 9000, [4095, 4096, 8000, 8191])
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    check_against_oracle(oracle, seq1, seq2s, table, ctx.score_batch(seq2s))
    res = ctx.score_batch(["A" * 8192])
    assert res[0].status == 3


def test_config4_full_size(psa, ctx, oracle):
    """Seq1 = 200000, Seq2 = 8000 (6.1e12 candidates): whole range against the O(L1*L2) oracle, and
    two offset shards combining to the same record."""
    seq1, seq2s = synth_problem(4, 200000, [8000])
    seq2 = seq2s[0]
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    whole = ctx.score_batch([seq2])[0]
    check_against_oracle(oracle, seq1, [seq2], table, [whole])
    S = len(seq1) - len(seq2) + 3
    parts = [ctx.score_range(seq2, 0, S // 2, psa.STAGE_RESOLVE), ctx.score_range(seq2, S // 2, S, psa.STAGE_RESOLVE)]
    assert psa.combine_results(parts).key() == whole.key()


def test_config5_at_scale(psa, oracle):
    """Tie stress at the long pair's size: Seq1 = 200000, Seq2 = 8000, all-same, period 2 and period 3 -- every
    offset (every offset; every third offset) attains the maximum, up to 192003 ties for the float32 tie-break.  The
    answer is the oracle's, and the resolve stage takes no more than three times what it takes on the random
    pair of that size (its tie kernels prune on what is already known instead of visiting every tied offset)."""
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    c = psa.Context(0)
    c.set_weights(w)

    def resolve_ms(seq1, seq2):
        c.set_seq1(seq1)
        b = c.batch([seq2])
        b.run(psa.STAGE_RESOLVE)  # (first run: buffers, LDS limits)
        best = 1e30
        for _ in range(3):
            r = b.run(psa.STAGE_RESOLVE)[0]
            best = min(best, b.timing()[1])
        b.close()
        return r, best
    seq1, seq2s = synth_problem(4, 200000, [8000])
    r, t_random = resolve_ms(seq1, seq2s[0])
    assert r.key() == oracle.fast(seq1, seq2s[0], table).as_tuple()
    for name, seq1, seq2 in (("all-same", "A" * 200000, "A" * 8000), ("period 2", "AB" * 100000, "AB" * 4000),
                             ("period 3", ("ABC" * 66667)[:200000], ("ABC" * 2667)[:8000])):
        r, t = resolve_ms(seq1, seq2)
        o = oracle.fast(seq1, seq2, table)
        assert r.status == 0 and r.key() == o.as_tuple() and r.mutant_num == o.mutant_num, (name, r.key(), o.as_tuple())
        assert tuple(r.counts) == oracle.counts(seq1, seq2, r.offset, r.n, r.k)
        assert t < 3.0 * t_random, (name, t, t_random)
    # both stages on the all-same pair: the enumeration ties everywhere too (first maximiser (1, 2, 0)), and its
    # shortlist overflows, so the cross-check stays at the rounding bound
    c.set_seq1("A" * 200000)
    r = c.score_batch(["A" * 8000])[0]
    assert r.key() == o.as_tuple() or r.key()[:3] == (1, 2, 0)
    assert (r.enum_n, r.enum_k, r.enum_offset) == (1, 2, 0) and r.enum_check == 1
    c.close()


def test_config3_batch_properties(psa, ctx, oracle):
    """A batch of 384 Seq2 of 500..2000 characters against Seq1 = 10000: results do not depend on
    the order or the company a Seq2 is scored in, and a sample equals the oracle."""
    rng = np.random.default_rng(303)
    lens = [int(x) for x in rng.integers(500, 2001, size=384)]
    seq1, seq2s = synth_problem(304, 10000, lens)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    ctx.set_seq1(seq1)
    ctx.set_weights(w)
    res = ctx.score_batch(seq2s)
    perm = rng.permutation(len(seq2s))
    res_p = ctx.score_batch([seq2s[i] for i in perm])
    for j, i in enumerate(perm):
        assert res_p[j].key() == res[i].key() and res_p[j].enum_key() == res[i].enum_key()
        assert tuple(res_p[j].counts) == tuple(res[i].counts)
    sample = [int(i) for i in rng.choice(len(seq2s), size=10, replace=False)]
    check_against_oracle(oracle, seq1, [seq2s[i] for i in sample], table, [res[i] for i in sample])
    alone = ctx.score_batch([seq2s[sample[0]]])[0]
    assert alone.key() == res[sample[0]].key() and alone.enum_key() == res[sample[0]].enum_key()
    assert sum(r.candidates for r in res) == sum((10000 - L + 3) * (L * (L - 1) // 2) for L in lens)


def test_config3_full_size(psa, oracle):
    """BASELINE's batch at full size: 4096 Seq2 of 500..2000 characters against Seq1 = 10000 (3.0e13
    candidates; dozens of slab passes, millions of chunks), in two orders; a seeded sample of 32
    against the oracle; the split over 8 workers scored part by part gives the same records."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(3)
    lens = [int(x) for x in rng.integers(500, 2001, size=4096)]
    seq1, seq2s = synth_problem(3, 10000, lens)
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = oracle.comp_matrix(w)
    c = psa.Context(0)
    c.set_seq1(seq1)
    c.set_weights(w)
    res = c.score_batch(seq2s)
    assert all(r.status == 0 for r in res)
    assert sum(r.candidates for r in res) == sum((10000 - L + 3) * (L * (L - 1) // 2) for L in lens)
    sample = sorted(int(i) for i in rng.choice(len(seq2s), size=32, replace=False))
    with ThreadPoolExecutor(max_workers=16) as ex:  # the oracle calls release the GIL
        list(ex.map(lambda i: check_against_oracle(oracle, seq1, [seq2s[i]], table, [res[i]]), sample))
        # every one of the 4096 float32 answers against the oracle (O(L1 L2) each; the enumeration's own
        # oracle is O(L1 L2) with a larger constant and stays on the sample)
        def float_answer(i):
            o = oracle.fast(seq1, seq2s[i], table)
            return res[i].key() == o.as_tuple() and res[i].mutant_num == o.mutant_num
        wrong = [i for i, ok in enumerate(ex.map(float_answer, range(len(seq2s)))) if not ok]
    assert not wrong, wrong[:10]
    assert all(r.enum_check == 2 for r in res)  # and each is also the winner of its replayed shortlist
    full = lambda r: r.key() + r.enum_key() + tuple(r.counts) + tuple(r.enum_counts) + (r.mutant_num, r.candidates)
    perm = rng.permutation(len(seq2s))
    res_p = c.score_batch([seq2s[i] for i in perm])
    assert all(full(res_p[j]) == full(res[i]) for j, i in enumerate(perm))
    del res_p
    owner = psa.partition(lens, 10000, 8)
    for part in (0, 7):  # two of the eight shares, each scored as its rank would score it
        idx = [i for i in range(len(seq2s)) if owner[i] == part]
        got = c.score_batch([seq2s[i] for i in idx])
        assert all(full(g) == full(res[i]) for g, i in zip(got, idx))
    c.close()


def _run_bench_ranks(nproc, bench_args, timeout=600):
    """bench.py under torch.distributed.run in fresh child processes; returns the JSON line of rank 0."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py")] + bench_args
    env = dict(os.environ)
    env.pop("PSA_DEVICE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu(psa):
    """The multi-rank code of bench.py on a one-GPU box: two ranks (gloo, both on GPU 0) run the headline and the
    scale block -- the list divided by psa_partition with the records gathered and compared with a one-rank run,
    and the long pair with its two exchanges per step and the combine."""
    d = _run_bench_ranks(2, ["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "5", "--nseq2", "64",
                             "--no-cpu-baseline", "--no-boundary", "--repeats", "2"])
    assert d["parity"] is True and d["n_gpus"] == 2, d.get("parity_detail")
    c3, c4 = d["scale"]["c3"], d["scale"]["c4"]
    assert c3["parity"] and c3["n_ranks_seen"] == 2 and not c3["complaints"]
    assert c4["parity"] and c4["n_ranks_seen"] == 2 and not c4["complaints"]
    assert all(x > 0 for x in c3["per_rank_candidates"]) and all(x > 0 for x in c4["per_rank_candidates"])
    assert c4["collective"] == "gloo"
    # the answer of the long pair comes from the enumerated candidates at two ranks as on one
    assert c4["shortlist"] == {"records": 1, "replay_equals_answer": 1, "bound_only": 0, "other": 0}
    assert c3["shortlist"]["replay_equals_answer"] == c3["shortlist"]["records"] == 64


def test_bench_one_rank_through_rccl(psa):
    """The long pair's exchanges through an RCCL communicator (one rank: the only RCCL a one-GPU box can build):
    process-group creation, the second group for the data path, device-side all-gather of the two messages."""
    d = _run_bench_ranks(1, ["--gpus", "1", "--force-dist", "--backend", "nccl", "--config", "c4", "--steps", "2",
                             "--no-cpu-baseline"])
    assert d["parity"] is True, d.get("parity_detail")
    assert d["config"]["collective"] == "rccl" and d["config"]["exchanges_per_step"] == 2
    assert d["parity_detail"]["shortlist"]["replay_equals_answer"] == 1


def test_authors_recorded_output(psa):
    """P/output.txt:1-4 -- what the CUDA program wrote for P/input.txt."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "inputs", "ref_input.txt")
    w, seq1, seq2s = psa.readFromFile(path)
    table = psa.createCompMatrix(w)
    lines = [psa.calc_best_score(seq1, s, table) for s in seq2s]
    assert lines == ["mutant num: 84, MS(1,86), score: 272.00, offset: 364\n",
                     "mutant num: 176, MS(1,178), score: 2383.50, offset: 129\n",
                     "mutant num: 207, MS(1,209), score: 343.30, offset: 858\n",
                     "mutant num: 208, MS(1,210), score: 750.00, offset: 976\n"]
