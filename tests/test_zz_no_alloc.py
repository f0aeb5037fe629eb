"""psa_score_range_noalloc -- the entry point that allocates nothing -- through the C ABI against
psa_score_range (resolve stage), the entry point it is a variant of: bit-exact n, k, offset, score
bits, class counts, mutant number and candidate count, on two calls in a row (the second with a
smaller input: the fixed array keeps the first call's contents), then shards, an empty shard, an
invalid residue found on the device, and the size limit.  Device memory for inputs and the record
comes from the HIP runtime the library itself is linked to, called through ctypes (a second HIP
runtime in the process -- torch bundles its own -- finds no GPU once the first has initialised, so
the test does not depend on which of the two came first)."""
import ctypes as C

import numpy as np
import pytest

from conftest import synth_problem

pytestmark = pytest.mark.gpu


def _fields(r):
    return (r.status, r.n, r.k, r.offset, r.score_bits, r.mutant_num, tuple(r.counts), r.candidates,
            r.enum_n, r.enum_k, r.enum_offset, r.enum_score, tuple(r.enum_counts))


class Hip:
    """The few HIP runtime calls the test needs, on the runtime libpsa_hip.so has already loaded."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.rt.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        self.rt.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self.rt.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        self.rt.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        self.rt.hipGraphExecDestroy.argtypes = [C.c_void_p]
        self.rt.hipGraphDestroy.argtypes = [C.c_void_p]
        self.bufs = []

    def free_bytes(self):
        free, total = C.c_size_t(), C.c_size_t()
        self.ok(self.rt.hipMemGetInfo(C.byref(free), C.byref(total)))
        return free.value

    def ok(self, rc):
        assert rc == 0, "HIP error %d" % rc

    def malloc(self, nbytes):
        p = C.c_void_p()
        self.ok(self.rt.hipMalloc(C.byref(p), max(nbytes, 1)))
        self.bufs.append(p)
        return p

    def upload(self, data):
        p = self.malloc(len(data))
        self.ok(self.rt.hipMemcpy(p, data, len(data), 1))  # hipMemcpyHostToDevice
        return p

    def download(self, p, nbytes):
        out = C.create_string_buffer(nbytes)
        self.ok(self.rt.hipMemcpy(out, p, nbytes, 2))  # hipMemcpyDeviceToHost
        return out.raw

    def free_all(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


def test_noalloc_entry_matches_score_range(oracle):
    import psa_amd
    L = psa_amd.lib()
    L.psa_score_range_noalloc.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                          C.c_int64, C.c_void_p, C.c_void_p]
    L.psa_noalloc_last_error_text.restype = C.c_char_p
    ctx = psa_amd.Context(0)  # makes GPU 0 current for this thread
    hip = Hip()
    assert L.psa_noalloc_init() == 0, L.psa_noalloc_last_error_text()
    w = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)
    table = psa_amd.createCompMatrix(w)
    d_table = hip.upload(table.tobytes())
    ctx.set_table(table)
    stream = C.c_void_p()
    hip.ok(hip.rt.hipStreamCreate(C.byref(stream)))
    d_out = hip.malloc(96)

    def noalloc(seq1, seq2, sb, se, expect_status=0, st_=stream):
        d1, d2 = hip.upload(seq1.encode()), hip.upload(seq2.encode())
        hip.ok(hip.rt.hipMemset(d_out, 0xAB, 96))  # stale bytes must all be overwritten
        st = L.psa_score_range_noalloc(d1, len(seq1), d2, len(seq2), d_table, sb, se, st_, d_out)
        assert st == expect_status, (st, L.psa_noalloc_last_error_text())
        if st != 0:
            return None
        hip.ok(hip.rt.hipStreamSynchronize(st_))
        return psa_amd.Result.from_buffer_copy(hip.download(d_out, 96))

    # two calls in a row, the second with a smaller input, then a larger one again
    for seed, L1, L2 in ((41, 6000, 2000), (42, 900, 130), (43, 20000, 4100), (44, 6, 4)):
        seq1, seq2s = synth_problem(seed, L1, [L2])
        seq2 = seq2s[0]
        S = L1 - L2 + 3
        ctx.set_seq1(seq1)
        want = ctx.score_range(seq2, 0, S, psa_amd.STAGE_RESOLVE)
        got = noalloc(seq1, seq2, 0, S)
        assert _fields(got) == _fields(want), (seed, _fields(got), _fields(want))
        assert got.enum_frac_bits == -1 and got.enum_exact == -1
        o = oracle.fast(seq1, seq2, oracle.comp_matrix(w))
        assert got.key() == o.as_tuple()
        # a shard of the offsets, and a range that is clipped to nothing
        lo, hi = S // 3, max(S // 3 + 1, 2 * S // 3)
        assert _fields(noalloc(seq1, seq2, lo, hi)) == _fields(ctx.score_range(seq2, lo, hi, psa_amd.STAGE_RESOLVE))
        empty = noalloc(seq1, seq2, S + 5, S + 9)
        assert _fields(empty) == _fields(ctx.score_range(seq2, S + 5, S + 9, psa_amd.STAGE_RESOLVE))
        assert empty.status == 0 and empty.n == 0 and empty.candidates == 0
    # tie-heavy input: the first maximum in (n, k, offset) order
    seq1, seq2 = "AB" * 300, "BA" * 40
    ctx.set_seq1(seq1)
    S = len(seq1) - len(seq2) + 3
    assert _fields(noalloc(seq1, seq2, 0, S)) == _fields(ctx.score_range(seq2, 0, S, psa_amd.STAGE_RESOLVE))
    # an invalid character is found on the device; the next call is clean again
    bad = noalloc("ABCDEFGH1JKLMNOP", "ABCDEF", 0, 13)
    assert bad.status == 4 and bad.n == 0 and bad.candidates == 0
    bad = noalloc("ABCDEFGHIJKLMNOP", "ABcDEF", 0, 13)
    assert bad.status == 4
    ctx.set_seq1("ABCDEFGHIJKLMNOP")
    assert _fields(noalloc("ABCDEFGHIJKLMNOP", "ABCDEF", 0, 13)) == \
        _fields(ctx.score_range("ABCDEF", 0, 13, psa_amd.STAGE_RESOLVE))
    # a table that is not symmetric (the entry point transposes it on the device, psa_set_table on the host)
    rng = np.random.default_rng(46)
    asym = rng.normal(0, 3, size=676).astype(np.float32)
    d_asym = hip.upload(asym.tobytes())
    seq1, seq2s = synth_problem(46, 3000, [700])
    ctx.set_seq1(seq1)
    ctx.set_table(asym)
    S = len(seq1) - 700 + 3
    d1, d2 = hip.upload(seq1.encode()), hip.upload(seq2s[0].encode())
    want = ctx.score_range(seq2s[0], 0, S, psa_amd.STAGE_RESOLVE)
    assert want.key() == oracle.fast(seq1, seq2s[0], asym).as_tuple()
    hip.ok(hip.rt.hipMemset(d_out, 0xAB, 96))
    # ... and the claim itself: the device's free memory is the same before and after the call
    hip.ok(hip.rt.hipStreamSynchronize(stream))
    free_before = hip.free_bytes()
    assert L.psa_score_range_noalloc(d1, len(seq1), d2, 700, d_asym, 0, S, stream, d_out) == 0
    hip.ok(hip.rt.hipStreamSynchronize(stream))
    assert hip.free_bytes() == free_before
    assert _fields(psa_amd.Result.from_buffer_copy(hip.download(d_out, 96))) == _fields(want)
    # ... and one call captured into a HIP graph (capture forbids allocation), replayed twice
    hip.ok(hip.rt.hipMemset(d_out, 0xAB, 96))
    graph, gexec = C.c_void_p(), C.c_void_p()
    hip.ok(hip.rt.hipStreamBeginCapture(stream, 2))  # hipStreamCaptureModeRelaxed
    st_cap = L.psa_score_range_noalloc(d1, len(seq1), d2, 700, d_asym, 0, S, stream, d_out)
    hip.ok(hip.rt.hipStreamEndCapture(stream, C.byref(graph)))
    assert st_cap == 0, L.psa_noalloc_last_error_text()
    hip.ok(hip.rt.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0))
    for _ in range(2):
        hip.ok(hip.rt.hipGraphLaunch(gexec, stream))
        hip.ok(hip.rt.hipStreamSynchronize(stream))
        assert _fields(psa_amd.Result.from_buffer_copy(hip.download(d_out, 96))) == _fields(want)
        hip.ok(hip.rt.hipMemset(d_out, 0xAB, 96))
    hip.ok(hip.rt.hipGraphExecDestroy(gexec))
    hip.ok(hip.rt.hipGraphDestroy(graph))
    # a table entry that is not finite is found on the device; the next call is clean again
    t_bad = table.copy()
    t_bad[5] = np.inf
    d_bad = hip.upload(t_bad.tobytes())
    ctx.set_table(table)
    ctx.set_seq1("ABCDEFGHIJKLMNOP")
    d1, d2 = hip.upload(b"ABCDEFGHIJKLMNOP"), hip.upload(b"ABCDEF")
    assert L.psa_score_range_noalloc(d1, 16, d2, 6, d_bad, 0, 13, stream, d_out) == 0
    hip.ok(hip.rt.hipStreamSynchronize(stream))
    assert psa_amd.Result.from_buffer_copy(hip.download(d_out, 96)).status == 1
    assert _fields(noalloc("ABCDEFGHIJKLMNOP", "ABCDEF", 0, 13)) == \
        _fields(ctx.score_range("ABCDEF", 0, 13, psa_amd.STAGE_RESOLVE))
    # the error text belongs to the most recent call: cleared by a call that succeeds
    assert L.psa_noalloc_last_error_text() == b""
    # sizes beyond the fixed capacity are refused with a clear message; so are impossible lengths
    big = hip.malloc(262145)
    d2 = hip.upload(b"ABCDEF")
    st = L.psa_score_range_noalloc(big, 262145, d2, 6, d_table, 0, 10, None, d_out)
    assert st == 10 and b"PSA_NOALLOC_MAX_SEQ1" in L.psa_noalloc_last_error_text()
    assert L.psa_score_range_noalloc(big, 100, d2, 1, d_table, 0, 10, None, d_out) == 2
    assert b"shorter than 2" in L.psa_noalloc_last_error_text()
    assert L.psa_score_range_noalloc(big, 3, d2, 6, d_table, 0, 10, None, d_out) == 3
    assert b"longer than" in L.psa_noalloc_last_error_text()
    # the largest Seq1 it takes, on the default stream
    seq1, seq2s = synth_problem(45, 262144, [600])
    ctx.set_seq1(seq1)
    S = len(seq1) - 600 + 3
    want = ctx.score_range(seq2s[0], 0, S, psa_amd.STAGE_RESOLVE)
    assert _fields(noalloc(seq1, seq2s[0], 0, S, st_=None)) == _fields(want)
    hip.ok(hip.rt.hipStreamDestroy(stream))
    hip.free_all()
    ctx.close()
