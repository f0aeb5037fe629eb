"""Python host side of libpsa_hip (ctypes over the C ABI declared in include/psa.h).

The names mirror the reference's host program ("Parallel Solution/main.c",
"cFunctions.c"): readFromFile, createCompMatrix, calc_best_score (the drop-in for
calc_best_score_CUDA), CPUGetNK, calcPortion, writeToFile.  Context and Batch
wrap the resident-data API used by the benchmark and the multi-GPU drivers.

There is no CPU fallback: if the shared library is missing or no GPU is visible
the calls raise.
"""
import ctypes as C
import os
import weakref

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# PSA_HIP_LIB names another build of the library to load (comparing two builds in one session)
LIB_PATH = os.environ.get("PSA_HIP_LIB") or os.path.join(os.path.dirname(_PKG), "libpsa_hip.so")

PSA_TABLE_SIZE = 676
STAGE_ENUMERATE = 1
STAGE_RESOLVE = 2
STAGE_BOTH = 3


class PsaError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        txt = lib().psa_strerror(status).decode()
        last = lib().psa_last_error_text().decode()
        super().__init__("%s: %s%s" % (where or "libpsa_hip", txt, (" [" + last + "]") if last else ""))


class Result(C.Structure):
    """struct psa_result of include/psa.h"""
    _fields_ = [("n", C.c_int32), ("k", C.c_int32), ("offset", C.c_int32), ("mutant_num", C.c_int32),
                ("score", C.c_float), ("score_bits", C.c_uint32), ("counts", C.c_int32 * 4),
                ("enum_n", C.c_int32), ("enum_k", C.c_int32), ("enum_offset", C.c_int32),
                ("enum_score", C.c_int32), ("enum_frac_bits", C.c_int32), ("enum_exact", C.c_int32),
                ("enum_counts", C.c_int32 * 4), ("status", C.c_int32), ("enum_check", C.c_int32),
                ("candidates", C.c_int64)]

    def key(self):
        return (self.n, self.k, self.offset, self.score_bits)

    def enum_key(self):
        return (self.enum_n, self.enum_k, self.enum_offset, self.enum_score)

    def as_dict(self):
        d = {f: getattr(self, f) for f, _ in self._fields_ if f not in ("counts", "enum_counts")}
        d["counts"] = list(self.counts)
        d["enum_counts"] = list(self.enum_counts)
        return d


# int (*psa_allgather_fn)(const void *send, void *recv, size_t bytes, void *comm)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None


def lib():
    """Loads libpsa_hip.so from the source tree; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libpsa_hip.so not built: run `make -C parallel-sequence-alignment_amd` "
                      "(or __graft_entry__.build()) first; there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, f32p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.psa_device_count.restype = C.c_int
    L.psa_default_device.restype = C.c_int
    L.psa_score_range_allgather.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int, C.c_int, ALLGATHER_FN, vp,
                                            C.c_uint32, C.POINTER(Result)]
    L.psa_batch_run_allgather.argtypes = [vp, C.c_int, ALLGATHER_FN, vp, C.c_uint32, C.POINTER(Result)]
    L.psa_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.psa_shutdown.argtypes = [vp]
    L.psa_shutdown.restype = None
    L.psa_set_seq1.argtypes = [vp, C.c_char_p, C.c_int64]
    L.psa_set_weights.argtypes = [vp, f32p]
    L.psa_set_table.argtypes = [vp, f32p]
    L.psa_clock_probe.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.psa_batch_create.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), i32p, C.POINTER(vp)]
    L.psa_batch_create_range.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(vp)]
    L.psa_batch_run.argtypes = [vp, C.c_uint32]
    L.psa_batch_enqueue.argtypes = [vp, C.c_uint32]
    L.psa_batch_wait.argtypes = [vp]
    L.psa_batch_results.argtypes = [vp, C.POINTER(Result)]
    L.psa_batch_timing.argtypes = [vp, f32p, f32p]
    L.psa_batch_event_times.argtypes = [vp, vp, f32p]
    L.psa_batch_shortlist.argtypes = [vp, i32p, i32p]
    L.psa_batch_size.argtypes = [vp]
    L.psa_batch_passes.argtypes = [vp]
    L.psa_batch_destroy.argtypes = [vp]
    L.psa_batch_destroy.restype = None
    L.psa_score_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), i32p, C.c_uint32, C.POINTER(Result)]
    L.psa_score_range.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_uint32, C.POINTER(Result)]
    L.psa_score_batch_answer.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), i32p, C.c_uint32, C.POINTER(Result),
                                         C.POINTER(C.c_int)]
    L.psa_parse_stages.argtypes = [C.c_char_p, C.POINTER(C.c_uint32)]
    L.psa_combine_results.argtypes = [C.POINTER(Result), C.c_int, C.POINTER(Result)]
    L.psa_calc_best_score.argtypes = [C.c_char_p, C.c_char_p, f32p]
    L.psa_calc_best_score.restype = C.c_void_p  # malloc'd char*, freed below
    L.psa_create_comp_matrix.argtypes = [f32p, C.c_int, f32p]
    L.psa_create_comp_matrix.restype = None
    L.psa_residue_class.argtypes = [C.c_char, C.c_char]
    L.psa_get_nk.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.psa_get_nk.restype = None
    L.psa_mutant_number.argtypes = [C.c_int, C.c_int, C.c_int]
    L.psa_mutant_number.restype = C.c_int64
    L.psa_format_result.argtypes = [C.POINTER(Result)]
    L.psa_format_result.restype = C.c_void_p
    L.psa_partition.argtypes = [C.c_int, i32p, C.c_int64, C.c_int, i32p]
    L.psa_partition_stages.argtypes = [C.c_int, i32p, C.c_int64, C.c_int, C.c_uint32, i32p]
    L.psa_cost_estimate.argtypes = [C.c_int32, C.c_int64, C.c_uint32]
    L.psa_cost_estimate.restype = C.c_double
    L.psa_calc_portion.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.psa_calc_portion.restype = None
    L.psa_strerror.argtypes = [C.c_int]
    L.psa_strerror.restype = C.c_char_p
    L.psa_last_error_text.restype = C.c_char_p
    L.psa_version.restype = C.c_char_p
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _take_string(ptr):
    if not ptr:
        return None
    s = C.string_at(ptr).decode()
    _libc.free(ptr)
    return s


def _check(st, where):
    if st != 0:
        raise PsaError(st, where)


def _bytes(s):
    return s.encode() if isinstance(s, str) else bytes(s)


# ---- the reference's host helpers ------------------------------------------------

def readFromFile(file_name):
    """Input grammar of P/cFunctions.c:17-37: four weights, Seq1, a count, then that
    many Seq2, separated by whitespace; anything after them is ignored.
    Returns (weights float32[4], seq1, [seq2...])."""
    with open(file_name) as f:
        tok = f.read().split()
    weights = np.array([np.float32(t) for t in tok[0:4]], dtype=np.float32)
    seq1 = tok[4]
    count = int(tok[5])
    return weights, seq1, tok[6:6 + count]


def writeToFile(file, final_result):
    """P/cFunctions.c:40-46"""
    if file is None:
        raise ValueError("writeToFile() - file is NULL")
    file.write(final_result)


def createCompMatrix(weights):
    """26x26 float32 table of P/cFunctions.c:55-64, row = Seq1 residue."""
    w = np.ascontiguousarray(weights, dtype=np.float32)
    out = np.zeros(PSA_TABLE_SIZE, dtype=np.float32)
    lib().psa_create_comp_matrix(out.ctypes.data_as(C.POINTER(C.c_float)), PSA_TABLE_SIZE,
                                 w.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def residue_class(a, b):
    return lib().psa_residue_class(_bytes(a)[:1], _bytes(b)[:1])


def CPUGetNK(mutant_num, seq2_len):
    """P/cFunctions.c:104-119: 1-based mutant number -> (n, k)."""
    n, k = C.c_int(), C.c_int()
    lib().psa_get_nk(mutant_num, seq2_len, C.byref(n), C.byref(k))
    return n.value, k.value


def mutant_number(n, k, seq2_len):
    return lib().psa_mutant_number(n, k, seq2_len)


def calcPortion(num):
    """P/cFunctions.c:49-52"""
    p = (C.c_int * 2)()
    lib().psa_calc_portion(p, num)
    return p[0], p[1]


def partition(len2, len1, nparts, stages=0):
    """Cost-balanced owner of each Seq2 (replaces the count-based calcPortion); the cost is the
    predicted GPU time of the stages that will run (0 = both)."""
    l2 = np.ascontiguousarray(len2, dtype=np.int32)
    owner = np.zeros(len(l2), dtype=np.int32)
    _check(lib().psa_partition_stages(len(l2), l2.ctypes.data_as(C.POINTER(C.c_int32)), len1, nparts, stages,
                                      owner.ctypes.data_as(C.POINTER(C.c_int32))), "psa_partition_stages")
    return owner


def cost_estimate(len2, len1, stages=0):
    return lib().psa_cost_estimate(len2, len1, stages)


def parse_stages(text):
    """"resolve" | "enumerate" | "both" (what PSA_STAGES and the host programs' --stages take) -> STAGE_* bits;
    None or "" -> STAGE_RESOLVE."""
    out = C.c_uint32()
    _check(lib().psa_parse_stages(None if text is None else _bytes(text), C.byref(out)), "psa_parse_stages")
    return out.value


def default_device():
    """GPU for this process: PSA_DEVICE, else the launcher's local rank modulo the GPU count, else 0."""
    return lib().psa_default_device()


def calc_best_score(seq1, seq2, comp_matrix):
    """Drop-in for calc_best_score_CUDA (P/cudaFunctions.cu:260): returns the result line."""
    t = np.ascontiguousarray(comp_matrix, dtype=np.float32)
    assert t.size == PSA_TABLE_SIZE
    ptr = lib().psa_calc_best_score(_bytes(seq1), _bytes(seq2), t.ctypes.data_as(C.POINTER(C.c_float)))
    return _take_string(ptr)


def format_result(r):
    return _take_string(lib().psa_format_result(C.byref(r)))


def combine_results(parts):
    arr = (Result * len(parts))(*parts)
    out = Result()
    _check(lib().psa_combine_results(arr, len(parts), C.byref(out)), "psa_combine_results")
    return out


# ---- resident-data API -------------------------------------------------------------

def _allgather_callback(allgather, nranks):
    """Wraps `allgather(send: bytes) -> bytes of all ranks in rank order` as a psa_allgather_fn."""
    if allgather is None:
        return ALLGATHER_FN()

    def _cb(send, recv, nbytes, _comm):
        try:
            got = allgather(C.string_at(send, nbytes))
            if len(got) != nbytes * nranks:
                return 1
            C.memmove(recv, got, len(got))
            return 0
        except Exception:  # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1
    return ALLGATHER_FN(_cb)


class Context:
    """One per GPU: Seq1 and the table stay in HBM across batches."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().psa_init(device, C.byref(self._h)), "psa_init")
        self.device = device
        self._batches = weakref.WeakSet()  # a batch must not outlive its context

    def close(self):
        if self._h:
            for b in list(self._batches):
                b.close()
            lib().psa_shutdown(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seq1(self, seq1):
        b = _bytes(seq1)
        _check(lib().psa_set_seq1(self._h, b, len(b)), "psa_set_seq1")
        self.len1 = len(b)

    def set_weights(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.float32)
        _check(lib().psa_set_weights(self._h, w.ctypes.data_as(C.POINTER(C.c_float))), "psa_set_weights")

    def set_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float32)
        assert t.size == PSA_TABLE_SIZE
        _check(lib().psa_set_table(self._h, t.ctypes.data_as(C.POINTER(C.c_float))), "psa_set_table")

    def clock_probe(self, enable):
        """Reads (clock MHz, wave cycles per unit) of the candidate loops run since the last call with
        the probe on, clears the counters, and switches the probe on or off for the runs that follow."""
        mhz, cyc = C.c_double(), C.c_double()
        _check(lib().psa_clock_probe(self._h, 1 if enable else 0, C.byref(mhz), C.byref(cyc)), "psa_clock_probe")
        return mhz.value, cyc.value

    def batch(self, seq2s):
        return Batch(self, seq2s)

    def batch_range(self, seq2, s_begin, s_end):
        return Batch(self, [seq2], (s_begin, s_end))

    def score_batch(self, seq2s, stages=STAGE_BOTH):
        b = Batch(self, seq2s)
        try:
            return b.run(stages)
        finally:
            b.close()

    def score_batch_answer(self, seq2s, stages=STAGE_RESOLVE):
        """psa_score_batch_answer: the reference's answer for every Seq2 by the path `stages` names -- the
        resolve stage, the enumeration stage (candidate loop + float32 replay of its shortlist, the resolve
        stage only for the Seq2 whose list overflowed) or both, which must agree.  Returns (records, number of
        Seq2 that fell back to the resolve stage)."""
        bs = [_bytes(s) for s in seq2s]
        n = len(bs)
        arr = (C.c_char_p * max(n, 1))(*bs)
        lens = np.array([len(b) for b in bs], dtype=np.int32)
        out = (Result * max(n, 1))()
        fb = C.c_int()
        _check(lib().psa_score_batch_answer(self._h, n, arr, lens.ctypes.data_as(C.POINTER(C.c_int32)), stages, out,
                                            C.byref(fb)), "psa_score_batch_answer")
        return list(out)[:n], fb.value

    def score_range_allgather(self, seq2, rank, nranks, allgather=None, stages=STAGE_BOTH):
        """One long pair over nranks ranks (psa_score_range_allgather): this rank's offset shard is
        scored here, `allgather(send: bytes) -> bytes of all ranks in rank order` moves the records,
        and every rank returns the combined record."""
        cb = _allgather_callback(allgather, nranks)
        b = _bytes(seq2)
        out = Result()
        _check(lib().psa_score_range_allgather(self._h, b, len(b), rank, nranks, cb, None, stages, C.byref(out)),
               "psa_score_range_allgather")
        return out

    def score_range(self, seq2, s_begin, s_end, stages=STAGE_BOTH):
        b = Batch(self, [seq2], (s_begin, s_end))
        try:
            return b.run(stages)[0]
        finally:
            b.close()


class Batch:
    """A set of Seq2 resident in HBM with its work lists and result buffers."""

    def __init__(self, ctx, seq2s, offset_range=None):
        self._h = C.c_void_p()
        self.ctx = ctx
        if not ctx._h:
            raise ValueError("the context is closed")
        ctx._batches.add(self)
        bs = [_bytes(s) for s in seq2s]
        self.n = len(bs)
        if offset_range is None:
            arr = (C.c_char_p * max(self.n, 1))(*bs)
            lens = np.array([len(b) for b in bs], dtype=np.int32)
            _check(lib().psa_batch_create(ctx._h, self.n, arr, lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                          C.byref(self._h)), "psa_batch_create")
        else:
            assert self.n == 1
            _check(lib().psa_batch_create_range(ctx._h, bs[0], len(bs[0]), offset_range[0], offset_range[1],
                                                C.byref(self._h)), "psa_batch_create_range")

    def launch(self, stages=STAGE_BOTH):
        """Runs the kernels on the resident data and waits; no host copies of inputs."""
        _check(lib().psa_batch_run(self._h, stages), "psa_batch_run")

    def enqueue(self, stages=STAGE_BOTH):
        """Puts one run on the context's streams and returns at once; wait() completes it."""
        _check(lib().psa_batch_enqueue(self._h, stages), "psa_batch_enqueue")

    def wait(self):
        _check(lib().psa_batch_wait(self._h), "psa_batch_wait")

    def results(self):
        out = (Result * max(self.n, 1))()
        _check(lib().psa_batch_results(self._h, out), "psa_batch_results")
        return list(out)[:self.n]

    def run(self, stages=STAGE_BOTH):
        self.launch(stages)
        return self.results()

    def run_allgather(self, nranks, allgather=None, stages=STAGE_BOTH):
        """A resident shard of one long pair (Context.batch_range): one run, one all-gather of one
        record per rank, the max-loc combine; every rank returns the same record."""
        out = Result()
        _check(lib().psa_batch_run_allgather(self._h, nranks, _allgather_callback(allgather, nranks), None, stages,
                                             C.byref(out)), "psa_batch_run_allgather")
        return out

    def shortlist(self):
        """Per Seq2 of the last run: (entries of its shortlist, state) -- state 0 no shortlist, 1 complete and
        replayed in float32, 2 overflowed."""
        cnt, st = (C.c_int32 * max(self.n, 1))(), (C.c_int32 * max(self.n, 1))()
        _check(lib().psa_batch_shortlist(self._h, cnt, st), "psa_batch_shortlist")
        return list(zip(cnt, st))[:self.n]

    def passes(self):
        """Candidate-loop launches of one enumerating run (slab passes)."""
        return lib().psa_batch_passes(self._h)

    def timing(self):
        """(enumeration kernel ms, first launch to last completion ms) of the last run"""
        a, b = C.c_float(), C.c_float()
        _check(lib().psa_batch_timing(self._h, C.byref(a), C.byref(b)), "psa_batch_timing")
        return a.value, b.value

    def event_times(self, ref=None):
        """Six times (ms) of the last run on the GPU's time axis, relative to the start of `ref`'s last run."""
        out = (C.c_float * 6)()
        _check(lib().psa_batch_event_times(self._h, (ref or self)._h, out), "psa_batch_event_times")
        return list(out)

    def close(self):
        if self._h:
            lib().psa_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
