"""ctypes access to oracle/libpsa_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, from __graft_entry__.smoke()
and from the cpu_baseline leg of bench.py, and from nowhere else.  The product
package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpsa_oracle.so")


class OracleResult(C.Structure):
    _fields_ = [("n", C.c_int32), ("k", C.c_int32), ("offset", C.c_int32),
                ("mutant_num", C.c_int32), ("score", C.c_float),
                ("score_bits", C.c_uint32), ("status", C.c_int32)]

    def as_tuple(self):
        return (self.n, self.k, self.offset, self.score_bits)


class OracleEnumResult(C.Structure):
    _fields_ = [("score", C.c_int64), ("n", C.c_int32), ("k", C.c_int32),
                ("offset", C.c_int32), ("status", C.c_int32)]


def build(force=False):
    """Compile the C oracle (and, where /root/reference exists, the reference binary)."""
    src = os.path.join(_HERE, "psa_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/Sequential Solution"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    # the reference's MPI host program bound to libpsa_hip (needs an MPI and the built library)
    psa_lib = os.path.join(os.path.dirname(_HERE), "parallel-sequence-alignment_amd", "libpsa_hip.so")
    mpi_home = os.environ.get("MPI_HOME", "/opt/conda")
    if (os.path.isdir("/root/reference/Parallel Solution") and os.path.exists(psa_lib)
            and os.path.exists(os.path.join(mpi_home, "include", "mpi.h"))):
        subprocess.check_call(["make", "-C", _HERE, "refhost"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        f32p = C.POINTER(C.c_float)
        i32p = C.POINTER(C.c_int32)
        L.psa_oracle_class.argtypes = [C.c_int, C.c_int]
        L.psa_oracle_class_table.argtypes = [u8p]
        L.psa_oracle_comp_matrix.argtypes = [f32p, f32p]
        L.psa_oracle_get_nk.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.psa_oracle_mutant_number.argtypes = [C.c_int, C.c_int, C.c_int]
        L.psa_oracle_mutant_number.restype = C.c_int64
        L.psa_oracle_brute.argtypes = [u8p, C.c_int64, u8p, C.c_int64, f32p, C.c_int64, C.c_int64,
                                       C.c_int64, C.c_int64, C.POINTER(OracleResult)]
        L.psa_oracle_brute_mt.argtypes = [u8p, C.c_int64, u8p, C.c_int64, f32p, C.c_int64, C.c_int64,
                                          C.c_int, C.POINTER(OracleResult)]
        L.psa_oracle_counts.argtypes = [u8p, u8p, C.c_int64, C.c_int64, C.c_int, C.c_int, i32p]
        L.psa_oracle_fast.argtypes = [u8p, C.c_int64, u8p, C.c_int64, f32p, C.c_int64, C.c_int64,
                                      C.POINTER(OracleResult)]
        L.psa_oracle_forward.argtypes = [u8p, C.c_int64, u8p, C.c_int64, f32p, f32p]
        L.psa_oracle_fixed_table.argtypes = [f32p, C.c_int64, C.c_int, i32p, C.POINTER(C.c_int)]
        L.psa_oracle_enum_brute.argtypes = [u8p, C.c_int64, u8p, C.c_int64, i32p, C.c_int64, C.c_int64,
                                            C.POINTER(OracleEnumResult)]
        L.psa_oracle_enum_fast.argtypes = L.psa_oracle_enum_brute.argtypes
        L.psa_oracle_enum_score.argtypes = [u8p, u8p, C.c_int64, i32p, C.c_int64, C.c_int, C.c_int]
        L.psa_oracle_enum_score.restype = C.c_int64
        L.psa_oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _u8(s):
    if isinstance(s, str):
        s = s.encode()
    a = np.frombuffer(bytes(s), dtype=np.uint8).copy()
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def comp_matrix(weights):
    w, wp = _f32(weights)
    out = np.zeros(676, dtype=np.float32)
    lib().psa_oracle_comp_matrix(wp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def class_table():
    out = np.zeros(676, dtype=np.uint8)
    lib().psa_oracle_class_table(out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def get_nk(t, L2):
    n, k = C.c_int(), C.c_int()
    lib().psa_oracle_get_nk(t, L2, C.byref(n), C.byref(k))
    return n.value, k.value


def mutant_number(n, k, L2):
    return lib().psa_oracle_mutant_number(n, k, L2)


def brute(seq1, seq2, table, t_begin=None, t_end=None, s_begin=None, s_end=None, threads=0):
    a, ap = _u8(seq1)
    b, bp = _u8(seq2)
    t, tp = _f32(table)
    L1, L2 = len(a), len(b)
    M = L2 * (L2 - 1) // 2
    S = L1 - L2 + 3
    t_begin = 1 if t_begin is None else t_begin
    t_end = M + 1 if t_end is None else t_end
    r = OracleResult()
    if threads and threads > 1:
        lib().psa_oracle_brute_mt(ap, L1, bp, L2, tp, t_begin, t_end, threads, C.byref(r))
    else:
        lib().psa_oracle_brute(ap, L1, bp, L2, tp, t_begin, t_end,
                               0 if s_begin is None else s_begin, S if s_end is None else s_end,
                               C.byref(r))
    return r


def fast(seq1, seq2, table, s_begin=0, s_end=None):
    a, ap = _u8(seq1)
    b, bp = _u8(seq2)
    t, tp = _f32(table)
    L1, L2 = len(a), len(b)
    r = OracleResult()
    lib().psa_oracle_fast(ap, L1, bp, L2, tp, s_begin, (L1 - L2 + 3) if s_end is None else s_end,
                          C.byref(r))
    return r


def forward(seq1, seq2, table):
    a, ap = _u8(seq1)
    b, bp = _u8(seq2)
    t, tp = _f32(table)
    S = len(a) - len(b) + 3
    F = np.zeros(max(S, 1), dtype=np.float32)
    st = lib().psa_oracle_forward(ap, len(a), bp, len(b), tp, F.ctypes.data_as(C.POINTER(C.c_float)))
    if st:
        raise ValueError("invalid input, status %d" % st)
    return F


def counts(seq1, seq2, s, n, k):
    a, ap = _u8(seq1)
    b, bp = _u8(seq2)
    out = (C.c_int32 * 4)()
    lib().psa_oracle_counts(ap, bp, len(b), s, n, k, out)
    return tuple(out)


SUM_BITS_F32 = 23  # float32 arithmetic of the enumeration stage (the library's default)
SUM_BITS_I32 = 29  # int32 arithmetic (PSA_ENUM_I32=1)


def fixed_table(table, L2, sum_bits=SUM_BITS_F32):
    t, tp = _f32(table)
    it = np.zeros(676, dtype=np.int32)
    exact = C.c_int()
    r = lib().psa_oracle_fixed_table(tp, L2, sum_bits, it.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(exact))
    return r, it, exact.value


def enum_best(seq1, seq2, itab, s_begin=0, s_end=None, brute_force=False):
    a, ap = _u8(seq1)
    b, bp = _u8(seq2)
    it = np.ascontiguousarray(itab, dtype=np.int32)
    L1, L2 = len(a), len(b)
    r = OracleEnumResult()
    fn = lib().psa_oracle_enum_brute if brute_force else lib().psa_oracle_enum_fast
    fn(ap, L1, bp, L2, it.ctypes.data_as(C.POINTER(C.c_int32)), s_begin,
       (L1 - L2 + 3) if s_end is None else s_end, C.byref(r))
    return r


def enum_score(seq1, seq2, itab, s, n, k):
    a, ap = _u8(seq1)
    b, bp = _u8(seq2)
    it = np.ascontiguousarray(itab, dtype=np.int32)
    return lib().psa_oracle_enum_score(ap, bp, len(b), it.ctypes.data_as(C.POINTER(C.c_int32)), s, n, k)


def max_threads():
    return lib().psa_oracle_max_threads()


REF_BIN = os.path.join(_HERE, "_ref", "seqref")


def run_reference(input_path):
    """Run the compiled reference (oracle/_ref/seqref) on an input file.

    Returns a list of dicts {n, k, offset, bits, text} in Seq2 order.  Only the
    reference's fixed limits apply (Seq1 < 5000, Seq2 < 3000 characters)."""
    env = dict(os.environ, PSA_REF_INPUT=os.path.abspath(input_path))
    out = subprocess.run([REF_BIN], env=env, cwd="/tmp", capture_output=True, text=True, check=True).stdout
    res, last = [], None
    for line in out.splitlines():
        if line.startswith("MS("):
            last = line
        elif line.startswith("PSA_REF_BITS"):
            kv = dict(x.split("=") for x in line.split()[1:])
            res.append({"n": int(kv["n"]), "k": int(kv["k"]), "offset": int(kv["offset"]),
                        "bits": int(kv["bits"], 16), "text": last})
    return res
