"""ctypes binding of the CPU oracle (oracle/libandi_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from andi_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libandi_oracle.so")

M_RAW, M_JC, M_KIMURA, M_LOGDET, M_ANI = range(5)


class Model(C.Structure):
    _fields_ = [("counts", C.c_uint32 * 16), ("seq_len", C.c_uint32)]


class Interval(C.Structure):
    _fields_ = [("l", C.c_int32), ("i", C.c_int32), ("j", C.c_int32), ("m", C.c_int32)]


class Esa(C.Structure):
    _fields_ = [
        ("S", C.c_void_p),
        ("SA", C.POINTER(C.c_int32)),
        ("LCP", C.POINTER(C.c_int32)),
        ("len", C.c_int32),
        ("cache", C.POINTER(Interval)),
        ("FVC", C.c_void_p),
        ("CLD", C.POINTER(C.c_int32)),
    ]


class Subject(C.Structure):
    _fields_ = [
        ("RS", C.c_void_p),
        ("RSlen", C.c_size_t),
        ("gc", C.c_double),
        ("threshold", C.c_size_t),
    ]


class ScanStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_uint64),
        ("esa_probes", C.c_uint64),
        ("lucky_tries", C.c_uint64),
        ("lucky_hits", C.c_uint64),
        ("anchor_pairs", C.c_uint64),
        ("gap_chars", C.c_uint64),
    ]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "andi_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "libandi_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_normalize.restype = C.c_size_t
        L.orc_normalize.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.orc_catcomp.restype = C.c_void_p
        L.orc_catcomp.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_gc.restype = C.c_double
        L.orc_gc.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_binomial.restype = C.c_size_t
        L.orc_binomial.argtypes = [C.c_size_t, C.c_size_t]
        L.orc_shustring_cum_prob.restype = C.c_double
        L.orc_shustring_cum_prob.argtypes = [C.c_size_t, C.c_double, C.c_size_t]
        L.orc_min_anchor_length.restype = C.c_size_t
        L.orc_min_anchor_length.argtypes = [C.c_double, C.c_double, C.c_size_t]
        L.orc_subject_init.restype = C.c_int
        L.orc_subject_init.argtypes = [C.POINTER(Subject), C.c_char_p, C.c_size_t, C.c_double]
        L.orc_subject_free.argtypes = [C.POINTER(Subject)]
        L.orc_suffix_array.restype = C.c_int
        L.orc_suffix_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_esa_init.restype = C.c_int
        L.orc_esa_init.argtypes = [C.POINTER(Esa), C.POINTER(Subject)]
        L.orc_esa_init_with_sa.restype = C.c_int
        L.orc_esa_init_with_sa.argtypes = [C.POINTER(Esa), C.POINTER(Subject), C.c_void_p]
        L.orc_esa_free.argtypes = [C.POINTER(Esa)]
        L.orc_get_match.restype = Interval
        L.orc_get_match.argtypes = [C.POINTER(Esa), C.c_char_p, C.c_size_t]
        L.orc_get_match_cached.restype = Interval
        L.orc_get_match_cached.argtypes = [C.POINTER(Esa), C.c_char_p, C.c_size_t]
        L.orc_model_count.argtypes = [C.POINTER(Model), C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_model_count_equal.argtypes = [C.POINTER(Model), C.c_char_p, C.c_size_t, C.c_int]
        L.orc_model_average.restype = Model
        L.orc_model_average.argtypes = [C.POINTER(Model), C.POINTER(Model)]
        L.orc_model_coverage.restype = C.c_double
        L.orc_model_coverage.argtypes = [C.POINTER(Model)]
        L.orc_estimate.restype = C.c_double
        L.orc_estimate.argtypes = [C.POINTER(Model), C.c_int]
        L.orc_dist_anchor.restype = Model
        L.orc_dist_anchor.argtypes = [
            C.POINTER(Esa), C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(ScanStats)]
        L.orc_dist_matrix.restype = C.c_int
        L.orc_dist_matrix.argtypes = [
            C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t,
            C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_scan_row.restype = None
        L.orc_scan_row.argtypes = [
            C.c_void_p, C.POINTER(Esa), C.c_size_t, C.POINTER(C.c_char_p),
            C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.orc_bootstrap_matrix.restype = None
        L.orc_bootstrap_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
        L.orc_ran_binomial.restype = C.c_uint64
        L.orc_ran_binomial.argtypes = [C.POINTER(C.c_uint64), C.c_double, C.c_uint64]
        _lib = L
    return _lib


class OracleEsa:
    """Subject + ESA built by the oracle; arrays exposed as numpy views."""

    def __init__(self, seq: bytes, p_value=0.025, sa=None):
        L = lib()
        self.seq = seq
        self.sub = Subject()
        if L.orc_subject_init(C.byref(self.sub), seq, len(seq), p_value):
            raise RuntimeError("orc_subject_init failed")
        self.esa = Esa()
        if sa is None:
            rc = L.orc_esa_init(C.byref(self.esa), C.byref(self.sub))
        else:
            sa = np.ascontiguousarray(sa, dtype=np.int32)
            rc = L.orc_esa_init_with_sa(C.byref(self.esa), C.byref(self.sub), sa.ctypes.data)
        if rc:
            raise RuntimeError("orc_esa_init failed")
        self.n = int(self.sub.RSlen)
        self.threshold = int(self.sub.threshold)
        self.gc = float(self.sub.gc)

    @property
    def RS(self) -> bytes:
        return C.string_at(self.sub.RS, self.n)

    def _arr(self, ptr, count, dtype):
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(
            C.cast(ptr, C.c_void_p).value)
        return np.frombuffer(buf, dtype=dtype, count=count)

    @property
    def SA(self):
        return self._arr(self.esa.SA, self.n, np.int32)

    @property
    def LCP(self):
        return self._arr(self.esa.LCP, self.n + 1, np.int32)

    @property
    def CLD(self):
        return self._arr(self.esa.CLD, self.n + 1, np.int32)

    @property
    def FVC(self):
        return self._arr(self.esa.FVC, self.n, np.uint8)

    @property
    def cache(self):
        return self._arr(self.esa.cache, 4 * (1 << 20), np.int32).reshape(-1, 4)

    def get_match(self, q: bytes, cached=True):
        L = lib()
        f = L.orc_get_match_cached if cached else L.orc_get_match
        r = f(C.byref(self.esa), q, len(q))
        return (r.l, r.i, r.j)

    def dist_anchor(self, q: bytes, model=M_JC, stats=False, threshold=None):
        L = lib()
        st = ScanStats()
        thr = self.threshold if threshold is None else threshold
        m = L.orc_dist_anchor(C.byref(self.esa), q, len(q), thr, model, C.byref(st))
        out = np.array(list(m.counts) + [m.seq_len], dtype=np.uint32)
        if stats:
            return out, {k: int(getattr(st, k)) for k, _ in ScanStats._fields_}
        return out

    def close(self):
        if self.esa.SA:
            lib().orc_esa_free(C.byref(self.esa))
        if self.sub.RS:
            lib().orc_subject_free(C.byref(self.sub))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dist_matrix(seqs, p_value=0.025, model=M_JC, threads=0, times=False):
    """n*n*17 uint32 count matrix (row = subject, column = query)."""
    L = lib()
    n = len(seqs)
    arr = (C.c_char_p * n)(*seqs)
    lens = (C.c_size_t * n)(*[len(s) for s in seqs])
    M = np.zeros((n, n, 17), dtype=np.uint32)
    t = (C.c_double * 2)()
    rc = L.orc_dist_matrix(M.ctypes.data, arr, lens, n, p_value, model, threads, t)
    if rc:
        raise RuntimeError("orc_dist_matrix failed")
    return (M, (t[0], t[1])) if times else M


def scan_row(esa, seqs, self_idx=-1, model=M_JC, threads=0):
    """dist_anchor of every sequence against one prepared subject (row of the matrix): (n, 17) uint32."""
    L = lib()
    n = len(seqs)
    arr = (C.c_char_p * n)(*seqs)
    lens = (C.c_size_t * n)(*[len(s) for s in seqs])
    row = np.zeros((n, 17), dtype=np.uint32)
    L.orc_scan_row(row.ctypes.data, C.byref(esa.esa), esa.threshold, arr, lens, n,
                   n if self_idx < 0 else self_idx, model, threads)
    return row


def estimate(counts17, model=M_JC):
    m = Model()
    for k in range(16):
        m.counts[k] = int(counts17[k])
    m.seq_len = int(counts17[16])
    return lib().orc_estimate(C.byref(m), model)


def coverage(counts17):
    m = Model()
    for k in range(16):
        m.counts[k] = int(counts17[k])
    m.seq_len = int(counts17[16])
    return lib().orc_model_coverage(C.byref(m))


def suffix_array(text: bytes):
    n = len(text)
    buf = C.create_string_buffer(text, n + 1)
    sa = np.empty(n, dtype=np.int32)
    if lib().orc_suffix_array(C.cast(buf, C.c_void_p), sa.ctypes.data, n):
        raise RuntimeError("orc_suffix_array failed")
    return sa


def bootstrap(M, replicates, seed=1):
    """calculate_bootstrap (src/process.c:289-321) with the restated gsl_ran_multinomial: (replicates, n, n, 17) uint32.
    Not a stream anyone else produces (the reference seeds from the clock): draws of the right DISTRIBUTION."""
    M = np.ascontiguousarray(M, dtype=np.uint32)
    n = M.shape[0]
    assert M.shape == (n, n, 17)
    B = np.zeros((replicates, n, n, 17), dtype=np.uint32)
    L = lib()
    for r in range(replicates):
        L.orc_bootstrap_matrix(B[r].ctypes.data, M.ctypes.data, n, (seed * 0x9E3779B97F4A7C15 + r * 0xD1B54A32D192ED03) & (2**64 - 1))
    return B
