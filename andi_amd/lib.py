"""ctypes binding of libandihip.so (include/andi_hip.h).

The names follow the reference's functions (seq_subject_init, esa_init,
get_match_cached, dist_anchor, distMatrix, estimate_*).  There is no CPU
fallback: if the library is missing or no HIP device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANDI_HIP_LIB") or os.path.join(_HERE, "libandihip.so")  # override: diagnostic builds

M_RAW, M_JC, M_KIMURA, M_LOGDET, M_ANI = range(5)
MODEL_NAMES = {"raw": M_RAW, "jc": M_JC, "kimura": M_KIMURA, "logdet": M_LOGDET, "ani": M_ANI}


class AndiHipError(RuntimeError):
    pass


class Seq(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("len", C.c_size_t)]


class Model(C.Structure):
    _fields_ = [("counts", C.c_uint32 * 16), ("seq_len", C.c_uint32)]


class Interval(C.Structure):
    _fields_ = [("l", C.c_int32), ("i", C.c_int32), ("j", C.c_int32), ("m", C.c_int32)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_size_t, C.c_size_t, C.c_void_p)


class Opts(C.Structure):
    _fields_ = [
        ("p_value", C.c_double),
        ("model", C.c_int),
        ("device", C.c_int),
        ("host_threads", C.c_int),
        ("low_memory", C.c_int),
        ("segment", C.c_uint32),
        ("progress", PROGRESS_FN),
        ("ud", C.c_void_p),
        ("sa_on_host", C.c_int),
        ("num_gpus", C.c_int),
        ("devices", C.POINTER(C.c_int)),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("build_ms", C.c_double),
        ("build_launches", C.c_uint64),
        ("scan_ms", C.c_double),
        ("scan_launches", C.c_uint64),
        ("stitch_ms", C.c_double),
        ("stitch_launches", C.c_uint64),
        ("scan_query_nt", C.c_uint64),
        ("scan_pairs", C.c_uint64),
        ("fixups", C.c_uint64),
        ("reference_subjects", C.c_uint64),
        ("sa_ms", C.c_double),
        ("sa_builds", C.c_uint64),
        ("sa_rounds", C.c_uint64),
        ("adaptive_calls", C.c_uint64),
        ("uniform_calls", C.c_uint64),
        ("coop_calls", C.c_uint64),
        ("coop_fallbacks", C.c_uint64),
        ("routed_calls", C.c_uint64),
        ("coop_query_nt", C.c_uint64),
        ("lane_query_nt", C.c_uint64),
        ("pool_calls", C.c_uint64),
    ]


# every symbol include/andi_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "andi_hip_default_opts": (None, [C.POINTER(Opts)]),
    "andi_hip_abi_version": (C.c_int, []),
    "andi_hip_trim": (C.c_size_t, []),
    "andi_hip_pack_symbols": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p]),
    "andi_hip_dist_matrix": (C.c_int, [_P, C.POINTER(Seq), C.c_size_t, C.POINTER(Opts), C.c_char_p, C.c_size_t]),
    "andi_hip_last_gather": (C.c_char_p, []),
    "andi_hip_row_block": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "andi_hip_copy_ceiling": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "andi_hip_subject_prepare": (C.c_int, [C.c_char_p, C.c_size_t, C.c_double, C.POINTER(_P),
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_double),
                                            C.POINTER(C.c_size_t)]),
    "andi_hip_free": (None, [_P]),
    "andi_hip_min_anchor_length": (C.c_size_t, [C.c_double, C.c_double, C.c_size_t]),
    "andi_hip_shustring_cum_prob": (C.c_double, [C.c_size_t, C.c_double, C.c_size_t]),
    "andi_hip_suffix_array": (C.c_int, [_P, _P, C.c_int32]),
    "andi_hip_suffix_sorter": (C.c_char_p, []),
    "andi_hip_model_average": (Model, [C.POINTER(Model), C.POINTER(Model)]),
    "andi_hip_model_coverage": (C.c_double, [C.POINTER(Model)]),
    "andi_hip_estimate": (C.c_double, [C.POINTER(Model), C.c_int]),
    "andi_hip_format_distances": (C.c_size_t, [_P, C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_int,
                                               C.c_int, C.c_int, _P, C.c_size_t, _P, C.c_size_t,
                                               C.POINTER(C.c_int)]),
    "andi_hip_device_count": (C.c_int, []),
    "andi_hip_reload_knobs": (None, []),
    "andi_hip_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_char_p, C.c_size_t]),
    "andi_hip_ctx_destroy": (None, [_P]),
    "andi_hip_ctx_expect_queries": (None, [_P, C.c_size_t]),
    "andi_hip_last_error": (C.c_char_p, [_P]),
    "andi_hip_sync": (C.c_int, [_P]),
    "andi_hip_esa_stage": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    "andi_hip_esa_stage_text": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    "andi_hip_esa_download_sa": (C.c_int, [_P, _P, _P]),
    "andi_hip_esa_build": (C.c_int, [_P, _P]),
    "andi_hip_esa_build_index": (C.c_int, [_P, _P]),
    "andi_hip_esa_build_index_batch": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "andi_hip_esa_flags": (C.c_int, [_P, _P, _P]),
    "andi_hip_esa_download": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "andi_hip_esa_download_index": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int)]),
    "andi_hip_esa_single_form": (C.c_int, [_P]),
    "andi_hip_esa_free": (None, [_P, _P]),
    "andi_hip_esa_bytes": (C.c_size_t, [_P]),
    "andi_hip_queries_stage": (C.c_int, [_P, C.POINTER(Seq), C.c_size_t, C.POINTER(_P)]),
    "andi_hip_queries_free": (None, [_P, _P]),
    "andi_hip_match_positions": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _P]),
    "andi_hip_scan_rows": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64), C.c_size_t, _P, C.c_int,
                                     C.c_uint32, _P]),
    "andi_hip_bootstrap": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, C.c_size_t, _P]),
    "andi_hip_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "andi_hip_dev_free": (None, [_P, _P]),
    "andi_hip_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "andi_hip_timings_get": (C.c_int, [_P, C.POINTER(Timings)]),
    "andi_hip_timings_reset": (None, [_P]),
}

_lib = None


def load():
    """Load libandihip.so; raises AndiHipError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AndiHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C andi_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
        import atexit
        atexit.register(L.andi_hip_trim)  # the arena's retained chunks go back before the interpreter is torn down
    return _lib


def _seq_array(seqs):
    arr = (Seq * len(seqs))()
    for k, s in enumerate(seqs):
        arr[k].seq = s
        arr[k].len = len(s)
    return arr


# ---------------------------------------------------------------- host pieces
def subject_prepare(seq: bytes, p_value=0.025):
    """seq_subject_init (src/sequence.c:210): returns (RS, gc, threshold)."""
    L = load()
    rs, n, gc, thr = _P(), C.c_size_t(), C.c_double(), C.c_size_t()
    if L.andi_hip_subject_prepare(seq, len(seq), p_value, C.byref(rs), C.byref(n), C.byref(gc), C.byref(thr)):
        raise AndiHipError("andi_hip_subject_prepare failed")
    try:
        RS = C.string_at(rs, n.value)
    finally:
        L.andi_hip_free(rs)
    return RS, gc.value, thr.value


def min_anchor_length(p, g, l):
    return load().andi_hip_min_anchor_length(p, g, l)


def shustring_cum_prob(x, p, l):
    return load().andi_hip_shustring_cum_prob(x, p, l)


def suffix_array(text: bytes):
    """divsufsort stand-in (src/esa.c:303)."""
    n = len(text)
    buf = C.create_string_buffer(text, n + 1)
    sa = np.empty(n, dtype=np.int32)
    if load().andi_hip_suffix_array(C.cast(buf, _P), sa.ctypes.data, n):
        raise AndiHipError("andi_hip_suffix_array failed")
    return sa


def _model(counts17):
    m = Model()
    for k in range(16):
        m.counts[k] = int(counts17[k])
    m.seq_len = int(counts17[16])
    return m


def estimate(counts17, model=M_JC):
    return load().andi_hip_estimate(C.byref(_model(counts17)), model)


def coverage(counts17):
    return load().andi_hip_model_coverage(C.byref(_model(counts17)))


def format_distances(M, names, model=M_JC, extra_verbose=False, truncate_names=False, warnings=True):
    """print_distances (src/io.c:246): returns (phylip_text, warning_text, flags)."""
    L = load()
    M = np.ascontiguousarray(M, dtype=np.uint32)
    n = M.shape[0]
    assert M.shape == (n, n, 17)
    cnames = (C.c_char_p * n)(*[s.encode() if isinstance(s, str) else s for s in names])
    cap = 64 + n * (64 + 16 * n)
    warn = C.create_string_buffer(1 << 20)
    flags = C.c_int()
    for _ in range(2):  # the call returns the bytes it needs: long names get a second, exact buffer
        out = C.create_string_buffer(cap)
        need = L.andi_hip_format_distances(M.ctypes.data, cnames, n, model, int(extra_verbose), int(truncate_names),
                                           int(warnings), C.cast(out, _P), cap, C.cast(warn, _P), len(warn),
                                           C.byref(flags))
        if need < cap:
            break
        cap = need + 1
    return out.value.decode(), warn.value.decode(), flags.value


# ---------------------------------------------------------------- device objects
class Context:
    def __init__(self, device=0):
        L = load()
        self._h = _P()
        err = C.create_string_buffer(512)
        if L.andi_hip_ctx_create(C.byref(self._h), device, err, len(err)):
            raise AndiHipError(err.value.decode())
        self.device = device

    def expect_queries(self, queries):
        """queries per subject from now on (decides the depth of the probe tables; results do not depend on it)"""
        load().andi_hip_ctx_expect_queries(self._h, int(queries))

    def _check(self, rc, what):
        if rc:
            raise AndiHipError(f"{what}: {load().andi_hip_last_error(self._h).decode()}")

    def sync(self):
        self._check(load().andi_hip_sync(self._h), "sync")

    def timings(self):
        t = Timings()
        self._check(load().andi_hip_timings_get(self._h, C.byref(t)), "timings")
        return {k: getattr(t, k) for k, _ in Timings._fields_}

    def timings_reset(self):
        load().andi_hip_timings_reset(self._h)

    def alloc(self, nbytes):
        p = _P()
        self._check(load().andi_hip_dev_alloc(self._h, nbytes, C.byref(p)), "dev_alloc")
        return p

    def free(self, p):
        load().andi_hip_dev_free(self._h, p)

    def close(self):
        if self._h:
            load().andi_hip_ctx_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Esa:
    """esa_s (src/esa.h:42) resident in HBM: host prepares RS + SA; the device
    builds the scan index (build="index", default) and/or the reference's own
    arrays LCP, CLD, FVC, 10-mer table (build="reference")."""

    def __init__(self, ctx: Context, seq: bytes, p_value=0.025, sa=None, build="index", prepared=None):
        """sa: None = suffix array by the host sorter; an int32 array = given; "device" = built on the device."""
        self.ctx = ctx
        if prepared is not None:  # (RS, gc, threshold, SA) from prepare_host(), e.g. made in a thread pool
            self.RS, self.gc, self.threshold, sa = prepared
        else:
            self.RS, self.gc, self.threshold = subject_prepare(seq, p_value)
        self.n = len(self.RS)
        self._h = _P()
        L = load()
        if isinstance(sa, str) and sa == "device":
            self._SA = None
            ctx._check(L.andi_hip_esa_stage_text(ctx._h, self.RS, self.n, self.threshold, C.byref(self._h)),
                       "esa_stage_text")
        else:
            self._SA = suffix_array(self.RS) if sa is None else np.ascontiguousarray(sa, dtype=np.int32)
            ctx._check(L.andi_hip_esa_stage(ctx._h, self.RS, self._SA.ctypes.data, self.n, self.threshold,
                                            C.byref(self._h)), "esa_stage")
        self.reference_built = False
        if build in ("index", "both", True):
            self.build()
        if build in ("reference", "both"):
            self.build_reference()

    @property
    def SA(self):
        if self._SA is None:  # built on the device: fetch it
            sa = np.empty(self.n, np.int32)
            self.ctx._check(load().andi_hip_esa_download_sa(self.ctx._h, self._h, sa.ctypes.data), "esa_download_sa")
            self._SA = sa
        return self._SA

    def build(self):
        """scan index (probe table)"""
        self.ctx._check(load().andi_hip_esa_build_index(self.ctx._h, self._h), "esa_build_index")

    def build_reference(self):
        """LCP, CLD, FVC, 10-mer table"""
        self.ctx._check(load().andi_hip_esa_build(self.ctx._h, self._h), "esa_build")
        self.reference_built = True

    def flags(self):
        out = np.zeros(4, np.int32)
        self.ctx._check(load().andi_hip_esa_flags(self.ctx._h, self._h, out.ctypes.data), "esa_flags")
        return out

    def download(self):
        if not self.reference_built:
            self.build_reference()
        n = self.n
        LCP = np.empty(n + 1, np.int32)
        CLD = np.empty(n + 1, np.int32)
        FVC = np.empty(n, np.uint8)
        cache = np.empty((1 << 20, 4), np.int32)
        self.ctx._check(load().andi_hip_esa_download(self.ctx._h, self._h, LCP.ctypes.data, CLD.ctypes.data,
                                                     FVC.ctypes.data, cache.ctypes.data), "esa_download")
        return LCP, CLD, FVC, cache

    def download_index(self):
        """(K, table): the probe table of the scan index, uint32[4^K, 2] (test hook; andi_hip.h has the layout)."""
        K = C.c_int(0)
        self.ctx._check(load().andi_hip_esa_download_index(self.ctx._h, self._h, None, C.byref(K)), "esa_download_index")
        table = np.empty((1 << (2 * K.value), 2), np.uint32)
        self.ctx._check(load().andi_hip_esa_download_index(self.ctx._h, self._h, table.ctypes.data, C.byref(K)),
                        "esa_download_index")
        return K.value, table

    def single_form(self):
        """form of the probe table's entries of K-mers that occur once: 0 plain, 1 extended, 2 short extended"""
        return load().andi_hip_esa_single_form(self._h)

    def nbytes(self):
        return load().andi_hip_esa_bytes(self._h)

    def close(self):
        if self._h and self.ctx._h:
            load().andi_hip_esa_free(self.ctx._h, self._h)
        self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare_host(seq: bytes, p_value=0.025):
    """The host part of esa_init for one subject: RS, gc, threshold and the suffix
    array.  ctypes releases the GIL, so a ThreadPool runs these in parallel (the
    role of the OpenMP subject loop, src/dist_hack.h:46-52)."""
    RS, gc, thr = subject_prepare(seq, p_value)
    return RS, gc, thr, suffix_array(RS)


class Queries:
    def __init__(self, ctx: Context, seqs):
        self.ctx = ctx
        self.seqs = [bytes(s) for s in seqs]
        self._arr = _seq_array(self.seqs)
        self._h = _P()
        ctx._check(load().andi_hip_queries_stage(ctx._h, self._arr, len(self.seqs), C.byref(self._h)),
                   "queries_stage")

    def __len__(self):
        return len(self.seqs)

    def close(self):
        if self._h and self.ctx._h:
            load().andi_hip_queries_free(self.ctx._h, self._h)
        self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match_positions(esa: Esa, queries: Queries, qidx, first, count, cached=True):
    """get_match(_cached) (src/esa.c:615-656) for consecutive suffixes.
    Returns int32 array (count, 4): l, i, j, SA[i]."""
    if not esa.reference_built:
        esa.build_reference()
    out = np.empty((count, 4), np.int32)
    esa.ctx._check(load().andi_hip_match_positions(esa.ctx._h, esa._h, queries._h, qidx, first, count,
                                                   int(cached), out.ctypes.data), "match_positions")
    return out


def build_indexes(ctx: Context, esas):
    """the scan indexes (probe tables) of several subjects in one pair of launches"""
    n = len(esas)
    hs = (_P * n)(*[e._h for e in esas])
    ctx._check(load().andi_hip_esa_build_index_batch(ctx._h, hs, n), "esa_build_index_batch")


def scan_rows_dev(ctx: Context, esas, selfs, queries: Queries, model, segment, dptr):
    n = len(esas)
    hs = (_P * n)(*[e._h for e in esas])
    sf = (C.c_int64 * n)(*[int(s) for s in selfs])
    ctx._check(load().andi_hip_scan_rows(ctx._h, hs, sf, n, queries._h, model, segment, dptr), "scan_rows")


def scan_rows(ctx: Context, esas, selfs, queries: Queries, model=M_JC, segment=0):
    """dist_anchor (src/process.c:141) for every (subject, query): uint32 array
    (nsub, nq, 17) = 16 counts + seq_len."""
    nsub, nq = len(esas), len(queries)
    out = np.empty((nsub, nq, 17), np.uint32)
    d = ctx.alloc(out.nbytes)
    try:
        scan_rows_dev(ctx, esas, selfs, queries, model, segment, d)
        ctx._check(load().andi_hip_copy_to_host(ctx._h, out.ctypes.data, d, out.nbytes), "copy_to_host")
    finally:
        ctx.free(d)
    return out


def device_count():
    return load().andi_hip_device_count()


def pack_symbols(seq: bytes):
    """(4-bit symbols of seq as a numpy uint8 array, whether a byte lies outside the alphabet) -- the host packer of the seam."""
    import numpy as np
    out = np.empty((len(seq) + 1) // 2, np.uint8)
    bad = load().andi_hip_pack_symbols(seq, len(seq), out.ctypes.data_as(C.c_void_p))
    return out, bool(bad)


def trim():
    """Give the device-memory chunks nobody holds a block of back to the driver; returns the bytes released."""
    return int(load().andi_hip_trim())


def reload_knobs():
    """The library reads its ANDI_* environment switches once; read them again (after changing os.environ)."""
    load().andi_hip_reload_knobs()


def last_gather():
    return load().andi_hip_last_gather().decode()


def row_block(total, parts, k):
    """the seam's own tiling of the subject rows over `parts` devices (api.hip: row_block): [first, last) of part k"""
    f, l = C.c_size_t(0), C.c_size_t(0)
    load().andi_hip_row_block(total, parts, k, C.byref(f), C.byref(l))
    return int(f.value), int(l.value)


def copy_ceiling(ctx, nbytes=1 << 30, reps=5):
    """GB/s (read + written) of the engine's 16-byte streaming copy kernel over nbytes: the measured ceiling beside the
    nominal HBM peak (include/andi_hip.h: andi_hip_copy_ceiling)"""
    g = C.c_double(0.0)
    ctx._check(load().andi_hip_copy_ceiling(ctx._h, nbytes, reps, C.byref(g)), "copy_ceiling")
    return float(g.value)


def bootstrap(ctx: Context, M, replicates, seed=0):
    """calculate_bootstrap (src/process.c:289): (replicates, n, n, 17) uint32."""
    M = np.ascontiguousarray(M, dtype=np.uint32)
    n = M.shape[0]
    assert M.shape == (n, n, 17)
    B = np.empty((replicates, n, n, 17), np.uint32)
    ctx._check(load().andi_hip_bootstrap(ctx._h, M.ctypes.data, n, seed, replicates, B.ctypes.data), "bootstrap")
    return B


def dist_matrix(seqs, p_value=0.025, model=M_JC, device=0, host_threads=0, segment=0, num_gpus=1, devices=None,
                low_memory=False, sa_on_host=False):
    """distMatrix (src/dist_hack.h:34): n*n*17 uint32, row = subject.  num_gpus / devices: the rows are
    tiled over several devices (or several contexts on one) behind the same call."""
    L = load()
    seqs = [bytes(s) for s in seqs]
    n = len(seqs)
    arr = _seq_array(seqs)
    o = Opts()
    L.andi_hip_default_opts(C.byref(o))
    o.p_value, o.model, o.device, o.host_threads, o.segment = p_value, model, device, host_threads, segment
    o.low_memory = int(low_memory)
    o.sa_on_host = int(sa_on_host)
    o.num_gpus = num_gpus
    if devices is not None:
        dl = (C.c_int * len(devices))(*devices)
        o.devices, o.num_gpus = dl, len(devices)
    M = np.zeros((n, n, 17), np.uint32)
    err = C.create_string_buffer(512)
    if L.andi_hip_dist_matrix(M.ctypes.data, arr, n, C.byref(o), err, len(err)):
        raise AndiHipError(err.value.decode())
    return M
