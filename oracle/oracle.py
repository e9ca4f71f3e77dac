"""ctypes binding of the CPU ORACLE (oracle/librf_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  The product package (rapidfuzz_rs_amd/) must never import this module.

The Python surface mirrors the reference's (rapidfuzz-rs v0.5.0) names so the known-answer tests read
like the reference's own test modules:  `oracle.levenshtein.distance(a, b, score_cutoff=3)`,
`oracle.levenshtein.BatchComparator(a).distance(b)`, ...   `None` is Rust's `None`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librf_oracle.so")

LEVENSHTEIN, INDEL, LCS_SEQ, JARO, JARO_WINKLER, FUZZ_RATIO, OSA = range(7)
OP_DISTANCE, OP_SIMILARITY, OP_NORMALIZED_DISTANCE, OP_NORMALIZED_SIMILARITY = range(4)
PATH_NAMES = ["none", "eq", "lendiff", "empty", "hyrroe2003", "small_band", "block", "mbleven", "wagner_fischer", "affix"]


class _Weights(C.Structure):
    _fields_ = [("insertion_cost", C.c_size_t), ("deletion_cost", C.c_size_t), ("substitution_cost", C.c_size_t)]


class CallArgs(C.Structure):
    _fields_ = [
        ("has_cutoff", C.c_int),
        ("has_hint", C.c_int),
        ("cutoff_usize", C.c_size_t),
        ("hint_usize", C.c_size_t),
        ("cutoff_f64", C.c_double),
        ("hint_f64", C.c_double),
        ("weights", _Weights),
        ("prefix_weight", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile librf_oracle.so with the committed Makefile (gcc, a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        import fcntl

        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:  # one builder at a time (parallel test workers / ranks)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.run(["make", "-C", _HERE, "librf_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        # RF_ORACLE_LIB: a differently built oracle (tests/test_oracle_sanitize.py runs the known answers on the ASan/UBSan build)
        L = C.CDLL(os.environ.get("RF_ORACLE_LIB") or _LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.rfo_batch_new.restype = C.c_void_p
        L.rfo_batch_new.argtypes = [C.c_int, u8p, C.c_size_t]
        L.rfo_batch_free.argtypes = [C.c_void_p]
        L.rfo_batch_pm.restype = C.POINTER(C.c_uint64)
        L.rfo_batch_pm.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.rfo_batch_usize.argtypes = [C.c_void_p, C.c_int, u8p, C.c_size_t, C.POINTER(CallArgs), C.POINTER(C.c_size_t)]
        L.rfo_batch_f64.argtypes = [C.c_void_p, C.c_int, u8p, C.c_size_t, C.POINTER(CallArgs), C.POINTER(C.c_double)]
        L.rfo_free_usize.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(CallArgs), C.POINTER(C.c_size_t)]
        L.rfo_free_f64.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(CallArgs), C.POINTER(C.c_double)]
        L.rfo_last_lev_path.restype = C.c_int
        L.rfo_last_lcs_q8_edges.restype = C.c_uint
        vp = C.c_void_p
        L.rfo_batch_many_usize.argtypes = [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(CallArgs), vp, C.c_int]
        L.rfo_batch_many_f64.argtypes = [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(CallArgs), vp, C.c_int]
        L.rfo_batch_rows_usize.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(CallArgs), vp, C.c_int]
        L.rfo_batch_rows_f64.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(CallArgs), vp, C.c_int]
        _lib = L
    return _lib


def _as_bytes(s) -> bytes:
    if isinstance(s, str):
        return s.encode("latin-1")  # one byte per element; tests pre-map wider alphabets to bytes
    if isinstance(s, np.ndarray):
        return s.astype(np.uint8, copy=False).tobytes()
    return bytes(s)


def _buf(b: bytes):
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


def make_args(score_cutoff=None, score_hint=None, weights=(1, 1, 1), prefix_weight=0.1, is_float=False) -> CallArgs:
    a = CallArgs()
    a.has_cutoff = int(score_cutoff is not None)
    a.has_hint = int(score_hint is not None)
    if is_float:
        a.cutoff_f64 = float(score_cutoff) if score_cutoff is not None else 0.0
        a.hint_f64 = float(score_hint) if score_hint is not None else 0.0
    else:
        a.cutoff_usize = int(score_cutoff) if score_cutoff is not None else 0
        a.hint_usize = int(score_hint) if score_hint is not None else 0
    a.weights = _Weights(*[int(w) for w in weights])
    a.prefix_weight = float(prefix_weight)
    return a


def last_lcs_q8_edges() -> int:
    """rows of the last banded multi-word LCS call on this thread that met quirk Q8's precondition (rfo_oracle.h)"""
    return int(lib().rfo_last_lcs_q8_edges())


def last_lev_path() -> str:
    return PATH_NAMES[lib().rfo_last_lev_path()]


class _Batch:
    """`<metric>::BatchComparator` of the reference (e.g. src/distance/levenshtein.rs:1636-1818)."""

    METRIC = -1
    FLOAT = False

    def __init__(self, s1):
        self._s1 = _as_bytes(s1)
        self._h = lib().rfo_batch_new(self.METRIC, _buf(self._s1), len(self._s1))

    def __del__(self):
        try:
            if self._h:
                lib().rfo_batch_free(self._h)
                self._h = None
        except Exception:
            pass

    def pm(self) -> np.ndarray:
        n = C.c_size_t()
        p = lib().rfo_batch_pm(self._h, C.byref(n))
        bc = max(n.value, 1)
        return np.ctypeslib.as_array(p, shape=(256 * bc,)).copy().reshape(256, bc)

    def _call(self, op, s2, **kw):
        b = _as_bytes(s2)
        is_f = self.FLOAT or op >= OP_NORMALIZED_DISTANCE
        a = make_args(is_float=is_f, **kw)
        if is_f:
            out = C.c_double()
            some = lib().rfo_batch_f64(self._h, op, _buf(b), len(b), C.byref(a), C.byref(out))
        else:
            out = C.c_size_t()
            some = lib().rfo_batch_usize(self._h, op, _buf(b), len(b), C.byref(a), C.byref(out))
        return out.value if some else None

    def distance(self, s2, **kw):
        return self._call(OP_DISTANCE, s2, **kw)

    def similarity(self, s2, **kw):
        return self._call(OP_SIMILARITY, s2, **kw)

    def normalized_distance(self, s2, **kw):
        return self._call(OP_NORMALIZED_DISTANCE, s2, **kw)

    def normalized_similarity(self, s2, **kw):
        return self._call(OP_NORMALIZED_SIMILARITY, s2, **kw)

    # ---- one-vs-many loops (None -> UINT64_MAX / NaN) ----
    def many(self, op, data: np.ndarray, offsets: np.ndarray, nthreads: int = 1, **kw) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        is_f = self.FLOAT or op >= OP_NORMALIZED_DISTANCE
        a = make_args(is_float=is_f, **kw)
        out = np.empty(n, dtype=np.float64 if is_f else np.uint64)
        fn = lib().rfo_batch_many_f64 if is_f else lib().rfo_batch_many_usize
        fn(self._h, op, data.ctypes.data, offsets.ctypes.data, n, C.byref(a), out.ctypes.data, nthreads)
        return out

    def rows(self, op, rows: np.ndarray, nthreads: int = 1, **kw) -> np.ndarray:
        """rows: uint8 [n, len] (C-contiguous): n candidates of one fixed length."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, ln = rows.shape
        is_f = self.FLOAT or op >= OP_NORMALIZED_DISTANCE
        a = make_args(is_float=is_f, **kw)
        out = np.empty(n, dtype=np.float64 if is_f else np.uint64)
        fn = lib().rfo_batch_rows_f64 if is_f else lib().rfo_batch_rows_usize
        fn(self._h, op, rows.ctypes.data, n, ln, rows.strides[0], C.byref(a), out.ctypes.data, nthreads)
        return out


class _Module:
    """One `rapidfuzz::distance::<metric>` module: free functions + BatchComparator."""

    def __init__(self, metric: int, is_float: bool):
        self.metric, self.is_float = metric, is_float
        self.BatchComparator = type("BatchComparator", (_Batch,), {"METRIC": metric, "FLOAT": is_float})

    def _call(self, op, s1, s2, **kw):
        b1, b2 = _as_bytes(s1), _as_bytes(s2)
        is_f = self.is_float or op >= OP_NORMALIZED_DISTANCE
        a = make_args(is_float=is_f, **kw)
        if is_f:
            out = C.c_double()
            some = lib().rfo_free_f64(self.metric, op, _buf(b1), len(b1), _buf(b2), len(b2), C.byref(a), C.byref(out))
        else:
            out = C.c_size_t()
            some = lib().rfo_free_usize(self.metric, op, _buf(b1), len(b1), _buf(b2), len(b2), C.byref(a), C.byref(out))
        return out.value if some else None

    def distance(self, s1, s2, **kw):
        return self._call(OP_DISTANCE, s1, s2, **kw)

    def similarity(self, s1, s2, **kw):
        return self._call(OP_SIMILARITY, s1, s2, **kw)

    def normalized_distance(self, s1, s2, **kw):
        return self._call(OP_NORMALIZED_DISTANCE, s1, s2, **kw)

    def normalized_similarity(self, s1, s2, **kw):
        return self._call(OP_NORMALIZED_SIMILARITY, s1, s2, **kw)


levenshtein = _Module(LEVENSHTEIN, False)
indel = _Module(INDEL, False)
lcs_seq = _Module(LCS_SEQ, False)
osa = _Module(OSA, False)
jaro = _Module(JARO, True)
jaro_winkler = _Module(JARO_WINKLER, True)


class _Fuzz:
    """`rapidfuzz::fuzz` (src/fuzz.rs:48-150)."""

    @staticmethod
    def ratio(s1, s2, score_cutoff: Optional[float] = None):
        return _Module(FUZZ_RATIO, True)._call(OP_NORMALIZED_SIMILARITY, s1, s2, score_cutoff=score_cutoff)

    class RatioBatchComparator(_Batch):
        METRIC = FUZZ_RATIO
        FLOAT = True

        def similarity(self, s2, **kw):  # the only method the reference gives it (src/fuzz.rs:115-149)
            return self._call(OP_NORMALIZED_SIMILARITY, s2, **kw)


fuzz = _Fuzz()
