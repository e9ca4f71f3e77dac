"""Host-side mirror of the reference's `BatchComparator` types and `Args` builders over the C ABI.

Names, argument meaning and `None` behaviour follow rapidfuzz-rs v0.5.0
(e.g. src/distance/levenshtein.rs:86-148 `Args`/`WeightTable`, :1636-1818 `BatchComparator`), so code and
tests written against the reference read the same here.  Every score is computed by the HIP kernels --
a single-candidate call is a one-candidate corpus -- and the module raises if the extension is missing.
"""
from __future__ import annotations

import copy
import ctypes as C
import math
from typing import NamedTuple, Optional, Sequence

import numpy as np

from . import _native as N
from .corpus import Corpus, _to_bytes, is_wide, to_u32


class WeightTable(NamedTuple):
    """src/distance/levenshtein.rs:128-148"""

    insertion_cost: int = 1
    deletion_cost: int = 1
    substitution_cost: int = 1


class Args:
    """`Args::default().score_cutoff(x).score_hint(y).weights(&w).prefix_weight(p)` (levenshtein.rs:86-126,
    jaro_winkler.rs:25-62).  Builder methods return a new object, like the Rust typestate builders."""

    def __init__(self):
        self._cutoff = None
        self._hint = None
        self._weights = WeightTable()
        self._prefix_weight = 0.1
        self._flags = 0

    def _with(self, **kw) -> "Args":
        a = copy.copy(self)
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def score_cutoff(self, v) -> "Args":
        return self._with(_cutoff=v)

    def score_hint(self, v) -> "Args":
        return self._with(_hint=v)

    def weights(self, w) -> "Args":
        return self._with(_weights=WeightTable(*w))

    def prefix_weight(self, p: float) -> "Args":
        return self._with(_prefix_weight=float(p))

    def ratio_indel_normalization(self, on: bool = True) -> "Args":
        """RatioBatchComparator only: the documented Indel ratio instead of the fuzz.rs:141 behaviour."""
        return self._with(_flags=(self._flags | N.FLAG_RATIO_INDEL_NORMALIZATION) if on else (self._flags & ~N.FLAG_RATIO_INDEL_NORMALIZATION))

    def slot_order(self, on: bool = True) -> "Args":
        """`*_many` only: results in the corpus' SLOT order (`corpus.slot_count` entries, `corpus.slot_index()` maps them back) --
        a length-bucketed corpus then skips its gather pass (RF_FLAG_SLOT_ORDER)."""
        return self._with(_flags=(self._flags | N.FLAG_SLOT_ORDER) if on else (self._flags & ~N.FLAG_SLOT_ORDER))

    def to_c(self, is_float: bool) -> N.RfArgs:
        a = N.RfArgs()
        N.lib().rf_args_default(C.byref(a))
        if self._cutoff is not None:
            if is_float:
                a.cutoff_f64 = float(self._cutoff)
            else:
                a.cutoff_usize = min(int(self._cutoff), N.NO_CUTOFF - 1)
        if self._hint is not None:
            if is_float:
                a.score_hint_f64 = float(self._hint)
            else:
                a.score_hint_usize = min(int(self._hint), N.NO_CUTOFF - 1)
        a.insertion_cost, a.deletion_cost, a.substitution_cost = (int(x) for x in self._weights)
        a.prefix_weight = self._prefix_weight
        a.flags = self._flags
        return a


def _mk_args(args: Optional[Args], score_cutoff, score_hint, weights, prefix_weight) -> Args:
    a = args if args is not None else Args()
    if score_cutoff is not None:
        a = a.score_cutoff(score_cutoff)
    if score_hint is not None:
        a = a.score_hint(score_hint)
    if weights is not None:
        a = a.weights(weights)
    if prefix_weight is not None:
        a = a.prefix_weight(prefix_weight)
    return a


def default_device() -> int:
    try:
        import torch

        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return 0


class BatchComparator:
    """`<metric>::BatchComparator::new(s1)`: owns a copy of the query and its pattern-match table."""

    METRIC = -1
    FLOAT = False  # jaro / jaro_winkler / ratio: every method is f64-valued

    def __init__(self, s1):
        h = C.c_void_p()
        self._wide = is_wide(s1)
        if self._wide:  # BatchComparator::new(s1.chars()) with symbols above 255: u32 elements
            self._s1 = to_u32(s1).copy()
            N.check(N.lib().rf_comparator_new_u32(self.METRIC, self._s1.ctypes.data, len(self._s1), C.byref(h)))
        else:
            self._s1 = _to_bytes(s1)
            buf = (C.c_uint8 * max(1, len(self._s1))).from_buffer_copy(self._s1 or b"\0")
            N.check(N.lib().rf_comparator_new(self.METRIC, buf, len(self._s1), C.byref(h)))
        self._h = h.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().rf_comparator_free(h)
            except Exception:
                pass

    def clone(self) -> "BatchComparator":  # #[derive(Clone)]
        return type(self)(self._s1)

    def pm(self) -> np.ndarray:
        """The cached BlockPatternMatchVector as uint64 [256, max(1, ceil(len/64))]."""
        n = C.c_size_t()
        p = N.lib().rf_comparator_pm(self._h, C.byref(n))
        w = max(1, n.value)
        return np.ctypeslib.as_array(p, shape=(256 * w,)).copy().reshape(256, w)

    # ------------------------------------------------------------------ one-vs-many (the GPU path)
    def many(self, op: int, corpus: Corpus, args: Optional[Args] = None, out=None, stream=None, *, score_cutoff=None,
             score_hint=None, weights=None, prefix_weight=None):
        """scores of every candidate of `corpus`, original order.  Returns a numpy array (uint32 with
        0xFFFFFFFF = None, or float64 with NaN = None); pass a CUDA torch tensor as `out` to keep the result
        on the device (the call is then asynchronous on `stream` / torch's current stream).  With host results the
        call runs on `stream` if given, else on the NULL stream (torch's current stream is NOT consulted: host threads
        that want to overlap pass their own `stream=torch.cuda.Stream().cuda_stream`)."""
        a = _mk_args(args, score_cutoff, score_hint, weights, prefix_weight)
        is_f = self.FLOAT or op >= N.OP_NORMALIZED_DISTANCE
        ca = a.to_c(is_f)
        n = corpus.slot_count if (a._flags & N.FLAG_SLOT_ORDER) else len(corpus)
        fn = N.lib().rf_many_f64 if is_f else N.lib().rf_many_u32
        if out is not None and hasattr(out, "is_cuda") and out.is_cuda:
            import torch

            want = torch.float64 if is_f else (torch.uint32 if out.dtype == torch.uint32 else torch.int32)
            assert out.dtype == want and out.numel() >= n and out.is_contiguous(), "out tensor has the wrong dtype/shape"
            st = stream if stream is not None else torch.cuda.current_stream(out.device).cuda_stream
            N.check(fn(self._h, corpus._h, op, C.byref(ca), out.data_ptr(), N.MEM_DEVICE, st))
            return out
        res = np.empty(n, dtype=np.float64 if is_f else np.uint32) if out is None else out
        N.check(fn(self._h, corpus._h, op, C.byref(ca), res.ctypes.data, N.MEM_HOST, stream))
        return res

    @classmethod
    def many_multi(cls, comparators: Sequence["BatchComparator"], op: int, corpus: Corpus, args: Optional[Args] = None, out=None,
                   stream=None, *, score_cutoff=None, score_hint=None, weights=None, prefix_weight=None):
        """[len(comparators), len(corpus)] scores: row j is `comparators[j].many(op, corpus, ...)` (rf_many_multi_*).
        Queries of <= 64 symbols are fused 4 at a time into one pass over the corpus."""
        a = _mk_args(args, score_cutoff, score_hint, weights, prefix_weight)
        is_f = cls.FLOAT or op >= N.OP_NORMALIZED_DISTANCE
        ca = a.to_c(is_f)
        q, n = len(comparators), len(corpus)
        hs = (C.c_void_p * max(q, 1))(*[c._h for c in comparators])
        fn = N.lib().rf_many_multi_f64 if is_f else N.lib().rf_many_multi_u32
        if out is not None and hasattr(out, "is_cuda") and out.is_cuda:
            import torch

            want = torch.float64 if is_f else (torch.uint32 if out.dtype == torch.uint32 else torch.int32)
            assert out.dtype == want and out.numel() >= q * n and out.is_contiguous(), "out tensor has the wrong dtype/shape"
            st = stream if stream is not None else torch.cuda.current_stream(out.device).cuda_stream
            N.check(fn(hs, q, corpus._h, op, C.byref(ca), out.data_ptr(), N.MEM_DEVICE, st))
            return out
        res = np.empty((q, n), dtype=np.float64 if is_f else np.uint32) if out is None else out
        N.check(fn(hs, q, corpus._h, op, C.byref(ca), res.ctypes.data, N.MEM_HOST, stream))
        return res

    def stream_many(self, op: int, path: str, n: Optional[int] = None, args: Optional[Args] = None, segment_bytes: int = 0, device: Optional[int] = None, *,
                    score_cutoff=None, score_hint=None, weights=None, prefix_weight=None) -> np.ndarray:
        """`many` over a corpus FILE written by `Corpus.save` that need not fit in HBM (rf_stream_many_*): scanned in
        segments of `segment_bytes` of payload, uploads overlapped with scans.  The number of candidates comes from the
        file itself (`n`, if given, must agree with it)."""
        a = _mk_args(args, score_cutoff, score_hint, weights, prefix_weight)
        is_f = self.FLOAT or op >= N.OP_NORMALIZED_DISTANCE
        ca = a.to_c(is_f)
        cnt = C.c_size_t()
        N.check(N.lib().rf_corpus_file_count(str(path).encode(), C.byref(cnt)))
        if n is not None and n != cnt.value:
            raise ValueError(f"{path} holds {cnt.value} candidates, not {n}")
        res = np.empty(cnt.value, dtype=np.float64 if is_f else np.uint32)
        fn = N.lib().rf_stream_many_f64 if is_f else N.lib().rf_stream_many_u32
        N.check(fn(self._h, str(path).encode(), op, C.byref(ca), res.ctypes.data, res.size, int(segment_bytes),
                   default_device() if device is None else device))
        return res

    def filter_many(self, op: int, corpus: Corpus, args: Optional[Args] = None, capacity: Optional[int] = None, order: int = N.FILTER_BY_INDEX, index_base: int = 0,
                    device_out: bool = False, stream=None, *, score_cutoff=None, score_hint=None, weights=None, prefix_weight=None):
        """The reference user's `corpus.iter().enumerate().filter_map(|(i, c)| scorer.<op>_with_args(c, &args).map(|v| (i, v)))` (rf_filter_u32 / rf_filter_f64):
        (indices uint64[m], scores[m]) of the candidates whose result is not None, `order` = FILTER_BY_INDEX / FILTER_BY_SCORE / FILTER_ANY.  `capacity` bounds
        the arrays; None = grow until everything fits (the call reports the true count, so at most one repeat).  With `device_out` the arrays are CUDA tensors."""
        a = _mk_args(args, score_cutoff, score_hint, weights, prefix_weight)
        is_f = self.FLOAT or op >= N.OP_NORMALIZED_DISTANCE
        ca = a.to_c(is_f)
        fn = N.lib().rf_filter_f64 if is_f else N.lib().rf_filter_u32
        cap = int(capacity) if capacity is not None else max(1024, len(corpus) // 64)
        while True:
            cnt = C.c_uint64()
            if device_out:
                import torch

                dev = torch.device("cuda", corpus.device)
                idx = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
                sc = torch.empty(max(cap, 1), dtype=torch.float64 if is_f else torch.int32, device=dev)
                st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
                N.check(fn(self._h, corpus._h, op, C.byref(ca), index_base, cap, idx.data_ptr(), sc.data_ptr(), C.byref(cnt), N.MEM_DEVICE, order, st))
            else:
                idx = np.empty(max(cap, 1), dtype=np.uint64)
                sc = np.empty(max(cap, 1), dtype=np.float64 if is_f else np.uint32)
                N.check(fn(self._h, corpus._h, op, C.byref(ca), index_base, cap, idx.ctypes.data, sc.ctypes.data, C.byref(cnt), N.MEM_HOST, order, stream))
            if cnt.value <= cap or capacity is not None:
                m = min(cnt.value, cap)
                self.last_filter_count = int(cnt.value)  # (the true number of candidates within the cutoff, also when `capacity` was too small)
                return idx[:m], sc[:m]
            cap = int(cnt.value)

    def distance_many(self, corpus, args=None, **kw):
        return self.many(N.OP_DISTANCE, corpus, args, **kw)

    def similarity_many(self, corpus, args=None, **kw):
        return self.many(N.OP_SIMILARITY, corpus, args, **kw)

    def normalized_distance_many(self, corpus, args=None, **kw):
        return self.many(N.OP_NORMALIZED_DISTANCE, corpus, args, **kw)

    def normalized_similarity_many(self, corpus, args=None, **kw):
        return self.many(N.OP_NORMALIZED_SIMILARITY, corpus, args, **kw)

    def topk(self, corpus: Corpus, k: int, op: Optional[int] = None, args: Optional[Args] = None, index_base: int = 0,
             out=None, stream=None, *, score_cutoff=None, score_hint=None, weights=None, prefix_weight=None):
        """(scores[m], indices uint64[m]), m <= k, ordered by (score, index) -- best first (ascending for the distance ops,
        descending for the similarity ops); see rf_topk_u32 / rf_topk_f64.  Scores are uint32 for distance / similarity of
        the usize metrics and float64 for everything else (jaro, jaro_winkler, ratio, normalized_*).  Any k >= 1.  `out` (a
        CUDA tensor or numpy array of len(corpus) scores) additionally receives every candidate's score from the same pass."""
        if op is None:
            op = N.OP_SIMILARITY if self.FLOAT else N.OP_DISTANCE
        a = _mk_args(args, score_cutoff, score_hint, weights, prefix_weight)
        is_f = self.FLOAT or op >= N.OP_NORMALIZED_DISTANCE
        ca = a.to_c(is_f)
        scores = np.empty(k, dtype=np.float64 if is_f else np.uint32)
        idx = np.empty(k, dtype=np.uint64)
        cnt = C.c_uint64() if is_f else C.c_uint32()
        out_ptr, out_mem = None, N.MEM_HOST
        if out is not None:
            if hasattr(out, "is_cuda") and out.is_cuda:
                import torch

                assert out.numel() >= len(corpus) and out.is_contiguous() and out.element_size() == (8 if is_f else 4)
                out_ptr, out_mem = out.data_ptr(), N.MEM_DEVICE
                if stream is None:
                    stream = torch.cuda.current_stream(out.device).cuda_stream
            else:
                assert out.dtype == (np.float64 if is_f else np.uint32) and out.size >= len(corpus)
                out_ptr = out.ctypes.data
        fn = N.lib().rf_topk_f64 if is_f else N.lib().rf_topk_u32
        N.check(fn(self._h, corpus._h, op, C.byref(ca), k, index_base, scores.ctypes.data, idx.ctypes.data, C.byref(cnt), out_ptr, out_mem, stream))
        return scores[: cnt.value], idx[: cnt.value]

    def topk_keys_device(self, corpus: Corpus, k: int, keys_out, op: int = N.OP_DISTANCE, args: Optional[Args] = None,
                         index_base: int = 0, out=None, stream=None, **kw):
        """Asynchronous top-k that never leaves the device: fills `keys_out` (CUDA int64 tensor, >= k entries)
        with (score << 32 | index_base + index) keys, best first, -1 = empty (rf_topk_keys_device)."""
        import torch

        a = _mk_args(args, kw.get("score_cutoff"), kw.get("score_hint"), kw.get("weights"), kw.get("prefix_weight"))
        ca = a.to_c(False)
        assert keys_out.is_cuda and keys_out.dtype == torch.int64 and keys_out.numel() >= k and keys_out.is_contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(keys_out.device).cuda_stream
        out_ptr, out_mem = (None, N.MEM_HOST) if out is None else (out.data_ptr(), N.MEM_DEVICE)
        N.check(N.lib().rf_topk_keys_device(self._h, corpus._h, op, C.byref(ca), k, index_base, keys_out.data_ptr(), out_ptr, out_mem, st))
        return keys_out

    def topk_entries_device(self, corpus: Corpus, k: int, entries_out, op: int = N.OP_DISTANCE, args: Optional[Args] = None,
                            index_base: int = 0, stream=None, **kw):
        """Top-k of ANY metric / op / k as 16-byte entries on the device (rf_topk_entries_device): `entries_out` is a CUDA int64
        tensor of shape [>= k, 2] -- (order-preserving key, index_base + index), best first, (-1, -1) = empty; entries of
        different shards merge by (key, index) (parallel.merge_entries_device / decode_entries)."""
        import torch

        a = _mk_args(args, kw.get("score_cutoff"), kw.get("score_hint"), kw.get("weights"), kw.get("prefix_weight"))
        is_f = self.FLOAT or op >= N.OP_NORMALIZED_DISTANCE
        ca = a.to_c(is_f)
        assert entries_out.is_cuda and entries_out.dtype == torch.int64 and entries_out.is_contiguous() and entries_out.numel() >= 2 * k
        st = stream if stream is not None else torch.cuda.current_stream(entries_out.device).cuda_stream
        N.check(N.lib().rf_topk_entries_device(self._h, corpus._h, op, C.byref(ca), k, index_base, entries_out.data_ptr(), st))
        return entries_out

    # ------------------------------------------------------------------ the reference's per-candidate methods
    def _one(self, op: int, s2, args, kw):
        # a u32 query needs the candidate in a u32 corpus even when the candidate itself is plain ASCII
        corpus = (Corpus.from_u32_list if self._wide else Corpus.from_list)([s2], device=default_device())
        r = self.many(op, corpus, args, **kw)[0]
        if r.dtype == np.uint32:
            return None if int(r) == N.NONE_U32 else int(r)
        return None if math.isnan(float(r)) else float(r)

    def distance(self, s2, args=None, **kw):
        return self._one(N.OP_DISTANCE, s2, args, kw)

    def similarity(self, s2, args=None, **kw):
        return self._one(N.OP_SIMILARITY, s2, args, kw)

    def normalized_distance(self, s2, args=None, **kw):
        return self._one(N.OP_NORMALIZED_DISTANCE, s2, args, kw)

    def normalized_similarity(self, s2, args=None, **kw):
        return self._one(N.OP_NORMALIZED_SIMILARITY, s2, args, kw)

    # the Rust spellings
    distance_with_args = distance
    similarity_with_args = similarity
    normalized_distance_with_args = normalized_distance
    normalized_similarity_with_args = normalized_similarity


class MetricModule:
    """One `rapidfuzz::distance::<metric>` module: `Args`, `BatchComparator` and the free functions.
    The free functions evaluate `BatchComparator(s1).<op>(s2)` on the device: in the reference both forms
    return the same value (its tests assert exactly that, levenshtein.rs:1847-1875)."""

    def __init__(self, name: str, metric: int, is_float: bool):
        self.__name__ = name
        self.Args = Args
        self.WeightTable = WeightTable
        self.BatchComparator = type("BatchComparator", (BatchComparator,), {"METRIC": metric, "FLOAT": is_float, "__doc__": BatchComparator.__doc__})

    def distance(self, s1, s2, args=None, **kw):
        return self.BatchComparator(s1).distance(s2, args, **kw)

    def similarity(self, s1, s2, args=None, **kw):
        return self.BatchComparator(s1).similarity(s2, args, **kw)

    def normalized_distance(self, s1, s2, args=None, **kw):
        return self.BatchComparator(s1).normalized_distance(s2, args, **kw)

    def normalized_similarity(self, s1, s2, args=None, **kw):
        return self.BatchComparator(s1).normalized_similarity(s2, args, **kw)

    distance_with_args = distance
    similarity_with_args = similarity
    normalized_distance_with_args = normalized_distance
    normalized_similarity_with_args = normalized_similarity
