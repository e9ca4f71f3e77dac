"""The device-resident packed candidate set (`rf_corpus`), i.e. what the user's
`for candidate in corpus { scorer.distance(candidate) }` loop iterates over in the reference
(rapidfuzz-benches/benches/bench_levenshtein.rs:51-60)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _native as N


def is_wide(s) -> bool:
    """True when `s` needs u32 elements: a str with a code point above 255, or an integer array wider than a byte."""
    if isinstance(s, str):
        return not s.isascii() and any(ord(ch) > 0xFF for ch in s)
    return isinstance(s, np.ndarray) and s.dtype.itemsize > 1


def to_u32(s) -> np.ndarray:
    """One u32 per element: code points of a str (`s.chars()`), bytes widened, integer arrays as they are."""
    if isinstance(s, str):
        return np.frombuffer(s.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
    if isinstance(s, np.ndarray):
        return s.astype(np.uint32, copy=False).reshape(-1)
    return np.frombuffer(bytes(s), dtype=np.uint8).astype(np.uint32)


def _to_bytes(s) -> bytes:
    if isinstance(s, str):
        return s.encode("latin-1")  # u8 elements: one byte per element
    if isinstance(s, np.ndarray):
        return s.astype(np.uint8, copy=False).tobytes()
    return bytes(s)


def ragged(candidates: Iterable) -> tuple[np.ndarray, np.ndarray]:
    """list of byte strings -> (concatenated uint8 data, uint64 offsets[n+1])."""
    bs = [_to_bytes(c) for c in candidates]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    data = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, dtype=np.uint8)
    return data, offsets


def _check_offsets(offsets: np.ndarray, n_elems: int) -> None:
    """The C ABI takes a pointer and trusts the offsets (rf_corpus_pack reads elems[offsets[i] .. offsets[i + 1])); this mirror owns the
    buffer, so it can refuse offsets that run past it instead of letting the packer read beyond the array."""
    if offsets.ndim != 1 or len(offsets) < 1:
        raise ValueError("offsets must be a 1-D array of n + 1 entries")
    if len(offsets) > 1 and bool(np.any(offsets[1:] < offsets[:-1])):
        raise ValueError("offsets must not decrease")
    if int(offsets[-1]) > n_elems or int(offsets[0]) > n_elems:
        raise ValueError(f"offsets run to element {int(offsets[-1])} of a buffer of {n_elems}")


class Corpus:
    """Owns an `rf_corpus*`.  Build with one of the constructors below; results of `*_many` calls always
    come back in the original candidate order."""

    def __init__(self, handle: int, device: int):
        self._h = handle
        self.device = device

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().rf_corpus_free(h)
            except Exception:
                pass

    def __len__(self) -> int:
        return N.lib().rf_corpus_count(self._h)

    @property
    def payload_bytes(self) -> int:
        return N.lib().rf_corpus_payload_bytes(self._h)

    @property
    def device_bytes(self) -> int:
        return N.lib().rf_corpus_device_bytes(self._h)

    @classmethod
    def from_ragged(cls, data: np.ndarray, offsets: np.ndarray, device: int = 0) -> "Corpus":
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check_offsets(offsets, len(data))
        h = C.c_void_p()
        N.check(N.lib().rf_corpus_pack(data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, device, C.byref(h)))
        return cls(h.value, device)

    @classmethod
    def from_list(cls, candidates: Sequence, device: int = 0) -> "Corpus":
        """bytes / str / uint8 arrays; if any candidate needs u32 elements (a `char` above 255) the corpus is packed
        over u32 elements with its own alphabet (rf_corpus_pack_u32)."""
        candidates = list(candidates)
        if any(is_wide(c) for c in candidates):
            return cls.from_u32_list(candidates, device=device)
        return cls.from_ragged(*ragged(candidates), device=device)

    @classmethod
    def from_u32_list(cls, candidates: Sequence, device: int = 0) -> "Corpus":
        parts = [to_u32(c) for c in candidates]
        offsets = np.zeros(len(parts) + 1, dtype=np.uint64)
        if parts:
            offsets[1:] = np.cumsum([len(x) for x in parts], dtype=np.uint64)
        data = np.concatenate(parts).astype(np.uint32) if parts else np.zeros(0, dtype=np.uint32)
        return cls.from_ragged_u32(data, offsets, device=device)

    @classmethod
    def from_ragged_u32(cls, data: np.ndarray, offsets: np.ndarray, device: int = 0) -> "Corpus":
        data = np.ascontiguousarray(data, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check_offsets(offsets, len(data))
        h = C.c_void_p()
        N.check(N.lib().rf_corpus_pack_u32(data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, device, C.byref(h)))
        return cls(h.value, device)

    @property
    def slot_count(self) -> int:
        """entries of a slot-ordered result vector (`Args.slot_order()`): len(self) for a single-length corpus, 64 per tile otherwise."""
        return N.lib().rf_corpus_slot_count(self._h)

    def slot_index(self) -> np.ndarray:
        """uint32[slot_count]: the original candidate index of every slot, 0xFFFFFFFF where a slot holds no candidate (rf_corpus_slot_index)."""
        out = np.empty(self.slot_count, dtype=np.uint32)
        N.check(N.lib().rf_corpus_slot_index(self._h, out.ctypes.data, N.MEM_HOST))
        return out

    def save(self, path: str) -> None:
        """Write the packed form to `path` (rf_corpus_save); `Corpus.load` maps it back without re-packing and
        `BatchComparator.stream_many` scans it segment by segment when it does not fit in HBM."""
        N.check(N.lib().rf_corpus_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path: str, device: int = 0) -> "Corpus":
        h = C.c_void_p()
        N.check(N.lib().rf_corpus_load(str(path).encode(), device, C.byref(h)))
        return cls(h.value, device)

    def alphabet_size(self) -> tuple[int, int]:
        """(symbols with an id of their own, symbols sharing the overflow id); (256, 0) for a byte corpus."""
        ov = C.c_size_t()
        return int(N.lib().rf_corpus_alphabet_size(self._h, C.byref(ov))), int(ov.value)

    @classmethod
    def from_rows(cls, rows: np.ndarray, device: int = 0) -> "Corpus":
        """host uint8 [n, len]: n candidates of one length."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, ln = rows.shape
        offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(ln)
        return cls.from_ragged(rows.reshape(-1), offsets, device=device)

    @classmethod
    def from_device_rows(cls, rows, stream: Optional[int] = None) -> "Corpus":
        """torch uint8 CUDA tensor [n, len] (row stride >= len, unit column stride): packed on the device."""
        import torch

        assert rows.is_cuda and rows.dtype == torch.uint8 and rows.dim() == 2 and (rows.shape[1] == 0 or rows.stride(1) == 1)
        dev = rows.device.index if rows.device.index is not None else torch.cuda.current_device()
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        h = C.c_void_p()
        N.check(N.lib().rf_corpus_pack_rows_device(rows.data_ptr(), rows.shape[0], rows.shape[1], rows.stride(0) if rows.shape[0] > 1 else max(rows.shape[1], 1), dev, st, C.byref(h)))
        return cls(h.value, dev)


def host_layout(data: np.ndarray, offsets: np.ndarray) -> dict:
    """The packed layout computed on the host only (no GPU): for the packer's CPU tests."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lay = N.RfHostLayout()
    N.check(N.lib().rf_corpus_layout_host(data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(lay)))
    try:
        nt = lay.n_tiles
        out = {
            "packed": np.ctypeslib.as_array(lay.packed, shape=(max(lay.packed_bytes, 1),))[: lay.packed_bytes].copy(),
            "tile_off": np.ctypeslib.as_array(lay.tile_off, shape=(max(nt, 1),))[:nt].copy(),
            "tile_len": np.ctypeslib.as_array(lay.tile_len, shape=(max(nt, 1),))[:nt].copy(),
            "tile_slot0": np.ctypeslib.as_array(lay.tile_slot0, shape=(max(nt, 1),))[:nt].copy(),
            "orig": np.ctypeslib.as_array(lay.orig, shape=(max(lay.n_slots, 1),))[: lay.n_slots].copy(),
            "identity": bool(lay.identity),
            "sigma": np.frombuffer(bytes(lay.sigma), dtype=np.uint8).copy(),
            "n_exact": int(lay.n_exact),
            "n_mixed": int(lay.n_mixed),
        }
    finally:
        N.lib().rf_host_layout_free(C.byref(lay))
    return out
