"""ctypes binding of librfgpu.so (the C ABI declared in include/rfgpu.h).

PyTorch is plumbing only: importing it first makes the process share ONE HIP runtime (torch ships a
libamdhip64 with the same SONAME the library links against), so torch tensors' device pointers and
streams are directly usable by the kernels.  The library itself has no torch dependency.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RF_LIB") or os.path.join(_HERE, "librfgpu.so")  # RF_LIB: an A/B build of the library (tools/ab.sh)

RF_OK, RF_ERR_INVALID_ARG, RF_ERR_HIP, RF_ERR_UNSUPPORTED, RF_ERR_NO_DEVICE, RF_ERR_OOM = range(6)
STATUS_NAMES = ["RF_OK", "RF_ERR_INVALID_ARG", "RF_ERR_HIP", "RF_ERR_UNSUPPORTED", "RF_ERR_NO_DEVICE", "RF_ERR_OOM"]
LEVENSHTEIN, INDEL, LCS_SEQ, JARO, JARO_WINKLER, FUZZ_RATIO, OSA = range(7)
OP_DISTANCE, OP_SIMILARITY, OP_NORMALIZED_DISTANCE, OP_NORMALIZED_SIMILARITY = range(4)
MEM_HOST, MEM_DEVICE = 0, 1
NO_CUTOFF = 2**64 - 1
NONE_U32 = 0xFFFFFFFF
FLAG_RATIO_INDEL_NORMALIZATION = 0x1


class RfError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else status}: {message}")
        self.status = status


class RfArgs(C.Structure):
    _fields_ = [
        ("cutoff_usize", C.c_uint64),
        ("score_hint_usize", C.c_uint64),
        ("cutoff_f64", C.c_double),
        ("score_hint_f64", C.c_double),
        ("insertion_cost", C.c_uint64),
        ("deletion_cost", C.c_uint64),
        ("substitution_cost", C.c_uint64),
        ("prefix_weight", C.c_double),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class RfHostLayout(C.Structure):
    _fields_ = [
        ("packed", C.POINTER(C.c_uint8)),
        ("tile_off", C.POINTER(C.c_uint64)),
        ("tile_len", C.POINTER(C.c_uint32)),
        ("tile_slot0", C.POINTER(C.c_uint32)),
        ("orig", C.POINTER(C.c_uint32)),
        ("packed_bytes", C.c_uint64),
        ("n_slots", C.c_uint64),
        ("n_tiles", C.c_uint32),
        ("identity", C.c_uint32),
        ("sigma", C.c_uint8 * 256),
        ("n_exact", C.c_uint32),
        ("n_mixed", C.c_uint32),
    ]


# every symbol include/rfgpu.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "rf_args_default", "rf_last_error", "rf_device_count",
    "rf_comparator_new", "rf_comparator_clone", "rf_comparator_free", "rf_comparator_metric",
    "rf_comparator_query_len", "rf_comparator_pm", "rf_comparator_new_u32", "rf_corpus_pack_u32", "rf_corpus_alphabet_size", "rf_corpus_save", "rf_corpus_load", "rf_stream_many_u32", "rf_stream_many_f64", "rf_corpus_file_count",
    "rf_corpus_pack", "rf_corpus_pack_rows_device", "rf_corpus_free", "rf_corpus_layout_host",
    "rf_host_layout_free", "rf_corpus_count", "rf_corpus_payload_bytes", "rf_corpus_device_bytes",
    "rf_corpus_device", "rf_many_u32", "rf_many_f64", "rf_one_u32", "rf_one_f64", "rf_many_multi_u32", "rf_many_multi_f64", "rf_topk_u32", "rf_topk_f64", "rf_topk_keys_device", "rf_topk_merge_keys_device", "rf_topk_merge_u32",
    "rf_probe_issue_rate", "rf_topk_allgather_merge",
    "rf_topk_entries_device", "rf_topk_merge_entries_device", "rf_topk_allgather_merge_entries", "rf_topk_merge_entries",
    "rf_topk_entry_score_u32", "rf_topk_entry_score_f64", "rf_probe_core_clock", "rf_release_caches",
]


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of librfgpu.so (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".hpp", ".inc")) or f == "Makefile"]
    deps.append(os.path.join(_HERE, "..", "include", "rfgpu.h"))
    def is_stale() -> bool:
        return force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps)

    if is_stale():
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            raise RuntimeError("librfgpu.so is stale or missing and hipcc is not available to rebuild it")
        # one builder at a time: the ranks of a multi-GPU launch import the package simultaneously
        import fcntl

        with open(os.path.join(src_dir, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if is_stale():
                r = subprocess.run(["make", "-C", src_dir, "-j4"], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("building librfgpu.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load the extension.  There is no fallback: without the HIP library nothing in this package works."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (first, so that one HIP runtime serves torch and the kernels)
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    vp, u8p, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.rf_last_error.restype = C.c_char_p
    L.rf_device_count.restype = C.c_int
    L.rf_args_default.argtypes = [C.POINTER(RfArgs)]
    L.rf_comparator_new.argtypes = [C.c_int, vp, C.c_size_t, C.POINTER(vp)]
    L.rf_comparator_new_u32.argtypes = [C.c_int, vp, C.c_size_t, C.POINTER(vp)]
    L.rf_comparator_clone.argtypes = [vp, C.POINTER(vp)]
    L.rf_comparator_free.argtypes = [vp]
    L.rf_comparator_metric.argtypes = [vp]
    L.rf_comparator_query_len.argtypes = [vp]
    L.rf_comparator_query_len.restype = C.c_size_t
    L.rf_comparator_pm.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.rf_comparator_pm.restype = u64p
    L.rf_corpus_pack.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.rf_corpus_pack_u32.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.rf_corpus_alphabet_size.restype = C.c_size_t
    L.rf_corpus_alphabet_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.rf_corpus_save.argtypes = [vp, C.c_char_p]
    L.rf_corpus_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rf_stream_many_u32.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(RfArgs), vp, C.c_size_t, C.c_uint64, C.c_int]
    L.rf_stream_many_f64.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(RfArgs), vp, C.c_size_t, C.c_uint64, C.c_int]
    L.rf_corpus_file_count.argtypes = [C.c_char_p, C.POINTER(C.c_size_t)]
    L.rf_release_caches.argtypes = []
    L.rf_corpus_pack_rows_device.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, vp, C.POINTER(vp)]
    L.rf_corpus_free.argtypes = [vp]
    L.rf_corpus_layout_host.argtypes = [vp, vp, C.c_size_t, C.POINTER(RfHostLayout)]
    L.rf_host_layout_free.argtypes = [C.POINTER(RfHostLayout)]
    L.rf_corpus_count.argtypes = [vp]
    L.rf_corpus_count.restype = C.c_size_t
    L.rf_corpus_payload_bytes.argtypes = [vp]
    L.rf_corpus_payload_bytes.restype = C.c_uint64
    L.rf_corpus_device_bytes.argtypes = [vp]
    L.rf_corpus_device_bytes.restype = C.c_uint64
    L.rf_corpus_device.argtypes = [vp]
    L.rf_many_u32.argtypes = [vp, vp, C.c_int, C.POINTER(RfArgs), vp, C.c_int, vp]
    L.rf_many_f64.argtypes = [vp, vp, C.c_int, C.POINTER(RfArgs), vp, C.c_int, vp]
    L.rf_one_u32.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(RfArgs), C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.rf_one_f64.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(RfArgs), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.rf_many_multi_u32.argtypes = [vp, C.c_uint32, vp, C.c_int, C.POINTER(RfArgs), vp, C.c_int, vp]
    L.rf_many_multi_f64.argtypes = [vp, C.c_uint32, vp, C.c_int, C.POINTER(RfArgs), vp, C.c_int, vp]
    L.rf_topk_u32.argtypes = [vp, vp, C.c_int, C.POINTER(RfArgs), C.c_uint32, C.c_uint64, vp, vp, u32p, vp, C.c_int, vp]
    L.rf_topk_f64.argtypes = [vp, vp, C.c_int, C.POINTER(RfArgs), C.c_uint64, C.c_uint64, vp, vp, u64p, vp, C.c_int, vp]
    L.rf_topk_keys_device.argtypes = [vp, vp, C.c_int, C.POINTER(RfArgs), C.c_uint32, C.c_uint32, vp, vp, C.c_int, vp]
    L.rf_topk_merge_keys_device.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_int, vp]
    L.rf_probe_issue_rate.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    L.rf_topk_allgather_merge.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_int, vp]
    L.rf_topk_merge_u32.argtypes = [C.c_int, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, u32p]
    L.rf_topk_entries_device.argtypes = [vp, vp, C.c_int, C.POINTER(RfArgs), C.c_uint64, C.c_uint64, vp, vp]
    L.rf_topk_merge_entries_device.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_int, vp]
    L.rf_topk_allgather_merge_entries.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp, vp, C.c_int, vp]
    L.rf_topk_merge_entries.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
    L.rf_topk_entry_score_u32.argtypes = [C.c_uint64, C.c_int]
    L.rf_topk_entry_score_u32.restype = C.c_uint32
    L.rf_topk_entry_score_f64.argtypes = [C.c_uint64, C.c_int]
    L.rf_topk_entry_score_f64.restype = C.c_double
    L.rf_probe_core_clock.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _lib = L
    return L


def check(status: int) -> None:
    if status != RF_OK:
        raise RfError(status, lib().rf_last_error().decode("utf-8", "replace"))
