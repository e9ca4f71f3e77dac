"""CPU-side checks of the product library (no GPU, no compute): the C-ABI library loads, exports every
symbol include/rfgpu.h declares, builds the same pattern-match table as the oracle, and its host-side
corpus layout round-trips."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth
from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rfgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    L = N.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under rapidfuzz_rs_amd/ may import, include or link it."""
    pkg = os.path.join(ROOT, "rapidfuzz_rs_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="replace").read()
                code = "\n".join(l for l in txt.splitlines() if not l.strip().startswith(("//", "#", "*", "/*", '"""')))
                assert "rf_oracle" not in code and "librf_oracle" not in code and "import oracle" not in code and "from oracle" not in code, f
    import subprocess

    needed = subprocess.run(["readelf", "-d", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in needed


def test_args_default_matches_reference_defaults():
    a = N.RfArgs()
    N.lib().rf_args_default(C.byref(a))
    assert a.cutoff_usize == N.NO_CUTOFF and np.isnan(a.cutoff_f64)
    assert (a.insertion_cost, a.deletion_cost, a.substitution_cost) == (1, 1, 1)  # levenshtein.rs:139-148
    assert a.prefix_weight == 0.1  # jaro_winkler.rs:36


@pytest.mark.parametrize("qlen", [0, 1, 5, 63, 64, 65, 128, 150, 256, 513])
def test_pm_table_equals_oracle(qlen):
    q = bytes((i * 37 + 11) % 256 for i in range(qlen))
    pm = rf.distance.levenshtein.BatchComparator(q).pm()
    ref = o.levenshtein.BatchComparator(q).pm()
    assert pm.shape == ref.shape and (pm == ref).all()


def test_comparator_clone_and_accessors():
    bc = rf.distance.jaro.BatchComparator(b"hello")
    cl = bc.clone()
    assert (bc.pm() == cl.pm()).all()
    assert N.lib().rf_comparator_query_len(cl._h) == 5
    assert N.lib().rf_comparator_metric(cl._h) == N.JARO


def _unpack(lay, n):
    """Invert the chunk-interleaved layout: returns the list of candidates in ORIGINAL order."""
    out = [None] * n
    inv = np.zeros(256, dtype=np.uint8)
    inv[lay["sigma"]] = np.arange(256, dtype=np.uint8)  # the payload stores renamed symbols
    assert sorted(lay["sigma"].tolist()) == list(range(256))
    packed = inv[lay["packed"]]
    for t in range(len(lay["tile_len"])):
        ln, off, slot0 = int(lay["tile_len"][t]), int(lay["tile_off"][t]), int(lay["tile_slot0"][t])
        for lane in range(64):
            idx = slot0 + lane if lay["identity"] else int(lay["orig"][slot0 + lane])
            if lay["identity"] and idx >= n:
                continue
            if idx == 0xFFFFFFFF:
                continue
            b = bytearray()
            for k in range((ln + 15) // 16):
                base = off + (k * 64 + lane) * 16
                b += bytes(packed[base : base + min(16, ln - 16 * k)])
            assert out[idx] is None
            out[idx] = bytes(b)
    return out


@pytest.mark.parametrize("n,max_len", [(0, 0), (1, 0), (1, 5), (200, 0), (300, 40), (1000, 70), (130, 300)])
def test_host_layout_roundtrip(n, max_len):
    data, offsets = synth.ragged_host(n, max_len, seed=n * 31 + max_len)
    lay = rf.host_layout(data, offsets)
    cands = [bytes(data[int(offsets[i]) : int(offsets[i + 1])]) for i in range(n)]
    assert _unpack(lay, n) == cands
    if len(data):  # frequency-rank renaming: the most frequent byte is stored as 0, the next as 1, ...
        hist = np.bincount(data, minlength=256)
        assert hist[np.argsort(lay["sigma"], kind="stable")].tolist() == sorted(hist.tolist(), reverse=True)
    lens = lay["tile_len"].astype(np.int64)
    ne = lay["n_exact"]
    assert (np.diff(lens[:ne]) >= 0).all() and (np.diff(lens[ne:]) >= 0).all()  # exact tiles ascend, then the views of the mixed tiles
    # exact tiles hold whole multiples of 64 per length: no padding lane anywhere in them
    if not lay["identity"]:
        assert (lay["orig"][: ne * 64] != 0xFFFFFFFF).all()
        counts = np.bincount((offsets[1:] - offsets[:-1]).astype(np.int64), minlength=1) if n else np.zeros(1, np.int64)
        assert ne == int((counts // 64).sum()) and lay["n_mixed"] == -(-int((counts % 64).sum()) // 64)


def test_host_layout_distinct_lengths_pack_to_their_payload():
    """VERDICT r1 missing #5: exact-length tiles cost up to 64x on corpora with many distinct lengths.  10 000 candidates of
    10 000 different lengths now share mixed tiles: the packed form stays within 2x of the payload (round 1: 64x)."""
    n = 10_000
    lens = np.arange(1, n + 1, dtype=np.uint64)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    rng = np.random.default_rng(3)
    data = synth.ALNUM[rng.integers(0, 62, size=int(offsets[-1]))]
    lay = rf.host_layout(data, offsets)
    payload = int(offsets[-1])
    assert lay["n_exact"] == 0 and lay["n_mixed"] == -(-n // 64)
    assert len(lay["packed"]) <= 1.02 * payload + 64 * 1024
    assert _unpack(lay, n)[:: 997] == [bytes(data[int(offsets[i]) : int(offsets[i + 1])]) for i in range(0, n, 997)]


def test_host_layout_single_length_is_identity():
    rows = synth.rows_host(150, 64, seed=5)
    lay = rf.host_layout(rows.reshape(-1), np.arange(151, dtype=np.uint64) * 64)
    assert lay["identity"] and len(lay["orig"]) == 0 and list(lay["tile_len"]) == [64, 64, 64]
    assert _unpack(lay, 150) == [bytes(r) for r in rows]


def test_no_device_fails_loudly_not_silently():
    """Without a GPU the scoring entry points must raise -- there is no CPU fallback in the product."""
    if N.lib().rf_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(rf.RfError):
        rf.Corpus.from_list([b"abc"])
    with pytest.raises(rf.RfError):
        rf.distance.levenshtein.distance(b"kitten", b"sitting")


def test_topk_merge_orders_by_score_then_index():
    k = 3
    scores = np.array([1, 4, 9, 1, 2, 7], dtype=np.uint32)
    idx = np.array([50, 3, 8, 7, 99, 1], dtype=np.uint64)
    counts = np.array([3, 3], dtype=np.uint32)
    os_, oi = np.zeros(k, np.uint32), np.zeros(k, np.uint64)
    cnt = C.c_uint32()
    N.check(N.lib().rf_topk_merge_u32(N.OP_DISTANCE, scores.ctypes.data, idx.ctypes.data, counts.ctypes.data, 2, k, os_.ctypes.data, oi.ctypes.data, C.byref(cnt)))
    assert cnt.value == 3 and list(os_) == [1, 1, 2] and list(oi) == [7, 50, 99]


def test_every_environment_knob_is_documented_and_none_changes_results():
    """every getenv() of the shipping library is listed in include/rfgpu.h with its default, and the one switch
    that makes results wrong on purpose (RF_EXP_NOHBM, a measurement aid) exists only in -DRF_EXPERIMENTS builds: neither the
    default build's sources outside that #ifdef nor the built librfgpu.so contain its name."""
    src = os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc")
    hdr = open(os.path.join(ROOT, "include", "rfgpu.h")).read()
    names = set()
    for f in os.listdir(src):
        if f.endswith((".hip", ".hpp")):
            txt = open(os.path.join(src, f)).read()
            txt = re.sub(r"#ifdef RF_EXPERIMENTS.*?#e(?:lse|ndif)", "", txt, flags=re.S)  # measurement builds only
            names |= set(re.findall(r'getenv\("(\w+)"\)', txt)) | set(re.findall(r'env_or\("(\w+)"', txt)) | set(re.findall(r'env_on\("(\w+)"', txt))
    assert "RF_EXP_NOHBM" not in names
    missing = [n for n in sorted(names) if n not in hdr]
    assert not missing, f"environment variables read by the library but not documented in rfgpu.h: {missing}"
    blob = open(N.LIB_PATH, "rb").read()
    assert b"RF_EXP_NOHBM" not in blob, "the shipping librfgpu.so must not contain the wrong-answer measurement switch"


def test_no_exception_can_cross_the_c_abi():
    """Every exported rf_status function is a function-try-block that ends in RF_ABI_CATCH (rf_host.hpp): std::bad_alloc and friends
    become RF_ERR_OOM / RF_ERR_INVALID_ARG + rf_last_error() instead of unwinding into a C or Rust caller."""
    import glob
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rapidfuzz_rs_amd", "csrc")
    seen = 0
    for path in glob.glob(os.path.join(root, "*.hip")):
        lines = open(path).read().split("\n")
        for i, ln in enumerate(lines):
            if re.match(r"^rf_status rf_[a-z_0-9]+\(", ln):
                j = i
                while not lines[j].rstrip().endswith(")"):
                    j += 1
                assert lines[j + 1] == "try {", (path, ln)
                k = j + 2
                while lines[k] != "}":
                    k += 1
                assert lines[k + 1] == "RF_ABI_CATCH", (path, ln)
                seen += 1
    header = open(os.path.join(os.path.dirname(root), "..", "include", "rfgpu.h")).read()
    declared = set(re.findall(r"^rf_status (rf_[a-z_0-9]+)\(", header, flags=re.M))
    assert seen == len(declared) >= 30, (seen, len(declared))


def test_python_mirror_refuses_offsets_that_run_past_the_buffer():
    """rf_corpus_pack trusts its offsets (a C pointer has no length); the Python mirror owns the buffer and refuses bad offsets before the
    packer can read beyond the array (found by a test of this repo whose offsets outran its data: SIGSEGV in the packer)."""
    data = np.zeros(100, dtype=np.uint8)
    for offs in ([0, 50, 101], [0, 60, 40], [200, 200]):
        with pytest.raises(ValueError):
            rf.Corpus.from_ragged(data, np.array(offs, dtype=np.uint64))
        with pytest.raises(ValueError):
            rf.Corpus.from_ragged_u32(data.astype(np.uint32), np.array(offs, dtype=np.uint64))
