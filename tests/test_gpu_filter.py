"""GPU parity tests (`-m gpu`) of round 6: the lane compaction of the head-plane cutoff scans on corpora that share prefixes with the
query, rf_filter_u32 / rf_filter_f64 (the reference user's filter_map over Option<T>: src/common.rs:18-46, :83-85) and RF_FLAG_SLOT_ORDER.
The HIP path through the C ABI against the CPU oracle on the same seeded inputs; nothing here reads /root/reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth
from oracle import oracle as o

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NONE32 = np.uint32(0xFFFFFFFF)
U64MAX = np.uint64(0xFFFFFFFFFFFFFFFF)
GPU = {"levenshtein": rf.distance.levenshtein, "indel": rf.distance.indel, "lcs_seq": rf.distance.lcs_seq, "jaro": rf.distance.jaro, "jaro_winkler": rf.distance.jaro_winkler, "osa": rf.distance.osa}
ORA = {"levenshtein": o.levenshtein, "indel": o.indel, "lcs_seq": o.lcs_seq, "jaro": o.jaro, "jaro_winkler": o.jaro_winkler, "osa": o.osa}


def _u32(x):
    return np.where(x == U64MAX, NONE32, x.astype(np.uint32))


def _same(got, exp):
    if got.dtype == np.uint32:
        return np.nonzero(got != _u32(exp))[0]
    return np.nonzero(~((got == exp) | (np.isnan(got) & np.isnan(exp))))[0]


def _some(exp):
    """(indices, values) of the oracle's Somes"""
    if exp.dtype == np.float64:
        keep = np.nonzero(~np.isnan(exp))[0]
        return keep.astype(np.uint64), exp[keep]
    keep = np.nonzero(exp != U64MAX)[0]
    return keep.astype(np.uint64), exp[keep].astype(np.uint32)


def _prefix_corpus(n, ln, share, seed, qlen=64):
    """n rows of `ln` alphanumerics; a fraction `share` start with the query's first 8..12 symbols (the rest random); ~400 planted near-duplicates whose edits
    include insertions and deletions (which shift the tail against the query) -- alone in their tiles or among the prefix sharers"""
    rng = np.random.default_rng(seed)
    q = synth.ALNUM[rng.integers(0, 62, size=qlen)]
    rows = synth.ALNUM[rng.integers(0, 62, size=(n, ln))]
    synth.head_share_rows_host(rows, q.tobytes(), share, seed + 1)
    other = np.uint8(126)
    for j, idx in enumerate(rng.choice(n, size=400, replace=False)):
        kind = j % 10
        base = np.resize(q, ln + 4)
        if kind <= 4:  # substitutions anywhere
            row = base[:ln].copy()
            row[rng.choice(ln, size=kind, replace=False)] = other
        elif kind <= 6:  # d deletions inside the first 12 symbols
            d = kind - 4
            keep = np.ones(ln + 4, dtype=bool)
            keep[rng.choice(12, size=d, replace=False)] = False
            row = base[keep][:ln]
        elif kind <= 8:  # d insertions inside the first 12 symbols
            d = kind - 6
            row = base.copy()
            for p in sorted(rng.choice(12, size=d, replace=False)):
                row = np.concatenate([row[:p], [other], row[p:]])
            row = row[:ln]
        else:  # a transposition in the head and one at the end
            row = base[:ln].copy()
            row[[2, 3]] = row[[3, 2]]
            row[[ln - 2, ln - 1]] = row[[ln - 1, ln - 2]]
        rows[idx] = row
    return q.tobytes(), rows


@pytest.mark.parametrize("share", [0.01, 0.2])
def test_prefix_sharing_corpora_through_the_lane_compaction(share):
    """2.1 M candidates (32 k tiles: the head plane, the band prefilter and the lane compaction all apply) of which 1 % / 20 % carry the
    query's first 8..12 symbols -- at 1 % nearly half of the TILES hold a survivor of the first pass, at 20 % all of them.  Every value of cutoffs 0..6, Levenshtein
    and OSA, u32 and normalized f64 results, the top-16, query lengths 64 / 60 / 30, against the oracle; then the same values with the second pass over surviving
    tiles (RF_LANE_COMPACT=0, round 5's path) in a child process."""
    if os.environ.get("RF_TEST_LANE_CHILD") is None and share == 0.01:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "test_prefix_sharing_corpora_through_the_lane_compaction"],
                           capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RF_TEST_LANE_CHILD="1", RF_LANE_COMPACT="0"))
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:]
    n = 2_100_037  # (odd tile count, partial last tile)
    for qlen, ln in ((64, 64), (60, 64), (30, 32)):
        q, rows = _prefix_corpus(n, ln, share, seed=int(share * 1000) + qlen, qlen=qlen)
        corpus = rf.Corpus.from_rows(rows)
        for metric in ("levenshtein", "osa"):
            bc, ob = GPU[metric].BatchComparator(q), ORA[metric].BatchComparator(q)
            full = ob.rows(N.OP_DISTANCE, rows, nthreads=8)
            for cutoff in (0, 1, 2, 3, 4, 5, 6):
                got = bc.distance_many(corpus, score_cutoff=cutoff)
                exp = np.where(full <= np.uint64(cutoff), full, U64MAX)
                bad = _same(got, exp)
                assert len(bad) == 0, (metric, qlen, cutoff, bad[:5], got[bad[:5]], exp[bad[:5]])
            for ncut in (0.05, 0.95):
                op = N.OP_NORMALIZED_DISTANCE if ncut < 0.5 else N.OP_NORMALIZED_SIMILARITY
                got = bc.many(op, corpus, score_cutoff=ncut)
                exp = ob.rows(op, rows, nthreads=8, score_cutoff=ncut)
                bad = _same(got, exp)
                assert len(bad) == 0, (metric, qlen, ncut, bad[:5], got[bad[:5]], exp[bad[:5]])
            order = np.lexsort((np.arange(n), full))
            for cutoff in (3, 5):
                s, i = bc.topk(corpus, 16, score_cutoff=cutoff)
                want = [(int(full[j]), int(j)) for j in order[:16] if full[j] <= cutoff]
                assert list(zip(s.tolist(), i.tolist())) == want, (metric, qlen, cutoff)
        del corpus


def _filter_check(bc, ob, op, corpus, exp, **kw):
    """rf_filter_* in its three orders, host and device outputs, against the oracle's Somes"""
    idx_e, val_e = _some(exp)
    desc = op in (N.OP_SIMILARITY, N.OP_NORMALIZED_SIMILARITY)
    for order in (N.FILTER_BY_INDEX, N.FILTER_BY_SCORE, N.FILTER_ANY):
        idx, val = bc.filter_many(op, corpus, order=order, **kw)
        assert bc.last_filter_count == len(idx_e), (order, bc.last_filter_count, len(idx_e))
        if order == N.FILTER_BY_INDEX:
            assert np.array_equal(idx, idx_e) and np.array_equal(val, val_e), (order, idx[:5], idx_e[:5])
        else:
            perm = np.argsort(idx, kind="stable")
            assert np.array_equal(idx[perm], idx_e) and np.array_equal(val[perm], val_e), order
            if order == N.FILTER_BY_SCORE:
                key = -val_e.astype(np.float64) if desc else val_e.astype(np.float64)
                want = np.lexsort((idx_e, key))
                assert np.array_equal(idx, idx_e[want]) and np.array_equal(val, val_e[want])
    # a capacity that is too small: the true count comes back, the entries delivered are valid pairs in order
    if len(idx_e) > 3:
        cap = len(idx_e) // 2
        idx, val = bc.filter_many(op, corpus, capacity=cap, **kw)
        assert bc.last_filter_count == len(idx_e) and len(idx) == cap
        pos = np.searchsorted(idx_e, idx)
        assert np.all(idx_e[pos] == idx) and np.array_equal(val_e[pos], val) and np.all(np.diff(idx.astype(np.int64)) > 0)
    # device outputs, index_base
    import torch

    idx, val = bc.filter_many(op, corpus, device_out=True, index_base=1 << 33, **kw)
    torch.cuda.synchronize()
    idx = idx.cpu().numpy().astype(np.uint64)
    val = val.cpu().numpy()
    val = val.view(np.uint32) if val.dtype == np.int32 else val
    assert np.array_equal(idx, idx_e + np.uint64(1 << 33)) and np.array_equal(val, val_e)
    # a pure count
    idx, _ = bc.filter_many(op, corpus, capacity=0, **kw)
    assert len(idx) == 0 and bc.last_filter_count == len(idx_e)


def test_filter_c1_shape_every_metric():
    """BASELINE configs[0]'s shape (query 32 x 10 k ragged candidates of <= 64 symbols): the compact pairs equal np.nonzero(oracle != None) for every usize metric x
    cutoffs, normalized ops, Jaro-Winkler >= 0.7 and fuzz::ratio."""
    q = synth.query(32, 5)
    data, offsets = synth.ragged_host(10_000, 64, seed=6)
    rng = np.random.default_rng(7)
    for r in rng.choice(10_000, size=60, replace=False):  # near-duplicates cut to the candidate's length
        a, b = int(offsets[r]), int(offsets[r + 1])
        row = np.resize(np.frombuffer(q, dtype=np.uint8), b - a).copy()
        if len(row) > 2:
            row[rng.integers(0, len(row), size=int(rng.integers(0, 4)))] = 122
        data[a:b] = row
    corpus = rf.Corpus.from_ragged(data, offsets)
    for metric in ("levenshtein", "osa", "indel", "lcs_seq"):
        bc, ob = GPU[metric].BatchComparator(q), ORA[metric].BatchComparator(q)
        for cutoff in (0, 3, 12, 40, None):
            kw = {} if cutoff is None else {"score_cutoff": cutoff}
            _filter_check(bc, ob, N.OP_DISTANCE, corpus, ob.many(N.OP_DISTANCE, data, offsets, nthreads=8, **kw), **kw)
        if metric != "levenshtein":  # (Q2: the reference's Levenshtein similarity above its cutoff is a wrapped value)
            _filter_check(bc, ob, N.OP_SIMILARITY, corpus, ob.many(N.OP_SIMILARITY, data, offsets, nthreads=8, score_cutoff=20), score_cutoff=20)
        _filter_check(bc, ob, N.OP_NORMALIZED_SIMILARITY, corpus, ob.many(N.OP_NORMALIZED_SIMILARITY, data, offsets, nthreads=8, score_cutoff=0.6), score_cutoff=0.6)
    for metric in ("jaro", "jaro_winkler"):
        bc, ob = GPU[metric].BatchComparator(q), ORA[metric].BatchComparator(q)
        _filter_check(bc, ob, N.OP_SIMILARITY, corpus, ob.many(N.OP_SIMILARITY, data, offsets, nthreads=8, score_cutoff=0.7), score_cutoff=0.7)
        _filter_check(bc, ob, N.OP_DISTANCE, corpus, ob.many(N.OP_DISTANCE, data, offsets, nthreads=8, score_cutoff=0.3), score_cutoff=0.3)
    bc, ob = rf.fuzz.RatioBatchComparator(q), o.fuzz.RatioBatchComparator(q)
    _filter_check(bc, ob, N.OP_SIMILARITY, corpus, ob.many(N.OP_NORMALIZED_SIMILARITY, data, offsets, nthreads=8, score_cutoff=0.5), score_cutoff=0.5)
    # an empty corpus, and a cutoff nothing passes
    empty = rf.Corpus.from_list([])
    idx, val = GPU["levenshtein"].BatchComparator(q).filter_many(N.OP_DISTANCE, empty, score_cutoff=3)
    assert len(idx) == 0
    idx, val = GPU["jaro"].BatchComparator(q).filter_many(N.OP_SIMILARITY, corpus, score_cutoff=1.5)
    assert len(idx) == 0


@pytest.mark.parametrize("share", [0.0, 0.05])
def test_filter_large_single_length_corpus_takes_the_lane_compaction(share):
    """2.1 M x 64 (the head-plane path): rf_filter under cutoffs 0..5 never builds the dense vector (rf_sparse.hip leaves the survivors' values at their numbers);
    cutoffs beyond the head plane's reach, Indel, normalized ops and Jaro-Winkler >= 0.9 take the general road.  With 5 % prefix sharers the survivors of the
    first pass outnumber the matches 250 : 1."""
    n = 2_100_037
    q, rows = _prefix_corpus(n, 64, share, seed=91)
    corpus = rf.Corpus.from_rows(rows)
    for metric in ("levenshtein", "osa"):
        bc, ob = GPU[metric].BatchComparator(q), ORA[metric].BatchComparator(q)
        full = ob.rows(N.OP_DISTANCE, rows, nthreads=8)
        for cutoff in (0, 2, 3, 5, 9, 30):
            _filter_check(bc, ob, N.OP_DISTANCE, corpus, np.where(full <= np.uint64(cutoff), full, U64MAX), score_cutoff=cutoff)
        _filter_check(bc, ob, N.OP_NORMALIZED_SIMILARITY, corpus, ob.rows(N.OP_NORMALIZED_SIMILARITY, rows, nthreads=8, score_cutoff=0.95), score_cutoff=0.95)
    bc, ob = GPU["indel"].BatchComparator(q), ORA["indel"].BatchComparator(q)
    _filter_check(bc, ob, N.OP_DISTANCE, corpus, ob.rows(N.OP_DISTANCE, rows, nthreads=8, score_cutoff=12), score_cutoff=12)
    bc, ob = GPU["jaro_winkler"].BatchComparator(q), ORA["jaro_winkler"].BatchComparator(q)
    _filter_check(bc, ob, N.OP_SIMILARITY, corpus, ob.rows(N.OP_SIMILARITY, rows, nthreads=8, score_cutoff=0.9), score_cutoff=0.9)
    bc, ob = rf.fuzz.RatioBatchComparator(q), o.fuzz.RatioBatchComparator(q)
    _filter_check(bc, ob, N.OP_SIMILARITY, corpus, ob.rows(N.OP_NORMALIZED_SIMILARITY, rows, nthreads=8, score_cutoff=0.9), score_cutoff=0.9)


def test_filter_when_the_survivors_outgrow_their_room():
    """Every candidate carries the query's head (share 1.0): the first pass keeps all 2.1 M lanes, more than the room the fast road gives its survivors
    (max(n / 4, 4 x capacity)) -- the call must notice on the device-side count and take the general road: same pairs."""
    n = 1_100_000
    q, rows = _prefix_corpus(n, 64, 1.0, seed=17)
    corpus = rf.Corpus.from_rows(rows)
    bc, ob = GPU["levenshtein"].BatchComparator(q), ORA["levenshtein"].BatchComparator(q)
    full = ob.rows(N.OP_DISTANCE, rows, nthreads=8)
    exp = np.where(full <= np.uint64(3), full, U64MAX)
    idx_e, val_e = _some(exp)
    idx, val = bc.filter_many(N.OP_DISTANCE, corpus, score_cutoff=3, capacity=4096)
    assert bc.last_filter_count == len(idx_e) and np.array_equal(idx, idx_e) and np.array_equal(val, val_e)
    got = bc.distance_many(corpus, score_cutoff=3)
    assert len(_same(got, exp)) == 0


def test_filter_and_slot_order_on_a_length_bucketed_corpus():
    """1.35 M candidates of lengths 1..70, log-normal (median 24) with Zipf symbols: (a) RF_FLAG_SLOT_ORDER -- results in slot order equal the default call's after
    the caller's own permutation through rf_corpus_slot_index, for full scans (asm tiles kernels, the gather path's kernels), cutoff scans (length-run views) and
    f64 metrics; (b) rf_filter_* over the slot-ordered temporary, every order."""
    n = 1_350_000
    data, offsets = synth.lognormal_ragged_host(n, 70, seed=3, median=24.0, sigma=0.5, zipf_s=1.1)
    q = bytes(data[int(offsets[11]): int(offsets[11]) + 40]) + b"Zq"
    corpus = rf.Corpus.from_ragged(data, offsets)
    slot_index = corpus.slot_index()
    assert len(slot_index) == corpus.slot_count >= n
    real = slot_index != NONE32
    assert real.sum() == n and np.array_equal(np.sort(slot_index[real]), np.arange(n, dtype=np.uint32))
    cases = [("levenshtein", N.OP_DISTANCE, {}), ("levenshtein", N.OP_DISTANCE, {"score_cutoff": 3}), ("levenshtein", N.OP_DISTANCE, {"score_cutoff": 20}),
             ("osa", N.OP_DISTANCE, {}), ("indel", N.OP_DISTANCE, {}), ("indel", N.OP_NORMALIZED_SIMILARITY, {}), ("lcs_seq", N.OP_SIMILARITY, {"score_cutoff": 10}),
             ("jaro_winkler", N.OP_SIMILARITY, {}), ("jaro", N.OP_SIMILARITY, {"score_cutoff": 0.8}), ("levenshtein", N.OP_NORMALIZED_DISTANCE, {"score_cutoff": 0.3})]
    for metric, op, kw in cases:
        bc, ob = GPU[metric].BatchComparator(q), ORA[metric].BatchComparator(q)
        exp = ob.many(op, data, offsets, nthreads=8, **kw)
        got = bc.many(op, corpus, **kw)
        assert len(_same(got, exp)) == 0, (metric, op, kw)
        slots = bc.many(op, corpus, rf.distance.levenshtein.Args().slot_order(), **kw)
        assert len(slots) == corpus.slot_count
        back = np.empty_like(got)
        back[slot_index[real]] = slots[real]
        assert len(_same(back, exp)) == 0, ("slot order", metric, op, kw)
        if kw:
            _filter_check(bc, ob, op, corpus, exp, **kw)
    # the query > 64 symbols (multi-word scans) in slot order
    q2 = (q * 3)[:100]
    bc, ob = GPU["levenshtein"].BatchComparator(q2), ORA["levenshtein"].BatchComparator(q2)
    exp = ob.many(N.OP_DISTANCE, data, offsets, nthreads=8)
    slots = bc.many(N.OP_DISTANCE, corpus, rf.distance.levenshtein.Args().slot_order())
    back = np.empty(n, dtype=np.uint32)
    back[slot_index[real]] = slots[real]
    assert len(_same(back, exp)) == 0
    # the flag belongs to rf_many_* alone: top-k, many-queries and filter calls write n-entry rows / pairs and must ignore it
    so = rf.distance.levenshtein.Args().slot_order()
    bc = GPU["levenshtein"].BatchComparator(q)
    s0, i0 = bc.topk(corpus, 7)
    s1, i1 = bc.topk(corpus, 7, args=so)
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1)
    rows0 = GPU["indel"].BatchComparator.many_multi([GPU["indel"].BatchComparator(q), GPU["indel"].BatchComparator(q[:20])], N.OP_DISTANCE, corpus)
    rows1 = GPU["indel"].BatchComparator.many_multi([GPU["indel"].BatchComparator(q), GPU["indel"].BatchComparator(q[:20])], N.OP_DISTANCE, corpus, so)
    assert rows0.shape == rows1.shape == (2, n) and np.array_equal(rows0, rows1)
    f0 = bc.filter_many(N.OP_DISTANCE, corpus, score_cutoff=5)
    f1 = bc.filter_many(N.OP_DISTANCE, corpus, so, score_cutoff=5)
    assert np.array_equal(f0[0], f1[0]) and np.array_equal(f0[1], f1[1])
    # a single-length corpus: slots are indices
    rows = synth.rows_host(5000, 24, seed=4)
    c2 = rf.Corpus.from_rows(rows)
    assert c2.slot_count == 5000 and np.array_equal(c2.slot_index(), np.arange(5000, dtype=np.uint32))
    a = GPU["indel"].BatchComparator(q).many(N.OP_DISTANCE, c2, rf.distance.indel.Args().slot_order())
    assert np.array_equal(a, GPU["indel"].BatchComparator(q).many(N.OP_DISTANCE, c2))


def _saved(corpus, path):
    corpus.save(str(path))
    return open(path, "rb").read()


@pytest.mark.parametrize("shape", ["ragged", "lognormal_zipf", "single_length", "long_tail", "offset_window"])
def test_device_packer_equals_the_host_layout_byte_for_byte(shape, tmp_path):
    """rf_corpus_pack does its per-candidate work on the device now (rf_pack_ragged.hip: length keys + histogram, stable radix sort by length,
    one scatter kernel per destination tile).  The layout's specification stays the host packer (rf_corpus_layout_host / build_layout): the same input packed with
    RF_DEVICE_PACK_MIN=0 (host) in a child process and on the device here must SAVE to identical files -- header, length table, tile descriptors, slot map, mixed
    section, every payload byte (file format unchanged) -- and scan to the oracle's values."""
    rng = np.random.default_rng(len(shape))
    n = 300_011
    if shape == "ragged":
        data, offsets = synth.ragged_host(n, 70, seed=5, min_len=0)
    elif shape == "lognormal_zipf":
        data, offsets = synth.lognormal_ragged_host(n, 200, seed=6, median=30.0, sigma=0.7, zipf_s=1.2)
    elif shape == "single_length":
        data, offsets = synth.ragged_host(n, 48, seed=7, min_len=48)
    elif shape == "long_tail":  # a few long candidates among short ones: leftovers of many lengths, several mixed tiles, partial last chunks
        lens = np.where(rng.random(n) < 0.001, rng.integers(300, 3000, size=n), rng.integers(0, 40, size=n))
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens, dtype=np.uint64)
        data = rng.integers(0, 256, size=int(offsets[-1]), dtype=np.uint8)  # every byte value: the renaming sees all 256 symbols
    else:  # offsets that do not start at 0 (a window into a larger buffer)
        data, offsets = synth.ragged_host(n, 33, seed=8, min_len=1)
        offsets = offsets + np.uint64(12345)
        data = np.concatenate([rng.integers(48, 123, size=12345, dtype=np.uint8), data])
    np.save(tmp_path / "data.npy", data)
    np.save(tmp_path / "offsets.npy", offsets)
    child = ("import sys, numpy as np; sys.path.insert(0, %r); import rapidfuzz_rs_amd as rf; "
             "rf.Corpus.from_ragged(np.load(%r), np.load(%r)).save(%r)") % (ROOT, str(tmp_path / "data.npy"), str(tmp_path / "offsets.npy"), str(tmp_path / "host.rfc"))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RF_DEVICE_PACK_MIN="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    corpus = rf.Corpus.from_ragged(data, offsets)
    dev_file, host_file = _saved(corpus, tmp_path / "dev.rfc"), open(tmp_path / "host.rfc", "rb").read()
    assert len(dev_file) == len(host_file)
    if dev_file != host_file:
        a, b = np.frombuffer(dev_file, dtype=np.uint8), np.frombuffer(host_file, dtype=np.uint8)
        bad = np.nonzero(a != b)[0]
        raise AssertionError(f"{len(bad)} bytes differ, first at {bad[:8]}")
    q = bytes(data[int(offsets[7]): int(offsets[7]) + 40]) or b"abc"
    for metric in ("levenshtein", "indel", "jaro_winkler"):
        op = N.OP_SIMILARITY if metric == "jaro_winkler" else N.OP_DISTANCE
        got = GPU[metric].BatchComparator(q).many(op, corpus)
        exp = ORA[metric].BatchComparator(q).many(op, data, offsets, nthreads=8)
        assert len(_same(got, exp)) == 0, metric


@pytest.mark.parametrize("seed", range(int(os.environ.get("RF_FUZZ_SEEDS", "16"))))
def test_randomized_corpus_models_vary_what_the_plan_looks_at(seed):
    """the plan layer picks kernels from symbol counts, length histograms and survivor shares, and every earlier GPU-vs-oracle corpus was
    iid-uniform with substitution-only near-duplicates.  Per seed: a symbol law (uniform 62 / Zipf / 2 symbols / all 256 bytes), a length law (one length / uniform /
    log-normal / bimodal), a share of candidates that carry the query's head, near-duplicates made by insertions, deletions and substitutions; 150 k .. 1.2 M
    candidates (so that the head plane, the 6-bit payload, the lane compaction, the device packer and the gather maps all come into play); then random metric x op
    x cutoff calls through rf_many_*, rf_filter_* and RF_FLAG_SLOT_ORDER -- every value against the oracle."""
    rng = np.random.default_rng(77_000 + seed)
    n = int(rng.choice([150_000, 400_000, 1_200_000]))
    law = str(rng.choice(["uniform", "zipf", "two", "bytes"]))
    draw = {"uniform": lambda m: synth.ALNUM[rng.integers(0, 62, size=m)], "zipf": lambda m: synth.zipf_alphabet_draw(rng, m, s=float(rng.choice([0.8, 1.3]))),
            "two": lambda m: np.frombuffer(b"ab", dtype=np.uint8)[rng.integers(0, 2, size=m)], "bytes": lambda m: rng.integers(0, 256, size=m, dtype=np.uint8)}[law]
    length_law = str(rng.choice(["one", "uniform", "lognormal", "bimodal"]))
    max_len = int(rng.choice([24, 64, 70, 130]))
    if length_law == "one":
        lens = np.full(n, int(rng.choice([16, 33, 48, 64])), dtype=np.int64)
    elif length_law == "uniform":
        lens = rng.integers(1, max_len + 1, size=n)
    elif length_law == "lognormal":
        lens = np.clip(np.rint(rng.lognormal(np.log(max_len / 3), 0.6, size=n)), 0, max_len).astype(np.int64)
    else:
        lens = np.where(rng.random(n) < 0.7, rng.integers(max_len - 4, max_len + 1, size=n), rng.integers(1, 9, size=n))
    qlen = int(rng.choice([8, 30, 48, 64, max(8, int(lens.max()))]))
    q = draw(qlen)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    data = draw(int(offsets[-1]))
    # the query's head on a share of the candidates (as far as they reach)
    share = float(rng.choice([0.0, 0.01, 0.3]))
    for i in np.nonzero(rng.random(n) < share)[0]:
        a, b = int(offsets[i]), int(offsets[i + 1])
        h = min(b - a, int(rng.integers(8, 13)), qlen)
        data[a: a + h] = q[:h]
    # near-duplicates by insertions / deletions / substitutions, cut or padded to the slot's length (the corpus keeps its length law)
    for i in rng.choice(n, size=300, replace=False):
        a, b = int(offsets[i]), int(offsets[i + 1])
        row = list(q)
        for _ in range(int(rng.integers(0, 6))):
            r, pos = int(rng.integers(0, 3)), int(rng.integers(0, len(row) + 1))
            if r == 0:
                row.insert(pos, int(draw(1)[0]))
            elif row:
                if r == 1:
                    del row[min(pos, len(row) - 1)]
                else:
                    row[min(pos, len(row) - 1)] = int(draw(1)[0])
        row = (row + list(draw(max(0, b - a - len(row)))))[: b - a]
        data[a:b] = np.array(row, dtype=np.uint8)
    corpus = rf.Corpus.from_ragged(data, offsets)
    slot_index = corpus.slot_index()
    real = slot_index != NONE32
    qb = q.tobytes()
    for _ in range(7):
        metric = str(rng.choice(["levenshtein", "osa", "indel", "lcs_seq", "jaro_winkler"]))
        is_float_metric = metric == "jaro_winkler"
        op = str(rng.choice(["distance", "similarity", "normalized_similarity"]))
        is_f = is_float_metric or op.startswith("normalized")
        kw = {}
        if rng.random() < 0.8:
            if is_f:
                kw["score_cutoff"] = float(rng.choice([0.1, 0.3, 0.7, 0.9, 0.97])) if op != "distance" else float(rng.choice([0.03, 0.1, 0.3]))
            elif op == "distance":
                kw["score_cutoff"] = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 20, qlen]))
            else:
                kw["score_cutoff"] = int(rng.choice([1, qlen // 2, max(1, qlen - 3)]))
        if metric == "levenshtein" and op == "similarity" and "score_cutoff" in kw:
            continue  # quirk Q2
        opn = {"distance": N.OP_DISTANCE, "similarity": N.OP_SIMILARITY, "normalized_similarity": N.OP_NORMALIZED_SIMILARITY}[op]
        bc, ob = GPU[metric].BatchComparator(qb), ORA[metric].BatchComparator(qb)
        exp = ob.many(opn, data, offsets, nthreads=8, **kw)
        got = bc.many(opn, corpus, **kw)
        bad = _same(got, exp)
        if len(bad) and metric in ("lcs_seq", "indel") and qlen > 64 and kw.get("score_cutoff") is not None:
            # Quirk Q8 (tests/test_gpu_parity.py _check_many has the rule): under a cutoff the reference's banded multi-word LCS can lose a match at the band's
            # edge; the device is exact -- where the two differ the device must hold the reference's UNCUT value, and the oracle, run on that one pair, must
            # show the defect's precondition (rfo_last_lcs_q8_edges).  The entries are taken out of the comparison of this call and of the ones below.
            uncut_raw = ob.many(opn, data, offsets, nthreads=8)
            uncut = _u32(uncut_raw) if got.dtype == np.uint32 else uncut_raw
            excused = []
            for i in bad:
                if got[i] != uncut[i]:
                    continue
                cand = np.ascontiguousarray(data[int(offsets[i]) : int(offsets[i + 1])])
                ob.many(opn, cand, np.array([0, len(cand)], dtype=np.uint64), nthreads=1, **kw)
                if o.last_lcs_q8_edges() > 0:
                    excused.append(int(i))
            if excused:
                exp = exp.copy()
                exp[excused] = uncut_raw[excused]
                bad = _same(got, exp)
        assert len(bad) == 0, (seed, law, length_law, share, metric, op, kw, qlen, bad[:5], got[bad[:5]], exp[bad[:5]])
        slots = bc.many(opn, corpus, rf.Args().slot_order(), **kw)
        back = np.empty_like(got)
        back[slot_index[real]] = slots[real]
        assert len(_same(back, exp)) == 0, ("slot order", seed, metric, op, kw)
        if kw:
            idx_e, val_e = _some(exp)
            order = int(rng.choice([N.FILTER_BY_INDEX, N.FILTER_BY_SCORE, N.FILTER_ANY]))
            idx, val = bc.filter_many(opn, corpus, order=order, **kw)
            perm = np.argsort(idx, kind="stable")
            assert np.array_equal(idx[perm], idx_e) and np.array_equal(val[perm], val_e), ("filter", seed, metric, op, kw, order, len(idx), len(idx_e))
            if order == N.FILTER_BY_INDEX:
                assert np.all(np.diff(idx.astype(np.int64)) > 0)


def _band_rows(n, len2, q, share, seed, kinds=8):
    """n rows of len2 symbols around query q (its own length may differ by a few): a fraction `share` near the query -- the query resized to len2 with 0..12
    substitutions, or with an insertion / deletion near the front (the rest of the row then runs ONE off the diagonal), or equal to it for the first 20..120 symbols
    and random from there on (in the band for a while, out of it long before the end) -- the others random."""
    rng = np.random.default_rng(seed)
    qa = np.frombuffer(q, dtype=np.uint8)
    rows = synth.ALNUM[rng.integers(0, 62, size=(n, len2))]
    near = np.nonzero(rng.random(n) < share)[0]
    for j, i in enumerate(near):
        kind = j % kinds  # (kinds = 6: no head-then-noise rows)
        base = np.resize(qa, len2 + 2)
        if kind <= 3:
            row = base[:len2].copy()
            k = int(rng.integers(0, 13))
            if k:
                row[rng.integers(0, len2, size=k)] = synth.ALNUM[rng.integers(0, 62, size=k)]
        elif kind == 4:  # one symbol missing near the front
            cut = int(rng.integers(0, 30))
            row = np.delete(base, cut)[:len2].copy()
        elif kind == 5:  # one symbol more near the front
            at = int(rng.integers(0, 30))
            row = np.insert(base, at, np.uint8(35))[:len2].copy()
        else:  # the query's head, then noise
            row = rows[i].copy()
            h = min(int(rng.integers(20, 121)), len2)
            row[:h] = base[:h]
        rows[i] = row
    return np.ascontiguousarray(rows)


@pytest.mark.parametrize("share", [0.02, 0.3, 0.5, 0.9])
def test_small_band_scan_hands_sparse_tiles_to_a_dense_second_pass(share):
    """rf_band.hip launch_band on a single-length corpus: a tile with few lanes left within the band at a chunk end (column 2k + 8, rounded up) is listed (tile, lane mask) and its lanes finish in
    band_sparse_kernel, 64 survivors to a wavefront (the reference's hyrroe2003_small_band_with_pm, levenshtein.rs:509-617, decides per candidate,
    so who shares a wavefront with whom must not show).  Near-duplicate shares from 2 % (nearly every tile listed with one or two lanes) to 90 % (nearly none listed),
    candidates that leave the band after the hand-over, queries shorter / longer than the rows, every cutoff range of the band kernel, n not a multiple of 64."""
    import torch

    for qlen, len2, n in ((256, 256, 150_011), (250, 256, 9_001), (300, 296, 9_001), (130, 128, 5_003)):
        q = synth.query(qlen, 0xBA2D + qlen)
        rows = _band_rows(n, len2, q, share, seed=qlen * 7 + int(share * 100))
        corpus = rf.Corpus.from_device_rows(torch.from_numpy(rows).cuda())
        bc, ob = rf.distance.levenshtein.BatchComparator(q), o.levenshtein.BatchComparator(q)
        for k in (4, 8, 17, 31):
            got = bc.distance_many(corpus, score_cutoff=k)
            exp = ob.rows(N.OP_DISTANCE, rows, nthreads=8, score_cutoff=k)
            bad = _same(got, exp)
            assert len(bad) == 0, (share, qlen, len2, k, bad[:5], got[bad[:5]], exp[bad[:5]])
            if qlen == 256 and k == 8:
                assert (got != NONE32).sum() > 0.3 * share * n  # (the near-duplicates are found)
        # the hinted scan runs the same band launch as its first pass (rf_api_scan.hip run_many_hinted)
        got = bc.distance_many(corpus, score_hint=8)
        exp = ob.rows(N.OP_DISTANCE, rows, nthreads=8)
        assert len(_same(got, exp)) == 0, (share, qlen, len2, "hint")


@pytest.mark.parametrize("env", [{"RF_BAND_DEFER": "0"}, {"RF_BAND_DEFER_AFTER": "0", "RF_BAND_DEFER_ADAPT": "0"},
                                 {"RF_BAND_DEFER_AT": "16", "RF_BAND_DEFER_MAX": "63", "RF_BAND_DEFER_AFTER": "0", "RF_BAND_DEFER_ADAPT": "0"},
                                 {"RF_BAND_DEFER_AT": "64", "RF_BAND_DEFER_MAX": "20", "RF_BAND_DEFER_AFTER": "0", "RF_BAND_DEFER_ADAPT": "0"},
                                 {"RF_ASM_BAND": "0", "RF_BAND_DEFER_AFTER": "3", "RF_BAND_DEFER_ADAPT": "0"}, {"RF_BAND_DEFER_AFTER": "1024"}])
def test_small_band_hand_over_switches(env):
    """The same test with the hand-over off, whatever the last launch listed (by default a stream whose last hand-over launch saved little takes the plain kernel), at column 16 for every tile with a lane left, at column 64, and on the compiled column (the switches are read once per process)."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "test_small_band_scan_hands_sparse_tiles"],
                       capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, **env))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("seed", range(int(os.environ.get("RF_FUZZ_SEEDS", "8"))))
def test_randomized_small_band_on_single_length_corpora(seed):
    """Randomized form of the test above (forced in the fuzz runs with RF_BAND_DEFER_AFTER=0 RF_BAND_DEFER_ADAPT=0): query 65..600 symbols, rows a few symbols shorter
    or longer (beyond the cutoff too: the length test), 4- and 62-symbol alphabets, any share of near candidates, cutoffs 0..31 and one score_hint per corpus."""
    import torch

    rng = np.random.default_rng(0xBA2D0000 + seed)
    for _ in range(3):
        qlen = int(rng.choice([65, 96, 128, 200, 256, 257, 300, 600, int(rng.integers(65, 600))]))
        len2 = max(1, qlen + int(rng.integers(-12, 13)))
        n = int(rng.choice([3_000, 70_001, 140_000]))
        share = float(rng.choice([0.0, 0.003, 0.05, 0.4, 0.6, 1.0]))
        alpha = synth.ALNUM if rng.random() < 0.6 else synth.ALNUM[:4]
        qa = alpha[rng.integers(0, len(alpha), size=qlen)]
        rows = alpha[rng.integers(0, len(alpha), size=(n, len2))]
        near = np.nonzero(rng.random(n) < share)[0]
        base = np.resize(qa, len2 + 3)
        for j, i in enumerate(near):
            kind = j % 6
            if kind <= 2:
                row = base[:len2].copy()
                e = int(rng.integers(0, 20))
                if e:
                    row[rng.integers(0, len2, size=e)] = alpha[rng.integers(0, len(alpha), size=e)]
            elif kind == 3:
                row = np.delete(base, int(rng.integers(0, min(40, len2))))[:len2].copy()
            elif kind == 4:
                row = np.insert(base, int(rng.integers(0, min(40, len2))), alpha[0])[:len2].copy()
            else:
                row = rows[i].copy()
                h = int(rng.integers(1, len2 + 1))
                row[:h] = base[:h]
            rows[i] = row
        rows = np.ascontiguousarray(rows)
        corpus = rf.Corpus.from_device_rows(torch.from_numpy(rows).cuda())
        q = qa.tobytes()
        bc, ob = rf.distance.levenshtein.BatchComparator(q), o.levenshtein.BatchComparator(q)
        for k in sorted(set(int(x) for x in rng.choice([0, 1, 3, 8, 12, 16, 25, 31], size=3))):
            got = bc.distance_many(corpus, score_cutoff=k)
            exp = ob.rows(N.OP_DISTANCE, rows, nthreads=8, score_cutoff=k)
            bad = _same(got, exp)
            assert len(bad) == 0, (seed, qlen, len2, n, share, len(alpha), k, bad[:5], got[bad[:5]], exp[bad[:5]])
        # the normalized ops under an f64 cutoff that leaves few raw edits ride the same kernel (run_many's norm_raw_cut) -- or, beyond 31 edits, the compiled f64 scan
        for op, cut in ((N.OP_NORMALIZED_SIMILARITY, 1.0 - float(rng.random()) * 0.16), (N.OP_NORMALIZED_DISTANCE, float(rng.choice([0.0, float(rng.random()) * 0.16, 25 / max(qlen, len2)])))):
            got = bc.many(op, corpus, score_cutoff=cut)
            exp = ob.rows(op, rows, nthreads=8, score_cutoff=cut)
            bad = _same(got, exp)
            assert len(bad) == 0, (seed, qlen, len2, n, share, len(alpha), op, cut, bad[:5], got[bad[:5]], exp[bad[:5]])
        hint = int(rng.choice([0, 4, 16, 40]))
        exp = ob.rows(N.OP_DISTANCE, rows, nthreads=8)
        for _rep in range(2):  # (the second call finds the hint credited where the first resolved >= 70 %: the list road of run_many_hinted)
            got = bc.distance_many(corpus, score_hint=hint)
            assert len(_same(got, exp)) == 0, (seed, qlen, len2, n, share, "hint", hint, _rep)


@pytest.mark.parametrize("share", [0.75, 0.97])
def test_score_hint_scan_walks_the_list_its_band_pass_leaves(share):
    """A credible score_hint on a single-length corpus (rf_api_scan.hip run_many_hinted, round 6): the band pass lists the lanes it answers None and the caller's own
    multi-word scan walks that list (rf_sparse.hip sparse_words_kernel) -- from the second hinted call on, once the first has shown the hint to resolve >= 70 % of the
    corpus.  Results never depend on the hint (levenshtein.rs:2153-2160): hints x cutoffs on both sides of max(hint, 31) x weights, queries of 2..8 words, rows a few
    symbols off the query's length, every value against the un-hinted oracle; RF_TRACE_PLAN (child process) shows that the list road was taken."""
    import torch

    if os.environ.get("RF_TRACE_PLAN") is None:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:xdist", "-s", "-k",
                            f"test_score_hint_scan_walks_the_list and {share}"], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RF_TRACE_PLAN="1"))
        assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
        assert (r.stdout + r.stderr).count("[rf plan] hint lists:") >= 8
        return
    for qlen, len2, n in ((256, 256, 120_011), (130, 128, 70_003), (300, 305, 70_003), (500, 512, 66_000)):
        q = synth.query(qlen, 0x4157 + qlen)
        rows = _band_rows(n, len2, q, share, seed=qlen + int(share * 1000), kinds=6)
        corpus = rf.Corpus.from_device_rows(torch.from_numpy(rows).cuda())
        bc, ob = rf.distance.levenshtein.BatchComparator(q), o.levenshtein.BatchComparator(q)
        for weights in (None, (2, 2, 2)):
            f = 1 if weights is None else 2
            kw = {} if weights is None else {"weights": rf.WeightTable(*weights)}
            okw = {} if weights is None else {"weights": weights}
            for cutoff in (None, 40 * f, 200 * f):
                exp = _u32(ob.rows(N.OP_DISTANCE, rows, nthreads=8, score_cutoff=cutoff, **okw))
                for hint in (8 * f, 0, 31 * f, 16 * f):
                    got = bc.many(N.OP_DISTANCE, corpus, score_cutoff=cutoff, score_hint=hint, **kw)
                    bad = np.nonzero(got != exp)[0]
                    assert len(bad) == 0, (share, qlen, len2, weights, cutoff, hint, len(bad), bad[:5], got[bad[:5]], exp[bad[:5]])


@pytest.mark.parametrize("kind", ["single_length", "ragged"])
def test_normalized_cutoffs_of_long_queries_through_the_band_kernel(kind):
    """normalized_distance / normalized_similarity of a Levenshtein scan with a query beyond 64 symbols under an f64 cutoff that leaves <= 31 raw edits (run_many's
    norm_raw_cut): the u32 scan under the implied raw cutoff (the small-band kernel) + the normalizing pass.  Some / None and every value must be the oracle's f64
    (details/distance.rs:246-250, :273; common.rs:43-45) -- cutoffs on both sides of the 31-edit line, exactly on representable quotients (k / maximum), 0 and 1,
    unit and (2, 2, 2) weights, rows shorter and longer than the query; the child process repeats it on the compiled f64 scan (RF_NORM_BAND=0)."""
    import torch

    if os.environ.get("RF_NORM_BAND") is None:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", f"test_normalized_cutoffs_of_long_queries and {kind}"],
                           capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RF_NORM_BAND="0"))
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:]
    rng = np.random.default_rng(77)
    for qlen in (100, 256, 300):
        q = synth.query(qlen, 0x4E02 + qlen)
        qa = np.frombuffer(q, dtype=np.uint8)
        if kind == "single_length":
            len2 = qlen + int(rng.integers(-6, 7))
            rows = _band_rows(50_003, len2, q, 0.3, seed=qlen)
            corpus = rf.Corpus.from_device_rows(torch.from_numpy(rows).cuda())
            data, offsets = rows.reshape(-1), np.arange(0, (len(rows) + 1) * len2, len2, dtype=np.uint64)
        else:
            cands = []
            for i in range(30_000):
                ln = int(rng.integers(max(1, qlen - 40), qlen + 41))
                if i % 3 == 0:
                    row = np.resize(qa, ln).copy()
                    e = int(rng.integers(0, 25))
                    if e:
                        row[rng.integers(0, ln, size=e)] = synth.ALNUM[rng.integers(0, 62, size=e)]
                else:
                    row = synth.ALNUM[rng.integers(0, 62, size=ln)]
                cands.append(row.tobytes())
            data, offsets = rf.ragged(cands)
            corpus = rf.Corpus.from_ragged(data, offsets)
        bc, ob = rf.distance.levenshtein.BatchComparator(q), o.levenshtein.BatchComparator(q)
        longest = max(qlen, int(np.diff(offsets).max()))
        for weights in (None, (2, 2, 2)):
            kw = {} if weights is None else {"weights": rf.WeightTable(*weights)}
            okw = {} if weights is None else {"weights": weights}
            for opname, op in (("normalized_distance", N.OP_NORMALIZED_DISTANCE), ("normalized_similarity", N.OP_NORMALIZED_SIMILARITY)):
                cuts = [0.0, 1.0, 0.02, 0.05, 0.1, 0.12, 0.2, 8 / longest, 9 / longest, 31 / longest, 32 / longest, float(rng.random()) * 0.15]
                for c in cuts:
                    cut = c if opname == "normalized_distance" else 1.0 - c
                    got = bc.many(op, corpus, score_cutoff=cut, **kw)
                    exp = ob.many(op, data, offsets, nthreads=8, score_cutoff=cut, **okw)
                    bad = _same(got, exp)
                    assert len(bad) == 0, (kind, qlen, weights, opname, cut, len(bad), bad[:5], got[bad[:5]], exp[bad[:5]])


def _bucketed_long_corpus(q, seed, per_length=20_000, share=0.05):
    """lengths len(q) - 6 .. len(q) + 6 with `per_length` candidates each (long runs of exact tiles), two sparsely populated lengths and leftovers for the mixed section;
    `share` of the candidates near the query (substitutions, one insertion or deletion near the front, or the query's head and noise), in original order shuffled"""
    rng = np.random.default_rng(seed)
    qa = np.frombuffer(q, dtype=np.uint8)
    qlen = len(qa)
    lens = np.concatenate([np.full(per_length + int(rng.integers(0, 64)), L) for L in range(qlen - 6, qlen + 7)] + [np.full(700, qlen - 20), np.full(90, qlen + 30), np.full(40, 3)])
    rng.shuffle(lens)
    cands = []
    for i, ln in enumerate(lens):
        ln = int(ln)
        if rng.random() < share:
            kind = i % 4
            base = np.resize(qa, ln + 2)
            if kind == 0:
                row = base[:ln].copy()
                e = int(rng.integers(0, 14))
                if e:
                    row[rng.integers(0, ln, size=e)] = synth.ALNUM[rng.integers(0, 62, size=e)]
            elif kind == 1:
                row = np.delete(base, int(rng.integers(0, min(30, ln))))[:ln].copy()
            elif kind == 2:
                row = np.insert(base, int(rng.integers(0, min(30, ln))), np.uint8(35))[:ln].copy()
            else:
                row = synth.ALNUM[rng.integers(0, 62, size=ln)]
                h = min(int(rng.integers(10, 121)), ln)
                row[:h] = base[:h]
        else:
            row = synth.ALNUM[rng.integers(0, 62, size=ln)]
        cands.append(row.tobytes())
    return rf.ragged(cands)


@pytest.mark.parametrize("env", [None, {"RF_BAND_RUNS": "0"}, {"RF_BAND_RUN_MIN_TILES": "256"}, {"RF_BAND_RUN_MIN_TILES": "256", "RF_BAND_DEFER_ADAPT": "0"},
                                 {"RF_BAND_RUN_MIN_TILES": "256", "RF_UNSCATTER_MIN": "1", "RF_BAND_DEFER_ADAPT": "0"}, {"RF_BAND_RUN_MIN_TILES": "1", "RF_BAND_DEFER_ADAPT": "0"},
                                 {"RF_BAND_RUN_MIN_TILES": "1", "RF_BAND_DEFER_ADAPT": "0", "RF_BAND_DEFER_MAX": "63", "RF_DEVICE_PACK_MIN": "1"}])
def test_small_band_scan_of_a_bucketed_corpus_walks_its_long_runs_as_single_length_corpora(env):
    """rf_api_scan.hip launch_band_runs: the long runs of exact tiles of a length-bucketed corpus go through launch_band as single-length corpora of their own (results
    through run_orig), with the hand-over of sparse tiles; short runs and the one-length views keep band_kernel<false>.  Query 200 / 300, 13 long runs + two short ones +
    leftovers, 5 % and 60 % near candidates, distance cutoffs across the band kernel's range and the normalized ops that ride it -- against the oracle.  Runs of one length
    count from 32768 tiles on (their launch sequence has a fixed price), so the in-process case is the tiles kernel; the child processes lower the line: runs of 312 tiles
    with the launch form following the last report and always handing over, through the slot-ordered temporary + gather, every run a run, every tile with a lane left
    handed over; and the runs off."""
    if env is not None:
        if os.environ.get("RF_TEST_CHILD") == "1":
            pytest.skip("the child runs the in-process case only")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "test_small_band_scan_of_a_bucketed_corpus and None"],
                           capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RF_TEST_CHILD="1", **env))
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:]
        return
    for qlen, share in ((200, 0.05), (300, 0.6)):
        q = synth.query(qlen, 0xB0C2 + qlen)
        data, offsets = _bucketed_long_corpus(q, seed=qlen, share=share)
        corpus = rf.Corpus.from_ragged(data, offsets)
        bc, ob = rf.distance.levenshtein.BatchComparator(q), o.levenshtein.BatchComparator(q)
        for k in (2, 8, 17, 31):
            for _rep in range(2):  # (the second call may take the other launch form: the stream's last report)
                got = bc.distance_many(corpus, score_cutoff=k)
                exp = ob.many(N.OP_DISTANCE, data, offsets, nthreads=8, score_cutoff=k)
                bad = _same(got, exp)
                assert len(bad) == 0, (qlen, share, k, _rep, len(bad), bad[:5], got[bad[:5]], exp[bad[:5]])
        for op, cut in ((N.OP_NORMALIZED_SIMILARITY, 0.95), (N.OP_NORMALIZED_DISTANCE, 0.08)):
            got = bc.many(op, corpus, score_cutoff=cut)
            exp = ob.many(op, data, offsets, nthreads=8, score_cutoff=cut)
            assert len(_same(got, exp)) == 0, (qlen, share, op, cut)
        idx, val = bc.filter_many(N.OP_DISTANCE, corpus, score_cutoff=8)
        idx_e, val_e = _some(ob.many(N.OP_DISTANCE, data, offsets, nthreads=8, score_cutoff=8))
        assert np.array_equal(idx, idx_e) and np.array_equal(val, val_e)
