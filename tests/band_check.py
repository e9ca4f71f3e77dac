"""Run by tests/test_gpu_parity.py::test_multiword_levenshtein_band_trimming in a subprocess (the forced-path modes are environment switches
the library reads once).

The multi-word Levenshtein scans (stream_levw{2,3,4}_*_kernel, tools/gen_stream_asm.py BlockKind; the compiled LevState<W>) do not run the
words whose rows lie outside the Ukkonen band of a chunk of columns (the reference's trimming, levenshtein.rs:810-825, :906-985).  That is
exact only if no optimal path leaves the band, so the corpus here is built to walk the band's EDGES: block shifts of the query by 63..65,
127..129 and len1 / 2 - 1 .. len1 / 2 + 1 symbols (the optimal path runs along the diagonal |i - j| = shift), cut or padded to every candidate
length 1..300, plus 4-symbol random strings (many equally good paths, wandering), prefixes / suffixes of the query, and plain random rows.
Queries of 100 / 192 / 200 / 256 symbols (the asm scans) and 320 / 449 / 512 (the compiled LevState<5..8>, whose step() takes the same band); every candidate length 1..300 (1..len1 + 90 for the longer queries; 70 of each: one exact tile + leftovers for the mixed tiles);
single-length corpora of a few lengths for the uniform kernels; no cutoff, distance cutoffs that reach the asm scans (they narrow the
band), similarity and the normalized ops.  Every value is compared with the oracle.

Exit status 0 = all equal.  With RF_SCAN_BLOCKS_PER_CU_FULL=1 a wavefront walks several tiles (band state re-armed per tile).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import rapidfuzz_rs_amd as rf  # noqa: E402
from rapidfuzz_rs_amd import _native as N  # noqa: E402
from oracle import oracle as o  # noqa: E402

NONE32, U64MAX = np.uint32(0xFFFFFFFF), np.uint64(0xFFFFFFFFFFFFFFFF)
OPS = {"distance": N.OP_DISTANCE, "similarity": N.OP_SIMILARITY, "normalized_distance": N.OP_NORMALIZED_DISTANCE, "normalized_similarity": N.OP_NORMALIZED_SIMILARITY}
failures = 0


def candidate(rng, q, len2, kind):
    """one candidate of length len2 around query q (uint8 array)"""
    len1 = len(q)
    if kind < 9:  # block shift: q[sh:] + q[:sh], resized to len2 (cut, or continued with the rotation again)
        sh = (63, 64, 65, 127, 128, 129, len1 // 2 - 1, len1 // 2, len1 // 2 + 1)[kind] % len1
        row = np.resize(np.roll(q, -sh), len2)
    elif kind == 9:  # tail of the query first, then noise
        sh = int(rng.integers(1, len1))
        row = np.resize(np.concatenate([q[sh:], rng.integers(48, 123, size=700, dtype=np.uint8)]), len2)
    elif kind == 10:  # noise first, then the head of the query
        k = int(rng.integers(0, len2 + 1))
        row = np.concatenate([rng.integers(48, 123, size=k, dtype=np.uint8), np.resize(q, len2 - k)])
    elif kind == 11:  # the query with scattered edits
        row = np.resize(q, len2).copy()
        row[rng.integers(0, len2, size=int(rng.integers(0, 12)))] = 35
    elif kind == 12:
        row = rng.integers(97, 101, size=len2, dtype=np.uint8)  # 4 symbols
    else:
        row = rng.integers(48, 123, size=len2, dtype=np.uint8)
    return row.astype(np.uint8).copy()


def same(got, exp):
    if got.dtype == np.uint32:
        exp = np.where(exp == U64MAX, NONE32, exp.astype(np.uint32))
        return np.nonzero(got != exp)[0]
    return np.nonzero(~((got == exp) | (np.isnan(got) & np.isnan(exp))))[0]


def check(tag, q, corpus, host=None, ragged=None):
    global failures
    bc, ob = rf.distance.levenshtein.BatchComparator(q), o.levenshtein.BatchComparator(q)

    def expect(op, **kw):
        return ob.rows(op, host, nthreads=8, **kw) if host is not None else ob.many(op, ragged[0], ragged[1], nthreads=8, **kw)

    bad_ops = []
    for opname, op in OPS.items():
        bad = same(bc.many(op, corpus), expect(op))
        if len(bad):
            bad_ops.append((opname, len(bad), bad[:4].tolist()))
    len1 = len(q)
    for cut in sorted({40, len1 // 3, len1 // 2 + 40, (3 * len1) // 4, (7 * len1) // 8, len1 - 1, len1, len1 + 30, 299, 1000}):  # (the smaller ones: the early-out kernels)
        got, exp = bc.many(N.OP_DISTANCE, corpus, score_cutoff=cut), expect(N.OP_DISTANCE, score_cutoff=cut)
        bad = same(got, exp)
        if len(bad):
            bad_ops.append((f"cutoff {cut}", len(bad), bad[:4].tolist(), got[bad[:4]].tolist(), exp[bad[:4]].tolist()))
    # normalized ops under loose f64 cutoffs (no early-out kernel: single-length corpora take the u32 asm scan + one normalizing pass, rf_api_scan.hip run_many)
    for opname, cut in (("normalized_distance", 0.9), ("normalized_distance", 0.7), ("normalized_similarity", 0.1), ("normalized_similarity", 0.35)):
        got, exp = bc.many(OPS[opname], corpus, score_cutoff=cut), expect(OPS[opname], score_cutoff=cut)
        bad = same(got, exp)
        if len(bad):
            bad_ops.append((f"{opname} cutoff {cut}", len(bad), bad[:4].tolist(), got[bad[:4]].tolist(), exp[bad[:4]].tolist()))
    # the in-scan / via-scores top-16 over the same scans
    exp = expect(N.OP_DISTANCE)
    order = np.lexsort((np.arange(len(exp)), exp))[:16]
    s, i = bc.topk(corpus, 16)
    if list(zip(s.tolist(), i.tolist())) != [(int(exp[j]), int(j)) for j in order]:
        bad_ops.append(("topk16",))
    print(f"{tag} len1={len1}: {'ok' if not bad_ops else bad_ops}", flush=True)
    failures += len(bad_ops)


rng = np.random.default_rng(5)
mode = sys.argv[1] if len(sys.argv) > 1 else "ragged"
for len1 in (100, 192, 200, 256, 320, 449, 512):  # W = 2 .. 4: the asm scans; 5 .. 8: the compiled LevState<W>
    q = np.random.default_rng(len1).integers(48, 123, size=len1, dtype=np.uint8)
    if mode == "rows":
        for len2 in (64, 100, 129, 200, 256, 300) + ((449, 512, 600) if len1 > 256 else ()):
            host = np.stack([candidate(rng, q, len2, r % 14) for r in range(3000)])
            corpus = rf.Corpus.from_device_rows(torch.from_numpy(host).cuda())
            check(f"rows len2={len2}", bytes(q), corpus, host=host)
            del corpus
    else:
        rows = [candidate(rng, q, len2, r % 14) for len2 in range(1, max(301, len1 + 91)) for r in range(70)]
        order = rng.permutation(len(rows))
        rows = [rows[i] for i in order]
        offsets = np.zeros(len(rows) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(r) for r in rows])
        data = np.concatenate(rows)
        corpus = rf.Corpus.from_ragged(data, offsets)
        check(f"ragged 1..{max(300, len1 + 90)}", bytes(q), corpus, ragged=(data, offsets))
        del corpus
print("FAILURES", failures)
sys.exit(1 if failures else 0)
