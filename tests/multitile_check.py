"""Run by tests/test_gpu_parity.py::test_asm_kernels_many_tiles_per_wavefront in a subprocess with RF_SCAN_BLOCKS_PER_CU_FULL=1.

With the default grid (256 workgroups per CU) a wavefront of a full scan owns a second tile only beyond 16.8 M candidates, so
the small parity tests never reach the hand-scheduled kernels' multi-tile machinery: the fetch ring running across tile
boundaries, the state re-arm, the cursor parked on the last valid chunk, tail chunks entered in the middle of the block, the
tile queue.  One workgroup per CU makes the grid 1024 wavefronts; with ~300 k candidates every wavefront walks >= 4 tiles.
Every value of every op is compared with the oracle.

Exit status 0 = all equal.  Prints one line per (shape, metric) so a failure names its kernel.
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

assert os.environ.get("RF_SCAN_BLOCKS_PER_CU_FULL") == "1", "run with RF_SCAN_BLOCKS_PER_CU_FULL=1"
OPS = {"distance": N.OP_DISTANCE, "similarity": N.OP_SIMILARITY, "normalized_distance": N.OP_NORMALIZED_DISTANCE, "normalized_similarity": N.OP_NORMALIZED_SIMILARITY}
NONE32, U64MAX = np.uint32(0xFFFFFFFF), np.uint64(0xFFFFFFFFFFFFFFFF)
GPU = {"levenshtein": rf.distance.levenshtein, "osa": rf.distance.osa, "indel": rf.distance.indel, "jaro": rf.distance.jaro, "jaro_winkler": rf.distance.jaro_winkler}
ORA = {"levenshtein": o.levenshtein, "osa": o.osa, "indel": o.indel, "jaro": o.jaro, "jaro_winkler": o.jaro_winkler}
failures = 0


def same(got, exp):
    if got.dtype == np.uint32:
        exp = np.where(exp == U64MAX, NONE32, exp.astype(np.uint32))
        return np.nonzero(got != exp)[0]
    return np.nonzero(~((got == exp) | (np.isnan(got) & np.isnan(exp))))[0]


def plant(rng, host, q, every):
    n, len2 = host.shape
    qa = np.frombuffer(q, dtype=np.uint8)
    for r in range(0, n, every):
        row = np.resize(qa, len2).copy()
        row[rng.integers(0, len2, size=r % 5)] = 122
        if r % 2:
            row = np.roll(row, 1 + r % 3)
        if r % 3 == 0 and len2 >= 4:
            row[[1, 2]] = row[[2, 1]]
        host[r] = row


def check(tag, corpus, host=None, ragged=None):
    """every metric / op of the hand-scheduled family against the oracle, plus the in-scan top-16"""
    global failures
    for len1 in (64, 20):  # 64-bit and 32-bit Levenshtein kernels
        q = QUERIES[len1]
        for metric in ("levenshtein", "osa", "indel", "jaro", "jaro_winkler"):
            if metric in ("jaro", "jaro_winkler", "osa", "indel") and len1 == 20 and tag.startswith("rows"):
                continue  # (one query length is enough for the kernels that do not switch on it)
            bc, ob = GPU[metric].BatchComparator(q), ORA[metric].BatchComparator(q)
            bad_ops = []
            for opname, op in OPS.items():
                got = bc.many(op, corpus)
                exp = ob.rows(op, host, nthreads=8) if host is not None else ob.many(op, ragged[0], ragged[1], nthreads=8)
                bad = same(got, exp)
                if len(bad):
                    bad_ops.append((opname, len(bad), bad[:4].tolist(), got[bad[:4]].tolist(), exp[bad[:4]].tolist()))
            if metric in ("levenshtein", "osa"):
                exp = ob.rows(N.OP_DISTANCE, host, nthreads=8) if host is not None else ob.many(N.OP_DISTANCE, ragged[0], ragged[1], nthreads=8)
                order = np.lexsort((np.arange(len(exp)), exp))[:16]
                s, i = bc.topk(corpus, 16)
                if list(zip(s.tolist(), i.tolist())) != [(int(exp[j]), int(j)) for j in order]:
                    bad_ops.append(("topk16", list(zip(s.tolist(), i.tolist()))[:4], [(int(exp[j]), int(j)) for j in order][:4]))
            print(f"{tag} len1={len1} {metric}: {'ok' if not bad_ops else bad_ops}", flush=True)
            failures += len(bad_ops)


    # multi-word Levenshtein (queries of 65 .. 256 symbols: the W = 2 / 4 / 4-word asm scans of rf_stream_asm.hip; round 4)
    for len1 in (100, 200, 256):
        q = QUERIES[len1]
        bc, ob = GPU["levenshtein"].BatchComparator(q), ORA["levenshtein"].BatchComparator(q)
        bad_ops = []
        for opname, op in OPS.items():
            got = bc.many(op, corpus)
            exp = ob.rows(op, host, nthreads=8) if host is not None else ob.many(op, ragged[0], ragged[1], nthreads=8)
            bad = same(got, exp)
            if len(bad):
                bad_ops.append((opname, len(bad), bad[:4].tolist(), got[bad[:4]].tolist(), exp[bad[:4]].tolist()))
        print(f"{tag} len1={len1} levenshtein (multi-word): {'ok' if not bad_ops else bad_ops}", flush=True)
        failures += len(bad_ops)

    # four queries fused per pass (scan_multi_kernel): 64-bit and 32-bit Levenshtein, LCS-family
    for metric, len1 in (("levenshtein", 64), ("levenshtein", 20), ("indel", 64)):
        base = QUERIES[len1]
        qs = [base, base[::-1], base[3:] + base[:3], base[: len1 // 2] + base[: len1 - len1 // 2]]
        cs = [GPU[metric].BatchComparator(q) for q in qs]
        got = GPU[metric].BatchComparator.many_multi(cs, N.OP_DISTANCE, corpus)
        bad_rows = []
        for j, q in enumerate(qs):
            ob = ORA[metric].BatchComparator(q)
            exp = ob.rows(N.OP_DISTANCE, host, nthreads=8) if host is not None else ob.many(N.OP_DISTANCE, ragged[0], ragged[1], nthreads=8)
            bad = same(got[j], exp)
            if len(bad):
                bad_rows.append((j, len(bad), bad[:4].tolist()))
        print(f"{tag} len1={len1} {metric} x4 fused: {'ok' if not bad_rows else bad_rows}", flush=True)
        failures += len(bad_rows)


rng = np.random.default_rng(20260929)
QUERIES = {64: bytes(rng.integers(48, 123, size=64, dtype=np.uint8)), 20: bytes(rng.integers(97, 123, size=20, dtype=np.uint8))}
for _len1 in (100, 200, 256):  # (a separate generator: the corpora below stay what they were)
    QUERIES[_len1] = bytes(np.random.default_rng(_len1).integers(48, 123, size=_len1, dtype=np.uint8))
n = int(os.environ.get("RF_MULTITILE_N", "300007"))
mode = sys.argv[1] if len(sys.argv) > 1 else "rows"
if mode == "rows":
    for len2 in (16, 32, 48, 64, 96, 160, 250):
        host = rng.integers(48, 123, size=(n if len2 < 200 else n // 3, len2), dtype=np.uint8)
        plant(rng, host, QUERIES[64], 997)
        plant(rng, host[500:], QUERIES[20], 991)
        plant(rng, host[250:], QUERIES[200], 1993)
        corpus = rf.Corpus.from_device_rows(torch.from_numpy(host).cuda())
        check(f"rows len2={len2}", corpus, host=host)
        del corpus
else:  # ragged corpora: every length 0..max_len (tails of every size), exact tiles of each length and a mixed section
    for max_len in (70, 33):
        lens = rng.integers(0, max_len + 1, size=n)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        data = rng.integers(48, 123, size=int(offsets[-1]), dtype=np.uint8)
        for r in range(0, n, 499):  # near-duplicates of both queries, cut to the candidate's length
            q = np.frombuffer(QUERIES[64 if r % 2 else 20], dtype=np.uint8)
            a, b = int(offsets[r]), int(offsets[r + 1])
            row = np.resize(q, b - a).copy() if b > a else np.zeros(0, dtype=np.uint8)
            if len(row) > 3:
                row[rng.integers(0, len(row), size=r % 4)] = 122
            data[a:b] = row
        corpus = rf.Corpus.from_ragged(data, offsets)
        check(f"ragged max_len={max_len}", corpus, ragged=(data, offsets))
        del corpus
print("FAILURES", failures)
sys.exit(1 if failures else 0)
