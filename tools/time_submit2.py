"""Submission cost against completion time of the device-output entry points (is every one of them asynchronous?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth
n = 20_000_000
rows = synth.rows_device(n, 64, seed=1)
uni = rf.Corpus.from_device_rows(rows); del rows
rng = np.random.default_rng(5)
lens = rng.integers(1, 65, size=n).astype(np.uint64)
offsets = np.zeros(n + 1, dtype=np.uint64); offsets[1:] = np.cumsum(lens)
rag = rf.Corpus.from_ragged(synth.ALNUM[rng.integers(0, 62, size=int(offsets[-1]))], offsets)
q = synth.query(64, 2)
out32 = torch.empty(4 * n, dtype=torch.int32, device="cuda")
out64 = torch.empty(n, dtype=torch.float64, device="cuda")
keys = torch.empty(64, dtype=torch.int64, device="cuda")
L, I, J, O = (getattr(rf.distance, m).BatchComparator(q) for m in ("levenshtein", "indel", "jaro_winkler", "osa"))
qs = [rf.distance.levenshtein.BatchComparator(synth.query(64, s)) for s in range(4)]
qi = [rf.distance.indel.BatchComparator(synth.query(64, s)) for s in range(4)]
cases = {
    "uniform lev many": lambda: L.distance_many(uni, out=out32),
    "uniform lev cutoff3": lambda: L.distance_many(uni, out=out32, score_cutoff=3),
    "uniform indel cutoff12": lambda: I.distance_many(uni, out=out32, score_cutoff=12),
    "uniform jw many": lambda: J.similarity_many(uni, out=out64),
    "uniform jw cutoff0.9": lambda: J.similarity_many(uni, out=out64, score_cutoff=0.9),
    "uniform lev topk keys": lambda: L.topk_keys_device(uni, 16, keys),
    "uniform lev topk keys cutoff3": lambda: L.topk_keys_device(uni, 16, keys, score_cutoff=3),
    "uniform multi4": lambda: rf.distance.levenshtein.BatchComparator.many_multi(qs, N.OP_DISTANCE, uni, out=out32),
    "ragged multi4 lev": lambda: rf.distance.levenshtein.BatchComparator.many_multi(qs, N.OP_DISTANCE, rag, out=out32),
    "ragged multi4 indel": lambda: rf.distance.indel.BatchComparator.many_multi(qi, N.OP_DISTANCE, rag, out=out32),
    "ragged lev many": lambda: L.distance_many(rag, out=out32),
    "ragged osa many": lambda: O.distance_many(rag, out=out32),
    "ragged indel many": lambda: I.distance_many(rag, out=out32),
    "ragged jw many": lambda: J.similarity_many(rag, out=out64),
    "ragged lev cutoff3": lambda: L.distance_many(rag, out=out32, score_cutoff=3),
    "ragged lev cutoff20": lambda: L.distance_many(rag, out=out32, score_cutoff=20),
    "ragged indel cutoff12": lambda: I.distance_many(rag, out=out32, score_cutoff=12),
    "ragged lev topk keys": lambda: L.topk_keys_device(rag, 16, keys),
    "ragged jw cutoff0.9": lambda: J.similarity_many(rag, out=out64, score_cutoff=0.9),
}
for name, fn in cases.items():
    try:
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        flag = "  <-- submission waits" if (t1 - t0) > 0.5 * (t2 - t0) else ""
        print(f"{name:32s} submit {1e6 * (t1 - t0) / 20:8.1f} us   done {1e6 * (t2 - t0) / 20:8.1f} us{flag}")
    except Exception as exc:
        print(f"{name:32s} {type(exc).__name__}: {exc}")
