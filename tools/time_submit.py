"""Host-side cost of submitting one ragged step (no synchronization inside the loop) against its GPU time, for a small and a large
corpus: does the gather path's stream-ordered temporary (hipMallocAsync / hipFreeAsync per call) keep the submission asynchronous?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth
for n in (2_000_000, int(os.environ.get("AB_N", 50_000_000))):
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 65, size=n).astype(np.uint64)
    offsets = np.zeros(n + 1, dtype=np.uint64); offsets[1:] = np.cumsum(lens)
    data = synth.ALNUM[rng.integers(0, 62, size=int(offsets[-1]))]
    corpus = rf.Corpus.from_ragged(data, offsets)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    for metric in ("indel", "levenshtein"):
        bc = getattr(rf.distance, metric).BatchComparator(synth.query(64, 2))
        for _ in range(20): bc.distance_many(corpus, out=out)
        torch.cuda.synchronize()
        reps = 200 if n < 10_000_000 else 40
        t0 = time.perf_counter()
        for _ in range(reps): bc.distance_many(corpus, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n} {metric}: submit {1e6 * (t1 - t0) / reps:.1f} us per call, done after {1e6 * (t2 - t0) / reps:.1f} us per call  (RF_UNSCATTER_MIN={os.environ.get('RF_UNSCATTER_MIN', 'default')})")
    del corpus
