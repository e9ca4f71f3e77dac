"""rf_topk_u32 with and without a score_hint (wall time per call, synchronous API): 100 M candidates of 64 symbols with 40 planted
near-duplicates of the query (0..9 substitutions), top-16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth
n = int(os.environ.get("AB_N", 100_000_000))
rows = synth.rows_device(n, 64, seed=1)
q = synth.query(64, 0xC0FFEE02)
qa = torch.from_numpy(np.frombuffer(q, dtype=np.uint8).copy()).cuda()
rng = np.random.default_rng(3)
for r, edits in zip(rng.choice(n, size=40, replace=False), list(range(10)) * 4):
    row = qa.clone()
    row[torch.from_numpy(rng.choice(64, size=edits, replace=False)).cuda()] = 126
    rows[int(r)] = row
corpus = rf.Corpus.from_device_rows(rows); del rows
bc = rf.distance.levenshtein.BatchComparator(q)
base = None
for hint in (None, 0, 1, 3, 8, 16, 1000):
    kw = {} if hint is None else {"score_hint": hint}
    for _ in range(3): s, i = bc.topk(corpus, 16, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): s, i = bc.topk(corpus, 16, **kw)
    dt = (time.perf_counter() - t0) / 10
    if base is None: base = (s.copy(), i.copy())
    same = np.array_equal(base[0], s) and np.array_equal(base[1], i)
    print(f"score_hint={hint}: {dt * 1e3:7.3f} ms per call  ({n / dt / 1e9:7.1f} Gpairs/s)  16th best distance {int(s[-1])}  same result: {same}")
