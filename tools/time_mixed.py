"""Times the distinct-lengths corpus of VERDICT r1 #7 (10 000 candidates, lengths 1..10 000) against a dense corpus of the same
bytes, with mixed tiles (default) and with the round-1 layout (RF_NO_MIXED_TILES=1 in the environment of a second run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth

n = 10_000
lens = np.arange(1, n + 1, dtype=np.uint64)
offsets = np.zeros(n + 1, dtype=np.uint64); offsets[1:] = np.cumsum(lens)
data = synth.ALNUM[np.random.default_rng(3).integers(0, 62, size=int(offsets[-1]))]
ragged = rf.Corpus.from_ragged(data, offsets)
rows = int(offsets[-1]) // 5000
dense = rf.Corpus.from_device_rows(torch.from_numpy(data[: rows * 5000].reshape(rows, 5000).copy()).cuda())
def timed(bc, corpus, out, **kw):
    for _ in range(2): bc.many(0, corpus, out=out, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): bc.many(0, corpus, out=out, **kw)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
layout = "round-1 layout (RF_NO_MIXED_TILES)" if os.environ.get("RF_NO_MIXED_TILES") else "mixed tiles"
print(f"{layout}: packed {ragged.device_bytes / 1e6:.1f} MB for {int(offsets[-1]) / 1e6:.1f} MB of payload")
for metric, qlen in (("levenshtein", 64), ("indel", 64), ("levenshtein", 256)):
    bc = getattr(rf.distance, metric).BatchComparator(synth.query(qlen, 9))
    tr = timed(bc, ragged, torch.empty(n, dtype=torch.int32, device="cuda"))
    td = timed(bc, dense, torch.empty(rows, dtype=torch.int32, device="cuda"))
    print(f"  {metric:12s} query {qlen:4d}: ragged {tr:8.3f} ms   dense {td:8.3f} ms   ratio {tr / td:5.2f}")
