"""normalized_similarity vs distance of the multi-word Levenshtein scans (query 256): a single-length corpus (10 M x 256) and a length-bucketed one (10 M candidates,
lengths uniform in [129, 256]).  RF_NORM_TWO_STEP=0 keeps the compiled f64 scan (the A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth

n = int(os.environ.get("AB_N", 10_000_000))
q = synth.query(256, 0xC0FFEE02)
bc = rf.distance.levenshtein.BatchComparator(q)
for kind in ("single-length", "ragged"):
    if kind == "single-length":
        rows = synth.rows_device(n, 256, seed=1); corpus = rf.Corpus.from_device_rows(rows); del rows
    else:
        rng = np.random.default_rng(5)
        lens = rng.integers(129, 257, size=n).astype(np.uint64)
        offsets = np.zeros(n + 1, dtype=np.uint64); offsets[1:] = np.cumsum(lens)
        data = synth.ALNUM[rng.integers(0, 62, size=int(offsets[-1]))]
        corpus = rf.Corpus.from_ragged(data, offsets); del data
    for opname, op, dt in (("distance", N.OP_DISTANCE, torch.int32), ("normalized_similarity", N.OP_NORMALIZED_SIMILARITY, torch.float64)):
        out = torch.empty(n, dtype=dt, device="cuda")
        for _ in range(5): bc.many(op, corpus, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): bc.many(op, corpus, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"two_step={os.environ.get('RF_NORM_TWO_STEP', '1')} {kind:14s} {opname:22s} {ms:7.3f} ms {n/ms/1e6:7.2f} Gpairs/s  chk {int(out.view(torch.int64 if dt == torch.float64 else torch.int32).sum().item()) & 0xFFFFFFFFFFFF:012x}", flush=True)
        del out
    del corpus
