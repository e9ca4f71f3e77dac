import sys, os, torch
sys.path.insert(0, os.getcwd())
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth
n = 100_000_000
rows = synth.rows_device(n, 64, seed=1); corpus = rf.Corpus.from_device_rows(rows); del rows
for metric, qlen in (("levenshtein", 64), ("levenshtein", 32), ("indel", 64), ("osa", 64)):
    bc = getattr(rf.distance, metric).BatchComparator(synth.query(qlen, 0xC0FFEE02))
    for opname, op, dt in (("distance", N.OP_DISTANCE, torch.int32), ("normalized_distance", N.OP_NORMALIZED_DISTANCE, torch.float64), ("normalized_similarity", N.OP_NORMALIZED_SIMILARITY, torch.float64)):
        out = torch.empty(n, dtype=dt, device="cuda")
        for _ in range(10): bc.many(op, corpus, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): bc.many(op, corpus, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{metric:12s} q{qlen} {opname:22s} {ms:7.3f} ms {n/ms/1e6:7.2f} Gpairs/s", flush=True)
        del out
