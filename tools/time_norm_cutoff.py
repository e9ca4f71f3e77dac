"""normalized_similarity >= 0.9 of a Levenshtein scan, query 256 x 10 M rows x 256 with a share of near-duplicates: the small-band kernel under the implied raw cutoff +
the normalizing pass (default) against the compiled f64 early-out scan (RF_NORM_BAND=0).  Run both on one box: for b in 1 0; do RF_NORM_BAND=$b python tools/time_norm_cutoff.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth
n, ln = 10_000_000, 256
q = synth.query(ln, 5)
for share in (0.0, 0.01, 0.5):
    rows = synth.rows_device(n, ln, seed=5, device=torch.device("cuda", 0))
    if share > 0:
        gen = torch.Generator(device="cuda"); gen.manual_seed(1)
        pidx = torch.nonzero(torch.rand(n, device="cuda", generator=gen) < share).flatten()
        qa = torch.from_numpy(np.frombuffer(q, dtype=np.uint8).copy()).cuda()
        rows[pidx] = qa
        # a few substitutions
        cols = torch.randint(0, ln, (len(pidx), 4), device="cuda", generator=gen)
        rows[pidx.unsqueeze(1), cols] = 35
    corpus = rf.Corpus.from_device_rows(rows)
    del rows
    bc = rf.distance.levenshtein.BatchComparator(q)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        bc.many(N.OP_NORMALIZED_SIMILARITY, corpus, score_cutoff=0.9, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        bc.many(N.OP_NORMALIZED_SIMILARITY, corpus, score_cutoff=0.9, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"RF_NORM_BAND={os.environ.get('RF_NORM_BAND','1')} share {share}: normalized_similarity >= 0.9, query 256 x 10 M x 256: {n / dt / 1e9:.2f} Gpairs/s ({dt * 1e3:.3f} ms), Somes {int((~torch.isnan(out)).sum())}", flush=True)
