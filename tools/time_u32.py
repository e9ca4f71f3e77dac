"""Times the two u32 paths on a CJK-like corpus: shared byte image vs per-call translated image (overflow symbols in the query)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N

n, ln = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 32
rng = np.random.default_rng(1)
w = 1.0 / np.arange(1, 3001) ** 1.05
cdf = np.cumsum(w / w.sum())
data = (0x4E00 + np.searchsorted(cdf, rng.random(n * ln))).astype(np.uint32)
offsets = np.arange(0, n * ln + 1, ln, dtype=np.uint64)
t0 = time.time(); corpus = rf.Corpus.from_ragged_u32(data, offsets); t_pack = time.time() - t0
own, overflow = corpus.alphabet_size()
counts = np.bincount(data - 0x4E00, minlength=3000)
order = np.argsort(-counts, kind="stable")
common = (0x4E00 + order[:200]).astype(np.uint32); rare = (0x4E00 + order[1500:1600]).astype(np.uint32)
q_common = rng.choice(common, size=32).astype(np.uint32)
q_rare = q_common.copy(); q_rare[[3, 17]] = rare[[5, 42]]
out = torch.empty(n, dtype=torch.int32, device="cuda")
print(f"n={n} x {ln} symbols, alphabet {own} + {overflow} overflow, pack {t_pack:.1f} s, device {corpus.device_bytes/1e9:.2f} GB")
for name, q in (("alphabet-only query", q_common), ("query with 2 overflow symbols", q_rare)):
    bc = rf.distance.levenshtein.BatchComparator(q)
    for _ in range(2): bc.distance_many(corpus, out=out)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): bc.distance_many(corpus, out=out)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms  {n/dt/1e9:.1f} Gpairs/s  min distance {int(out.min())}")
