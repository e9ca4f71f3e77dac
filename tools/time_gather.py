"""Times the ragged Indel step (a fast, memory-bound scan: the gather's share is largest) for the gather kernel's tuning knobs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth
n = int(os.environ.get("AB_N", 50_000_000))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
lens = torch.randint(1, 65, (n,), device=dev, generator=g)
rows = synth.rows_device(n, 64, seed=3, device=dev)
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lens, 0)
flat = torch.empty(int(off[-1]), dtype=torch.uint8, device=dev)
col = torch.arange(64, device=dev)[None, :]
for a in range(0, n, 1 << 23):
    b = min(n, a + (1 << 23))
    flat[int(off[a]):int(off[b])] = rows[a:b][col < lens[a:b, None]]
corpus = rf.Corpus.from_ragged(flat.cpu().numpy(), off.cpu().numpy().astype(np.uint64))
del rows, flat
out = torch.empty(n, dtype=torch.int32, device=dev)
bc = rf.distance.indel.BatchComparator(synth.query(64, 2))
for _ in range(20): bc.distance_many(corpus, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): bc.distance_many(corpus, out=out)
e1.record(); torch.cuda.synchronize()
print(f"windows={os.environ.get('RF_GATHER_WINDOWS','1')} variant={os.environ.get('RF_GATHER_VARIANT','0')}: {e0.elapsed_time(e1)/20:.4f} ms per step (n={n})")
