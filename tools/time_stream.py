"""Times rf_stream_many_u32 on a corpus file in /tmp (page-cache resident): the PCIe-inclusive rate of the streamed path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rows = synth.rows_device(n, 64, seed=1)
corpus = rf.Corpus.from_device_rows(rows)
del rows
path = "/tmp/stream_test.rfc"
t0 = time.time(); corpus.save(path); t_save = time.time() - t0
q = synth.query(64, 2)
bc = rf.distance.levenshtein.BatchComparator(q)
ref = bc.distance_many(corpus)
del corpus
torch.cuda.empty_cache()
for seg in (256 << 20, 1 << 30):
    for rep in range(2):
        t0 = time.time(); got = bc.stream_many(N.OP_DISTANCE, path, n, segment_bytes=seg); dt = time.time() - t0
    print(f"n={n} segment={seg >> 20} MiB: {dt*1e3:.1f} ms  {n/dt/1e9:.3f} Gpairs/s  {n*64/dt/1e9:.2f} GB/s payload  equal={bool((got == ref).all())}  (save {t_save:.1f} s)")
t0 = time.time(); c2 = rf.Corpus.load(path); print(f"load: {time.time()-t0:.2f} s for {os.path.getsize(path)/1e9:.2f} GB")
os.remove(path)
