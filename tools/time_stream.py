"""Times rf_stream_many_u32 on a corpus file in /tmp (page-cache resident): the PCIe-inclusive rate of the streamed path.
  python tools/time_stream.py [candidates] [segment MiB ...]     RF_STREAM_TIMING=1 prints the phases of every call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
segs = [int(x) << 20 for x in sys.argv[2:]] or [256 << 20, 1 << 30]
# link ceiling of this box: pinned host -> device
h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
d.copy_(h, non_blocking=True); torch.cuda.synchronize()
t0 = time.time()
for _ in range(4): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
print(f"link ceiling (pinned host -> device, 1 GiB copies): {4 * (1 << 30) / (time.time() - t0) / 1e9:.1f} GB/s")
del h, d
path = "/tmp/stream_test.rfc"
q = synth.query(64, 2)
bc = rf.distance.levenshtein.BatchComparator(q)
if n <= 400_000_000:
    rows = synth.rows_device(n, 64, seed=1)
    corpus = rf.Corpus.from_device_rows(rows)
    del rows
    t0 = time.time(); corpus.save(path); t_save = time.time() - t0
    ref = bc.distance_many(corpus)
    del corpus
else:  # (too big to keep rows + packed + results around comfortably: check a prefix instead)
    rows = synth.rows_device(n, 64, seed=1)
    corpus = rf.Corpus.from_device_rows(rows)
    del rows
    torch.cuda.empty_cache()
    t0 = time.time(); corpus.save(path); t_save = time.time() - t0
    ref = bc.distance_many(corpus)
    del corpus
torch.cuda.empty_cache()
size = os.path.getsize(path)
for seg in segs:
    for rep in range(3):
        t0 = time.time(); got = bc.stream_many(N.OP_DISTANCE, path, n, segment_bytes=seg); dt = time.time() - t0
        print(f"n={n} file {size / 1e9:.1f} GB segment={seg >> 20} MiB call {rep}: {dt*1e3:.1f} ms  {n/dt/1e9:.3f} Gpairs/s  {n*64/dt/1e9:.2f} GB/s payload  equal={bool((got == ref).all())}  (save {t_save:.1f} s)", flush=True)
os.remove(path)
