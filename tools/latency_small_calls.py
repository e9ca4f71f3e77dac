import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth
data, offsets = synth.ragged_host(10_000, 64, seed=3, min_len=1)
corpus = rf.Corpus.from_ragged(data, offsets)
bc = rf.distance.levenshtein.BatchComparator(synth.query(32, 4))
for _ in range(20): bc.distance_many(corpus)
t0 = time.perf_counter()
for _ in range(500): out = bc.distance_many(corpus)
t = (time.perf_counter() - t0) / 500
print("host-result distance_many, 10k ragged: %.1f us per call" % (t * 1e6))
t0 = time.perf_counter()
for _ in range(500): s = bc.topk(corpus, 10)
print("host-result topk(10): %.1f us per call" % ((time.perf_counter() - t0) / 500 * 1e6))
t0 = time.perf_counter()
for _ in range(200): v = bc.distance(b"hello world")
print("distance(one candidate): %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
