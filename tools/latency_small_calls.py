"""Latency of SMALL calls (BASELINE configs[0]: query 32 x 10 000 candidates of <= 64 symbols), one call at a time:
  python tools/latency_small_calls.py          (RF_LIB=<other build> for an A/B)"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch

import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth

data, offsets = synth.ragged_host(10_000, 64, seed=3, min_len=1)
corpus = rf.Corpus.from_ragged(data, offsets)
bc = rf.distance.levenshtein.BatchComparator(synth.query(32, 4))
dev_out = torch.empty(10_000, dtype=torch.int32, device="cuda")


def timed(fn, n=1000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def dev_sync():
    bc.distance_many(corpus, out=dev_out)
    torch.cuda.synchronize()


print("results left on the device, calls queued back to back : %6.1f us per call" % timed(lambda: bc.distance_many(corpus, out=dev_out)))
print("results left on the device, synchronized after each   : %6.1f us per call" % timed(dev_sync))
print("results on the host (numpy array)                     : %6.1f us per call" % timed(lambda: bc.distance_many(corpus)))
print("top-10 on the host                                    : %6.1f us per call" % timed(lambda: bc.topk(corpus, 10), 500))
print("one candidate (scorer.distance(s2))                   : %6.1f us per call" % timed(lambda: bc.distance(b"hello world"), 300))
