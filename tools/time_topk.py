"""Ad-hoc timing of the scan variants on one GPU (not part of the product)."""
import sys, time
sys.path.insert(0, ".")
import torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth

n = 100_000_000
q = synth.query(64, 0xC0FFEE02)
rows = synth.rows_device(n, 64, seed=1)
corpus = rf.Corpus.from_device_rows(rows)
del rows
bc = rf.distance.levenshtein.BatchComparator(q)
out = torch.empty(n, dtype=torch.int32, device="cuda")
keys = torch.empty(16, dtype=torch.int64, device="cuda")


def timeit(name, fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:40s} gpu {e0.elapsed_time(e1)/reps:7.3f} ms   wall {(time.perf_counter()-t0)*1e3/reps:7.3f} ms")


timeit("distance_many", lambda: bc.distance_many(corpus, out=out))
timeit("topk_keys_device (no out)", lambda: bc.topk_keys_device(corpus, 16, keys))
timeit("topk_keys_device + out", lambda: bc.topk_keys_device(corpus, 16, keys, out=out))
timeit("topk (host result) + out", lambda: bc.topk(corpus, 16, out=out))
timeit("distance_many cutoff 3", lambda: bc.distance_many(corpus, out=out, score_cutoff=3))
timeit("topk_keys_device cutoff 3", lambda: bc.topk_keys_device(corpus, 16, keys, score_cutoff=3))

import os
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
allk = torch.empty(16, dtype=torch.int64, device="cuda")
timeit("after nccl init: distance_many", lambda: bc.distance_many(corpus, out=out))
timeit("after nccl init: topk_keys + out", lambda: bc.topk_keys_device(corpus, 16, keys, out=out))
def with_gather():
    bc.topk_keys_device(corpus, 16, keys, out=out)
    dist.all_gather_into_tensor(allk, keys)
timeit("topk_keys + out + all_gather(sync op)", with_gather)
pend = [None]
def with_gather_async():
    bc.topk_keys_device(corpus, 16, keys, out=out)
    if pend[0] is not None: pend[0].wait()
    pend[0] = dist.all_gather_into_tensor(allk, keys, async_op=True)
timeit("topk_keys + out + all_gather(async)", with_gather_async)
dist.destroy_process_group()
