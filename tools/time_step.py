"""Where does the sharded bench step lose time against the library call alone?  Variants of the step, same process, same corpus."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N, parallel
from rapidfuzz_rs_amd.utils import synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 100_000_000
q = synth.query(64, 0xC0FFEE02)
seed = int(sys.argv[1], 0) if len(sys.argv) > 1 else 1
rows = synth.rows_device(n, 64, seed=seed)
corpus = rf.Corpus.from_device_rows(rows)
del rows
if len(sys.argv) > 2: torch.cuda.empty_cache()
print("seed", hex(seed), "empty_cache", len(sys.argv) > 2)
bc = rf.distance.levenshtein.BatchComparator(q)
out = torch.empty(n, dtype=torch.int32, device="cuda")
lk = [torch.empty(16, dtype=torch.int64, device="cuda") for _ in range(2)]
ak = [torch.empty(16, dtype=torch.int64, device="cuda") for _ in range(2)]
mk = torch.empty(16, dtype=torch.int64, device="cuda")
comp = torch.cuda.Stream(); xchg = torch.cuda.Stream(priority=-1)
torch.cuda.set_stream(comp)
def run(name, fn, reps=20):
    for _ in range(3): fn(0); fn(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(i & 1)
    torch.cuda.synchronize(); print(f"{name:48s} {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms")
cs = comp.cuda_stream
run("plain distance_many", lambda b: bc.distance_many(corpus, out=out, stream=cs))
run("topk_keys_device + out", lambda b: bc.topk_keys_device(corpus, 16, lk[b], out=out, stream=cs))
def v_event(b):
    bc.topk_keys_device(corpus, 16, lk[b], out=out, stream=cs)
    e = torch.cuda.Event(); e.record(comp)
run("  + event record", v_event)
def v_side(b):
    bc.topk_keys_device(corpus, 16, lk[b], out=out, stream=cs)
    e = torch.cuda.Event(); e.record(comp)
    with torch.cuda.stream(xchg):
        xchg.wait_event(e)
        ak[b].copy_(lk[b])
        parallel.merge_keys_device(ak[b], 16, mk, stream=xchg.cuda_stream)
run("  + side stream: copy + merge", v_side)
def v_rccl(b):
    bc.topk_keys_device(corpus, 16, lk[b], out=out, stream=cs)
    e = torch.cuda.Event(); e.record(comp)
    with torch.cuda.stream(xchg):
        xchg.wait_event(e)
        dist.all_gather_into_tensor(ak[b], lk[b])
        parallel.merge_keys_device(ak[b], 16, mk, stream=xchg.cuda_stream)
run("  + side stream: RCCL all_gather + merge", v_rccl)
def v_inline(b):
    bc.topk_keys_device(corpus, 16, lk[b], out=out, stream=cs)
    parallel.merge_keys_device(lk[b], 16, mk, stream=cs)
run("  + merge on the scan stream (no side stream)", v_inline)
bf = [None, None]
def v_bench(b):
    if bf[b] is not None and not bf[b].query():
        bf[b].synchronize()
    bc.topk_keys_device(corpus, 16, lk[b], rf._native.OP_DISTANCE if hasattr(rf, "_native") else 0, rf.Args(), index_base=0, out=out, stream=cs)
    e = torch.cuda.Event(); e.record(comp)
    with torch.cuda.stream(xchg):
        xchg.wait_event(e)
        dist.all_gather_into_tensor(ak[b], lk[b])
        parallel.merge_keys_device(ak[b], 16, mk, stream=xchg.cuda_stream)
        bf[b] = torch.cuda.Event(); bf[b].record(xchg)
run("  bench step verbatim", v_bench)
run("  bench step verbatim, again", v_bench)
dist.barrier(); torch.cuda.synchronize()
run("  bench step verbatim, after a barrier", v_bench)
run("  bench step verbatim, again (no barrier)", v_bench)
time.sleep(0.001); run("  bench step verbatim, after 1 ms host sleep", v_bench)
time.sleep(0.02); run("  bench step verbatim, after 20 ms host sleep", v_bench)
run("  bench step verbatim, again", v_bench)
run("plain distance_many again", lambda b: bc.distance_many(corpus, out=out, stream=cs))
dist.destroy_process_group()
