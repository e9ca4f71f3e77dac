"""Ad-hoc: cost of the k-entry all-gather + merge at world size 1 (RCCL loopback)."""
import os, sys, time
sys.path.insert(0, ".")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
a = torch.arange(16, dtype=torch.int64, device=dev); b = torch.empty(16, dtype=torch.int64, device=dev)
big = torch.empty(400_000_000, dtype=torch.uint8, device=dev)
imax = torch.tensor(2**63 - 1, dtype=torch.int64, device=dev)

def timeit(name, fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter()-t0)*1e6/reps:8.1f} us")

timeit("all_gather_into_tensor sync", lambda: dist.all_gather_into_tensor(b, a))
def asy():
    w = dist.all_gather_into_tensor(b, a, async_op=True); w.wait()
timeit("all_gather_into_tensor async+wait", asy)
timeit("where+sort", lambda: torch.sort(torch.where(b < 0, imax, b)).values[:16])
timeit("fill 400MB (2.8ms-ish kernel)", lambda: big.fill_(1), reps=20)
def both():
    big.fill_(1); w = dist.all_gather_into_tensor(b, a, async_op=True); big.fill_(2); w.wait(); torch.sort(torch.where(b < 0, imax, b))
timeit("fill + gather(async) + fill + merge", both, reps=20)
dist.destroy_process_group()
