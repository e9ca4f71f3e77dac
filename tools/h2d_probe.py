import time, numpy as np, torch, os
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
n = 3_250_000_000
a = np.random.default_rng(0).integers(0, 255, size=n, dtype=np.uint8)
t = torch.from_numpy(a)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); d.copy_(t); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pageable copy_: {dt:.3f} s  {n/dt/1e9:.1f} GB/s")
rt = torch.cuda.cudart()
t0 = time.perf_counter(); r = rt.cudaHostRegister(a.ctypes.data, n, 0); dt = time.perf_counter() - t0
print(f"hostRegister rc={r}: {dt:.3f} s")
for rep in range(2):
    t0 = time.perf_counter(); d.copy_(t, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"registered copy_: {dt:.3f} s  {n/dt/1e9:.1f} GB/s")
t0 = time.perf_counter(); rt.cudaHostUnregister(a.ctypes.data); print(f"unregister {time.perf_counter()-t0:.3f} s")
# threaded memcpy into pinned
import threading
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pn = pin.numpy()
for T in (4, 8, 16, 32):
    def work(k):
        lo, hi = n * k // T, n * (k + 1) // T
        pn[lo:hi] = a[lo:hi]
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"host memcpy into pinned, {T} threads: {dt:.3f} s {n/dt/1e9:.1f} GB/s")
t0 = time.perf_counter(); d.copy_(pin, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"pinned copy_: {dt:.3f} s  {n/dt/1e9:.1f} GB/s")
