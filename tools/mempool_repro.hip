// Is hipMallocAsync / hipFreeAsync safe across host threads with one stream each (ROCm 7.2, gfx950)?  Each thread: allocate from the
// default pool on its stream, fill with its own pattern, verify on the same stream, free on the stream -- while other threads do the
// same and (mode 1) a further thread calls hipMalloc / hipMemcpy / hipFree in a loop.  Prints the number of corrupted buffers.
//   hipcc --offload-arch=gfx950 -O2 -o build/plain/mempool_repro tools/mempool_repro.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>
__global__ void fill(uint32_t* p, size_t n, uint32_t pat) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pat + (uint32_t)i; }
__global__ void slow(uint32_t* p, size_t n, int spins) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t v = p[i]; for (int s = 0; s < spins; ++s) v = v * 1664525u + 1013904223u; if (v == 0x12345u) p[i] = v; } }
__global__ void check(const uint32_t* p, size_t n, uint32_t pat, unsigned long long* bad) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != pat + (uint32_t)i) atomicAdd(bad, 1ull); }
int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0, threads = 8, iters = 300;
    std::atomic<unsigned long long> corrupted{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            (void)hipSetDevice(0);
            hipStream_t st;
            (void)hipStreamCreate(&st);
            unsigned long long* d_bad;
            (void)hipMalloc(&d_bad, 8);
            (void)hipMemset(d_bad, 0, 8);
            for (int it = 0; it < iters; ++it) {
                const size_t n = (size_t)(1 + (it * 7 + t) % 5) << 19;  // 2 .. 10 MB
                uint32_t* p = nullptr;
                if (hipMallocAsync((void**)&p, n * 4, st) != hipSuccess) { corrupted += 1000000; break; }
                const uint32_t pat = (uint32_t)(t * 1000003 + it * 7919);
                fill<<<512, 256, 0, st>>>(p, n, pat);
                slow<<<512, 256, 0, st>>>(p, n, 200);
                check<<<512, 256, 0, st>>>(p, n, pat, d_bad);
                (void)hipFreeAsync(p, st);
                if (it % 3 == 0) (void)hipStreamSynchronize(st);
            }
            (void)hipStreamSynchronize(st);
            unsigned long long h = 0;
            (void)hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
            corrupted += h;
            (void)hipFree(d_bad);
            (void)hipStreamDestroy(st);
        });
    std::thread noise;
    if (mode == 1)
        noise = std::thread([&] {
            (void)hipSetDevice(0);
            std::vector<char> host(1 << 20);
            while (!stop.load()) {
                void* q = nullptr;
                (void)hipMalloc(&q, 4 << 20);
                (void)hipMemcpy(q, host.data(), host.size(), hipMemcpyHostToDevice);
                (void)hipFree(q);
            }
        });
    for (auto& x : th) x.join();
    stop = true;
    if (noise.joinable()) noise.join();
    printf("mode %d: %llu corrupted words\n", mode, corrupted.load());
    return corrupted.load() ? 1 : 0;
}
