// Streaming-read microbenchmark for the head-plane access pattern: every wavefront walks rows of 64 x B bytes (B = 8 or 16 per lane)
// with a grid stride, D rows in flight, nontemporal loads.  tools/bin/membw (built by: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/bin/membw)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <class V, int D, bool kScramble>
__global__ __launch_bounds__(256) void walk(const V* __restrict__ src, size_t rows, uint32_t* __restrict__ sink)
{
    // kScramble: row r is read at (r * odd) mod rows (rows a power of two): the same bytes in an order without DRAM page locality
    auto at = [&](size_t r) { return kScramble ? (r * 2654435761ull) & (rows - 1) : r; };
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t t = (size_t)blockIdx.x * 4 + wave;
    V buf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) buf[d] = __builtin_nontemporal_load(src + at(t + d * stride < rows ? t + d * stride : 0) * 64 + lane);
    uint32_t acc = 0;
    for (; t < rows; t += stride) {
        const V cur = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) buf[d] = buf[d + 1];
        const size_t tn = t + D * stride;
        buf[D - 1] = __builtin_nontemporal_load(src + at(tn < rows ? tn : 0) * 64 + lane);
        acc += cur.x ^ cur.y;
        if (__ballot(acc == 0x12345678u)) acc += 1;  // (a wave-uniform branch per row, like the scans' dead-tile test)
    }
    if (acc == 0xDEADBEEFu) sink[0] = acc;
}
template <class V, int D, bool kScramble = false>
static void run(const void* d, size_t bytes, int wg_per_cu, uint32_t* sink)
{
    size_t rows = bytes / (64 * sizeof(V));
    if (kScramble) { size_t p2 = 1; while (p2 * 2 <= rows) p2 *= 2; rows = p2; bytes = rows * 64 * sizeof(V); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<V, D, kScramble>), dim3(256 * wg_per_cu), dim3(256), 0, 0, (const V*)d, rows, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((walk<V, D, kScramble>), dim3(256 * wg_per_cu), dim3(256), 0, 0, (const V*)d, rows, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s%2zu B/lane  depth %d  %3d workgroups/CU: %7.3f ms per pass  %6.2f TB/s\n", kScramble ? "scrambled rows " : "", sizeof(V), D, wg_per_cu, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
}
int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 8e9);
    void* d; uint32_t* sink;
    hipMalloc(&d, bytes + 4096); hipMemset(d, 1, bytes + 4096); hipMalloc(&sink, 64);
    for (int wg : {8, 32, 128}) {
        run<v2u, 1>(d, bytes, wg, sink); run<v2u, 2>(d, bytes, wg, sink); run<v2u, 4>(d, bytes, wg, sink);
        run<v4u, 1>(d, bytes, wg, sink); run<v4u, 2>(d, bytes, wg, sink); run<v4u, 4>(d, bytes, wg, sink);
    }
    for (int wg : {8, 32}) {
        run<v2u, 2, true>(d, bytes, wg, sink); run<v4u, 2, true>(d, bytes, wg, sink);
    }
    return 0;
}
