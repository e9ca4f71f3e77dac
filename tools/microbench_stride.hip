// HBM read efficiency of the cutoff scans' access pattern: every wavefront reads 1 KiB (16 B per lane) at tile * STRIDE.
// STRIDE = 1 KiB is a contiguous stream, 4 KiB is "first chunk row of every len-64 tile" (what an early-out scan reads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ data, uint32_t n_tiles, uint32_t stride_u4, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (uint32_t t = blockIdx.x * 4 + wave; t < n_tiles; t += gridDim.x * 4) {
        const uint4 v = data[(size_t)t * stride_u4 + lane];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t bytes = 6400000000ull;
    uint4* d; uint32_t* o;
    (void)hipMalloc(&d, bytes + 4096); (void)hipMalloc(&o, 64);
    (void)hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid_per_cu : {8, 32}) for (uint32_t stride : {1024u, 2048u, 4096u, 8192u, 16384u}) {
        const uint32_t n_tiles = (uint32_t)(bytes / stride);
        const uint32_t use = stride == 1024 ? n_tiles / 4 : n_tiles;  // same number of KiB read as the 4 KiB case or fewer
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_read, dim3(256 * grid_per_cu), dim3(256), 0, 0, d, use, stride / 16, o);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid %2d/CU stride %5u B: %u KiB-reads in %.3f ms = %.2f TB/s of useful bytes\n", grid_per_cu, stride, use, ms, use * 1024.0 / (ms * 1e9));
        }
    }
    return 0;
}
