// rf_pack.hip -- device rows -> packed tiles (pack_rows_kernel) and the byte histogram behind the symbol renaming.
#include "rf_device.hpp"

namespace rf {

// ---------------------------------------------------------------------------------------------------
// corpus packing on the device: row-major fixed-length rows -> chunk-interleaved tiles
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t* __restrict__ rows, size_t n, uint32_t len,
                                                        size_t stride, uint8_t* __restrict__ packed, uint32_t n_tiles,
                                                        const uint8_t* __restrict__ sigma)
{
    __shared__ uint8_t lds_sigma[256];
    lds_sigma[threadIdx.x] = sigma[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t chunks = (len + kChunk - 1) / kChunk;
    const size_t tile_bytes = (size_t)chunks * kWave * kChunk;
    for (size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += (size_t)gridDim.x * 4) {
        const size_t row = t * kWave + lane;
        uint8_t* dst = packed + t * tile_bytes + (size_t)lane * kChunk;
        const uint8_t* src = rows + row * stride;
        for (uint32_t c = 0; c < chunks; ++c) {
            uint32_t w[4] = {0, 0, 0, 0};
            if (row < n) {
                const uint32_t base = c * kChunk;
                uint32_t raw[4] = {0, 0, 0, 0};
                uint32_t nb = min((uint32_t)kChunk, len - base);
                if (nb == kChunk && ((reinterpret_cast<uintptr_t>(src + base) & 3) == 0)) {
                    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + base);
                    raw[0] = s4[0];
                    raw[1] = s4[1];
                    raw[2] = s4[2];
                    raw[3] = s4[3];
                } else {
                    for (uint32_t b = 0; b < nb; ++b) raw[b / 4] |= (uint32_t)src[base + b] << (8 * (b % 4));
                }
#pragma unroll
                for (uint32_t b = 0; b < (uint32_t)kChunk; ++b)  // rename; bytes past the candidate's end stay 0
                    if (b < nb) w[b / 4] |= (uint32_t)lds_sigma[(raw[b / 4] >> (8 * (b % 4))) & 0xFFu] << (8 * (b % 4));
            }
            *reinterpret_cast<uint4*>(dst + (size_t)c * kWave * kChunk) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// byte histogram of (a prefix of) device rows, for the rename permutation
__global__ __launch_bounds__(256) void histogram_rows_kernel(const uint8_t* __restrict__ rows, size_t n, uint32_t len, size_t stride,
                                                             unsigned long long* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (size_t)gridDim.x * blockDim.x) {
        const uint8_t* src = rows + r * stride;
        for (uint32_t b = 0; b < len; ++b) atomicAdd(&h[src[b]], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

hipError_t launch_histogram_rows(const uint8_t* rows, size_t n, uint32_t len, size_t stride, unsigned long long* hist, hipStream_t stream)
{
    if (n == 0 || len == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(histogram_rows_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, len, stride, hist);
    return hipGetLastError();
}

hipError_t launch_pack_rows(const uint8_t* rows, size_t n, uint32_t len, size_t stride, uint8_t* packed, uint32_t n_tiles,
                            const uint8_t* sigma, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n_tiles + 3) / 4, 256 * 16);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, len, stride, packed, n_tiles, sigma);
    return hipGetLastError();
}

}  // namespace rf
