// rf_pack_ragged.hip -- the device half of rf_corpus_pack for ragged host input (round 6; VERDICT r5 item 3 / missing #6).
//
// north_star: "the host builds ... a length-bucketed packed candidate corpus".  Until round 5 the whole layout was a host counting sort (rf_api.hip build_layout:
// serial length and slot passes, a byte-by-byte threaded scatter, then one upload of the packed image).  What the layout IS does not change -- rf_corpus_layout_host
// stays the specification, and tests/test_gpu_filter.py::test_device_packer_* holds the device-packed corpus to it byte for byte through rf_corpus_save -- but the
// raw bytes and offsets now go up as they are and the GPU does the per-candidate work:
//   lengths      ragged_lengths_kernel: len[i] = offsets[i + 1] - offsets[i] as a sort key, the length histogram (LDS bins, then global), input validation
//   sample       ragged_byte_hist_kernel: the byte histogram of exactly the 64 KiB blocks the host packer samples (the symbol renaming must come out the same)
//   order        hipcub radix sort of (length, index) pairs over the significant bits of the longest length -- stable, so every length bucket keeps its candidates
//                in original order, which is what the host's `next++` pass computes one candidate at a time
//   scatter      ragged_scatter_tiles_kernel / ragged_scatter_mixed_kernel: one wavefront per DESTINATION tile -- lane r finds its candidate in the sorted order,
//                reads it 16 bytes at a time (unaligned loads from wherever it lies in the input), renames the symbols through the LDS copy of sigma, zeroes what
//                lies behind the candidate's end and stores chunk k of all 64 lanes as one contiguous KiB; orig[] / mixed_orig[] go out with it
// Everything between those kernels -- groups, tile descriptors, the mixed section and its views -- is a function of the length histogram alone and stays on the
// host (rf_api.hip plan_device_layout: microseconds).
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "rf_device.hpp"

namespace rf {

namespace {

constexpr uint32_t kLenBins = 4096;  // lengths below this are counted in LDS first

__global__ __launch_bounds__(256) void ragged_lengths_kernel(const uint64_t* __restrict__ offsets, uint32_t n, uint32_t max_len_allowed, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, unsigned long long* __restrict__ counts, uint32_t* __restrict__ status)
{
    __shared__ uint32_t bins[kLenBins];
    for (uint32_t b = threadIdx.x; b < kLenBins; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    uint32_t longest = 0, bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = offsets[i], b = offsets[i + 1];
        uint32_t len = 0;
        if (b < a)
            bad |= 1;  // offsets decrease: an error
        else if (b - a > max_len_allowed)
            bad |= 2;  // longer than this path's tables go: the host packer takes the corpus
        else
            len = (uint32_t)(b - a);
        keys[i] = len;
        vals[i] = (uint32_t)i;
        longest = max(longest, len);
        if (len < kLenBins)
            atomicAdd(&bins[len], 1u);
        else
            atomicAdd(&counts[len], 1ull);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kLenBins; b += blockDim.x)
        if (bins[b]) atomicAdd(&counts[b], (unsigned long long)bins[b]);
    if (longest) atomicMax(&status[0], longest);
    if (bad) atomicOr(&status[1], bad);
}

// the host packer's sample (rf_api.hip build_layout 3b): blocks of 64 KiB every `stride` blocks from `first`; one workgroup per sampled block
__global__ __launch_bounds__(256) void ragged_byte_hist_kernel(const uint8_t* __restrict__ bytes, uint64_t first, uint64_t total, uint64_t stride, unsigned long long* __restrict__ hist)
{
    constexpr uint64_t kBlock = 1u << 16;
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b0 = first + (uint64_t)blockIdx.x * kBlock * stride, e = min(total, b0 + kBlock);
    for (uint64_t b = b0 + threadIdx.x; b < e; b += blockDim.x) atomicAdd(&h[bytes[b]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

typedef uint32_t u4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

// chunk k of one candidate: 16 bytes from src + 16 k (the input buffer carries 16 readable bytes of padding), renamed, zero behind the candidate's end
__device__ __forceinline__ uint4 renamed_chunk(const uint8_t* src, uint32_t len, uint32_t k, const uint8_t* lds_sigma)
{
    const uint32_t base = k * kChunk;
    if (base >= len) return make_uint4(0, 0, 0, 0);
    const u4_unaligned raw = *reinterpret_cast<const u4_unaligned*>(src + base);
    const uint32_t in[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t w[4] = {0, 0, 0, 0};
    const uint32_t nb = min((uint32_t)kChunk, len - base);
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)kChunk; ++b)
        if (b < nb) w[b / 4] |= (uint32_t)lds_sigma[(in[b / 4] >> (8 * (b % 4))) & 0xFFu] << (8 * (b % 4));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// one wavefront per EXACT tile.  group tables (by distinct length, ascending): g_start = the bucket's first position in the sorted order, g_slot0 = its first
// slot, g_in_exact = how many of its candidates live in exact tiles (all of them in the identity layout, whose last tile is partial).  by_len: length -> group.
__global__ __launch_bounds__(256) void ragged_scatter_tiles_kernel(const uint8_t* __restrict__ bytes, uint64_t first, const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ sorted_idx,
                                                                   const TileDesc* __restrict__ tiles, uint32_t uniform_len, uint32_t n_exact, const uint32_t* __restrict__ by_len,
                                                                   const uint32_t* __restrict__ g_start, const uint32_t* __restrict__ g_slot0,
                                                                   const uint32_t* __restrict__ g_in_exact, const uint8_t* __restrict__ sigma, uint8_t* __restrict__ packed,
                                                                   uint32_t* __restrict__ orig)
{
    __shared__ uint8_t lds_sigma[256];
    lds_sigma[threadIdx.x] = sigma[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWave - 1);
    for (uint32_t t = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; t < n_exact; t += gridDim.x * kWavesPerBlock) {
        const uint32_t len = tiles ? tiles[t].len : uniform_len;
        const uint32_t slot0 = tiles ? tiles[t].slot0 : t * kWave;
        const uint64_t data_off = tiles ? tiles[t].data_off : (uint64_t)t * (((len + kChunk - 1) / kChunk) * kWave * kChunk);
        const uint32_t g = by_len[len];
        const uint32_t k_in_bucket = slot0 - g_slot0[g] + lane;
        const bool real = k_in_bucket < g_in_exact[g];
        const uint32_t i = real ? sorted_idx[g_start[g] + k_in_bucket] : kPad;
        if (orig) orig[slot0 + lane] = i;
        const uint8_t* src = bytes + (real ? offsets[i] - first : 0);
        const uint32_t my_len = real ? len : 0u;
        const uint32_t nch = (len + kChunk - 1) / kChunk;
        uint4* dst = reinterpret_cast<uint4*>(packed + data_off) + lane;
        for (uint32_t k = 0; k < nch; ++k) dst[(size_t)k * kWave] = renamed_chunk(src, my_len, k, lds_sigma);
    }
}

// one wavefront per MIXED tile: lane = pool position m * 64 + lane; pool_len / pool_spos (position in the sorted order) / pool_vslot0 (first slot of the one-length
// view this leftover is seen through) per pool position; the payload block is sized for the tile's longest candidate
__global__ __launch_bounds__(256) void ragged_scatter_mixed_kernel(const uint8_t* __restrict__ bytes, uint64_t first, const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ sorted_idx,
                                                                   const MixedDesc* __restrict__ mixed, uint32_t n_mixed, uint32_t pool_n, const uint32_t* __restrict__ pool_len,
                                                                   const uint32_t* __restrict__ pool_spos, const uint32_t* __restrict__ pool_vslot0,
                                                                   const uint8_t* __restrict__ sigma, uint8_t* __restrict__ packed, uint32_t* __restrict__ orig,
                                                                   uint32_t* __restrict__ mixed_orig)
{
    __shared__ uint8_t lds_sigma[256];
    lds_sigma[threadIdx.x] = sigma[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWave - 1);
    for (uint32_t m = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; m < n_mixed; m += gridDim.x * kWavesPerBlock) {
        const uint32_t q = m * kWave + lane;
        const bool real = q < pool_n;
        const uint32_t len = real ? pool_len[q] : 0u;
        const uint32_t i = real ? sorted_idx[pool_spos[q]] : kPad;
        mixed_orig[q] = i;
        if (real) orig[pool_vslot0[q] + lane] = i;
        const uint8_t* src = bytes + (real ? offsets[i] - first : 0);
        const uint32_t nch = (mixed[m].max_len + kChunk - 1) / kChunk;
        uint4* dst = reinterpret_cast<uint4*>(packed + mixed[m].data_off) + lane;
        for (uint32_t k = 0; k < nch; ++k) dst[(size_t)k * kWave] = renamed_chunk(src, len, k, lds_sigma);
    }
}

// the second tile order (rf_api.hip tiles_by_origin): the non-empty exact tiles [z, n_exact) ordered by their first candidate's original index -- keys and places
// here, a radix sort, then the descriptors gathered through the sorted places
__global__ void tile_first_index_kernel(const uint32_t* __restrict__ orig, const TileDesc* __restrict__ tiles, uint32_t z, uint32_t count, uint32_t* __restrict__ keys,
                                        uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        keys[i] = orig[tiles[z + i].slot0];
        vals[i] = z + i;
    }
}
__global__ void tile_gather_kernel(const TileDesc* __restrict__ tiles, const uint32_t* __restrict__ place, uint32_t z, uint32_t count, TileDesc* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[z + i] = tiles[place[i]];
}

}  // namespace

// out = tiles with [z, n_exact) in origin order (out already holds a copy of tiles); keys / vals / keys2 / vals2: n_exact u32 each
hipError_t launch_tiles_by_origin(const uint32_t* orig, const TileDesc* tiles, uint32_t z, uint32_t n_exact, uint32_t* keys, uint32_t* vals, uint32_t* keys2, uint32_t* vals2,
                                  void* temp, size_t temp_bytes, TileDesc* out, hipStream_t st)
{
    if (n_exact <= z) return hipSuccess;
    const uint32_t count = n_exact - z;
    const dim3 b(256), g((count + 255) / 256);
    hipLaunchKernelGGL(tile_first_index_kernel, g, b, 0, st, orig, tiles, z, count, keys, vals);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys2, vals, vals2, (int)count, 0, 32, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tile_gather_kernel, g, b, 0, st, tiles, vals2, z, count, out);
    return hipGetLastError();
}

hipError_t launch_ragged_lengths(const uint64_t* offsets, uint32_t n, uint32_t max_len_allowed, uint32_t* keys, uint32_t* vals, unsigned long long* counts, uint32_t* status,
                                 hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ragged_lengths_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 8192u)), dim3(256), 0, st, offsets, n, max_len_allowed, keys, vals, counts, status);
    return hipGetLastError();
}
hipError_t launch_ragged_byte_hist(const uint8_t* bytes, uint64_t first, uint64_t total, uint64_t stride, unsigned long long* hist, hipStream_t st)
{
    if (total <= first) return hipSuccess;
    const uint64_t block = 1u << 16, blocks = (total - first + block * stride - 1) / (block * stride);
    hipLaunchKernelGGL(ragged_byte_hist_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, bytes, first, total, stride, hist);
    return hipGetLastError();
}
size_t ragged_sort_temp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 32, nullptr);
    return std::max<size_t>((bytes + 255) / 256 * 256, 256);
}
hipError_t launch_ragged_sort(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, uint32_t bits, void* temp, size_t temp_bytes,
                              hipStream_t st)
{
    if (n == 0) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, (int)std::max(1u, bits), st);
}
hipError_t launch_ragged_scatter_tiles(const uint8_t* bytes, uint64_t first, const uint64_t* offsets, const uint32_t* sorted_idx, const TileDesc* tiles, uint32_t uniform_len, uint32_t n_exact,
                                       const uint32_t* by_len, const uint32_t* g_start, const uint32_t* g_slot0, const uint32_t* g_in_exact, const uint8_t* sigma, uint8_t* packed,
                                       uint32_t* orig, hipStream_t st)
{
    if (n_exact == 0) return hipSuccess;
    const uint32_t grid = std::min<uint32_t>((n_exact + kWavesPerBlock - 1) / kWavesPerBlock, 65536u);
    hipLaunchKernelGGL(ragged_scatter_tiles_kernel, dim3(grid), dim3(256), 0, st, bytes, first, offsets, sorted_idx, tiles, uniform_len, n_exact, by_len, g_start, g_slot0, g_in_exact, sigma,
                       packed, orig);
    return hipGetLastError();
}
hipError_t launch_ragged_scatter_mixed(const uint8_t* bytes, uint64_t first, const uint64_t* offsets, const uint32_t* sorted_idx, const MixedDesc* mixed, uint32_t n_mixed, uint32_t pool_n,
                                       const uint32_t* pool_len, const uint32_t* pool_spos, const uint32_t* pool_vslot0, const uint8_t* sigma, uint8_t* packed, uint32_t* orig,
                                       uint32_t* mixed_orig, hipStream_t st)
{
    if (n_mixed == 0) return hipSuccess;
    const uint32_t grid = std::min<uint32_t>((n_mixed + kWavesPerBlock - 1) / kWavesPerBlock, 65536u);
    hipLaunchKernelGGL(ragged_scatter_mixed_kernel, dim3(grid), dim3(256), 0, st, bytes, first, offsets, sorted_idx, mixed, n_mixed, pool_n, pool_len, pool_spos, pool_vslot0, sigma, packed,
                       orig, mixed_orig);
    return hipGetLastError();
}

}  // namespace rf
