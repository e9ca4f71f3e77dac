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

// ---------------------------------------------------------------------------------------------------
// u32 corpora with an overflow class: a query that contains overflow symbols gets its own byte image of the corpus.
// `raw` is parallel to the packed payload (raw[x] = the u32 symbol behind packed byte x, 0xFFFFFFFF in padding); every
// symbol of the query maps to its query-local id (1..r), everything else to 0 -- which is all a metric on this path
// needs to know about a candidate symbol.  The (symbol -> id) pairs arrive as an open-addressing table of `cap` slots
// (power of two, key 0xFFFFFFFF = empty) and are staged in LDS; 16 symbols in, one 16-byte chunk out per thread.
// ---------------------------------------------------------------------------------------------------
// 16 raw symbols of one output chunk (Sym = uint32_t or, for corpora inside the Basic Multilingual Plane, uint16_t)
template <class Sym>
__device__ __forceinline__ void load_raw16(const Sym* raw, uint64_t x, uint32_t (&sym)[16])
{
    if constexpr (sizeof(Sym) == 4) {
        const uint4* src = reinterpret_cast<const uint4*>(raw) + x * 4;
        const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
        const uint32_t v[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) sym[i] = v[i];
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(raw) + x * 2;
        const uint4 a = src[0], b = src[1];
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sym[2 * i] = v[i] & 0xFFFFu;
            sym[2 * i + 1] = v[i] >> 16;
        }
    }
}

template <class Sym>
__global__ __launch_bounds__(256) void translate_kernel(const Sym* __restrict__ raw, uint64_t n_chunks, const uint32_t* __restrict__ keys,
                                                        const uint8_t* __restrict__ vals, uint32_t cap, uint4* __restrict__ out)
{
    extern __shared__ uint32_t lds_keys[];  // cap keys, then cap ids (one byte each)
    uint8_t* lds_vals = reinterpret_cast<uint8_t*>(lds_keys + cap);
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) {
        lds_keys[i] = keys[i];
        lds_vals[i] = vals[i];
    }
    __syncthreads();
    const uint32_t mask = cap - 1;
    auto map = [&](uint32_t sym) -> uint32_t {
        uint32_t h = (sym * 2654435761u) & mask;
        while (true) {
            const uint32_t k = lds_keys[h];
            if (k == sym) return lds_vals[h];
            if (k == 0xFFFFFFFFu) return 0;
            h = (h + 1) & mask;
        }
    };
    constexpr uint32_t kPadSym = sizeof(Sym) == 4 ? 0xFFFFFFFFu : 0xFFFFu;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_chunks; x += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t sym[16], o[4] = {0, 0, 0, 0};
        load_raw16<Sym>(raw, x, sym);
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i / 4] |= (sym[i] == kPadSym ? 0u : map(sym[i])) << (8 * (i % 4));
        out[x] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// The same with a DIRECT table for the Basic Multilingual Plane (one id byte per code point, 64 KiB of LDS per
// workgroup, two workgroups per CU): one LDS byte read per symbol and no probe loop, whose trip count every lane of a
// wavefront would otherwise share.  Symbols above 0xFFFF (rare) still go through the small hash table, read from global.
template <class Sym>
__global__ __launch_bounds__(256) void translate_direct_kernel(const Sym* __restrict__ raw, uint64_t n_chunks, const uint32_t* __restrict__ keys,
                                                               const uint8_t* __restrict__ vals, uint32_t cap, uint4* __restrict__ out)
{
    extern __shared__ uint32_t lds_direct[];  // 16384 words = 65536 id bytes
    uint8_t* table = reinterpret_cast<uint8_t*>(lds_direct);
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds_direct[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) {
        const uint32_t k = keys[i];
        if (k < 0x10000u) table[k] = vals[i];
    }
    __syncthreads();
    const uint32_t mask = cap - 1;
    auto map = [&](uint32_t sym) -> uint32_t {
        if (sym < 0x10000u) return table[sym];
        if (sym == 0xFFFFFFFFu) return 0;
        uint32_t h = (sym * 2654435761u) & mask;
        while (true) {
            const uint32_t k = keys[h];
            if (k == sym) return vals[h];
            if (k == 0xFFFFFFFFu) return 0;
            h = (h + 1) & mask;
        }
    };
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_chunks; x += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t sym[16], o[4] = {0, 0, 0, 0};
        load_raw16<Sym>(raw, x, sym);  // all loads in flight before the lookups
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i / 4] |= map(sym[i]) << (8 * (i % 4));
        out[x] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <class Sym>
static hipError_t launch_translate_t(const Sym* raw, uint64_t n_chunks, const uint32_t* keys, const uint8_t* vals, uint32_t cap, uint4* out, hipStream_t stream)
{
    static const bool direct = [] { const char* e = getenv("RF_TRANSLATE_DIRECT"); return !e || atoi(e) != 0; }();  // A/B switch
    if (direct && n_chunks >= 1024) {  // (below that the table set-up -- 64 KiB per workgroup -- is the larger part)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(translate_direct_kernel<Sym>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return e;
        const int grid = (int)std::min<uint64_t>((n_chunks + 255) / 256, 512);  // two resident workgroups per CU
        hipLaunchKernelGGL(translate_direct_kernel<Sym>, dim3(grid), dim3(256), 65536, stream, raw, n_chunks, keys, vals, cap, out);
        return hipGetLastError();
    }
    const int grid = (int)std::min<uint64_t>((n_chunks + 255) / 256, (uint64_t)scan_max_grid() * 4);
    hipLaunchKernelGGL(translate_kernel<Sym>, dim3(grid), dim3(256), (size_t)cap * 5, stream, raw, n_chunks, keys, vals, cap, out);
    return hipGetLastError();
}

hipError_t launch_translate(const void* raw, uint32_t raw_elem, uint64_t n_bytes, const uint32_t* keys, const uint8_t* vals, uint32_t cap, uint8_t* out,
                            hipStream_t stream)
{
    const uint64_t n_chunks = n_bytes / 16;  // the payload is whole 16-byte chunks by construction
    if (n_chunks == 0) return hipSuccess;
    uint4* o = reinterpret_cast<uint4*>(out);
    if (raw_elem == 2) return launch_translate_t(static_cast<const uint16_t*>(raw), n_chunks, keys, vals, cap, o, stream);
    return launch_translate_t(static_cast<const uint32_t*>(raw), n_chunks, keys, vals, cap, o, stream);
}

// ---------------------------------------------------------------------------------------------------
// Results of a length-bucketed corpus in ORIGINAL candidate order without scattered stores.
// A tile holds 64 candidates of one length, i.e. 64 original indices spread over the whole corpus: writing out[orig[slot]]
// from the scan puts one 4-byte store into 64 different cache lines per tile, and the other 15..31 results of each line arrive
// from tiles of other lengths, much later -- every line goes to HBM as many partial writes (100 M ragged candidates: 1.5 ms of
// a 2.6 ms launch, measured; bench.py --ragged).  Seen from the OUTPUT side the same permutation is friendly: consecutive
// candidates read from as many sequential streams as there are lengths, and each stream's cache line is used up by neighbouring
// threads.  So for large ragged corpora the scans write tmp[slot] (coalesced: p.orig is replaced by a slot -> slot map that
// keeps the padding lanes' kPad) and one gather pass writes out[i] = tmp[slot_of[i]]: 12 bytes of streaming traffic per candidate.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slot_maps_kernel(const uint32_t* __restrict__ orig, uint32_t n_slots, uint32_t* __restrict__ slot_of,
                                                        uint32_t* __restrict__ ident)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += gridDim.x * blockDim.x) {
        const uint32_t o = orig[s];
        ident[s] = o == kPad ? kPad : s;
        if (o != kPad) slot_of[o] = s;  // (one-time scattered pass, per corpus)
    }
}
hipError_t launch_slot_maps(const uint32_t* orig, uint32_t n_slots, uint32_t* slot_of, uint32_t* ident, hipStream_t stream)
{
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(slot_maps_kernel, dim3(std::min<uint32_t>((n_slots + 255) / 256, 65536u)), dim3(256), 0, stream, orig, n_slots, slot_of, ident);
    return hipGetLastError();
}

// One workgroup walks kGatherSpan CONSECUTIVE candidates per trip: their slots advance along one sequential stream per candidate
// length, so with a span of 16 K candidates every 128-byte line of tmp (32 results of one length) is used up by this workgroup
// within a few iterations -- out of its own L1 / L2 -- instead of being fetched by whichever workgroups (on whichever XCDs)
// happen to hold the neighbouring candidates (first version: 4 consecutive candidates per thread, grid-stride: 0.65 ms for
// 100 M results = 1.9 TB/s of the 1.2 GB it has to move).
template <class T, uint32_t kGatherUnroll>
__global__ __launch_bounds__(256) void gather_results_kernel(const T* __restrict__ tmp, const uint32_t* __restrict__ slot_of, T* __restrict__ out, uint32_t n, uint32_t kGatherSpan)
{
    const uint32_t spans = (n + kGatherSpan - 1) / kGatherSpan;
    for (uint32_t sp = blockIdx.x; sp < spans; sp += gridDim.x) {
        const uint32_t base = sp * kGatherSpan, end = min(n, base + kGatherSpan);
        // software pipeline: the slot loads of trip k + 1 are in flight while trip k's dependent loads and stores run
        uint32_t s[kGatherUnroll], s_next[kGatherUnroll];
        uint32_t i0 = base + threadIdx.x;
#pragma unroll
        for (uint32_t j = 0; j < kGatherUnroll; ++j) s[j] = i0 + j * 256 < end ? __builtin_nontemporal_load(slot_of + i0 + j * 256) : kPad;
        for (; i0 < end; i0 += 256 * kGatherUnroll) {
            const uint32_t i1 = i0 + 256 * kGatherUnroll;
#pragma unroll
            for (uint32_t j = 0; j < kGatherUnroll; ++j) s_next[j] = i1 + j * 256 < end ? __builtin_nontemporal_load(slot_of + i1 + j * 256) : kPad;
            T v[kGatherUnroll];
#pragma unroll
            for (uint32_t j = 0; j < kGatherUnroll; ++j)
                if (s[j] != kPad) v[j] = tmp[s[j]];
#pragma unroll
            for (uint32_t j = 0; j < kGatherUnroll; ++j)
                if (s[j] != kPad) __builtin_nontemporal_store(v[j], out + i0 + j * 256);
#pragma unroll
            for (uint32_t j = 0; j < kGatherUnroll; ++j) s[j] = s_next[j];
        }
    }
}
hipError_t launch_gather_results(const void* tmp, const uint32_t* slot_of, void* out, uint32_t n, bool f64, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    static const uint32_t span = [] { const char* e = getenv("RF_GATHER_SPAN"); return e ? (uint32_t)atoi(e) : 16384u; }();  // tuning knobs
    static const int unroll = [] { const char* e = getenv("RF_GATHER_UNROLL"); return e ? atoi(e) : 8; }();
    const dim3 g(std::min<uint32_t>((n + span - 1) / span, (uint32_t)scan_max_grid())), b(256);
    if (f64)
        hipLaunchKernelGGL((gather_results_kernel<double, 8>), g, b, 0, stream, static_cast<const double*>(tmp), slot_of, static_cast<double*>(out), n, span);
    else if (unroll == 16)
        hipLaunchKernelGGL((gather_results_kernel<uint32_t, 16>), g, b, 0, stream, static_cast<const uint32_t*>(tmp), slot_of, static_cast<uint32_t*>(out), n, span);
    else if (unroll == 4)
        hipLaunchKernelGGL((gather_results_kernel<uint32_t, 4>), g, b, 0, stream, static_cast<const uint32_t*>(tmp), slot_of, static_cast<uint32_t*>(out), n, span);
    else
        hipLaunchKernelGGL((gather_results_kernel<uint32_t, 8>), g, b, 0, stream, static_cast<const uint32_t*>(tmp), slot_of, static_cast<uint32_t*>(out), n, span);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Head plane of a single-length corpus: the first 8 symbols of every candidate, tile t's 64 lanes at t * 512 B.
// A Levenshtein scan under a small cutoff k takes its first look at column k + 3 (rounded up to even) and on a typical corpus
// abandons nearly every tile there; with k <= 5 that look needs <= 8 symbols, so those scans stream 8 bytes per candidate from
// this plane instead of 16-byte chunk rows out of the tiles (rf_scan.hip early_head8_kernel).  An acceleration index beside the
// corpus (+ 12.5 % for 64-symbol candidates), built on the first such scan, never part of a corpus file.  NOT the layout
// experiment of profiles/head_plane_r03.txt (same bytes, denser: no gain) -- this one halves the bytes.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head8_plane_kernel(const uint8_t* __restrict__ data, uint32_t n_tiles, uint32_t tile_bytes, uint2* __restrict__ heads)
{
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += gridDim.x * 4)
        heads[(size_t)t * kWave + lane] = *reinterpret_cast<const uint2*>(data + (uint64_t)t * tile_bytes + (size_t)lane * kChunk);
}
hipError_t launch_head8_plane(const uint8_t* data, uint32_t n_tiles, uint32_t tile_bytes, uint8_t* heads, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(head8_plane_kernel, dim3(std::min<uint32_t>((n_tiles + 3) / 4, 65536u)), dim3(256), 0, stream, data, n_tiles, tile_bytes,
                       reinterpret_cast<uint2*>(heads));
    return hipGetLastError();
}

}  // namespace rf
