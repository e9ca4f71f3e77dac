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

// the largest STORED symbol of a packed payload (after the renaming sigma: the corpus' symbols are 0 .. alphabet - 1 by frequency
// rank).  Exact, over every byte -- the sample sigma was made from may have missed a rare symbol, and a kernel that sizes an LDS
// table by this number (rf_jaro.hip jaro_word_asm_kernel<true>) must not be told less.
__global__ __launch_bounds__(256) void max_byte_kernel(const uint4* __restrict__ data, uint64_t n16, uint32_t* __restrict__ out)
{
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = data[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = max(max(max(m, w[k] & 0xFFu), (w[k] >> 8) & 0xFFu), max((w[k] >> 16) & 0xFFu, w[k] >> 24));
    }
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicMax(out, m);
}
hipError_t launch_max_byte(const uint8_t* data, uint64_t bytes, uint32_t* out, hipStream_t stream)
{
    if (bytes < 16) return hipSuccess;
    hipLaunchKernelGGL(max_byte_kernel, dim3((uint32_t)std::min<uint64_t>((bytes / 16 + 255) / 256, 4096)), dim3(256), 0, stream,
                       reinterpret_cast<const uint4*>(data), bytes / 16, out);
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
        if (o != kPad && slot_of) slot_of[o] = s;  // (one-time scattered pass, per corpus; not needed when the window table serves)
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
// The same gather with every access coalesced.  The slots are a few RUNS in each of which the original indices ascend (one run
// per candidate length: the packer's counting sort is stable; the views of the mixed section add their own), so the candidates
// of one WINDOW of kGatherWindow consecutive original indices sit in one contiguous stretch of every run.  A table holds where
// each window starts in each run (built once per corpus, by binary search: window_table_kernel); a workgroup then reads its
// window's stretches of tmp and orig front to back, drops the values into an LDS image of the window at orig - base, and writes
// the image out in one piece.  12 bytes per candidate like gather_results_kernel (10 with the 2-byte offsets of launch_slot_off16 in place of orig[]: round 5),
// but that one issues 64 transactions per
// wavefront load of tmp (64 candidates of 64 lengths), this one 2-3.  Corpora with more than kMaxGatherRuns runs keep the other.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void run_starts_kernel(const uint32_t* __restrict__ orig, uint32_t n_slots, uint32_t* __restrict__ list, uint32_t cap,
                                                         uint32_t* __restrict__ count)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += gridDim.x * blockDim.x) {
        const uint32_t o = orig[s];
        if (o == kPad) continue;
        const uint32_t prev = s ? orig[s - 1] : kPad;
        if (prev == kPad || prev > o) {  // (a padding slot ends a run: inside a run padding can then only trail)
            const uint32_t k = atomicAdd(count, 1u);
            if (k < cap) list[k] = s;
        }
    }
}
hipError_t launch_run_starts(const uint32_t* orig, uint32_t n_slots, uint32_t* list, uint32_t cap, uint32_t* count, hipStream_t stream)
{
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(run_starts_kernel, dim3(std::min<uint32_t>((n_slots + 255) / 256, 65536u)), dim3(256), 0, stream, orig, n_slots, list, cap, count);
    return hipGetLastError();
}

// table[w * n_runs + r] = the first slot of run r (slots [runs[r], runs[r + 1])) whose original index is >= w * kGatherWindow;
// padding slots count as "beyond everything", so row n_rows - 1 (w * kGatherWindow >= n) holds each run's end without its padding
__global__ __launch_bounds__(256) void window_table_kernel(const uint32_t* __restrict__ orig, const uint32_t* __restrict__ runs, uint32_t n_runs,
                                                           uint32_t n_rows, uint32_t* __restrict__ table)
{
    const uint64_t total = (uint64_t)n_rows * n_runs;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = (uint32_t)(i / n_runs), r = (uint32_t)(i % n_runs);
        const uint64_t want = (uint64_t)w * kGatherWindow;
        uint32_t lo = runs[r], hi = runs[r + 1];
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if ((uint64_t)orig[mid] >= want) hi = mid; else lo = mid + 1;
        }
        table[i] = lo;
    }
}
hipError_t launch_window_table(const uint32_t* orig, const uint32_t* runs, uint32_t n_runs, uint32_t n_rows, uint32_t* table, hipStream_t stream)
{
    const uint64_t total = (uint64_t)n_rows * n_runs;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(window_table_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 65536u)), dim3(256), 0, stream, orig, runs, n_runs, n_rows, table);
    return hipGetLastError();
}

// kRows table windows per workgroup trip: 2 for u32 results, 1 for f64 (a 32 KiB image either way: 5 workgroups per CU)
template <class T, uint32_t kRows, uint32_t kFlight, bool kOff16>
__global__ __launch_bounds__(256) void window_gather_kernel(const T* __restrict__ tmp, const uint32_t* __restrict__ orig, const uint16_t* __restrict__ off16,
                                                            const uint32_t* __restrict__ table, uint32_t n_runs, uint32_t n_rows, T* __restrict__ out, uint32_t n,
                                                            uint32_t xcd_deal)
{
    constexpr uint32_t kSpan = kRows * kGatherWindow;
    static_assert(kSpan <= kGatherOff16Mod && kGatherOff16Mod % kSpan == 0, "a span lies inside one period of the 16-bit offsets");
    __shared__ T image[kSpan];
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const uint32_t spans = (n + kSpan - 1) / kSpan;
    // xcd_deal: workgroups go to the 8 XCDs round-robin, and neighbouring spans share the 128-byte lines their stretches begin and end in -- so XCD x walks the
    // x-th eighth of the spans, neighbours at the same time, and those lines are fetched into one L2 once (round 5)
    const uint32_t per_xcd = xcd_deal ? (spans + 7) / 8 : spans, stride = xcd_deal ? gridDim.x / 8 : gridDim.x;
    const uint32_t first = xcd_deal ? (blockIdx.x & 7) * per_xcd : 0;
    for (uint32_t j = xcd_deal ? blockIdx.x / 8 : blockIdx.x; j < per_xcd; j += stride) {
        const uint32_t sp = first + j;
        if (sp >= spans) break;
        const uint32_t base = sp * kSpan;
        const uint32_t* row0 = table + (size_t)(sp * kRows) * n_runs;
        const uint32_t* row1 = table + (size_t)min((sp + 1) * kRows, n_rows - 1) * n_runs;
        // kFlight runs per wavefront in flight: the 2 * kFlight loads of a trip do not depend on each other
        for (uint32_t r0 = wave * kFlight; r0 < n_runs; r0 += kFlight * (256 / kWave)) {
            uint32_t a[kFlight], b[kFlight];
#pragma unroll
            for (uint32_t j = 0; j < kFlight; ++j) {
                const bool live = r0 + j < n_runs;
                a[j] = live ? row0[r0 + j] : 0u;
                b[j] = live ? row1[r0 + j] : 0u;
            }
            if (kOff16) {
                // TWO slots per lane (an even-aligned pair: one dword of offsets, one 8- / 16-byte load of values) -- sub-dword lanes cost more per instruction than
                // their bytes save (profiles/grid_sweep_r05.txt), dword lanes do not
                uint32_t a2[kFlight], longest = 0;
#pragma unroll
                for (uint32_t j = 0; j < kFlight; ++j) {
                    a2[j] = a[j] & ~1u;
                    longest = max(longest, b[j] - a2[j]);
                }
                const uint32_t sub = base & (kGatherOff16Mod - 1);
                for (uint32_t k = 2 * lane; k < longest + 2 * lane; k += 2 * kWave) {  // (uniform trip count)
                    uint32_t o[kFlight];
                    T v0[kFlight], v1[kFlight];
#pragma unroll
                    for (uint32_t j = 0; j < kFlight; ++j) {
                        const uint32_t s0 = a2[j] + k;  // (even; slot s0 + 1 exists: the slot count is a multiple of 64)
                        const bool in = s0 < b[j];
                        o[j] = in ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(off16 + s0)) : 0xFFFFFFFFu;
                        if (in) {
                            typedef T Pair __attribute__((ext_vector_type(2)));
                            const Pair v = __builtin_nontemporal_load(reinterpret_cast<const Pair*>(tmp + s0));  // (aligned: s0 is even)
                            v0[j] = v.x;
                            v1[j] = v.y;
                        }
                        if (s0 < a[j]) o[j] |= 0xFFFFu;             // the slot before the stretch
                        if (s0 + 1 >= b[j]) o[j] |= 0xFFFF0000u;    // the slot behind it
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kFlight; ++j) {
                        if ((o[j] & 0xFFFFu) != 0xFFFFu) image[(o[j] & 0xFFFFu) - sub] = v0[j];
                        if ((o[j] >> 16) != 0xFFFFu) image[(o[j] >> 16) - sub] = v1[j];
                    }
                }
                continue;
            }
            uint32_t longest = 0;
#pragma unroll
            for (uint32_t j = 0; j < kFlight; ++j) longest = max(longest, b[j] - a[j]);
            for (uint32_t k = lane; k < longest + lane; k += kWave) {  // (uniform trip count; k - lane < longest)
                uint32_t o[kFlight];
                T v[kFlight];
#pragma unroll
                for (uint32_t j = 0; j < kFlight; ++j) {
                    const bool in = a[j] + k < b[j];
                    o[j] = in ? __builtin_nontemporal_load(orig + a[j] + k) : kPad;
                    if (in) v[j] = __builtin_nontemporal_load(tmp + a[j] + k);
                }
#pragma unroll
                for (uint32_t j = 0; j < kFlight; ++j)
                    if (o[j] != kPad) image[o[j] - base] = v[j];
            }
        }
        __syncthreads();
        const uint32_t count = min(kSpan, n - base);
        for (uint32_t i = threadIdx.x; i < count; i += 256) __builtin_nontemporal_store(image[i], out + base + i);
        __syncthreads();
    }
}
hipError_t launch_window_gather(const void* tmp, const uint32_t* orig, const uint16_t* off16, const uint32_t* table, uint32_t n_runs, uint32_t n_rows, void* out, uint32_t n,
                                bool f64, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    // (measured, 100 M u32 results of 64 lengths, whole Indel step: 2 windows x 8 runs in flight 1.125 ms; 2 x 4: 1.155; 4 x 4: 1.19;
    // 4 x 8: 1.15; 1 x 4: 1.165; 2 x 16: 1.15; 1 x 8: 1.185 -- gather_results_kernel: 1.226)
    const uint32_t span = (f64 ? 1u : 2u) * kGatherWindow;
    static const bool use_deal = [] { const char* e = getenv("RF_GATHER_XCD"); return !e || atoi(e) != 0; }();
    uint32_t grid = std::min<uint32_t>((n + span - 1) / span, (uint32_t)scan_max_grid());
    const uint32_t deal = (use_deal && grid >= 64) ? 1u : 0u;
    if (deal) grid = (grid + 7) / 8 * 8;
    const dim3 g(grid), b(256);
    if (f64 && off16)
        hipLaunchKernelGGL((window_gather_kernel<double, 1, 8, true>), g, b, 0, stream, static_cast<const double*>(tmp), orig, off16, table, n_runs, n_rows, static_cast<double*>(out), n, deal);
    else if (f64)
        hipLaunchKernelGGL((window_gather_kernel<double, 1, 8, false>), g, b, 0, stream, static_cast<const double*>(tmp), orig, off16, table, n_runs, n_rows, static_cast<double*>(out), n, deal);
    else if (off16)
        hipLaunchKernelGGL((window_gather_kernel<uint32_t, 2, 8, true>), g, b, 0, stream, static_cast<const uint32_t*>(tmp), orig, off16, table, n_runs, n_rows,
                           static_cast<uint32_t*>(out), n, deal);
    else
        hipLaunchKernelGGL((window_gather_kernel<uint32_t, 2, 8, false>), g, b, 0, stream, static_cast<const uint32_t*>(tmp), orig, off16, table, n_runs, n_rows,
                           static_cast<uint32_t*>(out), n, deal);
    return hipGetLastError();
}

// The f64 a normalized op returns for every u32 distance: exactly emit_fin's f64 branch (rf_device.hpp tile_fin / emit_fin; details/distance.rs:246-250, :273;
// the cutoff compare of src/common.rs:43-45 / :83-85) -- run_many's two-step path for the multi-word Levenshtein scans.  The candidate's length comes from
// len_of[] (original order; launch_len_of) or is the one length of a single-length corpus.
__global__ __launch_bounds__(256) void normalize_kernel(const uint32_t* __restrict__ dist, const uint32_t* __restrict__ len_of, uint32_t uniform_len, double* __restrict__ out,
                                                        uint32_t n, uint32_t len1, int32_t fin_mS, int32_t fin_mM, uint32_t op, uint32_t has_cutoff, double cutoff)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = __builtin_nontemporal_load(dist + i);
        const uint32_t len2 = len_of ? __builtin_nontemporal_load(len_of + i) : uniform_len;
        const uint32_t maximum = (uint32_t)fin_mS * (len1 + len2) + (uint32_t)fin_mM * max(len1, len2);
        const double nd = maximum == 0 ? 0.0 : (double)d / (double)maximum;
        double v;
        bool keep;
        if (op == RF_OP_NORMALIZED_DISTANCE) {
            v = nd;
            keep = !has_cutoff || v <= cutoff;
        } else {
            v = 1.0 - nd;
            keep = !has_cutoff || v >= cutoff;
        }
        if (d == RF_NONE_U32) keep = false;  // (the u32 scan ran under a raw cutoff and answered None: beyond anything the f64 test can pass -- run_many's norm_raw_cut)
        __builtin_nontemporal_store(keep ? v : __longlong_as_double(0x7FF8000000000000ll), out + i);
    }
}
hipError_t launch_normalize(const uint32_t* dist, const uint32_t* len_of, uint32_t uniform_len, double* out, uint32_t n, uint32_t len1, int32_t fin_mS, int32_t fin_mM, uint32_t op,
                            uint32_t has_cutoff, double cutoff, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_kernel, dim3(std::min<uint32_t>((n + 255) / 256, (uint32_t)scan_max_grid() * 4)), dim3(256), 0, stream, dist, len_of, uniform_len, out, n, len1,
                       fin_mS, fin_mM, op, has_cutoff, cutoff);
    return hipGetLastError();
}
// len_of[orig[slot]] = the length of the slot's tile, over the exact tiles and the one-length views of the mixed section (every candidate has a slot in one of them)
__global__ __launch_bounds__(256) void len_of_kernel(const TileDesc* __restrict__ tiles, uint32_t n_tiles, const uint32_t* __restrict__ orig, uint32_t* __restrict__ len_of)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    for (uint32_t t = blockIdx.x * 4 + threadIdx.x / kWave; t < n_tiles; t += gridDim.x * 4) {
        const TileDesc d = tiles[t];
        const uint32_t o = orig[d.slot0 + lane];
        if (o != kPad) len_of[o] = d.len;
    }
}
hipError_t launch_len_of(const TileDesc* tiles, uint32_t n_tiles, const uint32_t* orig, uint32_t* len_of, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(len_of_kernel, dim3(std::min<uint32_t>((n_tiles + 3) / 4, 65536u)), dim3(256), 0, stream, tiles, n_tiles, orig, len_of);
    return hipGetLastError();
}

// off16[slot] = orig[slot] mod kGatherOff16Mod (0xFFFF on padding slots): all the window gather needs to know about a slot's candidate, since its span of
// original indices is known -- 2 bytes per slot instead of the 4 of orig[] (once per corpus)
__global__ __launch_bounds__(256) void slot_off16_kernel(const uint32_t* __restrict__ orig, uint32_t n_slots, uint16_t* __restrict__ off16)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += gridDim.x * blockDim.x) {
        const uint32_t o = orig[s];
        off16[s] = o == kPad ? (uint16_t)0xFFFFu : (uint16_t)(o & (kGatherOff16Mod - 1));
    }
}
hipError_t launch_slot_off16(const uint32_t* orig, uint32_t n_slots, uint16_t* off16, hipStream_t stream)
{
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(slot_off16_kernel, dim3(std::min<uint32_t>((n_slots + 255) / 256, 65536u)), dim3(256), 0, stream, orig, n_slots, off16);
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
// The head plane at 6 bits per symbol (round 4): a corpus that stores fewer than 64 distinct symbols (symbols are stored as frequency
// ranks) needs 48 bits for a candidate's first 8 symbols, and the band prefilter pass is a pure stream over the plane -- 6 instead of
// 8 bytes per candidate.  Built from the 8-byte plane; layout in rf_internal.hpp (ScanParams::heads6).
__global__ __launch_bounds__(256) void head6_plane_kernel(const uint2* __restrict__ heads8, uint32_t n_tiles, uint32_t* __restrict__ heads6)
{
    const uint32_t lane = threadIdx.x & 63, pairs = (n_tiles + 1) / 2;
    auto pack = [](uint2 h) {  // 8 bytes -> 48 bits, symbol i on bits 6 i ..
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= (uint64_t)(((i < 4 ? h.x : h.y) >> (8 * (i & 3))) & 63u) << (6 * i);
        return v;
    };
    for (uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6); q < pairs; q += gridDim.x * 4) {
        const size_t c0 = (size_t)q * 2 * kWave + 2 * lane;  // candidates 2l, 2l + 1 of the pair (the 8-byte plane has a pad row behind the last tile)
        const uint64_t a = pack(heads8[c0]), b = pack(heads8[c0 + 1]);
        uint32_t* row = heads6 + (size_t)q * 3 * kWave;
        row[lane] = (uint32_t)a;
        row[kWave + lane] = (uint32_t)(a >> 32) | ((uint32_t)b << 16);
        row[2 * kWave + lane] = (uint32_t)(b >> 16);
    }
}
hipError_t launch_head6_plane(const uint8_t* heads8, uint32_t n_tiles, uint32_t* heads6, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(head6_plane_kernel, dim3(std::min<uint32_t>(((n_tiles + 1) / 2 + 3) / 4, 65536u)), dim3(256), 0, stream, reinterpret_cast<const uint2*>(heads8), n_tiles, heads6);
    return hipGetLastError();
}
// The PAYLOAD at 6 bits per symbol (round 5): a single-length corpus that stores fewer than 64 distinct symbols kept a second
// time with 16 symbols in 12 bytes -- symbol j of a chunk on bits 6 j .. 6 j + 5 of the 96-bit little-endian value -- chunk k of lane r of tile t
// at ((t * nch + k) * 64 + r) * 12: a wavefront's load of "my next 16 columns" is one contiguous 768-byte read (global_load_dwordx3) instead of 1 KiB.
// The HBM-bound scans (Indel / LCS, single word) stream it instead of the 8-bit payload: stream6_kernel, rf_scan.hip.
// Positions behind the candidates' end (a length that is not a multiple of 16) hold the code 63, which such a corpus does not store (the builder's condition) and
// whose table row the scans zero: an LCS column over it changes nothing, so the scans run whole chunks only.
__global__ __launch_bounds__(256) void pack6_kernel(const uint4* __restrict__ data, uint32_t n_tiles, uint32_t nch, uint32_t len, uint32_t* __restrict__ data6)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t rows = (uint64_t)n_tiles * nch;  // chunk rows of 64 lanes
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (uint64_t)gridDim.x * 4) {
        const uint4 c = data[r * kWave + lane];
        const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
        const uint32_t col0 = (uint32_t)(r % nch) * kChunk;
        uint32_t w[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t sym = col0 + j < len ? (dw[j / 4] >> (8 * (j % 4))) & 63u : 63u;
            const int o = 6 * j;
            w[o / 32] |= sym << (o % 32);
            if (o % 32 > 26) w[o / 32 + 1] |= sym >> (32 - o % 32);
        }
        uint32_t* dst = data6 + (r * kWave + lane) * 3;
        dst[0] = w[0];
        dst[1] = w[1];
        dst[2] = w[2];
    }
}
hipError_t launch_pack6(const uint8_t* data, uint32_t n_tiles, uint32_t len, uint32_t* data6, hipStream_t stream)
{
    const uint32_t nch = (len + kChunk - 1) / kChunk;
    if (n_tiles == 0 || nch == 0) return hipSuccess;
    const uint64_t rows = (uint64_t)n_tiles * nch;
    hipLaunchKernelGGL(pack6_kernel, dim3((uint32_t)std::min<uint64_t>((rows + 3) / 4, 262144u)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(data), n_tiles, nch, len,
                       data6);
    return hipGetLastError();
}
// the same over the EXACT tiles of a length-bucketed corpus (round 4): row t = the first 8 stored bytes of tile t's 64 lanes, whatever
// the tile's length (a candidate shorter than 8 symbols contributes its zero padding -- the cutoff scans only take their first look
// from the plane for runs of >= 16 symbols, rf_api_scan.hip launch_scan_runs)
__global__ __launch_bounds__(256) void head8_plane_tiles_kernel(const uint8_t* __restrict__ data, const TileDesc* __restrict__ tiles, uint32_t n_tiles,
                                                                uint2* __restrict__ heads)
{
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += gridDim.x * 4)
        heads[(size_t)t * kWave + lane] = *reinterpret_cast<const uint2*>(data + tiles[t].data_off + (size_t)lane * kChunk);
}
hipError_t launch_head8_plane_tiles(const uint8_t* data, const TileDesc* tiles, uint32_t n_tiles, uint8_t* heads, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(head8_plane_tiles_kernel, dim3(std::min<uint32_t>((n_tiles + 3) / 4, 65536u)), dim3(256), 0, stream, data, tiles, n_tiles,
                       reinterpret_cast<uint2*>(heads));
    return hipGetLastError();
}

}  // namespace rf
