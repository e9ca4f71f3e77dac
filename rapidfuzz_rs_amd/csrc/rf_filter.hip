// rf_filter.hip -- compact (index, score) results of a thresholded scan (rf_filter_u32 / rf_filter_f64, round 6).
//
// reference: every `<op>_with_args` returns Option<T> (src/common.rs:18-46, :83-85) and the caller of a thresholded dedup / record-linkage loop keeps the Somes:
//   corpus.iter().enumerate().filter_map(|(i, c)| scorer.distance_with_args(c, &args).map(|d| (i, d)))
// These kernels are that filter_map over a device vector of results (None = 0xFFFFFFFF / NaN), ORDER PRESERVING and without atomics:
//   count   one wavefront per segment of 1024 entries: how many Somes (coalesced rounds of 64 entries, one ballot each)
//   sums    hipcub exclusive sum over the segments' counts (the last entry is the total)
//   emit    the same walk again -- only over segments that hold a Some at all, so a sparse result costs ONE read of the vector -- each Some written at
//           (its segment's sum + its rank inside the segment) while that is below the caller's capacity
// An entry's index is its position, or map[position] (slot -> original index of a length-bucketed corpus, survivor number -> candidate of the lane-compacted cutoff
// scans; 0xFFFFFFFF = no candidate there).  Then, where the caller's order asks for it, hipcub radix sorts over the (few) results, and a last kernel widens the
// indices to u64 + index_base.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "rf_internal.hpp"

namespace rf {

namespace {

constexpr uint32_t kSegRounds = 16, kSeg = kSegRounds * kWave;  // entries per wavefront segment

__device__ __forceinline__ bool some(uint32_t v) { return v != RF_NONE_U32; }
__device__ __forceinline__ bool some(double v) { return v == v; }

// entries [0, m): m = min(m_bound, *m_dev) when m_dev is given.  map (nullable): entry -> index, kPad = not a candidate; entries below map_from are known to be
// candidates (the exact tiles of a bucketed corpus have no padding lane), so the COUNT need not read the map there.
template <class T>
__global__ __launch_bounds__(256) void filter_count_kernel(const T* __restrict__ val, const uint32_t* __restrict__ map, uint32_t map_from, uint32_t m_bound,
                                                           const uint32_t* __restrict__ m_dev, uint32_t n_seg, uint32_t* __restrict__ seg_cnt)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t s = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (s > n_seg) return;
    const uint32_t m = m_dev ? min(m_bound, *m_dev) : m_bound;
    uint32_t cnt = 0;
    if (s < n_seg) {
        const uint32_t base = s * kSeg;
        constexpr uint32_t kPer = 16 / sizeof(T);  // entries per 16-byte load
        if (base + kSeg <= m && !(map && base + kSeg > map_from)) {
            // a whole segment that needs no map: 16 bytes per lane and load (counting does not care which lane sees which entry)
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u* src = reinterpret_cast<const v4u*>(val + base);
#pragma unroll 4
            for (uint32_t r = 0; r < kSegRounds / kPer; ++r) {
                const v4u v = __builtin_nontemporal_load(src + r * kWave + lane);
                uint32_t mine;
                if constexpr (sizeof(T) == 4) {
                    mine = (v.x != RF_NONE_U32) + (v.y != RF_NONE_U32) + (v.z != RF_NONE_U32) + (v.w != RF_NONE_U32);
                } else {  // a NaN: exponent all ones and a mantissa that is not zero
                    const uint64_t a = ((uint64_t)v.y << 32) | v.x, b = ((uint64_t)v.w << 32) | v.z;
                    mine = ((a & 0x7FFFFFFFFFFFFFFFull) <= 0x7FF0000000000000ull) + ((b & 0x7FFFFFFFFFFFFFFFull) <= 0x7FF0000000000000ull);
                }
                cnt += mine;
            }
#pragma unroll
            for (uint32_t d = kWave / 2; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, kWave);
        } else {
#pragma unroll 4
            for (uint32_t r = 0; r < kSegRounds; ++r) {
                const uint32_t e = base + r * kWave + lane;
                bool sel = false;
                if (e < m) {
                    sel = some(val[e]);
                    if (sel && map && e >= map_from) sel = map[e] != kPad;
                }
                cnt += (uint32_t)__popcll(__ballot(sel));
            }
        }
    }
    if (lane == 0) seg_cnt[s] = cnt;  // (seg_cnt[n_seg] = 0: the exclusive sum's last entry is the total)
}

template <class T>
__global__ __launch_bounds__(256) void filter_emit_kernel(const T* __restrict__ val, const uint32_t* __restrict__ map, uint32_t m_bound, const uint32_t* __restrict__ m_dev,
                                                          uint32_t n_seg, const uint32_t* __restrict__ seg_sum, uint32_t capacity, uint32_t* __restrict__ out_idx,
                                                          T* __restrict__ out_val)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t s = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (s >= n_seg) return;
    uint32_t at = seg_sum[s];
    const uint32_t end = seg_sum[s + 1];
    if (end == at || at >= capacity) return;  // nothing here, or everything here lies beyond the capacity
    const uint32_t m = m_dev ? min(m_bound, *m_dev) : m_bound;
    const uint32_t base = s * kSeg;
    for (uint32_t r = 0; r < kSegRounds && at < end; ++r) {
        const uint32_t e = base + r * kWave + lane;
        bool sel = false;
        T v = T();
        uint32_t idx = e;
        if (e < m) {
            v = val[e];
            sel = some(v);
            if (sel && map) {
                idx = map[e];
                sel = idx != kPad;
            }
        }
        const uint64_t b = __ballot(sel);
        const uint32_t pos = at + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (sel && pos < capacity) {
            out_idx[pos] = idx;
            out_val[pos] = v;
        }
        at += (uint32_t)__popcll(b);
    }
}

// order-preserving keys of the scores (smaller = better): u32 -> the score, or its complement for the similarity ops; f64 -> the IEEE bits with the sign handled
// (rf_topk_entry's map), complemented likewise
__global__ void filter_keys_u32_kernel(const uint32_t* __restrict__ val, uint32_t n, bool desc, uint32_t* __restrict__ key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = desc ? ~val[i] : val[i];
}
__global__ void filter_keys_f64_kernel(const double* __restrict__ val, uint32_t n, bool desc, uint64_t* __restrict__ key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t b = (uint64_t)__double_as_longlong(val[i]);
        b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        key[i] = desc ? ~b : b;
    }
}
// the caller's arrays: u64 indices (+ index_base) and the scores -- from the values, or (key != nullptr) decoded from their sort keys
// (n_dev: the number of results when only the device knows it yet -- min(n, *n_dev) entries are written)
__global__ void filter_finish_u32_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val, const uint32_t* __restrict__ key, bool desc, uint32_t n,
                                         const uint32_t* __restrict__ n_dev, uint64_t index_base, uint64_t* __restrict__ out_index, uint32_t* __restrict__ out_val)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i < n) {
        out_index[i] = index_base + idx[i];
        out_val[i] = key ? (desc ? ~key[i] : key[i]) : val[i];
    }
}
__global__ void filter_finish_f64_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ val, const uint64_t* __restrict__ key, bool desc, uint32_t n,
                                         const uint32_t* __restrict__ n_dev, uint64_t index_base, uint64_t* __restrict__ out_index, double* __restrict__ out_val)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i < n) {
        out_index[i] = index_base + idx[i];
        if (key) {
            uint64_t b = desc ? ~key[i] : key[i];
            b = (b >> 63) ? (b & 0x7FFFFFFFFFFFFFFFull) : ~b;
            out_val[i] = __longlong_as_double((long long)b);
        } else {
            out_val[i] = val[i];
        }
    }
}

// ---- the whole of it in ONE workgroup when the entries are few (the usual case behind the lane compaction: a cutoff of a few edits leaves hundreds of survivors
// in 100 M candidates): select the Somes, order them -- by index, or by (score, index) -- with a bitonic sort in LDS, widen the indices, write the caller's arrays
// and the count.  res: [0] the Somes, [1] the entries there were, [2] 1 = too many entries for this kernel (nothing written: the general road).
constexpr uint32_t kSmallMax = 2048, kSmallThreads = 256;
template <class T>
__global__ __launch_bounds__(kSmallThreads) void filter_small_kernel(const T* __restrict__ val, const uint32_t* __restrict__ map, uint32_t m_bound, const uint32_t* __restrict__ m_dev,
                                                                     bool by_score, bool desc, uint32_t capacity, uint64_t index_base, uint64_t* __restrict__ out_index,
                                                                     T* __restrict__ out_val, uint32_t* __restrict__ res, uint32_t seq, const uint32_t* __restrict__ aux_dev)
{
    __shared__ uint64_t k1[kSmallMax];  // primary key: 0 (by index) or the order-preserving image of the score
    __shared__ uint32_t k2[kSmallMax];  // secondary key: the index
    __shared__ uint64_t pv[kSmallMax];  // the value's bits
    __shared__ uint32_t n_some;
    const uint32_t m_all = m_dev ? *m_dev : m_bound;
    // res may be pinned HOST memory the caller spins on (rf_api_filter.hip): the words first, then -- system scope, release -- the call's sequence number
    if (m_all > kSmallMax || m_all > m_bound) {
        if (threadIdx.x == 0) {
            res[0] = 0, res[1] = m_all, res[2] = 1, res[4] = aux_dev ? *aux_dev : 0u;
            __hip_atomic_store(&res[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (threadIdx.x == 0) n_some = 0;
    __syncthreads();
    // the Somes, in whatever order the atomic hands out places: the sort below is what orders them
    for (uint32_t e = threadIdx.x; e < m_all; e += kSmallThreads) {
        const T v = val[e];
        const uint32_t idx = map ? map[e] : e;
        if (some(v) && idx != kPad) {
            uint64_t a, bits;
            if constexpr (sizeof(T) == 8) {
                bits = (uint64_t)__double_as_longlong(v);
                const uint64_t b = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
                a = by_score ? (desc ? ~b : b) : 0ull;
            } else {
                bits = v;
                a = by_score ? (uint64_t)(desc ? ~(uint32_t)v : (uint32_t)v) : 0ull;
            }
            const uint32_t at = atomicAdd(&n_some, 1u);
            k1[at] = a, k2[at] = idx, pv[at] = bits;
        }
    }
    __syncthreads();
    const uint32_t count = n_some;
    uint32_t width = 1;  // the sort's width: the next power of two (hundreds of results at most, usually: a few dozen compare-exchange rounds)
    while (width < count) width <<= 1;
    for (uint32_t e = count + threadIdx.x; e < width; e += kSmallThreads) k1[e] = ~0ull, k2[e] = kPad, pv[e] = 0;  // padding sorts behind every result
    __syncthreads();
    // bitonic sort of `width` entries by (k1, k2) ascending
    for (uint32_t k = 2; k <= width; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < width / 2; t += kSmallThreads) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;  // the pair (i, i + j) this thread compares
                const bool up = (i & k) == 0;
                const uint64_t a1 = k1[i], b1 = k1[l];
                const uint32_t a2 = k2[i], b2 = k2[l];
                const bool greater = a1 > b1 || (a1 == b1 && a2 > b2);
                if (greater == up) {
                    const uint64_t pa = pv[i], pb = pv[l];
                    k1[i] = b1, k1[l] = a1, k2[i] = b2, k2[l] = a2, pv[i] = pb, pv[l] = pa;
                }
            }
            __syncthreads();
        }
    for (uint32_t e = threadIdx.x; e < count && e < capacity; e += kSmallThreads) {
        out_index[e] = index_base + k2[e];
        if constexpr (sizeof(T) == 8)
            out_val[e] = __longlong_as_double((long long)pv[e]);
        else
            out_val[e] = (uint32_t)pv[e];
    }
    __threadfence_system();  // (the caller's arrays are complete before the host can see the sequence number)
    __syncthreads();
    if (threadIdx.x == 0) {
        res[0] = count, res[1] = m_all, res[2] = 0, res[4] = aux_dev ? *aux_dev : 0u;  // (aux: a device word that comes home with the report)
        __hip_atomic_store(&res[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- the same finish for MANY entries with few results (the survivors of the lane compaction: 100 k band-test survivors per 100 M random candidates, a few dozen of
// them within the cutoff): every workgroup walks its share of the entries and appends the Somes it finds to a small workspace (one atomic per RESULT, not per entry),
// the LAST workgroup to arrive (a ticket) orders them in LDS exactly like filter_small_kernel, writes the caller's arrays and the report, and leaves the workspace
// zeroed for the next call.  More than kSmallMax results: report [2] = 1 and nothing written -- the general compaction takes over.
struct SelectWork {
    uint32_t count, done, pad[2];
    uint32_t idx[kSmallMax];
    uint64_t bits[kSmallMax];
};
template <class T>
__global__ __launch_bounds__(kSmallThreads) void filter_select_kernel(const T* __restrict__ val, const uint32_t* __restrict__ map, uint32_t m_bound, const uint32_t* __restrict__ m_dev,
                                                                      bool by_score, bool desc, uint32_t capacity, uint64_t index_base, uint64_t* __restrict__ out_index,
                                                                      T* __restrict__ out_val, SelectWork* __restrict__ ws, uint32_t* __restrict__ res, uint32_t seq,
                                                                      const uint32_t* __restrict__ aux_dev)
{
    __shared__ uint64_t k1[kSmallMax];
    __shared__ uint32_t k2[kSmallMax];
    __shared__ uint64_t pv[kSmallMax];
    __shared__ uint32_t last;
    const uint32_t m_all = m_dev ? *m_dev : m_bound, m = min(m_all, m_bound);
    for (uint64_t e = (uint64_t)blockIdx.x * kSmallThreads + threadIdx.x; e < m; e += (uint64_t)gridDim.x * kSmallThreads) {
        const T v = val[e];
        if (some(v)) {
            const uint32_t idx = map ? map[e] : (uint32_t)e;
            if (idx != kPad) {
                const uint32_t at = atomicAdd(&ws->count, 1u);
                if (at < kSmallMax) {  // (agent-scope stores: the reader is a workgroup on another XCD, whose L2 is not coherent with this one's)
                    uint64_t bits;
                    if constexpr (sizeof(T) == 8)
                        bits = (uint64_t)__double_as_longlong(v);
                    else
                        bits = v;
                    __hip_atomic_store(&ws->idx[at], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ws->bits[at], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&ws->done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    const uint32_t count = __hip_atomic_load(&ws->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool too_many = count > kSmallMax || m_all > m_bound;  // (entries beyond the room the scan had: the caller takes another road)
    if (!too_many) {
        for (uint32_t e = threadIdx.x; e < count; e += kSmallThreads) {
            const uint64_t bits = __hip_atomic_load(&ws->bits[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t a = 0;
            if (by_score) {
                if constexpr (sizeof(T) == 8) {
                    const uint64_t b = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
                    a = desc ? ~b : b;
                } else {
                    a = (uint64_t)(desc ? ~(uint32_t)bits : (uint32_t)bits);
                }
            }
            k1[e] = a, k2[e] = __hip_atomic_load(&ws->idx[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pv[e] = bits;
        }
        uint32_t width = 1;
        while (width < count) width <<= 1;
        for (uint32_t e = count + threadIdx.x; e < width; e += kSmallThreads) k1[e] = ~0ull, k2[e] = kPad, pv[e] = 0;
        __syncthreads();
        for (uint32_t k = 2; k <= width; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < width / 2; t += kSmallThreads) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const bool up = (i & k) == 0;
                    const uint64_t a1 = k1[i], b1 = k1[l];
                    const uint32_t a2 = k2[i], b2 = k2[l];
                    const bool greater = a1 > b1 || (a1 == b1 && a2 > b2);
                    if (greater == up) {
                        const uint64_t pa = pv[i], pb = pv[l];
                        k1[i] = b1, k1[l] = a1, k2[i] = b2, k2[l] = a2, pv[i] = pb, pv[l] = pa;
                    }
                }
                __syncthreads();
            }
        for (uint32_t e = threadIdx.x; e < count && e < capacity; e += kSmallThreads) {
            out_index[e] = index_base + k2[e];
            if constexpr (sizeof(T) == 8)
                out_val[e] = __longlong_as_double((long long)pv[e]);
            else
                out_val[e] = (uint32_t)pv[e];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&ws->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for the next call on this host thread)
        __hip_atomic_store(&ws->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        res[0] = too_many ? 0u : count, res[1] = m_all, res[2] = too_many ? 1u : 0u, res[4] = aux_dev ? *aux_dev : 0u;
        __hip_atomic_store(&res[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

uint32_t filter_segments(uint32_t m_bound) { return (m_bound + kSeg - 1) / kSeg; }
size_t filter_scan_temp_bytes(uint32_t n_seg)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_seg + 1, nullptr);
    return std::max<size_t>((bytes + 255) / 256 * 256, 256);
}
// seg: n_seg + 1 counts, overwritten by their exclusive sums (seg[n_seg] = the number of Somes)
hipError_t launch_filter_compact(const void* val, bool f64, const uint32_t* map, uint32_t map_from, uint32_t m_bound, const uint32_t* m_dev, uint32_t* seg, void* temp,
                                 size_t temp_bytes, uint32_t capacity, uint32_t* out_idx, void* out_val, hipStream_t st)
{
    const uint32_t n_seg = filter_segments(m_bound);
    const dim3 b(256), g((n_seg + 1 + 3) / 4);
    if (f64)
        hipLaunchKernelGGL(filter_count_kernel<double>, g, b, 0, st, (const double*)val, map, map_from, m_bound, m_dev, n_seg, seg);
    else
        hipLaunchKernelGGL(filter_count_kernel<uint32_t>, g, b, 0, st, (const uint32_t*)val, map, map_from, m_bound, m_dev, n_seg, seg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, seg, seg, (int)n_seg + 1, st);
    if (e != hipSuccess) return e;
    if (capacity == 0 || n_seg == 0) return hipSuccess;
    const dim3 g2((n_seg + 3) / 4);
    if (f64)
        hipLaunchKernelGGL(filter_emit_kernel<double>, g2, b, 0, st, (const double*)val, map, m_bound, m_dev, n_seg, seg, capacity, out_idx, (double*)out_val);
    else
        hipLaunchKernelGGL(filter_emit_kernel<uint32_t>, g2, b, 0, st, (const uint32_t*)val, map, m_bound, m_dev, n_seg, seg, capacity, out_idx, (uint32_t*)out_val);
    return hipGetLastError();
}

// two device words to the report slot (count and an auxiliary word), for the roads that need no ordering kernel
__global__ void filter_report_kernel(const uint32_t* __restrict__ a_dev, const uint32_t* __restrict__ aux_dev, uint32_t* __restrict__ res, uint32_t seq)
{
    res[0] = a_dev ? *a_dev : 0u, res[1] = res[0], res[2] = 0, res[4] = aux_dev ? *aux_dev : 0u;
    __hip_atomic_store(&res[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_filter_report(const uint32_t* a_dev, const uint32_t* aux_dev, uint32_t* res, uint32_t seq, hipStream_t st)
{
    hipLaunchKernelGGL(filter_report_kernel, dim3(1), dim3(1), 0, st, a_dev, aux_dev, res, seq);
    return hipGetLastError();
}
hipError_t launch_filter_small(const void* val, bool f64, const uint32_t* map, uint32_t m_bound, const uint32_t* m_dev, bool by_score, bool desc, uint32_t capacity,
                               uint64_t index_base, uint64_t* out_index, void* out_val, uint32_t* res, uint32_t seq, const uint32_t* aux_dev, hipStream_t st)
{
    if (f64)
        hipLaunchKernelGGL(filter_small_kernel<double>, dim3(1), dim3(kSmallThreads), 0, st, (const double*)val, map, m_bound, m_dev, by_score, desc, capacity, index_base, out_index,
                           (double*)out_val, res, seq, aux_dev);
    else
        hipLaunchKernelGGL(filter_small_kernel<uint32_t>, dim3(1), dim3(kSmallThreads), 0, st, (const uint32_t*)val, map, m_bound, m_dev, by_score, desc, capacity, index_base,
                           out_index, (uint32_t*)out_val, res, seq, aux_dev);
    return hipGetLastError();
}
uint32_t filter_small_max() { return kSmallMax; }
size_t filter_select_work_bytes() { return sizeof(SelectWork); }
// ws: filter_select_work_bytes() of device memory, zeroed ONCE (the kernel leaves it zeroed); one call at a time per workspace
hipError_t launch_filter_select(const void* val, bool f64, const uint32_t* map, uint32_t m_bound, const uint32_t* m_dev, bool by_score, bool desc, uint32_t capacity,
                                uint64_t index_base, uint64_t* out_index, void* out_val, void* ws, uint32_t* res, uint32_t seq, const uint32_t* aux_dev, uint32_t expected,
                                hipStream_t st)
{
    // (every workgroup takes a ticket on ONE address when it is done: 2048 of them cost the call 75 us.  The grid follows the number of entries the caller expects --
    // the previous call's -- between 32 and 512 workgroups; a workgroup strides over whatever there is)
    const uint32_t grid = std::max(32u, std::min((std::min(expected, m_bound) + 8 * kSmallThreads - 1) / (8 * kSmallThreads), 512u));
    if (f64)
        hipLaunchKernelGGL(filter_select_kernel<double>, dim3(grid), dim3(kSmallThreads), 0, st, (const double*)val, map, m_bound, m_dev, by_score, desc, capacity, index_base, out_index,
                           (double*)out_val, (SelectWork*)ws, res, seq, aux_dev);
    else
        hipLaunchKernelGGL(filter_select_kernel<uint32_t>, dim3(grid), dim3(kSmallThreads), 0, st, (const uint32_t*)val, map, m_bound, m_dev, by_score, desc, capacity, index_base,
                           out_index, (uint32_t*)out_val, (SelectWork*)ws, res, seq, aux_dev);
    return hipGetLastError();
}

// ---- ordering `count` compact results (all arrays on the device; the sorts are hipcub's stable radix sorts)
size_t filter_sort_temp_bytes(uint32_t count)
{
    size_t a = 0, b = 0, c = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)count);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)count);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, c, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)count);
    return std::max<size_t>((std::max(a, std::max(b, c)) + 255) / 256 * 256, 256);
}
// ascending index (results of a length-bucketed corpus arrive in slot order)
hipError_t launch_filter_sort_by_index(const uint32_t* idx_in, const void* val_in, bool f64, uint32_t count, uint32_t* idx_out, void* val_out, void* temp, size_t temp_bytes,
                                       hipStream_t st)
{
    if (count == 0) return hipSuccess;
    if (f64) return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, idx_in, idx_out, (const uint64_t*)val_in, (uint64_t*)val_out, (int)count, 0, 32, st);
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, idx_in, idx_out, (const uint32_t*)val_in, (uint32_t*)val_out, (int)count, 0, 32, st);
}
// best score first, ties by ascending index: (idx_in, val_in) must be in index order.  key_in / key_out: count u64 each; the scores come back out of key_out
// (launch_filter_finish decodes them)
hipError_t launch_filter_sort_by_score(const uint32_t* idx_in, const void* val_in, bool f64, bool desc, uint32_t count, void* key_in, void* key_out, uint32_t* idx_out,
                                       void* temp, size_t temp_bytes, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    const dim3 b(256), g((count + 255) / 256);
    if (f64) {
        hipLaunchKernelGGL(filter_keys_f64_kernel, g, b, 0, st, (const double*)val_in, count, desc, (uint64_t*)key_in);
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint64_t*)key_in, (uint64_t*)key_out, idx_in, idx_out, (int)count, 0, 64, st);
    }
    hipLaunchKernelGGL(filter_keys_u32_kernel, g, b, 0, st, (const uint32_t*)val_in, count, desc, (uint32_t*)key_in);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t*)key_in, (uint32_t*)key_out, idx_in, idx_out, (int)count, 0, 32, st);
}
// count_dev != nullptr: at most `count` results, *count_dev of them real (the host has not seen the count yet: the grid covers a guess, a grid-stride loop the rest)
hipError_t launch_filter_finish(const uint32_t* idx, const void* val, const void* key, bool f64, bool desc, uint32_t count, const uint32_t* count_dev, uint64_t index_base,
                                uint64_t* out_index, void* out_val, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    const dim3 b(256), g((count + 255) / 256);
    if (f64)
        hipLaunchKernelGGL(filter_finish_f64_kernel, g, b, 0, st, idx, (const double*)val, (const uint64_t*)key, desc, count, count_dev, index_base, out_index, (double*)out_val);
    else
        hipLaunchKernelGGL(filter_finish_u32_kernel, g, b, 0, st, idx, (const uint32_t*)val, (const uint32_t*)key, desc, count, count_dev, index_base, out_index, (uint32_t*)out_val);
    return hipGetLastError();
}

}  // namespace rf
