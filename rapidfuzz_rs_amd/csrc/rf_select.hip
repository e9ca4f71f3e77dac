// rf_select.hip -- exact top-k SELECTION over a device-resident score vector: the general top-k path
// (any k, u32 or f64 scores), next to the in-scan wavefront lists of rf_scan.hip (k <= 64, u32 scores).
//
// Definition (rfgpu.h "top-k"): drop None, order by (score ascending for distances / descending for similarities, index
// ascending), keep the first k.  The reference has no extract API; the oracle is "sort the full result".
//
// A score becomes an order-preserving unsigned key (u32: the score, or 0xFFFFFFFE minus it; f64: the sign-flipped IEEE bits or
// their complement), None becomes the maximum key and is never selected.  Then
//   1. minmax pass            the key range; the radix passes below start at its first differing bit
//   2. digit passes (11 bits) histogram of the next digit among keys matching the prefix found so far -> the k-th smallest
//                             KEY value T exactly, and how many keys are below it
//   3. block-count pass       per block of 2048 scores: #keys < T and #keys == T
//   4. block scan             exclusive prefix sums (one workgroup)
//   5. emit pass              every key < T, and the first (k - #below) keys == T in INDEX order (scores are stored in
//                             original candidate order, so index order is position order) -> exactly min(k, #valid) pairs
// The k pairs are sorted by (key, index) on the host (they are going to host arrays anyway).  Every pass streams the
// score vector once: 4 + ceil(significant bits / 11) reads of 4 or 8 bytes per candidate.
#include <algorithm>

#include "rf_device.hpp"

namespace rf {

constexpr int kSelThreads = 256;
constexpr int kSelPerThread = 8;
constexpr int kSelBlock = kSelThreads * kSelPerThread;  // scores per block
constexpr int kSelBins = 2048;                          // 11-bit digits

// order-preserving keys; the all-ones key is None
template <class Key>
struct KeyOf;
template <>
struct KeyOf<uint32_t> {
    using Score = uint32_t;
    // valid scores are 0 .. 0xFFFFFFFE (0xFFFFFFFF is None), so the descending key is 0xFFFFFFFE - s: a similarity of 0 must
    // not land on the None key (`~s` did, and the selection path lost every zero-similarity candidate)
    static __device__ __forceinline__ uint32_t get(uint32_t s, bool desc) { return s == RF_NONE_U32 ? ~0u : (desc ? 0xFFFFFFFEu - s : s); }
};
template <>
struct KeyOf<uint64_t> {
    using Score = double;
    static __device__ __forceinline__ uint64_t get(double d, bool desc)
    {
        if (d != d) return ~0ull;  // NaN = None
        uint64_t b = (uint64_t)__double_as_longlong(d);
        if (b == 0x8000000000000000ull) b = 0;                      // -0.0 == +0.0
        b ^= (b >> 63) ? ~0ull : 0x8000000000000000ull;             // ascending in the value
        return desc ? ~b : b;                                       // (never all ones for a number: that pattern is a NaN)
    }
};

struct SelectCtl {  // device-resident, 64-bit fields
    unsigned long long min_key, max_key;  // over valid keys
    unsigned long long valid;             // number of valid (not None) scores
};

template <class Key>
__global__ __launch_bounds__(kSelThreads) void sel_minmax_kernel(const typename KeyOf<Key>::Score* __restrict__ s, uint32_t n, bool desc, SelectCtl* ctl)
{
    Key lo = ~(Key)0, hi = 0;
    uint32_t valid = 0;
    for (uint32_t i = blockIdx.x * kSelThreads + threadIdx.x; i < n; i += gridDim.x * kSelThreads) {
        const Key k = KeyOf<Key>::get(s[i], desc);
        if (k != ~(Key)0) {
            lo = k < lo ? k : lo;
            hi = k > hi ? k : hi;
            ++valid;
        }
    }
    __shared__ unsigned long long s_lo, s_hi, s_valid;
    if (threadIdx.x == 0) s_lo = ~0ull, s_hi = 0, s_valid = 0;
    __syncthreads();
    if (valid) {
        atomicMin(&s_lo, (unsigned long long)lo);
        atomicMax(&s_hi, (unsigned long long)hi);
        atomicAdd(&s_valid, (unsigned long long)valid);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) {
        atomicMin(&ctl->min_key, s_lo);
        atomicMax(&ctl->max_key, s_hi);
        atomicAdd(&ctl->valid, s_valid);
    }
}

// histogram of digit (key >> shift) & 2047 among keys with (key & prefix_mask) == prefix; per-workgroup LDS histogram,
// wavefront-aggregated (integer scores concentrate on a handful of bins: one LDS atomic per DISTINCT bin per wavefront)
template <class Key>
__global__ __launch_bounds__(kSelThreads) void sel_hist_kernel(const typename KeyOf<Key>::Score* __restrict__ s, uint32_t n, bool desc, Key prefix_mask,
                                                               Key prefix, uint32_t shift, uint32_t digit_mask, unsigned long long* __restrict__ hist)
{
    __shared__ uint32_t lh[kSelBins];
    for (int i = threadIdx.x; i < kSelBins; i += kSelThreads) lh[i] = 0;
    __syncthreads();
    for (uint32_t base = blockIdx.x * kSelThreads; base < n; base += gridDim.x * kSelThreads) {
        const uint32_t i = base + threadIdx.x;
        Key k = ~(Key)0;
        if (i < n) k = KeyOf<Key>::get(s[i], desc);
        bool active = k != ~(Key)0 && (k & prefix_mask) == prefix;
        const uint32_t bin = (uint32_t)(k >> shift) & digit_mask;  // (the last digit may be narrower than 11 bits)
        uint64_t todo = __ballot(active);
        while (todo) {
            const uint32_t leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t b = __builtin_amdgcn_readlane(bin, leader);
            const uint64_t same = __ballot(active && bin == b);
            if ((threadIdx.x & 63) == leader) atomicAdd(&lh[b], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelBins; i += kSelThreads)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// per block of kSelBlock scores: how many keys are below T, how many equal T
template <class Key>
__global__ __launch_bounds__(kSelThreads) void sel_count_kernel(const typename KeyOf<Key>::Score* __restrict__ s, uint32_t n, bool desc, Key T,
                                                                uint32_t* __restrict__ cnt_less, uint32_t* __restrict__ cnt_eq)
{
    uint32_t less = 0, eq = 0;
    const uint32_t b0 = blockIdx.x * kSelBlock;
#pragma unroll
    for (int r = 0; r < kSelPerThread; ++r) {
        const uint32_t i = b0 + r * kSelThreads + threadIdx.x;
        if (i < n) {
            const Key k = KeyOf<Key>::get(s[i], desc);
            less += k < T;
            eq += k == T && k != ~(Key)0;
        }
    }
    __shared__ uint32_t s_less, s_eq;
    if (threadIdx.x == 0) s_less = s_eq = 0;
    __syncthreads();
    if (less) atomicAdd(&s_less, less);
    if (eq) atomicAdd(&s_eq, eq);
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt_less[blockIdx.x] = s_less;
        cnt_eq[blockIdx.x] = s_eq;
    }
}

// exclusive scan of both count arrays, in place (one workgroup; nb is n / 2048: tens of thousands at most)
__global__ __launch_bounds__(kSelThreads) void sel_scan_kernel(uint32_t* cnt_less, uint32_t* cnt_eq, uint32_t nb)
{
    __shared__ unsigned long long part[2][kSelThreads];
    const uint32_t per = (nb + kSelThreads - 1) / kSelThreads;
    const uint32_t lo = threadIdx.x * per, hi = min(nb, lo + per);
    unsigned long long a = 0, b = 0;
    for (uint32_t i = lo; i < hi; ++i) a += cnt_less[i], b += cnt_eq[i];
    part[0][threadIdx.x] = a;
    part[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ra = 0, rb = 0;
        for (int t = 0; t < kSelThreads; ++t) {
            const unsigned long long xa = part[0][t], xb = part[1][t];
            part[0][t] = ra;
            part[1][t] = rb;
            ra += xa;
            rb += xb;
        }
    }
    __syncthreads();
    unsigned long long ra = part[0][threadIdx.x], rb = part[1][threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t xa = cnt_less[i], xb = cnt_eq[i];
        cnt_less[i] = (uint32_t)ra;  // (both totals fit 32 bits: they are bounded by n)
        cnt_eq[i] = (uint32_t)std::min<unsigned long long>(rb, 0xFFFFFFFFull);
        ra += xa;
        rb += xb;
    }
}

// emit: keys below T go to out[off_less[block] + rank], keys equal to T with global equal-rank < need_eq go to
// out[n_less + rank]; ranks inside a block follow index order (row by row, lane prefix inside a row)
template <class Key>
__global__ __launch_bounds__(kSelThreads) void sel_emit_kernel(const typename KeyOf<Key>::Score* __restrict__ s, uint32_t n, bool desc, Key T,
                                                               const uint32_t* __restrict__ off_less, const uint32_t* __restrict__ off_eq, uint32_t n_less,
                                                               uint32_t need_eq, Key* __restrict__ out_key, uint32_t* __restrict__ out_idx)
{
    __shared__ uint32_t w_less[kSelThreads / kWave], w_eq[kSelThreads / kWave];
    const uint32_t b0 = blockIdx.x * kSelBlock;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t run_less = off_less[blockIdx.x], run_eq = off_eq[blockIdx.x];
    if (run_eq >= need_eq && off_less[blockIdx.x] == (blockIdx.x + 1 < gridDim.x ? off_less[blockIdx.x + 1] : 0xFFFFFFFFu)) {
        // nothing below T in this block and the equal quota is already used up: nothing to emit
        // (cheap early exit for the common case of a tight T; correctness does not depend on it)
        return;
    }
    for (int r = 0; r < kSelPerThread; ++r) {  // index order: row r covers indices b0 + r*256 .. +255
        const uint32_t i = b0 + r * kSelThreads + threadIdx.x;
        Key k = ~(Key)0;
        if (i < n) k = KeyOf<Key>::get(s[i], desc);
        const bool is_less = k < T, is_eq = k == T && k != ~(Key)0;
        const uint64_t ml = __ballot(is_less), me = __ballot(is_eq);
        if (lane == 0) w_less[wave] = (uint32_t)__popcll(ml), w_eq[wave] = (uint32_t)__popcll(me);
        __syncthreads();
        uint32_t pl = 0, pe = 0, tl = 0, te = 0;
        for (uint32_t w = 0; w < (uint32_t)(kSelThreads / kWave); ++w) {
            if (w < wave) pl += w_less[w], pe += w_eq[w];
            tl += w_less[w], te += w_eq[w];
        }
        const uint64_t below = (1ull << lane) - 1;
        if (is_less) {
            const uint32_t pos = run_less + pl + (uint32_t)__popcll(ml & below);
            out_key[pos] = k;
            out_idx[pos] = i;
        }
        if (is_eq) {
            const uint32_t rank = run_eq + pe + (uint32_t)__popcll(me & below);
            if (rank < need_eq) {
                out_key[n_less + rank] = k;
                out_idx[n_less + rank] = i;
            }
        }
        run_less += tl;
        run_eq += te;
        __syncthreads();
    }
}

// ---- host-callable launchers ----------------------------------------------------------------------------------------
static int sel_grid(uint32_t n) { return (int)std::min<uint32_t>((n + kSelThreads - 1) / kSelThreads, (uint32_t)scan_max_grid()); }

template <class Key>
static hipError_t launch_minmax_t(const void* s, uint32_t n, bool desc, void* ctl, hipStream_t st)
{
    hipLaunchKernelGGL((sel_minmax_kernel<Key>), dim3(sel_grid(n)), dim3(kSelThreads), 0, st, (const typename KeyOf<Key>::Score*)s, n, desc, (SelectCtl*)ctl);
    return hipGetLastError();
}
hipError_t launch_select_minmax(const void* s, bool f64, uint32_t n, bool desc, void* ctl, hipStream_t st)
{
    return f64 ? launch_minmax_t<uint64_t>(s, n, desc, ctl, st) : launch_minmax_t<uint32_t>(s, n, desc, ctl, st);
}
hipError_t launch_select_hist(const void* s, bool f64, uint32_t n, bool desc, uint64_t prefix_mask, uint64_t prefix, uint32_t shift, uint32_t bits,
                              unsigned long long* hist, hipStream_t st)
{
    const uint32_t digit_mask = (1u << bits) - 1;
    if (f64)
        hipLaunchKernelGGL((sel_hist_kernel<uint64_t>), dim3(sel_grid(n)), dim3(kSelThreads), 0, st, (const double*)s, n, desc, prefix_mask, prefix, shift, digit_mask, hist);
    else
        hipLaunchKernelGGL((sel_hist_kernel<uint32_t>), dim3(sel_grid(n)), dim3(kSelThreads), 0, st, (const uint32_t*)s, n, desc, (uint32_t)prefix_mask,
                           (uint32_t)prefix, shift, digit_mask, hist);
    return hipGetLastError();
}
uint32_t select_blocks(uint32_t n) { return (n + kSelBlock - 1) / kSelBlock; }
hipError_t launch_select_count(const void* s, bool f64, uint32_t n, bool desc, uint64_t T, uint32_t* cnt_less, uint32_t* cnt_eq, hipStream_t st)
{
    const dim3 g(select_blocks(n)), b(kSelThreads);
    if (f64)
        hipLaunchKernelGGL((sel_count_kernel<uint64_t>), g, b, 0, st, (const double*)s, n, desc, T, cnt_less, cnt_eq);
    else
        hipLaunchKernelGGL((sel_count_kernel<uint32_t>), g, b, 0, st, (const uint32_t*)s, n, desc, (uint32_t)T, cnt_less, cnt_eq);
    hipLaunchKernelGGL(sel_scan_kernel, dim3(1), b, 0, st, cnt_less, cnt_eq, select_blocks(n));
    return hipGetLastError();
}
hipError_t launch_select_emit(const void* s, bool f64, uint32_t n, bool desc, uint64_t T, const uint32_t* off_less, const uint32_t* off_eq, uint32_t n_less,
                              uint32_t need_eq, void* out_key, uint32_t* out_idx, hipStream_t st)
{
    const dim3 g(select_blocks(n)), b(kSelThreads);
    if (f64)
        hipLaunchKernelGGL((sel_emit_kernel<uint64_t>), g, b, 0, st, (const double*)s, n, desc, T, off_less, off_eq, n_less, need_eq, (uint64_t*)out_key, out_idx);
    else
        hipLaunchKernelGGL((sel_emit_kernel<uint32_t>), g, b, 0, st, (const uint32_t*)s, n, desc, (uint32_t)T, off_less, off_eq, n_less, need_eq,
                           (uint32_t*)out_key, out_idx);
    return hipGetLastError();
}

// ---- top-k entries (rfgpu.h rf_topk_entry): 16 bytes {order-preserving key, 64-bit global index} ---------------------------------
// in-scan keys ((score or ~score) << 32 | local index, UINT64_MAX = empty) -> entries
__global__ void keys_to_entries_kernel(const uint64_t* __restrict__ keys, uint32_t k, uint64_t index_base, rf_topk_entry* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const uint64_t x = keys[i];
    rf_topk_entry e;
    e.key = x == ~0ull ? ~0ull : (x >> 32);  // distance: the score; similarity: ~score as u32 = 0xFFFFFFFF - score
    e.index = x == ~0ull ? ~0ull : index_base + (uint32_t)x;
    out[i] = e;
}
// ---------------------------------------------------------------------------------------------------
// Top-k (k <= 64, u32 scores) as ONE streaming pass over a score vector in original candidate order (round 4).  The scans that have a
// whole-kernel asm form (Levenshtein up to 4 words, OSA; any corpus) carry no top-k epilogue -- their per-tile code is a store -- so a
// top-k over such a shape is that scan into a score vector + this pass: 4 bytes per candidate read once (0.08 ms per 100 M) with the
// wavefront lists, the launch-wide bound and the in-launch selection of the scan kernels' top-k mode (rf_device.hpp WaveTopK,
// topk_block_publish).  Replaces round 2's fixed-shape hybrid kernels (single-length corpora, lengths that are multiples of 16) and
// the compiled scans that served every other shape in top-k mode; key = (score or ~score) << 32 | (key_index_base + index), as there.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave* kWavesPerBlock) void topk_scores_kernel(const ScanParams p, const uint32_t* __restrict__ scores, uint32_t n)
{
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = uniform(threadIdx.x / kWave);
    WaveTopK best;
    best.init();
    uint64_t limit = ~0ull, published = ~0ull;
    // a lane takes 4 consecutive scores (one 16-byte load; rows of 1 KiB per wavefront), kRows rows in flight: a small grid -- few lists to
    // warm up and to merge -- still keeps ~8 MB on its way
    constexpr uint32_t kRows = 4, kPer = 4;
    const bool vec = (reinterpret_cast<uintptr_t>(scores) & 15u) == 0;
    const uint64_t stride = (uint64_t)gridDim.x * kWavesPerBlock * kWave * kPer;
    uint32_t trips = 0;
    for (uint64_t base = ((uint64_t)blockIdx.x * kWavesPerBlock + wave) * kWave * kPer; base < n; base += kRows * stride) {
        uint32_t v[kRows][kPer];
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint64_t i = base + (uint64_t)r * stride + (uint64_t)lane * kPer;
            if (vec && i + kPer <= n) {
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                const v4u q = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(scores + i));
                v[r][0] = q.x, v[r][1] = q.y, v[r][2] = q.z, v[r][3] = q.w;
            } else {
#pragma unroll
                for (uint32_t j = 0; j < kPer; ++j) v[r][j] = i + j < n ? scores[i + j] : RF_NONE_U32;
            }
        }
        // The launch-wide bound is published and re-read every 4th trip only, and only by a wavefront whose own k-th best beats what it
        // published before: this pass has no sampled bound to start from, so every list warms up with ~k ln(n / k) improvements, and one
        // atomic per improvement (what the scan kernels do behind their sampled bound) was 870 k same-address atomics = 2 ms.
        if ((trips++ & 3u) == 0) {
            const uint64_t w = best.worst(p.topk_k);
            if (w < published && w <= limit) {
                if (lane == 0) asm volatile("global_atomic_umin_x2 %0, %1, off" ::"v"(p.topk_bound), "v"(w) : "memory");
                published = w;
            }
            topk_refresh_bound(p, limit);
        }
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r)
#pragma unroll
            for (uint32_t j = 0; j < kPer; ++j) {
                const uint32_t idx = (uint32_t)(base + (uint64_t)r * stride + (uint64_t)lane * kPer + j);
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v[r][j] : v[r][j]) << 32) | (p.key_index_base + idx);
                if (best.offer(mine, v[r][j] != RF_NONE_U32, p.topk_k, lane, limit)) {
                    const uint64_t w = best.worst(p.topk_k);
                    limit = w < limit ? w : limit;
                }
            }
    }
    topk_block_publish(p, best, lds_topk, wave, lane, limit);
}
hipError_t launch_topk_scores(const ScanParams& p, const uint32_t* scores, uint32_t n, hipStream_t st)
{
    // 2 workgroups per CU (scan_max_grid = 32 per CU): 2048 lists at most; never more workgroups than there are 4 KiB stretches of scores
    const uint32_t chunks = (n + kWave * kWavesPerBlock * 4 - 1) / (kWave * kWavesPerBlock * 4);
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(chunks, (uint32_t)std::max(1, scan_max_grid() / 16)));
    hipLaunchKernelGGL(topk_scores_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0, st, p, scores, n);
    return hipGetLastError();
}

hipError_t launch_keys_to_entries(const uint64_t* keys, uint32_t k, uint64_t index_base, rf_topk_entry* out, hipStream_t st)
{
    hipLaunchKernelGGL(keys_to_entries_kernel, dim3((k + 63) / 64), dim3(64), 0, st, keys, k, index_base, out);
    return hipGetLastError();
}
// merge by ranking: (key, index) pairs are unique, so an entry's position in the merged list is the number of entries below it.
// n is (ranks x k): hundreds to a few thousand entries -- every workgroup ranks 256 of them against all n.
__global__ __launch_bounds__(256) void merge_entries_kernel(const rf_topk_entry* __restrict__ in, uint32_t n, uint32_t k, rf_topk_entry* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < k && i >= n) out[i] = rf_topk_entry{~0ull, ~0ull};  // (fewer inputs than k)
    if (i >= n) return;
    const rf_topk_entry me = in[i];
    uint32_t rank = 0, valid = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const rf_topk_entry o = in[j];
        rank += (o.key < me.key || (o.key == me.key && o.index < me.index)) ? 1u : 0u;
        valid += (o.key != ~0ull || o.index != ~0ull) ? 1u : 0u;
    }
    const bool empty = me.key == ~0ull && me.index == ~0ull;
    if (!empty && rank < k) out[rank] = me;
    // the tail behind the valid entries: one writer per position (the thread whose index equals the position)
    if (i < k && i >= valid) out[i] = rf_topk_entry{~0ull, ~0ull};
}
hipError_t launch_merge_entries(const rf_topk_entry* in, uint32_t n, uint32_t k, rf_topk_entry* out, hipStream_t st)
{
    const uint32_t cover = n > k ? n : k;
    hipLaunchKernelGGL(merge_entries_kernel, dim3((cover + 255) / 256), dim3(256), 0, st, in, n, k, out);
    return hipGetLastError();
}

}  // namespace rf
