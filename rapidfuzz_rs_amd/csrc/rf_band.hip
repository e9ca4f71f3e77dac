// rf_band.hip -- Levenshtein with a long query and a small distance cutoff: one 64-bit word sliding down the diagonal.
//
// The reference's hyrroe2003_small_band_with_pm (src/distance/levenshtein.rs:509-617; Hyyro's band variant, after Ukkonen):
// if the distance is to be at most k, only the cells within k of the main diagonal matter, and 2k + 1 <= 64 of them per
// column fit ONE machine word whatever the query length.  The word covers pattern rows [j + k - 63, j + k] at column j and
// moves down one row per column: where the full-matrix recurrence shifts the horizontal deltas up by one (HP << 1,
// HN << 1), the band recurrence shifts the diagonal term down instead (D0 >> 1).  The multi-word scan kernel spends
// ceil(len1 / 64) words of VALU per column on such a query (4 for BASELINE.json configs[2], 8 at 512 symbols); this one
// spends one, plus fetching the pattern bits at a bit offset:
//   * the PM table is staged in LDS as 32-bit words with 64 zero bits below row 0 and above the last row of every symbol's
//     bit-vector, row stride odd (bank spread), so the 64 pattern bits at ANY window position v = start_pos + 64 are three
//     consecutive dwords d0 d1 d2 at index v / 32 funnel-shifted by v % 32: two v_alignbit_b32 with a scalar shift amount
//     (the window position depends on the column only: wavefront-uniform);
//   * the running score follows the reference: down the diagonal while the window's top row is above the last pattern row
//     (score += the diagonal delta, bit 63 of D0 clear), then along the last row (+HP, -HN at a bit that moves down).
// Exact whenever the distance is <= k; any value > k otherwise, which the finishing compare turns into None -- the same
// contract the reference relies on (levenshtein.rs:1059-1066).  A wavefront abandons a tile as soon as every lane is past
// break_score (levenshtein.rs:519-523, :568-570), and candidates whose length differs from the query's by more than k are
// None without being read.
#include "rf_device.hpp"
#include "rf_band_asm.inc"

namespace rf {

constexpr uint32_t band_row_dwords(uint32_t words) { return 2 * words + 5; }  // 2 zero dwords + 2W + 2 zero dwords + 1 (odd stride)

// One tile down the band: the 64 lanes' candidates (this lane's chunk c at src + c * kWave) against the staged table.  Returns 0 when the lanes' results have been
// written; kDefer: at column p.band_defer_at a tile with few lanes left within break_score gives up -- the lanes that are out get their None, the mask of the others
// is returned (nothing written for them) and the caller lists it for band_sparse_kernel.
template <bool kDefer, bool kList = false>
__device__ __forceinline__ uint64_t band_tile(const ScanParams& p, const uint32_t* lds_band, uint32_t stride, uint32_t pitch_bytes, bool asm_run, const uint4* src,
                                              uint32_t len2, bool valid, uint32_t idx, bool& open)
{
    const uint32_t len1 = p.len1, k = p.band_k;
    // levenshtein.rs:515-531
    uint64_t vp = ~0ull << (63 - k), vn = 0;
    const uint32_t break_score = 2 * k + len2 - len1;  // (diff <= k: never negative)
    const uint32_t first = min(len1 - k, len2);        // columns walked down the diagonal (len1 > 64 > k)
    uint32_t diag_hits = 0;                            // set diagonal deltas seen in phase 1: score = k + columns - hits
    uint32_t score = k;                                // phase 2 continues from the phase-1 total
    uint32_t v = k + 1;                                // window position + 64: bit v - 64 + b of the pattern is window bit b
    bool dead = false;
    uint64_t handed = 0;
    const uint32_t nch = (len2 + kChunk - 1) / kChunk;
    uint4 cur = nch ? load_chunk(src) : make_uint4(0, 0, 0, 0);
    auto chunks = [&](uint32_t c_begin, uint32_t c_end) {
    for (uint32_t c = c_begin; c < c_end && !dead; ++c) {
        uint4 nxt = cur;
        if (c + 1 < nch) nxt = load_chunk(src + (size_t)(c + 1) * kWave);
        const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
        // 16 columns = 4 dwords x 4 bytes with COMPILE-TIME byte positions: `((dw >> 8k) & 0xFF) * stride` is one
        // v_mul_u32_u24_sdwa; walking the chunk with a running byte shift cost three v_alignbit_b32 + a shift + an and + a
        // v_mul_lo_u32 per column -- six instructions, five of them half-rate on gfx950 (profiles/issue_rates_r02.txt).
        // Columns go in runs of 8 (the reference's break test, :568-570 / :607-609, is evaluated per wavefront after each run);
        // a run that lies entirely on the diagonal walk and inside the chunk -- the common case -- is a straight line with no
        // per-column scalar tests: the first version spent 340 scalar instructions per tile next to 550 vector ones, and four
        // SIMDs share one scalar unit.
        const uint32_t dws[4] = {cur.x, cur.y, cur.z, cur.w};
        auto column = [&](uint32_t sym, uint32_t j, bool diagonal) {
            const uint32_t* row = lds_band + sym * stride + (v >> 5);
            const uint32_t d0w = row[0], d1w = row[1], d2w = row[2];
            const uint32_t sh = v & 31;
            const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(d2w, d1w, sh) << 32) | __builtin_amdgcn_alignbit(d1w, d0w, sh);
            const uint64_t sum = (x & vp) + vp;
            const uint64_t e = lut3<T_XOR_OR>(sum, vp, x);
            const uint64_t d0 = e | vn;                     // levenshtein.rs:556 / :593
            const uint64_t hp = lut3<T_OR_NOR>(vn, e, vp);  // vn | ~(d0 | vp)
            const uint64_t hn = e & vp;                     // d0 & vp (vp & vn == 0)
            if (diagonal) {                                 // :560-562
                diag_hits += (uint32_t)(d0 >> 63);
            } else {                                        // :597-600: the last row, at a bit that moves down
                const uint64_t hmask = 1ull << (62 - (j - first));
                score += (hp & hmask) != 0;
                score -= (hn & hmask) != 0;
            }
            const uint64_t d0s = d0 >> 1;
            vp = lut3<T_OR_NOR>(hn, d0s, hp);               // :571 / :611
            vn = d0s & hp;
            ++v;
        };
#pragma unroll
        for (int run = 0; run < 2; ++run) {
            const uint32_t b0 = (uint32_t)run * 8, j0 = c * kChunk + b0;
            if (b0 >= cols || dead) break;  // wavefront-uniform
            if (b0 + 8 <= cols && j0 + 8 <= first) {
                if (asm_run) {
                    // the eight columns as one asm block (tools/gen_band_asm.py: 23 VALU per column against the ~30 hipcc writes); the diagonal bits come back
                    // as a shift register
                    uint32_t vpl = (uint32_t)vp, vph = (uint32_t)(vp >> 32), vnl = (uint32_t)vn, vnh = (uint32_t)(vn >> 32), acc = 0;
                    asm volatile(RF_BAND_RUN8_ASM
                                 : [vpl] "+v"(vpl), [vph] "+v"(vph), [vnl] "+v"(vnl), [vnh] "+v"(vnh), [acc] "+v"(acc)
                                 : [dw0] "v"(dws[run * 2]), [dw1] "v"(dws[run * 2 + 1]), [pitch] "v"(pitch_bytes), [v0] "s"(uniform(v))
                                 : RF_BAND_RUN8_CLOBBERS);
                    vp = ((uint64_t)vph << 32) | vpl;
                    vn = ((uint64_t)vnh << 32) | vnl;
                    diag_hits += (uint32_t)__popc(acc & 0xFFu);
                    v += 8;
                } else {
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) column((dws[run * 2 + kb / 4] >> (8 * (kb % 4))) & 0xFFu, j0 + kb, true);
                }
                score = k + (j0 + 8) - diag_hits;           // :561: the running total of the diagonal walk
            } else {
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    if (b0 + kb < cols) {
                        const uint32_t j = j0 + kb;
                        const bool diagonal = j < first;
                        column((dws[run * 2 + kb / 4] >> (8 * (kb % 4))) & 0xFFu, j, diagonal);
                        if (diagonal) score = k + (j + 1) - diag_hits;
                    }
                }
            }
            if (__ballot(valid && score <= break_score) == 0) dead = true;
        }
        cur = nxt;
    }
    };
    if constexpr (kDefer) {
        // few lanes left with most of the columns to go: hand them to the dense second pass (band_sparse_kernel) and take the next tile.  Asked once, at a chunk end,
        // BETWEEN two instances of the chunk loop: the test inside the loop -- after every run of 8, or after every chunk -- cost every column 3..5 %.
        const uint32_t dc = min(p.band_defer_at / (uint32_t)kChunk, nch);
        chunks(0, dc);
        if (dc < nch && !dead) {
            const uint64_t alive = __ballot(valid && score <= break_score);
            if ((uint32_t)__popcll(alive) <= p.band_defer_max) {
                // ... once the LAUNCH has seen band_defer_after such tiles: a second pass over a handful of survivors is one wavefront walking every column
                // alone (~25 us at 256 columns) behind a first pass that would have hidden them -- a random corpus with ten planted near-duplicates.  The
                // launch's tiles are counted in one word until the count is reached; a wavefront that has seen it reached stops asking.
                if (!open) {
                    uint32_t seen = 0;
                    if ((threadIdx.x & (kWave - 1)) == 0) seen = __hip_atomic_fetch_add(p.band_defer_seen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    open = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen) >= p.band_defer_after;
                }
                if (open) handed = alive;
            }
        }
        if (handed == 0) chunks(dc, nch);
    } else {
        chunks(0, nch);
    }
    // (a lane that is handed on has score <= break_score and writes nothing here; the others of its tile are past break_score >= k: None)
    if (valid && !(kDefer && handed != 0 && score <= break_score)) {
        if (dead || score > k)
            emit_none(p, idx);
        else
            emit_usize(p, score, len2, idx);
    }
    if constexpr (kList) return __ballot(valid && (dead || score > k));  // (score_hint's first pass: the lanes it answered None -- the second pass' work list)
    return handed;
}

__device__ __forceinline__ void band_stage_table(const ScanParams& p, uint32_t* lds_band, uint32_t W, uint32_t stride)
{
    for (uint32_t i = threadIdx.x; i < 256 * stride; i += kWave * kWavesPerBlock) {
        const uint32_t c = i / stride, d = i % stride;
        uint32_t v = 0;
        if (d >= 2 && d < 2 + 2 * W) {
            const uint64_t word = p.pm[(size_t)c * W + (d - 2) / 2];
            v = (d & 1) ? (uint32_t)(word >> 32) : (uint32_t)word;  // d - 2 even -> low half
        }
        lds_band[(uint32_t)p.sigma[c] * stride + d] = v;  // the corpus stores renamed symbols
    }
    __syncthreads();
}

// kDefer (single-length corpora, round 6): tiles that still hold a few live lanes at column p.band_defer_at are LISTED (tile, lane mask: the 16-byte entries of
// rf_scan.hip's lane lists, one segment of `cap` entries per wavefront, counts at buf + 4) instead of being run to the end for those few.
template <bool kUniform, bool kDefer, bool kList = false>
__device__ __forceinline__ void band_kernel_body(const ScanParams& p, uint32_t* __restrict__ buf, uint32_t cap)
{
    extern __shared__ uint32_t lds_band[];  // 256 rows x band_row_dwords(words)
    const uint32_t W = p.words, stride = band_row_dwords(W);
    band_stage_table(p, lds_band, W, stride);

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t len1 = p.len1, k = p.band_k;
    const uint32_t pitch_bytes = stride * 4u;
    const bool asm_run = p.band_asm != 0;  // (RF_ASM_BAND=0: the compiled column everywhere, the A/B switch)
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave, n_waves = gridDim.x * kWavesPerBlock;
    uint4* seg = (kDefer || kList) ? reinterpret_cast<uint4*>(buf + 4 + 2 * (size_t)n_waves) + (size_t)gw * cap : nullptr;
    uint32_t kept = 0, kept_lanes = 0;
    bool open = p.band_defer_after == 0;  // this wavefront has seen the launch's count of hand-over candidates reached (0: nothing to count)
    for (uint32_t t = p.tile_begin + gw; t < p.tile_end; t += n_waves) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        bool valid = kUniform ? slot < p.n : idx != kPad;
        if (kUniform && p.run_orig) {  // a length run of a bucketed corpus walked as a single-length corpus of its own: the slot's original index
            idx = p.run_orig[slot];
            valid = idx != kPad;
        }
        const uint32_t diff = len1 > len2 ? len1 - len2 : len2 - len1;
        if (diff > k) {  // levenshtein.rs:1389-1391: the distance is at least the length difference
            if (valid) emit_none(p, idx);
            continue;
        }
        const uint64_t left = band_tile<kDefer, kList>(p, lds_band, stride, pitch_bytes, asm_run, tv.src + lane, len2, valid, idx, open);
        if constexpr (kDefer || kList) {
            if (left != 0) {
                if (lane == 0) seg[kept] = make_uint4(t, (uint32_t)left, (uint32_t)(left >> 32), 0u);
                ++kept;
                kept_lanes += (uint32_t)__popcll(left);
            }
        }
    }
    if constexpr (kDefer || kList) {
        if (lane == 0) reinterpret_cast<uint2*>(buf + 4)[gw] = make_uint2(kept, kept_lanes);
    }
}

template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void band_kernel(const ScanParams p)
{
    band_kernel_body<kUniform, false>(p, nullptr, 0u);
}
// (its own kernel rather than a template argument of the one above: an occupancy attribute on the shared template cost the plain kernel 5 % of its columns)
#ifndef RF_BAND_DEFER_NOATTR
__attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ __launch_bounds__(kWave* kWavesPerBlock) void band_defer_kernel(const ScanParams p, uint32_t* __restrict__ buf, uint32_t cap)
{
    band_kernel_body<true, true>(p, buf, cap);
}

// score_hint's first pass over a single-length corpus (rf_api_scan.hip run_many_hinted): the plain kernel, and every tile that holds lanes it answered None goes on the
// wavefront's list with their mask -- the work list of the caller's own scan (rf_sparse.hip sparse_words_kernel)
__global__ __launch_bounds__(kWave* kWavesPerBlock) void band_list_kernel(const ScanParams p, uint32_t* __restrict__ buf, uint32_t cap)
{
    band_kernel_body<true, false, true>(p, buf, cap);
}

// The second pass of a deferring launch: DENSE tiles of the listed lanes, 64 to a wavefront (rf_sparse.hip has the scheme: the packed entries carry the survivors'
// running count, lane l of dense tile j finds its entry by a binary search across the lanes and its candidate as the n-th set bit of that tile's mask), each lane
// reading its own candidate's chunk rows, columns from 0 (the state a tile had reached when it gave up is not kept: band_defer_at columns out of len2).
__global__ __launch_bounds__(kWave* kWavesPerBlock) void band_sparse_kernel(const ScanParams p)
{
    extern __shared__ uint32_t lds_band[];
    const uint4* __restrict__ list = reinterpret_cast<const uint4*>(p.tile_list);
    const uint32_t entries = uniform(p.tile_list_count[0]), total = uniform(p.tile_list_count[1]);
    const uint32_t n_dense = (total + kWave - 1) / kWave;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (p.band_defer_seen) *p.band_defer_seen = 0;  // (the first pass' count of hand-over candidates: zero again for this stream's next launch)
        if (p.band_report) {  // what the first pass listed, for the host's choice of the next launch's form (rf_api_scan.hip run_many; read there without waiting)
            const uint32_t words[7] = {entries, total, p.tile_end - p.tile_begin, 1u, p.band_defer_at, p.band_defer_max, p.uniform_len};
#pragma unroll
            for (int i = 0; i < 7; ++i) __hip_atomic_store(p.band_report + i, words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (blockIdx.x * kWavesPerBlock >= n_dense) return;  // (the survivors' number is only known here)
    const uint32_t W = p.words, stride = band_row_dwords(W);
    band_stage_table(p, lds_band, W, stride);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t pitch_bytes = stride * 4u;
    const bool asm_run = p.band_asm != 0;
    for (uint32_t j = blockIdx.x * kWavesPerBlock + wave; j < n_dense; j += gridDim.x * kWavesPerBlock) {
        const uint32_t g = j * kWave + lane;
        const bool have = g < total;
        const uint32_t gg = have ? g : total - 1;  // (idle lanes of the last dense tile shadow its last survivor: defined bytes, no result)
        const uint32_t e0 = uniform(p.lane_first[j]);
        const uint4 ent = list[min(e0 + lane, entries - 1)];
        uint32_t lo = 0, hi = kWave;
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t wm = (uint32_t)__shfl((int)ent.w, (int)mid, kWave);
            if (wm <= gg)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t t = (uint32_t)__shfl((int)ent.x, (int)lo, kWave), mlo = (uint32_t)__shfl((int)ent.y, (int)lo, kWave), mhi = (uint32_t)__shfl((int)ent.z, (int)lo, kWave),
                       w = (uint32_t)__shfl((int)ent.w, (int)lo, kWave);
        const uint32_t ls = nth_set_bit(((uint64_t)mhi << 32) | mlo, gg - w);
        uint32_t idx = t * kWave + ls;
        bool valid = have && idx < p.n;
        if (p.run_orig) {
            idx = p.run_orig[idx];
            valid = have && idx != kPad;
        }
        const uint4* src = reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes) + ls;
        bool unused = false;
        (void)band_tile<false>(p, lds_band, stride, pitch_bytes, asm_run, src, p.uniform_len, valid, idx, unused);
    }
}

hipError_t launch_band(const ScanParams& p, hipStream_t stream)
{
    if (p.tile_end <= p.tile_begin) return hipSuccess;
    const size_t lds = (size_t)256 * band_row_dwords(p.words) * sizeof(uint32_t);
    // every workgroup stages the 32-bit band table (256 x (2W + 5) dwords: 13 KiB at 256 symbols) before its first tile, and a
    // tile dies after 8-16 columns: half the general launches' workgroups per CU measured best (configs[2] corpus, cutoff 8:
    // 8 or 16 per CU 76.7 Gpairs/s, 24: 72.7, 32: 67.8, 64: 55.5)
    const int band_grid = std::max(1, std::min(scan_grid(p.tile_end - p.tile_begin), (scan_max_grid() + 1) / 2));
    const dim3 g(band_grid), b(kWave * kWavesPerBlock);
    static const bool use_asm = [] { const char* e = getenv("RF_ASM_BAND"); return !e || atoi(e) != 0; }();
    ScanParams pa = p;
    pa.band_asm = use_asm ? 1u : 0u;
    // LANE COMPACTION (single-length corpora): on a corpus where a good share of the candidates IS near the query no tile dies -- every one of
    // them holds a few -- and each ran all its columns on 64 lanes for those few.  With a list buffer at hand the first launch gives such a tile up at column
    // band_defer_at (a random candidate is past break_score by then) unless most of its lanes are still in, and band_sparse_kernel runs the listed lanes 64 to a
    // wavefront from column 0.  RF_BAND_DEFER=0 switches it off, RF_BAND_DEFER_AT / RF_BAND_DEFER_MAX move the column and the lane count (default k + 8 rounded up to a whole chunk / 44),
    // RF_BAND_DEFER_AFTER the number of such tiles a launch runs in place first (0).
    static const bool defer_on = [] { const char* e = getenv("RF_BAND_DEFER"); return !e || atoi(e) != 0; }();
    static const uint32_t defer_at_env = [] { const char* e = getenv("RF_BAND_DEFER_AT"); return e ? ((uint32_t)atoi(e) + 15u) / 16u * 16u : 0u; }();
    static const uint32_t defer_max = [] { const char* e = getenv("RF_BAND_DEFER_MAX"); return e ? (uint32_t)atoi(e) : 44u; }();
    // (0 since the launch form follows the last launch's report, rf_api_scan.hip: a corpus with a handful of near candidates takes the plain kernel on 15 launches of 16,
    // and counting cost the others one same-address atomic per wavefront -- 16 K of them at the start of the launch, 250 -> 190 us at 1 % near candidates)
    static const uint32_t defer_after = [] { const char* e = getenv("RF_BAND_DEFER_AFTER"); return e ? (uint32_t)atoi(e) : 0u; }();
    // (the column: a candidate unrelated to the query adds nearly one to its score per column from k on and is out past break_score = 2k (+ the length difference):
    // k + 8 columns see the random lanes off, rounded up to a chunk end -- 16 at k = 8, 48 at the k = 31 of a hinted scan's first pass.  Where they last longer
    // -- four-symbol alphabets -- the tile is simply not sparse yet at that column and runs on in place.)
    const uint32_t defer_at = defer_at_env ? defer_at_env : std::min(64u, std::max(16u, (p.band_k + 8u + 15u) / 16u * 16u));
    const uint32_t G = (uint32_t)band_grid * kWavesPerBlock, n_tiles = p.tile_end - p.tile_begin;
    if (p.band_list) {  // (the caller has checked band_list_geometry() and holds the list buffer; it launches the second pass itself)
        if (!p.tile_list_buf || p.tiles || G > 16384u) return hipErrorInvalidValue;
        const uint32_t cap = (n_tiles + G - 1) / G;
        if (lds > 48 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_list_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(band_list_kernel, g, b, lds, stream, pa, p.tile_list_buf, cap);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        uint32_t* packed_at = p.tile_list_buf + 4 + 2 * (size_t)G + 4 * (size_t)G * cap;
        return launch_lane_list_pack(p.tile_list_buf, G, cap, packed_at + 4 * ((size_t)n_tiles + 2), stream);
    }
    const bool defer = defer_on && defer_at && p.lane_list && p.tile_list_buf && p.band_defer_seen && !p.tiles && G <= 16384u && p.uniform_len >= defer_at + 64u &&
                       p.len1 >= defer_at + 64u + p.band_k;
    if (defer) {
        const uint32_t cap = (n_tiles + G - 1) / G;  // the tiles one wavefront walks
        pa.band_defer_at = defer_at;
        pa.band_defer_max = std::min(defer_max, 63u);
        pa.band_defer_after = defer_after;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_defer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_sparse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(band_defer_kernel, g, b, lds, stream, pa, p.tile_list_buf, cap);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        uint32_t* packed_at = p.tile_list_buf + 4 + 2 * (size_t)G + 4 * (size_t)G * cap;
        uint32_t* first_at = packed_at + 4 * ((size_t)n_tiles + 2);
        e = launch_lane_list_pack(p.tile_list_buf, G, cap, first_at, stream);
        if (e != hipSuccess) return e;
        ScanParams p2 = pa;
        p2.tile_list = packed_at;
        p2.lane_first = first_at;
        p2.tile_list_count = p.tile_list_buf;
        // (half the first pass' grid: the workgroups beyond the survivors leave at once, but launching them is 2 us of an empty second pass)
        hipLaunchKernelGGL(band_sparse_kernel, dim3(std::max(1, band_grid / 2)), b, lds, stream, p2);
        return hipGetLastError();
    }
    auto kern = p.tiles ? band_kernel<false> : band_kernel<true>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, g, b, lds, stream, pa);
    return hipGetLastError();
}

// where a band_list launch over p's tiles leaves its packed list and first[] inside p.tile_list_buf (false: such a launch does not fit the buffer's 16 K segments)
bool band_list_geometry(const ScanParams& p, uint32_t** packed_at, uint32_t** first_at)
{
    if (p.tile_end <= p.tile_begin || !p.tile_list_buf) return false;
    const uint32_t n_tiles = p.tile_end - p.tile_begin;
    const int band_grid = std::max(1, std::min(scan_grid(n_tiles), (scan_max_grid() + 1) / 2));
    const uint32_t G = (uint32_t)band_grid * kWavesPerBlock, cap = (n_tiles + G - 1) / G;
    if (G > 16384u) return false;
    *packed_at = p.tile_list_buf + 4 + 2 * (size_t)G + 4 * (size_t)G * cap;
    *first_at = *packed_at + 4 * ((size_t)n_tiles + 2);
    return true;
}

}  // namespace rf
