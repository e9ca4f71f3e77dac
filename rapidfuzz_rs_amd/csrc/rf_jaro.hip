// rf_jaro.hip -- Jaro / Jaro-Winkler: flagging and transposition passes (jaro.rs:147-190, :192-420, :339-368), the
// f64 epilogue replaying jaro.rs:516-598 / jaro_winkler.rs:103-141 / details/distance.rs:277-385 bit for bit.
#include <cstddef>
#include "rf_device.hpp"

namespace rf {

// ---------------------------------------------------------------------------------------------------
// Jaro / Jaro-Winkler, single-word path (jaro.rs:516-598 with len1, len2 <= 64 after the window truncation
// of :550-565).  Per lane: P_flag / T_flag in two VGPR pairs; the candidate's <= 64 bytes stay in 16 VGPRs for
// the second (transposition) pass.  f64 epilogue in the reference's operation order, -ffp-contract=off.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t blsi64(uint64_t v) { return v & (0 - v); }  // intrinsics.rs:35-37
__device__ __forceinline__ uint64_t mask_lsb64(uint32_t n) { return n < 64 ? (1ull << n) - 1 : ~0ull; }  // :28-34

struct JaroRaw {
    uint32_t common, transpositions, prefix;
    bool eq11;  // the two single characters are equal (only meaningful for 1 x 1)
};

// jaro.rs:106-119
__device__ __forceinline__ double jaro_calculate_similarity(uint32_t p_len, uint32_t t_len, uint32_t common, uint32_t transposition)
{
    transposition /= 2;
    double sim = 0.0;
    sim += (double)common / (double)p_len;
    sim += (double)common / (double)t_len;
    sim += ((double)common - (double)transposition) / (double)common;
    return sim / 3.0;
}
// jaro.rs:122-131
__device__ __forceinline__ bool jaro_length_filter(uint32_t p_len, uint32_t t_len, double cutoff)
{
    if (t_len == 0 || p_len == 0) return false;
    const double min_len = (double)min(p_len, t_len);
    double sim = min_len / (double)p_len + min_len / (double)t_len + 1.0;
    sim /= 3.0;
    return sim >= cutoff;
}
// jaro.rs:134-145
__device__ __forceinline__ bool jaro_common_char_filter(uint32_t p_len, uint32_t t_len, uint32_t common, double cutoff)
{
    if (common == 0) return false;
    double sim = 0.0;
    sim += (double)common / (double)p_len;
    sim += (double)common / (double)t_len;
    sim += 1.0;
    sim /= 3.0;
    return sim >= cutoff;
}
// jaro::similarity_with_pm (jaro.rs:516-598) given the flag counts; every early `return 0.0` of the reference
// is a select here because the flags were computed unconditionally.
__device__ __forceinline__ double jaro_similarity(uint32_t len1, uint32_t len2, const JaroRaw& r, double cutoff)
{
    if (cutoff > 1.0) return 0.0;                              // :533-535
    if (len1 == 0 && len2 == 0) return 1.0;                     // :537-539
    if (!jaro_length_filter(len1, len2, cutoff)) return 0.0;    // :542-544
    if (len1 == 1 && len2 == 1) return r.eq11 ? 1.0 : 0.0;      // :546-548
    if (!jaro_common_char_filter(len1, len2, r.common, cutoff)) return 0.0;  // :579-581
    return jaro_calculate_similarity(len1, len2, r.common, r.transpositions);
}
// jaro_winkler::similarity_with_pm (jaro_winkler.rs:103-141)
__device__ __forceinline__ double jw_similarity(uint32_t len1, uint32_t len2, const JaroRaw& r, double prefix_weight, double cutoff)
{
    double jaro_cutoff = cutoff;
    if (jaro_cutoff > 0.7) {  // :125-133
        const double prefix_sim = (double)r.prefix * prefix_weight;
        jaro_cutoff = prefix_sim >= 1.0 ? 0.7 : fmax(0.7, (prefix_sim - jaro_cutoff) / (prefix_sim - 1.0));
    }
    double sim = jaro_similarity(len1, len2, r, jaro_cutoff);
    if (sim > 0.7) sim += (double)r.prefix * prefix_weight * (1.0 - sim);  // :136-138
    return sim;
}

// Metricf64 (details/distance.rs:277-385) with maximum == 1.0, then score() (common.rs:43-45 / :83-85)
__device__ __forceinline__ double f64_metric_value(const ScanParams& p, uint32_t len2, const JaroRaw& r, bool* keep)
{
    const bool has = p.has_cutoff != 0;
    const double c = p.cutoff_f64;
    auto sim_with = [&](bool has_c, double cc) {  // _similarity: score_cutoff.unwrap_or(0.0)
        const double cut = has_c ? cc : 0.0;
        return p.finish == FIN_JW ? jw_similarity(p.len1, len2, r, p.prefix_weight, cut) : jaro_similarity(p.len1, len2, r, cut);
    };
    auto dist_with = [&](bool has_c, double cc) {  // _distance, :280-302
        const double cs = has_c ? (1.0 >= cc ? 1.0 - cc : 0.0) : 0.0;
        return 1.0 - sim_with(has_c, cs);
    };
    auto ndist_with = [&](bool has_c, double cc) {  // _normalized_distance, :336-361 (maximum = 1.0)
        const double d = dist_with(has_c, 1.0 * cc);
        return d / 1.0;
    };
    double v;
    switch (p.op) {
    case RF_OP_SIMILARITY:
        v = sim_with(has, c);
        *keep = !has || v >= c;
        break;
    case RF_OP_DISTANCE:
        v = dist_with(has, c);
        *keep = !has || v <= c;
        break;
    case RF_OP_NORMALIZED_DISTANCE:
        v = ndist_with(has, c);
        *keep = !has || v <= c;
        break;
    default:  // _normalized_similarity, :363-384
        v = 1.0 - ndist_with(has, has ? norm_sim_to_norm_dist(c) : 0.0);
        *keep = !has || v >= c;
        break;
    }
    return v;
}

constexpr uint32_t T_AND_ANDN = TA & TB & ~TC;        // a & b & ~c
constexpr uint32_t T_OR_ANDN_B = TA | (TB & ~TC);     // a | (b & ~c)
constexpr uint32_t T_ANDN_AND = TA & ~TB & TC;        // a & ~b & c
constexpr uint32_t T_OR_AND = TA | (TB & TC);         // a | (b & c)
constexpr uint32_t T_AND_ORN = TA & (TB | ~TC);       // a & (b | ~c)

// Per-lane state of the single-word Jaro passes.  Pass 1 = flag_similar_characters_word (jaro.rs:147-190); pass 2 =
// count_transpositions_word (:339-368) restated without data-dependent control flow:
//   * blsi(x) = x & ~(x - 1): one 64-bit decrement + one v_bitop3 per half instead of a carry-chained negate;
//   * every flagged text character consumes the lowest remaining pattern flag (P &= P - 1); it is a MATCH when the
//     PM word of the text character has that bit -- matched bits are OR-ed into `hits`, so
//     transpositions = common - popcount(hits) with no per-column compare or count.
struct JaroWordState {
    uint64_t p_flag, hits;
    uint32_t t_lo, t_hi;  // T bits of columns 0..31 / 32..63; chunk k's sixteen sit at bits (k & 1) * 16 .., column j on bit 15 - j
                          // (two words steered by a scalar mask: a 64-bit conditional update compiles to a PAIR of VOP2
                          // v_cndmask_b32, and two of those back to back cost ~10 ns each on this chip, profiles/issue_rates_r02.txt)
};

// (PM rows are fetched kJaroGroup symbols ahead of their use; 2 rather than the scans' 4 saves 8 VGPRs, and with 7-8
// wavefronts per SIMD the LDS latency is covered either way)
constexpr int kJaroGroup = 2;
// Row of the (single-block) pattern table for column n of a chunk.  The byte offset sym * 8 is ONE v_lshlrev_b32_sdwa (the
// byte select and the shift in one half-rate instruction); written as `(dw >> 8k) & 0xFF` and indexed, hipcc emits
// v_bfe_u32 + v_lshl_add_u32 here -- two half-rate instructions (tools/microbench_issue: 1.85 ns each per wavefront and SIMD).
__device__ __forceinline__ uint64_t jaro_pm_row(const uint64_t* lds_pm0, const uint4& c, int n)
{
    const uint32_t dw = n < 4 ? c.x : (n < 8 ? c.y : (n < 12 ? c.z : c.w));
    uint32_t off;
    switch (n % 4) {
    case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(3u), "v"(dw)); break;
    case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(3u), "v"(dw)); break;
    case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(3u), "v"(dw)); break;
    default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(3u), "v"(dw)); break;
    }
    return *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(lds_pm0) + off);
}

// c ? a : b for doubles through an SGPR-pair lane mask: as `c ? a : b` hipcc emits two VOP2 v_cndmask_b32 ..., vcc back to back,
// which cost ~10 ns each on this chip (profiles/issue_rates_r02.txt); the e64 form with the mask in SGPRs is 1.9 ns
__device__ __forceinline__ double select_f64(bool c, double a, double b)
{
    const uint64_t m = __ballot(c);
    const uint64_t ua = (uint64_t)__double_as_longlong(a), ub = (uint64_t)__double_as_longlong(b);
    uint32_t lo, hi;
    asm("v_cndmask_b32 %0, %2, %3, %6\n\tv_cndmask_b32 %1, %4, %5, %6"
        : "=&v"(lo), "=&v"(hi)
        : "v"((uint32_t)ub), "v"((uint32_t)ua), "v"((uint32_t)(ub >> 32)), "v"((uint32_t)(ua >> 32)), "s"(m));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// x - 1 on an aligned register pair (hipcc otherwise feeds v_lshl_add_u64 a pair with a stale high half and repairs the
// result with an extra v_add_u32 per column)
__device__ __forceinline__ uint64_t dec64(uint64_t x)
{
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, -1" : "=v"(r) : "v"(x));
    return r;
}

// x / 3.0, correctly rounded, without the division (the table epilogue's one remaining quotient, jaro.rs:119): z = RN(1/3) =
// (1/3)(1 - 2^-54), q = RN(x z) is within 2 ulp of x / 3, r = x - 3 q is exact in one fma, and q + r z = x/3 - (r/3) 2^-54 rounds to
// RN(x / 3) because x / 3 is never closer than ulp / 6 to a rounding boundary (3 x an odd 54-bit integer is not a double).  Three
// f64 instructions instead of the ~23 of the IEEE division sequence; tests/test_div3.py checks every value the epilogue can produce
// against the host's divide.
__device__ __forceinline__ double div3(double x)
{
    const double z = 0x1.5555555555555p-2;
    const double q = x * z;
    const double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, z, q);
}

// Compile-time knobs of the compiled single-word kernels (A/B builds: tools/build_jaro_variant.sh)
#ifndef RF_JARO_PREFETCH
#define RF_JARO_PREFETCH 1  // the next tile's descriptor and first chunk row are fetched at the top of the current tile
#endif
#ifndef RF_JARO_TAB2G
#define RF_JARO_TAB2G 1  // common / len2 from the device table (rf_api.hip jaro_device_table) instead of a per-wavefront LDS table
#endif
#ifndef RF_JARO_DIV3
#define RF_JARO_DIV3 1
#endif
constexpr uint32_t kJaroTab2Rows = 130;  // common / len2 for len2 < 130 (beyond that no candidate takes the single-word path)

// Pass 1 over one 16-column chunk.  The sliding window mask of column j (jaro.rs:168,176,185) depends on j and the
// tile's bound only, so it comes from a 64-entry table this wavefront keeps in LDS (jaro_window_table, rebuilt when the
// bound changes): one broadcast ds_read per column.  Round 1 advanced the mask with 6 scalar instructions per column and
// built the T bit with 3 more; the CU's one scalar unit serves four SIMDs, and at 9 SALU per column x 4 wavefronts it --
// not the vector unit (10 VALU per column) -- was what bounded this pass (ISA tally of the r01 kernel: 144 SALU next to
// 160 VALU per chunk).  T bits are gathered at compile-time positions 0..15 and shifted into place once per chunk.
// kCols = 4, 8, 12 or 16 columns (round 4): a tile's last, partial chunk runs the unrolled, pipelined block rounded up to four columns
// instead of one branch per column -- the columns behind the candidate's end are no-ops because their window rows are zero
// (jaro_window_table), so no T bit and no pattern flag is set there.  On a corpus of lengths 1..64 a quarter of all columns sit in
// partial chunks.
template <int kCols>
__device__ __forceinline__ void jaro_flag_chunk(JaroWordState& st, const uint64_t* lds_pm0, const uint64_t* wtab, const uint4 c, uint32_t j0)
{
    static_assert(kCols % kJaroGroup == 0 && kCols >= kJaroGroup && kCols <= kChunk, "whole groups");
    constexpr int G = kJaroGroup, NG = kCols / G;
    const uint64_t* wrow = wtab + j0;
    uint32_t t16 = 0;
    uint64_t cur[G], nxt[G];
#pragma unroll
    for (int b = 0; b < G; ++b) cur[b] = jaro_pm_row(lds_pm0, c, b);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int b = 0; b < G; ++b) nxt[b] = jaro_pm_row(lds_pm0, c, (g + 1) * G + b);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < G; ++b) {
            const uint64_t pm_j = lut3<T_AND_ANDN>(cur[b], wrow[g * G + b], st.p_flag);  // PM & window & ~P
            const uint64_t below = dec64(pm_j);
            st.p_flag = lut3<T_OR_ANDN_B>(st.p_flag, pm_j, below);  // P |= blsi(pm_j)
            // T bit of this column = (pm_j != 0) = the sign of pm_j | -pm_j, and -pm_j == ~below: one LUT on the high
            // halves and one funnel shift that pushes the sign into t16 -- a 64-bit compare + select + or costs 4.7 ns
            // per wavefront and SIMD on this chip, this 3.0 (tools/microbench_issue).  Column j lands on bit 15 - j.
            const uint32_t y = (uint32_t)(pm_j >> 32) | ~(uint32_t)(below >> 32);
            t16 = __builtin_amdgcn_alignbit(t16, y, 31);           // (t16 << 1) | (y >> 31)      jaro.rs:174 / :183
        }
#pragma unroll
        for (int b = 0; b < G; ++b) cur[b] = nxt[b];
    }
    t16 <<= kChunk - kCols;  // (the columns not run)
    const uint32_t v = t16 << (j0 & 16), in_lo = (j0 & 32) ? 0u : ~0u;  // j0 is wavefront-uniform: the mask is scalar
    st.t_lo |= v & in_lo;
    st.t_hi |= v & ~in_lo;
}

// the window mask of every column for this bound (closed form of the recurrence bm' = (bm << 1) | (j < bound) started
// at the low bound + 1 bits): bits [max(j - bound, 0), min(j + bound + 1, 64))
// Rows of columns at or behind the (truncated) candidate length are ZERO: a partial last chunk runs whole groups of four columns, and a
// zero window row makes a column a no-op (no pattern flag, no T bit).
__device__ __forceinline__ void jaro_window_table(uint64_t* wtab, uint32_t lane, uint32_t bound, uint32_t len2)
{
    const uint32_t lo = lane > bound ? lane - bound : 0, hi = min(lane + bound + 1, 64u);
    const uint64_t row = (hi >= 64 ? ~0ull : ((1ull << hi) - 1)) & (~0ull << lo);
    wtab[lane] = lane < len2 ? row : 0ull;
}

template <int kCols>  // (columns behind the candidate's end have no T bit: no-ops here too)
__device__ __forceinline__ void jaro_transpose_chunk(JaroWordState& st, const uint64_t* lds_pm0, const uint4 c, uint32_t j0)
{
    constexpr int G = kJaroGroup, NG = kCols / G;
    const uint32_t in_lo = (j0 & 32) ? 0u : ~0u;
    const uint32_t thalf = (st.t_lo & in_lo) | (st.t_hi & ~in_lo);
    const uint32_t t16 = thalf >> (j0 & 16);  // this chunk's T bits, column j at position 15 - j (no scalar address arithmetic per column)
    uint64_t cur[G], nxt[G];
#pragma unroll
    for (int b = 0; b < G; ++b) cur[b] = jaro_pm_row(lds_pm0, c, b);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int b = 0; b < G; ++b) nxt[b] = jaro_pm_row(lds_pm0, c, (g + 1) * G + b);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < G; ++b) {
            const uint32_t f32 = (uint32_t)__builtin_amdgcn_sbfe((int)t16, 15 - (g * G + b), 1);  // all ones iff T bit j
            const uint64_t f = ((uint64_t)f32 << 32) | f32;
            const uint64_t below = dec64(st.p_flag);
            const uint64_t m = lut3<T_ANDN_AND>(st.p_flag, below, f);  // lowest remaining pattern flag, if flagged
            st.hits = lut3<T_OR_AND>(st.hits, cur[b], m);              // match iff PM[text char] has that bit
            st.p_flag = lut3<T_AND_ORN>(st.p_flag, below, f);          // consume it, if flagged
        }
#pragma unroll
        for (int b = 0; b < G; ++b) cur[b] = nxt[b];
    }
}

// The f64 epilogue without a cutoff (kFast), bit for bit the reference's arithmetic but without its divisions: every
// quotient of jaro.rs:106-119 has a small integer numerator and denominator, so it is LOOKED UP instead of computed --
//   common / len1            tab1[common]      LDS, filled once per workgroup (len1 is the query's length)
//   common / len2            tab2[common]      LDS, per wavefront, refilled when the tile length changes (one division per lane)
//   (common - t/2) / common  p.jaro_tab[common * 33 + t/2]   global, 65 x 33 doubles built once per device on the host
// with the same IEEE divide, so the looked-up values ARE the reference's; only `sim / 3.0` is still divided per candidate.
// The general epilogue (f64_metric_value: filters, cutoff back-translation) costs ~8 f64 divisions per candidate, a
// quarter of this kernel's VALU time (rocprofv3: 24.8 VALU per column, ~21 of them the two passes).
constexpr int kJaroTabStride = 66;  // doubles per table (65 used)
// One STATIC LDS object per kernel: its address is a compile-time constant, so the pattern-table reads take their base in
// the instruction's offset field (as dynamic `extern __shared__` memory the base was a relocation the compiler added to every
// gather address with a v_add_u32 -- one VALU instruction per column and pass).
struct JaroWordLds {
    uint64_t pm0[256];                                   // block 0 of every row
    double tabs[kJaroTabStride * (1 + kWavesPerBlock)];  // tab1 + one tab2 per wavefront (kFast)
    uint64_t wtab[kWavesPerBlock * kWave];               // one window-mask table per wavefront
};
template <bool kUniform, bool kEarly, bool kFast>
__device__ __forceinline__ void jaro_word_body(const ScanParams& p, JaroWordLds& lds)
{
    const uint32_t W = p.words;  // PM row stride; only block 0 is read on this path (jaro.rs:172, pm.get(0, ..))
    uint64_t* lds_pm0 = lds.pm0;
    double* tab1 = lds.tabs;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds_pm0[p.sigma[i]] = p.pm[(size_t)i * W];  // renamed rows
    if (kFast && threadIdx.x < 65) tab1[threadIdx.x] = (double)threadIdx.x / (double)p.len1;
    __syncthreads();
    double* tab2 = tab1 + kJaroTabStride * (1 + uniform(threadIdx.x / kWave));
    uint32_t tab2_len = 0xFFFFFFFFu;
    // this wavefront's window-mask table (64 x u64), behind the f64 tables
    uint64_t* wtab = lds.wtab + kWave * uniform(threadIdx.x / kWave);
    uint32_t wtab_bound = 0xFFFFFFFFu;

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t q4 = p.query_head;  // first four query bytes, little endian (Winkler prefix)

    // The next tile's descriptor and first chunk row are on their way while this tile runs (RF_JARO_PREFETCH): a tile used to begin
    // with a full HBM latency in front of its first column.
    uint32_t t = p.tile_begin + dealt_workgroup(p) * kWavesPerBlock + wave;
    TileView tv_ahead;
    uint4 head_ahead;
    if (RF_JARO_PREFETCH && t < p.tile_end) {
        tv_ahead = load_tile<kUniform>(p, t);
        head_ahead = tv_ahead.src[lane];
    }
    for (; t < p.tile_end; t += stride) {
        TileView tv;
        uint4 cur;
        if (RF_JARO_PREFETCH) {
            tv = tv_ahead;
            cur = head_ahead;
            if (t + stride < p.tile_end) {  // (wavefront-uniform)
                tv_ahead = load_tile<kUniform>(p, t + stride);
                head_ahead = tv_ahead.src[lane];
            }
        } else {
            tv = load_tile<kUniform>(p, t);
            cur = tv.src[lane];
        }
        const uint32_t len2_orig = tv.len, len1_orig = p.len1;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];

        // window truncation, jaro.rs:550-565 (wavefront-uniform)
        uint32_t len1 = len1_orig, len2 = len2_orig, bound = 0;
        if (len2 > len1) {
            bound = len2 / 2 - 1;
            if (len2 > len1 + bound) len2 = len1 + bound;
        } else if (len1 >= 2) {
            bound = len1 / 2 - 1;
            if (len1 > len2 + bound) len1 = len2 + bound;
        }
        // (len1 <= 1 with len2 <= len1 never reaches the flags: the length filter / 1x1 rule decide)
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;  // <= 4 on this path

        // the candidate (<= 64 bytes = 4 chunk rows) is streamed twice: HBM once, the second pass hits L1/L2
        JaroRaw r;
        r.eq11 = (cur.x & 0xFFu) == (q4 & 0xFFu);
        {  // Winkler prefix: equal leading bytes among the first min(4, len1_orig, len2_orig), jaro_winkler.rs:118-123
            const uint32_t lim = min(4u, min(len1_orig, len2_orig));
            const uint32_t diff = cur.x ^ q4;
            const uint32_t first_diff = diff ? (uint32_t)(__ffs(diff) - 1) / 8 : 4u;
            r.prefix = min(first_diff, lim);
        }

        JaroWordState st;
        st.p_flag = st.hits = 0;
        st.t_lo = st.t_hi = 0;
        if ((bound | (len2 << 8)) != wtab_bound) {  // wavefront-uniform; this wavefront's own table: no barrier
            jaro_window_table(wtab, lane, bound, len2);
            wtab_bound = bound | (len2 << 8);
        }
        // Early-out under a tight cutoff: after j text symbols the number of common characters can still grow by at most
        // one per remaining symbol, the similarity is at most (m/len1 + m/len2 + 1) / 3 (common_char_filter, jaro.rs:134-145)
        // and the Winkler boost at most prefix * weight * (1 - sim) with the prefix already known.  If no lane can reach
        // `jaro_need` any more (1e-9 of slack covers the reciprocal arithmetic of this bound), the rest of pass 1 and all
        // of pass 2 are skipped and the tile is None -- exactly what the replayed filters would say.
        const bool early = kEarly && nch > 0;  // (a separate instantiation: the plain kernel keeps its registers and loop shape)
        double inv1 = 0.0, inv2 = 0.0;
        if (early) {
            inv1 = 1.0 / (double)len1_orig;
            inv2 = 1.0 / (double)len2_orig;
        }
        bool dead = false;
        for (uint32_t k = 0; k < nch; ++k) {  // pass 1
            const uint4 nxt = tv.src[(size_t)(k + 1 < nch ? k + 1 : 0) * kWave + lane];  // next chunk, then chunk 0 again
            const uint32_t cols = len2 - k * kChunk;
            if (cols > 12u)
                jaro_flag_chunk<16>(st, lds_pm0, wtab, cur, k * kChunk);
            else if (cols > 8u)
                jaro_flag_chunk<12>(st, lds_pm0, wtab, cur, k * kChunk);
            else if (cols > 4u)
                jaro_flag_chunk<8>(st, lds_pm0, wtab, cur, k * kChunk);
            else
                jaro_flag_chunk<4>(st, lds_pm0, wtab, cur, k * kChunk);
            if (early) {
                const uint32_t j = min(len2, (k + 1) * kChunk);
                const uint32_t m_ub = min((uint32_t)__popcll(st.p_flag) + (len2 - j), min(len1, len2));
                const double m = (double)m_ub;
                const double sim_ub = (m * inv1 + m * inv2 + 1.0) / 3.0;
                const double boosted = p.finish == FIN_JW ? sim_ub + (double)r.prefix * p.prefix_weight * (1.0 - sim_ub) : sim_ub;
                if (__ballot(m_ub != 0 && boosted + 1e-9 >= p.jaro_need) == 0) {
                    dead = true;
                    break;
                }
            }
            cur = nxt;
        }
        if (dead) {
            const bool valid_d = kUniform ? slot < p.n : idx != kPad;
            if (valid_d) reinterpret_cast<double*>(p.out)[idx] = __longlong_as_double(0x7FF8000000000000ll);
            continue;
        }
        r.common = __popcll(st.p_flag);
        double quot2 = 0.0;  // common / len2 from the device table, on its way during pass 2
        const bool tab2_global = kFast && RF_JARO_TAB2G && len2_orig < kJaroTab2Rows;
        if (tab2_global) quot2 = p.jaro_tab[65u * 33u + len2_orig * 65u + r.common];
        for (uint32_t k = 0; k < nch; ++k) {  // pass 2
            uint4 nxt = cur;
            if (k + 1 < nch) nxt = tv.src[(size_t)(k + 1) * kWave + lane];
            const uint32_t cols = len2 - k * kChunk;
            if (cols > 12u)
                jaro_transpose_chunk<16>(st, lds_pm0, cur, k * kChunk);
            else if (cols > 8u)
                jaro_transpose_chunk<12>(st, lds_pm0, cur, k * kChunk);
            else if (cols > 4u)
                jaro_transpose_chunk<8>(st, lds_pm0, cur, k * kChunk);
            else
                jaro_transpose_chunk<4>(st, lds_pm0, cur, k * kChunk);
            cur = nxt;
        }
        r.transpositions = r.common - __popcll(st.hits);

        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (kFast) {
            // jaro::similarity_with_pm with score_cutoff = 0.0 (jaro.rs:533-598): the filters reduce to the empty-string
            // and common == 0 cases; calculate_similarity (:106-119) from the tables, in the reference's order
            if (!tab2_global && len2_orig != tab2_len) {  // wavefront-uniform; this wavefront's own table: no barrier
                tab2[lane] = (double)lane / (double)len2_orig;
                if (lane == 0) tab2[64] = 64.0 / (double)len2_orig;
                tab2_len = len2_orig;
            }
            double sim;
            if (len1_orig == 0 || len2_orig == 0) {
                sim = (len1_orig == 0 && len2_orig == 0) ? 1.0 : 0.0;       // :537-544
            } else if (len1_orig == 1 && len2_orig == 1) {
                sim = r.eq11 ? 1.0 : 0.0;                                   // :546-548
            } else {
                double acc = 0.0;
                acc += tab1[r.common];
                acc += tab2_global ? quot2 : tab2[r.common];
                acc += p.jaro_tab[r.common * 33u + r.transpositions / 2u];
                acc = RF_JARO_DIV3 ? div3(acc) : acc / 3.0;
                sim = select_f64(r.common == 0, 0.0, acc);                  // :579-581
            }
            if (p.finish == FIN_JW) sim = select_f64(sim > 0.7, sim + (double)r.prefix * p.prefix_weight * (1.0 - sim), sim);  // jaro_winkler.rs:136-138
            double v = sim;                                                 // Metricf64 defaults, details/distance.rs:277-385
            if (p.op == RF_OP_DISTANCE || p.op == RF_OP_NORMALIZED_DISTANCE) v = 1.0 - sim;
            if (p.op == RF_OP_NORMALIZED_SIMILARITY) v = 1.0 - (1.0 - sim);
            if (valid) reinterpret_cast<double*>(p.out)[idx] = v;
        } else if (valid) {
            bool keep;
            const double v = f64_metric_value(p, len2_orig, r, &keep);
            reinterpret_cast<double*>(p.out)[idx] = keep ? v : __longlong_as_double(0x7FF8000000000000ll);
        }
    }
}

// Two entry points over the same body (the table epilogue is a different instantiation).  Both keep the compiler's own
// register budget: pinned to 8 wavefronts per SIMD the fast kernel spills 20 bytes and measures 1.5 % slower than at 7.
template <bool kUniform, bool kEarly>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void jaro_word_kernel(const ScanParams p)
{
    __shared__ JaroWordLds lds;
    jaro_word_body<kUniform, kEarly, false>(p, lds);
}
template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void jaro_word_fast_kernel(const ScanParams p)
{
    __shared__ JaroWordLds lds;
    jaro_word_body<kUniform, false, true>(p, lds);
}

// ---------------------------------------------------------------------------------------------------
// Jaro / Jaro-Winkler, multi-word path (jaro.rs:192-337 flag_similar_characters_block / _step, :370-420
// count_transpositions_block) for strings of up to 512 symbols after the window truncation.  P_flag / T_flag are
// 8 + 8 VGPR pairs per lane; the sliding search window (SearchBoundMask, jaro.rs:99-104) is wavefront-uniform;
// the candidate is streamed twice (flags, then transpositions).
// ---------------------------------------------------------------------------------------------------
constexpr int kJaroWords = 8;

template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void jaro_block_kernel(const ScanParams p)
{
    const uint32_t W = p.words;  // PM row stride (<= 8 here)
    extern __shared__ uint64_t lds_pmw[];  // 256 x W (+ one pad row: an exhausted window may index word W)
    for (uint32_t i = threadIdx.x; i < 256 * W + W + 1; i += kWave * kWavesPerBlock) {
        if (i < 256 * W)
            lds_pmw[(uint32_t)p.sigma[i / W] * W + i % W] = p.pm[i];  // renamed rows
        else
            lds_pmw[i] = 0;
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t q4 = p.query_head;

    for (uint32_t t = p.tile_begin + dealt_workgroup(p) * kWavesPerBlock + wave; t < p.tile_end; t += stride) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2_orig = tv.len, len1_orig = p.len1;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];

        uint32_t len1 = len1_orig, len2 = len2_orig, bound = 0;  // jaro.rs:550-565
        if (len2 > len1) {
            bound = len2 / 2 - 1;
            if (len2 > len1 + bound) len2 = len1 + bound;
        } else if (len1 >= 2) {
            bound = len1 / 2 - 1;
            if (len1 > len2 + bound) len1 = len2 + bound;
        }

        const uint4 head = len2_orig ? tv.src[lane] : make_uint4(0, 0, 0, 0);
        JaroRaw r;
        r.eq11 = (head.x & 0xFFu) == (q4 & 0xFFu);
        {
            const uint32_t lim = min(4u, min(len1_orig, len2_orig));
            const uint32_t diff = head.x ^ q4;
            const uint32_t first_diff = diff ? (uint32_t)(__ffs(diff) - 1) / 8 : 4u;
            r.prefix = min(first_diff, lim);
        }

        uint64_t P[kJaroWords], T[kJaroWords];
#pragma unroll
        for (int w = 0; w < kJaroWords; ++w) P[w] = T[w] = 0;

        // ---- pass 1: flag_similar_characters_block (jaro.rs:286-337); window state is wavefront-uniform
        const uint32_t start_range = min(bound + 1, len1);
        uint32_t win_words = 1 + start_range / 64, empty_words = 0;
        uint64_t last_mask = (1ull << (start_range % 64)) - 1, first_mask = ~0ull;
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        uint64_t tcur = 0;
        // the same early-out as jaro_word_kernel: after every chunk, can any lane still reach `jaro_need`?
        const bool early = p.jaro_need >= 0.0 && nch > 0;
        const double inv1 = early ? 1.0 / (double)len1_orig : 0.0, inv2 = early ? 1.0 / (double)len2_orig : 0.0;
        bool dead = false;
        for (uint32_t c = 0; c < nch && !dead; ++c) {
            uint4 data = tv.src[(size_t)c * kWave + lane];
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            for (uint32_t b = 0; b < cols; ++b) {
                const uint32_t j = c * kChunk + b;
                const uint32_t ch = data.x & 0xFFu;
                const uint32_t last_word = empty_words + win_words - 1;
                bool found = false;
#pragma unroll
                for (int w = 0; w < kJaroWords; ++w) {  // flag_similar_characters_step, jaro.rs:192-284
                    if ((uint32_t)w >= empty_words && (uint32_t)w <= last_word) {
                        uint64_t mask = ~0ull;
                        if ((uint32_t)w == empty_words) mask &= first_mask;
                        if ((uint32_t)w == last_word) mask &= last_mask;
                        const uint64_t pm_j = lds_pmw[ch * W + w] & mask & ~P[w];
                        const bool hit = !found && pm_j != 0;
                        P[w] |= hit ? blsi64(pm_j) : 0ull;
                        found = found || hit;
                    }
                }
                tcur |= (uint64_t)found << (j & 63);
                if ((j & 63) == 63 || j + 1 == len2) {
#pragma unroll
                    for (int k = 0; k < kJaroWords; ++k)
                        if ((uint32_t)k == (j >> 6)) T[k] = tcur;
                    tcur = 0;
                }
                if (j + bound + 1 < len1) {  // jaro.rs:318-324
                    last_mask = (last_mask << 1) | 1;
                    if (j + bound + 2 < len1 && last_mask == ~0ull) {
                        last_mask = 0;
                        win_words += 1;
                    }
                }
                if (j >= bound) {  // jaro.rs:326-333
                    first_mask <<= 1;
                    if (first_mask == 0) {
                        first_mask = ~0ull;
                        win_words -= 1;
                        empty_words += 1;
                    }
                }
                data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                data.w >>= 8;
            }
            if (early) {
                uint32_t so_far = 0;
#pragma unroll
                for (int w = 0; w < kJaroWords; ++w) so_far += __popcll(P[w]);
                const uint32_t j = min(len2, (c + 1) * kChunk);
                const uint32_t m_ub = min(so_far + (len2 - j), min(len1, len2));
                const double m = (double)m_ub;
                const double sim_ub = (m * inv1 + m * inv2 + 1.0) / 3.0;
                const double boosted = p.finish == FIN_JW ? sim_ub + (double)r.prefix * p.prefix_weight * (1.0 - sim_ub) : sim_ub;
                dead = __ballot(m_ub != 0 && boosted + 1e-9 >= p.jaro_need) == 0;
            }
        }
        if (dead) {
            const bool valid_d = kUniform ? slot < p.n : idx != kPad;
            if (valid_d) reinterpret_cast<double*>(p.out)[idx] = __longlong_as_double(0x7FF8000000000000ll);
            continue;
        }
        uint32_t common = 0;
#pragma unroll
        for (int w = 0; w < kJaroWords; ++w) common += __popcll(P[w]);
        r.common = common;

        // ---- pass 2: count_transpositions_block (jaro.rs:370-420): every flagged text character, in text order,
        //      consumes the lowest remaining pattern flag
        uint32_t transpositions = 0;
        for (uint32_t c = 0; c < nch; ++c) {
            uint4 data = tv.src[(size_t)c * kWave + lane];
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            uint64_t tw = 0;
#pragma unroll
            for (int k = 0; k < kJaroWords; ++k)
                if ((uint32_t)k == ((c * kChunk) >> 6)) tw = T[k];
            for (uint32_t b = 0; b < cols; ++b) {
                const uint32_t j = c * kChunk + b;
                const uint32_t ch = data.x & 0xFFu;
                const bool flagged = (tw >> (j & 63)) & 1;
                int sel = -1;
                uint64_t pw = 0;
#pragma unroll
                for (int w = kJaroWords - 1; w >= 0; --w)
                    if (P[w] != 0) {
                        sel = w;
                        pw = P[w];
                    }
                const uint64_t m = blsi64(pw);
                const uint64_t pmv = lds_pmw[ch * W + (sel < 0 ? 0 : sel)];
                transpositions += (flagged && (pmv & m) == 0) ? 1u : 0u;
#pragma unroll
                for (int w = 0; w < kJaroWords; ++w)
                    if (flagged && w == sel) P[w] ^= m;
                data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                data.w >>= 8;
            }
        }
        r.transpositions = transpositions;

        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
            bool keep;
            const double v = f64_metric_value(p, len2_orig, r, &keep);
            reinterpret_cast<double*>(p.out)[idx] = keep ? v : __longlong_as_double(0x7FF8000000000000ll);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Jaro / Jaro-Winkler beyond 512 symbols (after the window truncation): the same two passes (jaro.rs:192-337, :370-420) with
// the flag words in a global scratch strip per wavefront instead of registers -- P_flag [word][lane], then T_flag [word][lane],
// coalesced across the lanes -- and PM words fetched from the L2-resident global table.  Only the words inside the sliding
// search window are touched per text symbol (the window bounds are wavefront-uniform).  A completeness path: no length
// limit, no attempt at the roofline.
// ---------------------------------------------------------------------------------------------------
template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void jaro_long_kernel(const ScanParams p)
{
    const uint32_t W = p.words;  // PM row stride
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t q4 = p.query_head;
    __shared__ uint8_t lds_unrename[256];  // stored symbol -> original symbol (the PM table is indexed by original symbols)
    lds_unrename[p.sigma[threadIdx.x & 255]] = (uint8_t)(threadIdx.x & 255);
    __syncthreads();
    const uint32_t wp_max = (p.len1 + 63) / 64 + 1, wt_max = p.long_chunks_max;  // words of P / T per lane (T: ceil(max len2 / 64))
    uint64_t* P = reinterpret_cast<uint64_t*>(p.long_scratch) + (size_t)gw * (wp_max + wt_max) * kWave + lane;  // P[w * kWave]
    uint64_t* T = P + (size_t)wp_max * kWave;

    for (uint32_t t = p.tile_begin + gw; t < p.tile_end; t += stride) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2_orig = tv.len, len1_orig = p.len1;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        uint32_t len1 = len1_orig, len2 = len2_orig, bound = 0;  // jaro.rs:550-565
        if (len2 > len1) {
            bound = len2 / 2 - 1;
            if (len2 > len1 + bound) len2 = len1 + bound;
        } else if (len1 >= 2) {
            bound = len1 / 2 - 1;
            if (len1 > len2 + bound) len1 = len2 + bound;
        }
        const uint4 head = len2_orig ? tv.src[lane] : make_uint4(0, 0, 0, 0);
        JaroRaw r;
        r.eq11 = (head.x & 0xFFu) == (q4 & 0xFFu);
        {
            const uint32_t lim = min(4u, min(len1_orig, len2_orig));
            const uint32_t diff = head.x ^ q4;
            const uint32_t first_diff = diff ? (uint32_t)(__ffs(diff) - 1) / 8 : 4u;
            r.prefix = min(first_diff, lim);
        }
        const uint32_t wp = (len1 + 63) / 64, wt = (len2 + 63) / 64;
        for (uint32_t w = 0; w < wp; ++w) P[(size_t)w * kWave] = 0;
        for (uint32_t w = 0; w < wt; ++w) T[(size_t)w * kWave] = 0;

        // ---- pass 1 (jaro.rs:286-337); the window state is wavefront-uniform
        const uint32_t start_range = min(bound + 1, len1);
        uint32_t win_words = 1 + start_range / 64, empty_words = 0;
        uint64_t last_mask = (1ull << (start_range % 64)) - 1, first_mask = ~0ull;
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        uint64_t tcur = 0;
        for (uint32_t c = 0; c < nch; ++c) {
            uint4 data = tv.src[(size_t)c * kWave + lane];
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            for (uint32_t b = 0; b < cols; ++b) {
                const uint32_t j = c * kChunk + b;
                const uint64_t* row = p.pm + (size_t)lds_unrename[data.x & 0xFFu] * W;
                const uint32_t last_word = empty_words + win_words - 1;
                bool found = false;
                for (uint32_t w = empty_words; w <= last_word && w < wp; ++w) {  // flag_similar_characters_step, jaro.rs:192-284
                    uint64_t mask = ~0ull;
                    if (w == empty_words) mask &= first_mask;
                    if (w == last_word) mask &= last_mask;
                    const uint64_t pw = P[(size_t)w * kWave];
                    const uint64_t pm_j = row[w] & mask & ~pw;
                    const bool hit = !found && pm_j != 0;
                    if (hit) P[(size_t)w * kWave] = pw | blsi64(pm_j);
                    found = found || hit;
                }
                tcur |= (uint64_t)found << (j & 63);
                if ((j & 63) == 63 || j + 1 == len2) {
                    T[(size_t)(j >> 6) * kWave] = tcur;
                    tcur = 0;
                }
                if (j + bound + 1 < len1) {  // jaro.rs:318-324
                    last_mask = (last_mask << 1) | 1;
                    if (j + bound + 2 < len1 && last_mask == ~0ull) {
                        last_mask = 0;
                        win_words += 1;
                    }
                }
                if (j >= bound) {  // jaro.rs:326-333
                    first_mask <<= 1;
                    if (first_mask == 0) {
                        first_mask = ~0ull;
                        win_words -= 1;
                        empty_words += 1;
                    }
                }
                data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                data.w >>= 8;
            }
        }
        uint32_t common = 0;
        for (uint32_t w = 0; w < wp; ++w) common += __popcll(P[(size_t)w * kWave]);
        r.common = common;

        // ---- pass 2 (jaro.rs:370-420): every flagged text character, in text order, consumes the lowest remaining pattern flag
        uint32_t transpositions = 0, wsel = 0;  // wsel: this lane's first word that may still hold a flag (only ever grows)
        uint64_t pcur = wp ? P[0] : 0;
        for (uint32_t c = 0; c < nch; ++c) {
            uint4 data = tv.src[(size_t)c * kWave + lane];
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            const uint64_t tw = T[(size_t)((c * kChunk) >> 6) * kWave];
            for (uint32_t b = 0; b < cols; ++b) {
                const uint32_t j = c * kChunk + b;
                if ((tw >> (j & 63)) & 1) {  // (per lane)
                    while (pcur == 0 && wsel + 1 < wp) pcur = P[(size_t)(++wsel) * kWave];
                    const uint64_t m = blsi64(pcur);
                    const uint64_t pmv = p.pm[(size_t)lds_unrename[data.x & 0xFFu] * W + wsel];
                    transpositions += (pmv & m) == 0;
                    pcur ^= m;
                }
                data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                data.w >>= 8;
            }
        }
        r.transpositions = transpositions;
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
            bool keep;
            const double v = f64_metric_value(p, len2_orig, r, &keep);
            reinterpret_cast<double*>(p.out)[idx] = keep ? v : __longlong_as_double(0x7FF8000000000000ll);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The single-word Jaro kernel for the headline shape -- single-length corpus, no cutoff, candidate length (after the window
// truncation) a multiple of 16 -- with both passes over a 16-column chunk as hand-scheduled asm blocks
// (tools/gen_jaro_chunk_asm.py -> rf_jaro_chunk_asm.inc; same reasons and technique as rf_lev_asm.hip: dependent chains of
// full- and half-rate instructions whose issue stalls eight wavefronts do not hide, s_nop placement the compiler cannot be told).
// P, hits and the T flags live in physical VGPRs (v58..v63) as register-asm variables that only asm statements touch; the
// pattern table is the first member of the kernel's only LDS object, so table rows are at symbol * 8 and the window rows at
// offsetof(wtab).  Everything else -- tile loop, Winkler prefix, table epilogue -- is jaro_word_body<true, false, true> again.
// ---------------------------------------------------------------------------------------------------
#include "rf_jaro_chunk_asm.inc"
// kPriv (round 4): corpora whose stored symbols are all < 64 (exact: ScanParams::max_stored_sym) gather the table rows from a second,
// CONFLICT-FREE copy -- row of symbol s for lane l at s * 256 + (l & 31) * 8, i.e. every lane of a half-wavefront on a bank pair of
// its own (16 KiB per workgroup).  This kernel keeps the LDS ~80 % busy (2 table gathers + 1 window row per column against the
// Levenshtein scan's 1 gather), and 62 symbols on 32 bank pairs make every gather a 2-way conflict: the copy halves their cost.
struct JaroWordLdsPriv {
    JaroWordLds w;                     // at LDS address 0 (the asm blocks' window / table addresses have no base)
    uint64_t priv[64 * 32];            // [symbol][lane & 31]
};
static_assert(sizeof(JaroWordLds) == RF_JARO_PRIV_OFF, "tools/gen_jaro_chunk_asm.py RF_GEN_JPRIV_OFF must equal sizeof(JaroWordLds)");
template <bool kPriv>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void jaro_word_asm_kernel(const ScanParams p)
{
    __shared__ typename std::conditional<kPriv, JaroWordLdsPriv, JaroWordLds>::type lds_obj;
    JaroWordLds& lds = *reinterpret_cast<JaroWordLds*>(&lds_obj);
    const uint32_t W = p.words;
    double* tab1 = lds.tabs;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds.pm0[p.sigma[i]] = p.pm[(size_t)i * W];  // renamed rows
    if (threadIdx.x < 65) tab1[threadIdx.x] = (double)threadIdx.x / (double)p.len1;
    __syncthreads();
    if constexpr (kPriv) {
        uint64_t* priv = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(&lds_obj) + sizeof(JaroWordLds));
        for (int i = threadIdx.x; i < 64 * 32; i += kWave * kWavesPerBlock) priv[i] = lds.pm0[i >> 5];
        __syncthreads();
    }
    const uint32_t lane_bank = (threadIdx.x & 31u) * 8u;                                      // [lb]: this lane's bank pair inside a table row
    const uint32_t sel0 = 0x0C0C0400u, sel1 = 0x0C0C0500u, sel2 = 0x0C0C0600u, sel3 = 0x0C0C0700u;  // v_perm_b32: {0, 0, chunk byte k, lane_bank byte 0}
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    double* tab2 = tab1 + kJaroTabStride * (1 + wave);
    uint64_t* wtab = lds.wtab + kWave * wave;
    const uint32_t wtab_addr = (uint32_t)offsetof(JaroWordLds, wtab) + kWave * wave * 8u;  // LDS byte address (the struct is at 0)
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t q4 = p.query_head;
    const uint32_t three = 3u;

    // single-length corpus: lengths, window and tables are the same for every tile
    const uint32_t len2_orig = p.uniform_len, len1_orig = p.len1;
    uint32_t len1 = len1_orig, len2 = len2_orig, bound = 0;  // window truncation, jaro.rs:550-565
    if (len2 > len1) {
        bound = len2 / 2 - 1;
        if (len2 > len1 + bound) len2 = len1 + bound;
    } else if (len1 >= 2) {
        bound = len1 / 2 - 1;
        if (len1 > len2 + bound) len1 = len2 + bound;
    }
    const uint32_t nch = len2 / kChunk;  // the launcher sends only whole-chunk lengths here
    jaro_window_table(wtab, lane, bound, len2);
    tab2[lane] = (double)lane / (double)len2_orig;
    if (lane == 0) tab2[64] = 64.0 / (double)len2_orig;

    register uint32_t pl asm("v60"), ph asm("v61"), hl asm("v62"), hh asm("v63"), tlo asm("v58"), thi asm("v59");
    // The candidate's chunk rows (<= 4) stay in registers for BOTH passes, and the next tile's rows are fetched into them one
    // by one as pass 2 finishes with each: every load has the rest of pass 2, the epilogue and the earlier chunks of pass 1
    // (>= 3 us) to arrive.  (Fetching chunk k + 1 while chunk k is processed, as the compiled kernel does, gives a load ~1.2 us
    // -- a pass-1 chunk is short -- which is less than the loaded HBM latency: every chunk started with a stall.)
    uint4 b0, b1, b2, b3;
    uint32_t t = p.tile_begin + blockIdx.x * kWavesPerBlock + wave;
    if (t >= p.tile_end) return;
    const uint4* src = reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes);
    b0 = src[lane];
    b1 = b2 = b3 = b0;
    if (nch > 1) b1 = src[kWave + lane];
    if (nch > 2) b2 = src[2 * kWave + lane];
    if (nch > 3) b3 = src[3 * kWave + lane];
#define RF_P1(buf, j0)                                                                                                          \
    do {                                                                                                                        \
        if constexpr (kPriv)                                                                                                    \
            asm volatile(RF_JARO_PASS1_PRIV_ASM                                                                                 \
                         : "+v"(pl), "+v"(ph), "+v"(tlo), "+v"(thi)                                                             \
                         : [c0] "v"(buf.x), [c1] "v"(buf.y), [c2] "v"(buf.z), [c3] "v"(buf.w), [lb] "v"(lane_bank), [sel0] "s"(sel0), [sel1] "s"(sel1), \
                           [sel2] "s"(sel2), [sel3] "s"(sel3), [wa] "v"(wtab_addr + (j0) * 8u), [sh] "s"((j0) & 16u), [lo] "s"(((j0) & 32u) ? 0u : ~0u) \
                         : RF_JARO_CHUNK_CLOBBERS);                                                                             \
        else                                                                                                                    \
            asm volatile(RF_JARO_PASS1_ASM                                                                                      \
                         : "+v"(pl), "+v"(ph), "+v"(tlo), "+v"(thi)                                                             \
                         : [c0] "v"(buf.x), [c1] "v"(buf.y), [c2] "v"(buf.z), [c3] "v"(buf.w), [k3] "v"(three), [wa] "v"(wtab_addr + (j0) * 8u), \
                           [sh] "s"((j0) & 16u), [lo] "s"(((j0) & 32u) ? 0u : ~0u)                                              \
                         : RF_JARO_CHUNK_CLOBBERS);                                                                             \
    } while (0)
#define RF_P2(buf, j0)                                                                                                          \
    do {                                                                                                                        \
        if constexpr (kPriv)                                                                                                    \
            asm volatile(RF_JARO_PASS2_PRIV_ASM                                                                                 \
                         : "+v"(pl), "+v"(ph), "+v"(hl), "+v"(hh)                                                               \
                         : [c0] "v"(buf.x), [c1] "v"(buf.y), [c2] "v"(buf.z), [c3] "v"(buf.w), [lb] "v"(lane_bank), [sel0] "s"(sel0), [sel1] "s"(sel1), \
                           [sel2] "s"(sel2), [sel3] "s"(sel3), "v"(tlo), "v"(thi), [sh] "s"((j0) & 16u), [lo] "s"(((j0) & 32u) ? 0u : ~0u) \
                         : RF_JARO_CHUNK_CLOBBERS);                                                                             \
        else                                                                                                                    \
            asm volatile(RF_JARO_PASS2_ASM                                                                                      \
                         : "+v"(pl), "+v"(ph), "+v"(hl), "+v"(hh)                                                               \
                         : [c0] "v"(buf.x), [c1] "v"(buf.y), [c2] "v"(buf.z), [c3] "v"(buf.w), [k3] "v"(three), "v"(tlo), "v"(thi), \
                           [sh] "s"((j0) & 16u), [lo] "s"(((j0) & 32u) ? 0u : ~0u)                                              \
                         : RF_JARO_CHUNK_CLOBBERS);                                                                             \
    } while (0)
    while (true) {
        const uint32_t idx = t * kWave + lane;
        JaroRaw r;
        r.eq11 = (b0.x & 0xFFu) == (q4 & 0xFFu);
        {  // Winkler prefix, jaro_winkler.rs:118-123
            const uint32_t lim = min(4u, min(len1_orig, len2_orig));
            const uint32_t diff = b0.x ^ q4;
            const uint32_t first_diff = diff ? (uint32_t)(__ffs(diff) - 1) / 8 : 4u;
            r.prefix = min(first_diff, lim);
        }
        asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %5, 0"
                     : "=v"(pl), "=v"(ph), "=v"(hl), "=v"(hh), "=v"(tlo), "=v"(thi));
        RF_P1(b0, 0u);  // pass 1
        if (nch > 1) RF_P1(b1, 16u);
        if (nch > 2) RF_P1(b2, 32u);
        if (nch > 3) RF_P1(b3, 48u);
        uint32_t common;
        asm volatile("v_bcnt_u32_b32 %0, %1, 0\n\tv_bcnt_u32_b32 %0, %2, %0" : "=&v"(common) : "v"(pl), "v"(ph));
        const uint32_t t_next = t + stride;
        const bool more = t_next < p.tile_end;
        const uint4* nsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)(more ? t_next : t) * p.uniform_tile_bytes);
        RF_P2(b0, 0u);  // pass 2, and the next tile's rows behind it
        b0 = nsrc[lane];
        if (nch > 1) {
            RF_P2(b1, 16u);
            b1 = nsrc[kWave + lane];
        }
        if (nch > 2) {
            RF_P2(b2, 32u);
            b2 = nsrc[2 * kWave + lane];
        }
        if (nch > 3) {
            RF_P2(b3, 48u);
            b3 = nsrc[3 * kWave + lane];
        }
        uint32_t nhits;
        asm volatile("v_bcnt_u32_b32 %0, %1, 0\n\tv_bcnt_u32_b32 %0, %2, %0" : "=&v"(nhits) : "v"(hl), "v"(hh));
        r.common = common;
        r.transpositions = common - nhits;

        // jaro::similarity_with_pm with score_cutoff = 0.0, from the tables (see jaro_word_body)
        double sim;
        if (len1_orig == 0 || len2_orig == 0) {
            sim = (len1_orig == 0 && len2_orig == 0) ? 1.0 : 0.0;
        } else if (len1_orig == 1 && len2_orig == 1) {
            sim = r.eq11 ? 1.0 : 0.0;
        } else {
            double acc = 0.0;
            acc += tab1[r.common];
            acc += tab2[r.common];
            acc += p.jaro_tab[r.common * 33u + r.transpositions / 2u];
            acc = RF_JARO_DIV3 ? div3(acc) : acc / 3.0;
            sim = select_f64(r.common == 0, 0.0, acc);
        }
        if (p.finish == FIN_JW) sim = select_f64(sim > 0.7, sim + (double)r.prefix * p.prefix_weight * (1.0 - sim), sim);
        double v = sim;
        if (p.op == RF_OP_DISTANCE || p.op == RF_OP_NORMALIZED_DISTANCE) v = 1.0 - sim;
        if (p.op == RF_OP_NORMALIZED_SIMILARITY) v = 1.0 - (1.0 - sim);
        if (idx < p.n) reinterpret_cast<double*>(p.out)[idx] = v;
        if (!more) break;
        t = t_next;
    }
#undef RF_P1
#undef RF_P2
}

static hipError_t launch_jaro_word(const ScanParams& p, ScanParams q, hipStream_t stream)
{
    const dim3 b(kWave * kWavesPerBlock);
    const bool early = p.jaro_need >= 0.0;
    const dim3 g(early ? scan_grid(q.tile_end - q.tile_begin) : scan_grid_full(q.tile_end - q.tile_begin));
    const bool fast = !p.has_cutoff && p.jaro_tab != nullptr;  // the table epilogue (no cutoff to replay)
    auto k = p.tiles ? (early ? jaro_word_kernel<false, true> : (fast ? jaro_word_fast_kernel<false> : jaro_word_kernel<false, false>))
                     : (early ? jaro_word_kernel<true, true> : (fast ? jaro_word_fast_kernel<true> : jaro_word_kernel<true, false>));
    // the hand-scheduled kernel: single-length corpus, table epilogue, truncated candidate length a multiple of 16
    static const bool use_asm = [] { const char* e = getenv("RF_ASM_CHUNK"); return !e || atoi(e) != 0; }();
    bool asm_ok = use_asm && fast && !early && !p.tiles && p.len1 >= 2;
    if (asm_ok) {
        uint32_t len1 = p.len1, len2 = p.uniform_len, bound = 0;  // jaro.rs:550-565, as in the kernel
        if (len2 > len1) {
            bound = len2 / 2 - 1;
            if (len2 > len1 + bound) len2 = len1 + bound;
        }
        asm_ok = len2 >= (uint32_t)kChunk && len2 % kChunk == 0 && len2 <= 64;
    }
    // RF_JARO_PRIV=1: the conflict-free table copy for corpora of <= 64 stored symbols.  OFF by default -- measured, it buys nothing
    // (profiles/jaro_lds_r04.txt: bank conflicts 400 M -> 0.5 M, LDS-active cycles -38 %, SQ_WAIT_INST_LDS -46 %, and the same 2.94 ms:
    // the kernel is bound by VALU issue, not by the LDS, and the copy's 16 KiB cost it one resident workgroup per CU)
    static const bool use_priv = [] { const char* e = getenv("RF_JARO_PRIV"); return e && atoi(e) != 0; }();
    if (asm_ok && use_priv && p.max_stored_sym < 64u)
        hipLaunchKernelGGL(jaro_word_asm_kernel<true>, g, b, 0, stream, q);
    else if (asm_ok)
        hipLaunchKernelGGL(jaro_word_asm_kernel<false>, g, b, 0, stream, q);
    else
        hipLaunchKernelGGL(k, g, b, 0, stream, q);
    return hipGetLastError();
}
static hipError_t launch_jaro_block(const ScanParams& p, ScanParams q, hipStream_t stream)
{
    const dim3 b(kWave * kWavesPerBlock);
    if (p.jaro_long) {  // some string is beyond 512 symbols: flags in the global scratch strips
        const dim3 g(std::max(1u, std::min<uint32_t>(p.long_grid, (uint32_t)scan_grid(q.tile_end - q.tile_begin))));
        if (p.tiles)
            hipLaunchKernelGGL(jaro_long_kernel<false>, g, b, 0, stream, q);
        else
            hipLaunchKernelGGL(jaro_long_kernel<true>, g, b, 0, stream, q);
    } else {
        const dim3 g(scan_grid(q.tile_end - q.tile_begin));
        const size_t lds = ((size_t)256 * p.words + p.words + 1) * sizeof(uint64_t);
        if (p.tiles)
            hipLaunchKernelGGL(jaro_block_kernel<false>, g, b, lds, stream, q);
        else
            hipLaunchKernelGGL(jaro_block_kernel<true>, g, b, lds, stream, q);
    }
    return hipGetLastError();
}

hipError_t launch_jaro(const ScanParams& p, hipStream_t stream)
{
    // Two runs of tiles, each ascending by length: the exact tiles [0, n_exact) and the one-length views of the mixed section
    // [n_exact, n_tiles).  In each run the tiles below the split (jaro_split / jaro_split2, plan()) take the single-word path and
    // the rest the multi-word path -- so the short leftovers behind a long exact tile are not dragged into the block kernel.
    // All of it clipped to [tile_begin, tile_end), the cutoff's length window.
    const uint32_t n_exact = std::min(p.n_exact, p.n_tiles);
    const uint32_t lo[4] = {0, std::min(p.jaro_split, n_exact), n_exact, std::max(n_exact, std::min(p.jaro_split2, p.n_tiles))};
    const uint32_t hi[4] = {lo[1], n_exact, lo[3], p.n_tiles};
    for (int r = 0; r < 4; ++r) {
        ScanParams q = p;
        q.tile_begin = std::max(lo[r], p.tile_begin);
        q.tile_end = std::min(hi[r], p.tile_end);
        if (q.tile_end <= q.tile_begin) continue;
        const hipError_t e = (r % 2 == 0) ? launch_jaro_word(p, q, stream) : launch_jaro_block(p, q, stream);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

}  // namespace rf
