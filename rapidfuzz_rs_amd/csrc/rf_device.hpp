// rf_device.hpp -- device-side building blocks shared by the gfx950 kernels (rf_scan.hip, rf_long.hip, rf_jaro.hip):
// boolean LUT / shift helpers, the per-lane recurrence states (Levenshtein, OSA, LCS and their 32-bit forms), the
// finishing arithmetic, the wavefront-local top-k list, and the tile / chunk access helpers.
// Everything here is __device__ __forceinline__ or a template: the header is included by several translation units.
//
//
// Execution shape: one candidate per wavefront lane, one 64-candidate tile per wavefront
// at a time, 4 wavefronts (one per SIMD) per workgroup, grid-stride over tiles.  The query's
// pattern-match table (256 x W u64, src/details/pattern_match_vector.rs:194-321) is staged once per
// workgroup into LDS; candidate bytes arrive as one coalesced 1 KiB global_load_dwordx4 per wavefront per
// 16 columns; the bit-vectors of the recurrence (VP/VN, S, P/T flags) never leave VGPRs.
// Integer/bitwise work only: no MFMA.  3-input boolean terms use v_bitop3_b32 (new on gfx950).
//
// Reference algorithms restated here for the device (cited per function):
//   hyrroe2003 / hyrroe2003_block      src/distance/levenshtein.rs:435-507, :769-1019 (advance_block :838-875)
//   lcs_unroll                         src/distance/lcs_seq.rs:199-261
//   flag_similar_characters_word,
//   count_transpositions_word          src/distance/jaro.rs:147-190, :339-368
//   MetricUsize / Metricf64 defaults   src/details/distance.rs:154-385
#pragma once

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "rf_internal.hpp"

namespace rf {

// ---------------------------------------------------------------------------------------------------
// v_bitop3_b32: arbitrary 3-input boolean function; the truth table is f(0xF0, 0xCC, 0xAA)
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <uint32_t TT>
__device__ __forceinline__ uint64_t lut3(uint64_t a, uint64_t b, uint64_t c)
{
    uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT & 0xFF);
    uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), TT & 0xFF);
    return ((uint64_t)hi << 32) | lo;
}
constexpr uint32_t T_XOR_OR = (TA ^ TB) | TC;       // (a ^ b) | c
constexpr uint32_t T_OR_NOR = TA | (~(TB | TC));    // a | ~(b | c)
constexpr uint32_t T_OR_ANDN = TA | (TB & ~TC);      // a | (b & ~c)
constexpr uint32_t T_AND_OR = TA & (TB | TC);        // a & (b | c)
constexpr uint32_t T_NOR3 = ~(TA | TB | TC);         // ~(a | b | c)
constexpr uint32_t T_OR3 = TA | TB | TC;             // a | b | c   (as a LUT: hipcc picks v_or3_b32 for `a | b | c`, a half-rate
                                                     //  instruction on gfx950; v_bitop3_b32 is full rate, profiles/issue_rates_r02.txt)
constexpr uint32_t T_ANDN_BA = TB & ~TA;             // ~a & b      (likewise instead of v_bfi_b32)

// (x << 1) | carry_in.  Measured on gfx950 (tools/microbench.hip, profiles/microbench_r01.txt): the 64-bit VALU
// forms v_lshl_add_u64 / v_lshlrev_b64 issue at the same (half) rate as ONE v_alignbit_b32 / v_lshl_or_b32, so a
// single 64-bit instruction beats the two-instruction 32-bit pair hipcc otherwise builds from split halves.
// Plain VALU on VGPR pairs: no memory counters, no hazard padding needed (guide 5.7).
// (hipcc canonicalises x + x + 1 back into shift-or on split halves, hence the asm; it is plain VALU on VGPR
// pairs: nothing to count, no hazard padding needed -- guide 5.7.)
template <int CIN>
__device__ __forceinline__ uint64_t shl1_const(uint64_t x)
{
    uint64_t r;
    if (CIN)
        asm("v_lshl_add_u64 %0, %1, 1, 1" : "=v"(r) : "v"(x));
    else
        asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
// (x << 1) + y in one v_lshl_add_u64 (callers guarantee the terms are disjoint where an OR is meant)
__device__ __forceinline__ uint64_t shl1_add(uint64_t x, uint64_t y)
{
    // (asm: written as (x << 1) + y hipcc sometimes splits the add into {lo, 0} + {0, hi} partial sums -- two
    // v_lshl_add_u64 and two v_mov; the price of the asm is one s_nop of hazard padding after it)
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint64_t shl1_var(uint64_t x, uint32_t cin)
{
    uint64_t r;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return r | cin;
}

// One 16-byte chunk of this lane.  Candidate bytes are read once per launch: the non-temporal hint (`nt`) keeps the
// stream from displacing the PM / descriptor lines in L2 (A/B measured; compile with -DRF_NO_NT to drop the hint).
__device__ __forceinline__ uint4 load_chunk(const uint4* src)
{
#ifdef RF_NO_NT
    return *src;
#else
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src));
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
}

// ---------------------------------------------------------------------------------------------------
// Levenshtein: one column of Hyyro's recurrence over W 64-bit words (levenshtein.rs:466-490 for W == 1,
// advance_block :838-875 for the carries between words).  The running score of :476-477 is NOT tracked:
// after the last column VP/VN hold the vertical deltas of column len2, so
//   D[len1][len2] = len2 + popcount(VP & valid) - popcount(VN & valid)        (D[0][len2] = len2)
// which removes the per-column mask tests from the hot loop.
// ---------------------------------------------------------------------------------------------------
// does a State trim its columns to a band (LevState<W >= 2>::set_band)?
template <class S, class = void>
struct has_band : std::false_type {};
template <class S>
struct has_band<S, std::void_t<decltype(S::kHasBand)>> : std::bool_constant<S::kHasBand> {};

template <int W>
struct LevState {
    using Word = uint64_t;
    static constexpr int kWords = W;
    uint64_t vp[W], vn[W];
    // W >= 2, the Ukkonen band (round 5; the reference's trimming, levenshtein.rs:810-825, :906-985; tools/gen_stream_asm.py BlockKind has the
    // derivation): bit w = word w runs in the current 16-column chunk.  Wavefront-uniform; all ones (what init() leaves, and what every kernel
    // that never calls set_band() computes with: the tests below fold away) = every word in every column.
    static constexpr bool kHasBand = W >= 2;
    uint32_t live;
    __device__ __forceinline__ void init()
    {
        live = ~0u;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            vp[w] = ~0ull;  // levenshtein.rs:454-455
            vn[w] = 0;
        }
    }
    // chunk c (columns 16c + 1 .. 16c + 16) of a tile of candidates of len2 symbols: D <= k := min(k_bound, max(len1, len2)), so only rows
    // j + dlo <= i <= j + dhi of column j can lie on a path that matters (s = (k - |len1 - len2|) / 2, dlo = min(0, len1 - len2) - s,
    // dhi = max(0, len1 - len2) + s).  A word outside is not run: one that has not started keeps VP = ~0, VN = 0, one that is done keeps its
    // deltas and the first live word above it runs with hp_c = 1, hn_c = 0 -- upper bounds of the cells they stand for (the recurrence is
    // monotone), and result() stays len2 + the popcounts.  Values <= k are exact, values beyond come out > k.  All arguments are scalars.
    __device__ __forceinline__ void set_band(uint32_t len1, uint32_t len2, uint32_t k_bound, uint32_t c)
    {
        if constexpr (W >= 2) {
            const int32_t d = (int32_t)len1 - (int32_t)len2;
            const uint32_t k = min(k_bound, max(len1, len2));
            const int32_t s = max((int32_t)k - (d < 0 ? -d : d), 0) >> 1;
            const int32_t dhi = max(d, 0) + s, dlo = min(d, 0) - s;
            const uint32_t hi = min((uint32_t)W, ((uint32_t)((int32_t)(16u * c + 15u) + dhi) >> 6) + 1u);
            const uint32_t lo = (uint32_t)max((int32_t)(16u * c) + dlo, 0) >> 6;
            live = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        }
    }
    // One column, after algebra on levenshtein.rs:467-484 (D0 is never materialised):
    //   e   = (((x & VP) + VP) ^ VP) | x           D0 = e | VN
    //   HN  = e & VP                               (== D0 & VP because VP & VN == 0)
    //   HP  = VN | ~(e | VP)                       (== VN | ~(D0 | VP))
    //   HP' = (HP << 1) + carry-in                 one v_lshl_add_u64
    //   VN' = HP' & (e | VN)                       one 3-input LUT per half
    //   T   = ~(e | VN | HP')                      one 3-input LUT per half
    //   VP' = (HN << 1) + T (+ carry-in)           one v_lshl_add_u64: the two terms are disjoint -- a set bit i of HN << 1
    //         says the cell above-left..above lost one, D[i-1][j] = D[i-1][j-1] - 1, which forces D[i][j] = D[i-1][j-1],
    //         i.e. bit i of D0 is set and bit i of T is clear -- so the add never carries and equals the reference's OR.
    // 12 full-rate + 3 half-rate VALU instructions per word (the textbook order costs 14 + 3: D0 and the separate shift
    // of HN are gone).
    __device__ __forceinline__ void step(const uint64_t (&pm_row)[W])
    {
        uint32_t hp_c = 1, hn_c = 0;  // levenshtein.rs:824-825
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (W >= 2 && !((live >> w) & 1u)) continue;     // outside the band (set_band): a skipped word leaves hp_c = 1, hn_c = 0 to the next
            uint64_t x = pm_row[w];
            if (w > 0) x |= hn_c;                            // :847
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);    // (sum ^ vp) | x;  d0 = e | vn (:848)
            // (source order matters to the schedule: with hn computed AFTER the asm shift the s_nop of hazard padding
            // behind the asm disappears, and the kernel gets 1.5 % slower -- measured A/B, twice)
            const uint64_t hn = e & p;                       // :852
            const uint64_t hp = lut3<T_OR_NOR>(n, e, p);     // vn | ~(d0 | vp) == vn | ~(e | vp)   (:851)
            const uint64_t hps = w == 0 ? shl1_const<1>(hp) : shl1_add(hp, (uint64_t)hp_c);  // :865-866
            if (w + 1 < W) hp_c = (uint32_t)(hp >> 63);      // :857-858
            vn[w] = lut3<T_AND_OR>(hps, e, n);               // hp & d0                            (:869)
            const uint64_t t = lut3<T_NOR3>(e, n, hps);      // ~(d0 | hp)
            uint64_t v = shl1_add(hn, t);                    // hn | ~(d0 | hp), hn shifted          (:868)
            if (w > 0) v |= hn_c;                            // bit 0 of T is clear when hn_c is set (x |= hn_c above)
            if (w + 1 < W) hn_c = (uint32_t)(hn >> 63);
            vp[w] = v;
        }
    }
    // the same column with the horizontal deltas entering word 0 / leaving word W-1 as variables: one 512-row
    // group of a longer pattern (long_kernel); levenshtein.rs:838-875 with hp_carry / hn_carry crossing groups
    __device__ __forceinline__ void step_carry(const uint64_t (&pm_row)[W], uint32_t& hp_c, uint32_t& hn_c)
    {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t x = pm_row[w] | hn_c;
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);
            const uint64_t hn = e & p;
            const uint64_t hp = lut3<T_OR_NOR>(n, e, p);
            const uint64_t hps = shl1_add(hp, (uint64_t)hp_c);
            hp_c = (uint32_t)(hp >> 63);
            vn[w] = lut3<T_AND_OR>(hps, e, n);
            const uint64_t t = lut3<T_NOR3>(e, n, hps);
            vp[w] = shl1_add(hn, t) | hn_c;
            hn_c = (uint32_t)(hn >> 63);
        }
    }
    // popcount contribution of this group's words to D[len1][j]; word w is absolute word (word0 + w)
    __device__ __forceinline__ int32_t delta_sum(uint32_t len1, uint32_t word0) const
    {
        int32_t d = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int32_t bits = (int32_t)len1 - 64 * (int32_t)(word0 + w);
            const uint64_t valid = bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1));
            d += __popcll(vp[w] & valid) - __popcll(vn[w] & valid);
        }
        return d;
    }
    // Early-out bound under a distance cutoff (the reference applies its cutoff only after the loop,
    // levenshtein.rs:492-496, so this is value-preserving pruning): adjacent cells of the last row differ by at
    // most 1, hence D[len1][len2] >= D[len1][j] - (len2 - j).  D[len1][j] comes from the same popcount identity
    // as result().  The kernels feed the bound to may_pass() (below): no lane may pass -> the tile is abandoned.
    // A second bound comes from the diagonal through (len1, len2): values never decrease along a diagonal of the
    // Levenshtein matrix, so D[len1][len2] >= D[j + len1 - len2][j] -- the same popcount identity with a shorter row
    // mask.  For equal lengths that is the distance between the two j-prefixes, which for unrelated strings grows by
    // almost 1 per column: nearly every wavefront of a random corpus is past a small cutoff after 8 columns.
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ uint32_t bound(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t last_row = (int32_t)result(len1, j) - (int32_t)(len2 - j);
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;  // <= len1 because j <= len2
        const int32_t diag = i > 0 ? (int32_t)result((uint32_t)i, j) : 0;
        return (uint32_t)max(max(last_row, diag), 0);  // a LOWER bound on D[len1][len2]
    }
    // the bound for the look taken in the MIDDLE of a tile's first chunk (scan_body): the diagonal term alone -- early in the
    // tile the last-row term is just the length difference, which the host's length window has already applied -- at half the
    // popcounts.  Weaker than bound() at worst, never wrong.
    __device__ __forceinline__ uint32_t bound_first(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;
        return i > 0 ? result((uint32_t)i, j) : bound(len1, j, len2);
    }
    // D[len1][len2] from the final column's vertical deltas (any row count <= len1 gives D[rows][len2])
    __device__ __forceinline__ uint32_t result(uint32_t len1, uint32_t len2) const
    {
        int32_t d = (int32_t)len2;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int32_t bits = (int32_t)len1 - 64 * w;  // valid pattern rows in this word
            uint64_t valid = bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1));
            d += __popcll(vp[w] & valid) - __popcll(vn[w] & valid);
        }
        return (uint32_t)d;
    }
};

// ---------------------------------------------------------------------------------------------------
// OSA (optimal string alignment, src/distance/osa.rs:60-226): Hyyro's recurrence plus the transposition term
//   tr = ((~D0_old & PM) << 1 | carry from the word below) & PM_old          (osa.rs:86, :180)
// OR-ed into D0, which costs two more bit-vectors of state per word (D0 and the previous column's PM).
// ---------------------------------------------------------------------------------------------------
template <int W>
struct OsaState {
    using Word = uint64_t;
    static constexpr int kWords = W;
    uint64_t vp[W], vn[W], d0[W], pm_old[W];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            vp[w] = ~0ull;  // osa.rs:74-77, :125-135
            vn[w] = 0;
            d0[w] = 0;
            pm_old[w] = 0;
        }
    }
    __device__ __forceinline__ void step(const uint64_t (&pm_row)[W])
    {
        uint32_t hp_c = 1, hn_c = 0, tr_c = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t pm_j = pm_row[w];
            const uint64_t t = lut3<T_ANDN_BA>(d0[w], pm_j, pm_j);         // ~d0 & pm_j: candidates for a transposition
            const uint64_t tr = (w == 0 ? shl1_const<0>(t) : shl1_var(t, tr_c)) & pm_old[w];  // osa.rs:180
            if (w + 1 < W) tr_c = (uint32_t)(t >> 63);                      // ((~d0_last) & pm_last) >> 63 for the next word
            uint64_t x = pm_j;
            if (w > 0) x |= hn_c;                                           // osa.rs:182
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);
            const uint64_t d = lut3<T_OR3>(e, n, tr);                       // osa.rs:183
            const uint64_t hn = d & p;
            const uint64_t hp = lut3<T_OR_NOR>(n, d, p);
            const uint64_t hps = w == 0 ? shl1_const<1>(hp) : shl1_var(hp, hp_c);
            const uint64_t hns = w == 0 ? shl1_const<0>(hn) : shl1_var(hn, hn_c);
            if (w + 1 < W) {
                hp_c = (uint32_t)(hp >> 63);
                hn_c = (uint32_t)(hn >> 63);
            }
            vn[w] = hps & d;
            vp[w] = lut3<T_OR_NOR>(hns, hps, d);
            d0[w] = d;
            pm_old[w] = pm_j;
        }
    }
    // one 512-row group of a longer pattern (long_kernel): the horizontal deltas and the transposition candidate bit enter
    // word 0 from the group below and leave word W-1 for the group above (osa.rs:156-226 across block boundaries)
    __device__ __forceinline__ void step_carry(const uint64_t (&pm_row)[W], uint32_t& hp_c, uint32_t& hn_c, uint32_t& tr_c)
    {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t pm_j = pm_row[w];
            const uint64_t t = lut3<T_ANDN_BA>(d0[w], pm_j, pm_j);
            const uint64_t tr = shl1_var(t, tr_c) & pm_old[w];  // osa.rs:180
            tr_c = (uint32_t)(t >> 63);
            const uint64_t x = pm_j | hn_c;                       // osa.rs:182
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);
            const uint64_t d = lut3<T_OR3>(e, n, tr);             // osa.rs:183
            const uint64_t hn = d & p;
            const uint64_t hp = lut3<T_OR_NOR>(n, d, p);
            const uint64_t hps = shl1_var(hp, hp_c);
            const uint64_t hns = shl1_var(hn, hn_c);
            hp_c = (uint32_t)(hp >> 63);
            hn_c = (uint32_t)(hn >> 63);
            vn[w] = hps & d;
            vp[w] = lut3<T_OR_NOR>(hns, hps, d);
            d0[w] = d;
            pm_old[w] = pm_j;
        }
    }
    __device__ __forceinline__ int32_t delta_sum(uint32_t len1, uint32_t word0) const
    {
        int32_t d = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int32_t bits = (int32_t)len1 - 64 * (int32_t)(word0 + w);
            const uint64_t valid = bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1));
            d += __popcll(vp[w] & valid) - __popcll(vn[w] & valid);
        }
        return d;
    }
    // The vertical-delta identity and both bounds of LevState::bound hold for the OSA matrix as well: steps are unit,
    // and values never decrease along a diagonal (drop the last symbol of both strings from an optimal restricted
    // alignment: a pair aligned to each other disappears, a transposed pair (a_i a_i+1)/(b_j b_j+1) becomes one
    // substitution, and in every other case a deletion or insertion of a last symbol is saved for at most one new one).
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ uint32_t bound(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t last_row = (int32_t)result(len1, j) - (int32_t)(len2 - j);
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;
        const int32_t diag = i > 0 ? (int32_t)result((uint32_t)i, j) : 0;
        return (uint32_t)max(max(last_row, diag), 0);
    }
    // the bound for the look taken in the MIDDLE of a tile's first chunk (scan_body): the diagonal term alone -- early in the
    // tile the last-row term is just the length difference, which the host's length window has already applied -- at half the
    // popcounts.  Weaker than bound() at worst, never wrong.
    __device__ __forceinline__ uint32_t bound_first(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;
        return i > 0 ? result((uint32_t)i, j) : bound(len1, j, len2);
    }
    __device__ __forceinline__ uint32_t result(uint32_t len1, uint32_t len2) const
    {
        int32_t d = (int32_t)len2;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int32_t bits = (int32_t)len1 - 64 * w;
            const uint64_t valid = bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1));
            d += __popcll(vp[w] & valid) - __popcll(vn[w] & valid);
        }
        return (uint32_t)d;
    }
};

// ---------------------------------------------------------------------------------------------------
// LCS: Hyyro's bit-parallel LCS length (lcs_seq.rs:222-252): S' = (S + (S & M)) | (S - (S & M)) with the
// add's carry chained across words; similarity = sum popcount(~S).
// ---------------------------------------------------------------------------------------------------
template <int W>
struct LcsState {
    using Word = uint64_t;
    static constexpr int kWords = W;
    uint64_t s[W];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int w = 0; w < W; ++w) s[w] = ~0ull;  // lcs_seq.rs:215
    }
    __device__ __forceinline__ void step(const uint64_t (&pm_row)[W])
    {
        uint64_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t sw = s[w];
            const uint64_t u = sw & pm_row[w];
            uint64_t x = sw + u;  // carrying_add, src/details/intrinsics.rs:22-26
            uint64_t c = x < sw;
            if (w > 0) {
                const uint64_t x2 = x + carry;
                c |= (uint64_t)(x2 < x);
                x = x2;
            }
            carry = c;
            // lcs_seq.rs:230 `x | (s - u)`: u is a subset of s, so the subtraction never borrows and
            // s - u == s & ~u -- one v_bitop3 per half instead of a carry-chained 64-bit subtract
            s[w] = lut3<T_OR_ANDN>(x, sw, u);
        }
    }
    // one group of a longer pattern: the adder carry enters word 0 and leaves word W-1 (lcs_seq.rs:313-318)
    __device__ __forceinline__ void step_carry(const uint64_t (&pm_row)[W], uint32_t& carry_io)
    {
        uint64_t carry = carry_io;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t sw = s[w];
            const uint64_t u = sw & pm_row[w];
            uint64_t x = sw + u;
            uint64_t c = x < sw;
            const uint64_t x2 = x + carry;
            c |= (uint64_t)(x2 < x);
            x = x2;
            carry = c;
            s[w] = lut3<T_OR_ANDN>(x, sw, u);
        }
        carry_io = (uint32_t)carry;
    }
    // After j columns the LCS can still grow by at most one per remaining candidate symbol, and never beyond the shorter
    // string: an UPPER bound on the final LCS length (the favourable side for every op built on it).
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ uint32_t bound(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        return min(result(len1, j) + (len2 - j), min(len1, len2));
    }
    __device__ __forceinline__ uint32_t bound_first(uint32_t len1, uint32_t j, uint32_t len2) const { return bound(len1, j, len2); }
    __device__ __forceinline__ uint32_t result(uint32_t, uint32_t) const
    {
        uint32_t sim = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) sim += __popcll(~s[w]);  // lcs_seq.rs:254-257
        return sim;
    }
};

// ---------------------------------------------------------------------------------------------------
// 32-bit specialisations for queries of at most 32 symbols (BASELINE.json configs[0] shape, most real-world
// names/titles): the same recurrences on ONE VGPR per bit-vector -- 10 instead of 18 VALU instructions per column
// for Levenshtein -- reading the low half of each PM entry from a 1 KiB LDS table.
// ---------------------------------------------------------------------------------------------------
template <uint32_t TT>
__device__ __forceinline__ uint32_t lut3w(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT & 0xFF);
}

struct Lev32State {
    using Word = uint32_t;
    static constexpr int kWords = 1;
    uint32_t vp, vn;
    __device__ __forceinline__ void init()
    {
        vp = ~0u;
        vn = 0;
    }
    __device__ __forceinline__ void step(const uint32_t (&pm_row)[1])
    {
        const uint32_t x = pm_row[0];
        const uint32_t sum = (x & vp) + vp;
        const uint32_t e = lut3w<T_XOR_OR>(sum, vp, x);
        // the same algebra as LevState::step: D0 = e | VN is never materialised, HN is never shifted on its own, and
        // VP' = (HN << 1) + T with T = ~(D0 | HP') -- the two terms are disjoint, so the add equals the reference's OR
        // (7 full-rate + 2 half-rate instructions instead of 8 + 2, and one of the half-rate shifts gone)
        const uint32_t hn = e & vp;
        const uint32_t hp = lut3w<T_OR_NOR>(vn, e, vp);   // vn | ~(e | vp)
        uint32_t hps;
        asm("v_lshl_or_b32 %0, %1, 1, 1" : "=v"(hps) : "v"(hp));
        const uint32_t t = lut3w<T_NOR3>(e, vn, hps);
        vn = lut3w<T_AND_OR>(hps, e, vn);
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(vp) : "v"(hn), "v"(t));
    }
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ uint32_t bound(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t last_row = (int32_t)result(len1, j) - (int32_t)(len2 - j);
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;  // the diagonal bound, see LevState::bound
        const int32_t diag = i > 0 ? (int32_t)result((uint32_t)i, j) : 0;
        return (uint32_t)max(max(last_row, diag), 0);
    }
    // the bound for the look taken in the MIDDLE of a tile's first chunk (scan_body): the diagonal term alone -- early in the
    // tile the last-row term is just the length difference, which the host's length window has already applied -- at half the
    // popcounts.  Weaker than bound() at worst, never wrong.
    __device__ __forceinline__ uint32_t bound_first(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;
        return i > 0 ? result((uint32_t)i, j) : bound(len1, j, len2);
    }
    __device__ __forceinline__ uint32_t result(uint32_t len1, uint32_t len2) const
    {
        const uint32_t valid = len1 >= 32 ? ~0u : ((1u << len1) - 1);
        return (uint32_t)((int32_t)len2 + __popc(vp & valid) - __popc(vn & valid));
    }
};

// OSA on 32-bit words (osa.rs:156-226 as OsaState<1>::step): the first look of the cutoff scans runs on it (rf_scan.hip
// early_lean_body: the diagonal bound at column kFirst only needs the first <= 32 pattern rows)
struct Osa32State {
    using Word = uint32_t;
    static constexpr int kWords = 1;
    uint32_t vp, vn, d0, pm_old;
    __device__ __forceinline__ void init()
    {
        vp = ~0u;
        vn = 0;
        d0 = 0;
        pm_old = 0;
    }
    __device__ __forceinline__ void step(const uint32_t (&pm_row)[1])
    {
        const uint32_t pm_j = pm_row[0];
        const uint32_t t = lut3w<T_ANDN_BA>(d0, pm_j, pm_j);  // ~d0 & pm_j
        const uint32_t tr = (t << 1) & pm_old;                  // osa.rs:180
        const uint32_t sum = (pm_j & vp) + vp;
        const uint32_t e = lut3w<T_XOR_OR>(sum, vp, pm_j);
        const uint32_t d = lut3w<T_OR3>(e, vn, tr);             // osa.rs:183
        const uint32_t hn = d & vp;
        const uint32_t hp = lut3w<T_OR_NOR>(vn, d, vp);
        uint32_t hps;
        asm("v_lshl_or_b32 %0, %1, 1, 1" : "=v"(hps) : "v"(hp));
        const uint32_t hns = hn << 1;
        vn = hps & d;
        vp = lut3w<T_OR_NOR>(hns, hps, d);
        d0 = d;
        pm_old = pm_j;
    }
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ uint32_t bound(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t last_row = (int32_t)result(len1, j) - (int32_t)(len2 - j);
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;  // the diagonal bound, see OsaState::bound
        const int32_t diag = i > 0 ? (int32_t)result((uint32_t)i, j) : 0;
        return (uint32_t)max(max(last_row, diag), 0);
    }
    __device__ __forceinline__ uint32_t bound_first(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;
        return i > 0 ? result((uint32_t)i, j) : bound(len1, j, len2);
    }
    __device__ __forceinline__ uint32_t result(uint32_t len1, uint32_t len2) const
    {
        const uint32_t valid = len1 >= 32 ? ~0u : ((1u << len1) - 1);
        return (uint32_t)((int32_t)len2 + __popc(vp & valid) - __popc(vn & valid));
    }
};

struct Lcs32State {
    using Word = uint32_t;
    static constexpr int kWords = 1;
    uint32_t s;
    __device__ __forceinline__ void init() { s = ~0u; }
    __device__ __forceinline__ void step(const uint32_t (&pm_row)[1])
    {
        const uint32_t u = s & pm_row[0];
        s = lut3w<T_OR_ANDN>(s + u, s, u);  // (s + u) | (s - u) with s - u == s & ~u
    }
    // After j columns the LCS can still grow by at most one per remaining candidate symbol, and never beyond the shorter
    // string: an UPPER bound on the final LCS length (the favourable side for every op built on it).
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ uint32_t bound(uint32_t len1, uint32_t j, uint32_t len2) const
    {
        return min(result(len1, j) + (len2 - j), min(len1, len2));
    }
    __device__ __forceinline__ uint32_t bound_first(uint32_t len1, uint32_t j, uint32_t len2) const { return bound(len1, j, len2); }
    __device__ __forceinline__ uint32_t result(uint32_t, uint32_t) const { return __popc(~s); }
};

// ---------------------------------------------------------------------------------------------------
// finishing arithmetic: raw primitive -> the value `<op>_with_args` returns (or None)
// ---------------------------------------------------------------------------------------------------
// norm_sim_to_norm_dist, src/details/common.rs:4-7
__device__ __forceinline__ double norm_sim_to_norm_dist(double c) { return fmin(1.0 - c + 0.00001, 1.0); }

// Finishing.  For every (metric, weights) this path serves, distance and maximum are affine in
//   S = len1 + len2,  Mx = max(len1, len2)  and the raw recurrence result (Levenshtein distance or LCS length):
//     uniform Levenshtein (f,f,f)      dist = f*raw            maximum = f*Mx   (levenshtein.rs:263-277, :1308-1316)
//     lcs_seq                          dist = Mx - raw         maximum = Mx     (details/distance.rs:157-179)
//     indel                            dist = S - 2*raw        maximum = S      (indel.rs:365-367)
//     Levenshtein (f,f,>=2f)           dist = f*(S - 2*raw)    maximum = f*S    (levenshtein.rs:1321-1327)
// so the host folds metric, weights and op into a few coefficients (rf_api_scan.hip plan()) and the kernels do one
// multiply-add per candidate with tile-uniform (scalar) S and Mx -- no per-tile branching on the metric.
// All arithmetic is mod 2^32 like the reference's usize arithmetic is mod 2^64.
struct TileFin {
    uint32_t v0;       // value at raw == 0: fin_vS * S + fin_vM * Mx
    uint32_t d0, max;  // distance at raw == 0 and the maximum (normalized ops only)
};
__device__ __forceinline__ TileFin tile_fin(const ScanParams& p, uint32_t len1, uint32_t len2)
{
    const uint32_t S = len1 + len2, Mx = max(len1, len2);
    TileFin f;
    f.v0 = (uint32_t)p.fin_vS * S + (uint32_t)p.fin_vM * Mx;
    f.d0 = (uint32_t)p.fin_dS * S + (uint32_t)p.fin_dM * Mx;
    f.max = (uint32_t)p.fin_mS * S + (uint32_t)p.fin_mM * Mx;
    return f;
}
// Which value the op yields and whether `score()` (src/common.rs:43-45 / :83-85) keeps it.  All kernels on
// this path are exact, so the CPU-side cutoff plumbing (details/distance.rs:157-274) reduces to
// "compute the value, then compare with the user's cutoff" -- see DESIGN.md "cutoff equivalence".
// distance keeps v <= cutoff, similarity keeps v >= cutoff: one compare after xor-ing both sides with fin_flip.
__device__ __forceinline__ uint32_t usize_value(const ScanParams& p, const TileFin& f, uint32_t raw, bool* keep)
{
    const uint32_t v = f.v0 + (uint32_t)p.fin_vR * raw;
    *keep = (v ^ p.fin_flip) <= p.fin_cflip;
    return v;
}
__device__ __forceinline__ uint32_t usize_value(const ScanParams& p, uint32_t raw, uint32_t len2, bool* keep, uint32_t len1)
{
    return usize_value(p, tile_fin(p, len1, len2), raw, keep);
}

// one result from a tile's finishing terms; `out` is the launch's (or, in the multi-query kernel, the query's) row
__device__ __forceinline__ void emit_fin(const ScanParams& p, const TileFin& f, uint32_t raw, uint32_t idx, void* out)
{
    if (!p.out_f64) {
        bool keep;
        const uint32_t v = usize_value(p, f, raw, &keep);
        reinterpret_cast<uint32_t*>(out)[idx] = keep ? v : RF_NONE_U32;
    } else {
        const uint32_t dist = f.d0 + (uint32_t)p.fin_dR * raw;
        // details/distance.rs:246-250: dist / maximum (0.0 when maximum == 0)
        const double nd = f.max == 0 ? 0.0 : (double)dist / (double)f.max;
        double v;
        bool keep;
        if (p.op == RF_OP_NORMALIZED_DISTANCE) {
            v = nd;
            keep = !p.has_cutoff || v <= p.cutoff_f64;
        } else {  // details/distance.rs:273: 1.0 - norm_dist
            v = 1.0 - nd;
            keep = !p.has_cutoff || v >= p.cutoff_f64;
        }
        reinterpret_cast<double*>(out)[idx] = keep ? v : __longlong_as_double(0x7FF8000000000000ll);
    }
}
__device__ __forceinline__ void emit_usize(const ScanParams& p, uint32_t raw, uint32_t len2, uint32_t idx, void* out, uint32_t len1)
{
    emit_fin(p, tile_fin(p, len1, len2), raw, idx, out);
}
// Could a candidate whose raw result is at best `raw_bound` still pass the cutoff?  `raw_bound` is the favourable
// bound of State::bound() (a lower bound on a distance, an upper bound on an LCS length); every finishing map here is
// monotone in raw and runs the very arithmetic of emit_fin, so "the bound fails" implies "every reachable value
// fails" -- for the u32 and the f64 (normalized_*) outputs alike.  This is what makes the early-out value-preserving.
__device__ __forceinline__ bool may_pass(const ScanParams& p, const TileFin& f, uint32_t raw_bound)
{
    bool keep;
    if (!p.out_f64) {
        (void)usize_value(p, f, raw_bound, &keep);
    } else {
        const uint32_t dist = f.d0 + (uint32_t)p.fin_dR * raw_bound;
        const double nd = f.max == 0 ? 0.0 : (double)dist / (double)f.max;
        keep = p.op == RF_OP_NORMALIZED_DISTANCE ? nd <= p.cutoff_f64 : (1.0 - nd) >= p.cutoff_f64;
    }
    return keep;
}
__device__ __forceinline__ void emit_none(const ScanParams& p, uint32_t idx)
{
    if (!p.out_f64)
        reinterpret_cast<uint32_t*>(p.out)[idx] = RF_NONE_U32;
    else
        reinterpret_cast<double*>(p.out)[idx] = __longlong_as_double(0x7FF8000000000000ll);
}
__device__ __forceinline__ void emit_usize(const ScanParams& p, uint32_t raw, uint32_t len2, uint32_t idx)
{
    emit_usize(p, raw, len2, idx, p.out, p.len1);
}

// ---------------------------------------------------------------------------------------------------
// wavefront-local top-k (k <= 64): lane l holds the l-th smallest 64-bit key, ~0 = empty.  A key is
// (score << 32 | local index) for "smaller is better" and (~score << 32 | local index) for similarities, so
// the order is exactly (score, index) and keys are unique.  Everything stays in two VGPRs per lane.
// ---------------------------------------------------------------------------------------------------
struct WaveTopK {
    uint64_t key;
    __device__ __forceinline__ void init() { key = ~0ull; }
    __device__ __forceinline__ uint64_t worst(uint32_t k) const
    {
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)key, k - 1), hi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), k - 1);
        return ((uint64_t)hi << 32) | lo;
    }
    // x is wavefront-uniform
    __device__ __forceinline__ void insert(uint64_t x, uint32_t lane)
    {
        const uint32_t pos = __popcll(__ballot(key < x));  // sorted ascending: the smaller keys are a lane prefix
        // the key of lane - 1: DPP wave_shr:1 (one v_mov_b32_dpp per half, whole-wavefront shift on gfx9-family parts).
        // __shfl_up goes through ds_bpermute_b32 -- two LDS round trips per insertion, which made every selection phase
        // (a wavefront's first tile, the workgroup merge, the collectors) latency-bound at ~0.2 us per inserted key.
        const uint32_t up_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)key, 0x138, 0xF, 0xF, false);
        const uint32_t up_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(key >> 32), 0x138, 0xF, 0xF, false);
        // key = lane > pos ? up : (lane == pos ? x : key), with the two lane masks built on the scalar unit (pos is uniform) and
        // selected through SGPR pairs: written as `?:` on 64-bit values hipcc emits pairs of VOP2 v_cndmask_b32 ..., vcc, and two
        // of those back to back cost ~10 ns each on this chip (profiles/issue_rates_r02.txt)
        const uint64_t at = pos < 64 ? 1ull << pos : 0ull, above = pos < 63 ? ~0ull << (pos + 1) : 0ull;
        uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
        asm("v_cndmask_b32 %0, %0, %2, %6\n\tv_cndmask_b32 %1, %1, %3, %6\n\tv_cndmask_b32 %0, %0, %4, %7\n\tv_cndmask_b32 %1, %1, %5, %7"
            : "+v"(lo), "+v"(hi)
            : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"(up_lo), "v"(up_hi), "s"(at), "s"(above));  // (one SGPR operand per VALU instruction: x travels in VGPRs)
        key = ((uint64_t)hi << 32) | lo;
        (void)lane;
    }
    // Offer one key per lane (valid lanes only).  `limit` is wavefront-uniform: min(this list's worst key, any upper
    // bound on the launch's k-th best key) -- keys at or above it can never be in the answer.  The common case (no
    // lane below the limit) is one compare and one scalar branch; returns true when the list changed.
    __device__ __forceinline__ bool offer(uint64_t mine, bool valid, uint32_t k, uint32_t lane, uint64_t limit)
    {
        uint64_t m = __ballot(valid && mine < limit);
        if (m == 0) return false;
        bool changed = false;
        while (m) {  // rare: a handful per wavefront over a whole launch once the bound is tight
            const uint32_t l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine, l), hi = __builtin_amdgcn_readlane((uint32_t)(mine >> 32), l);
            const uint64_t x = ((uint64_t)hi << 32) | lo;
            if (x < worst(k)) {
                insert(x, lane);
                changed = true;
            }
        }
        return changed;
    }
};

// ---------------------------------------------------------------------------------------------------
// End of a scan kernel in top-k mode: from per-wavefront lists to the launch's k best keys, inside the scan launch
// (the selection used to be launches of its own).  Workgroups are dealt into kTopkWays "ways" (way = blockIdx % 64):
//   * every workgroup merges its 4 wavefront lists (LDS) and appends the keys that can still be in the answer (at or
//     below the pruning bound as it last saw it) to its WAY's segment of the candidate buffer, behind that way's counter;
//   * it then ARRIVES at its way with one fire-and-forget atomic increment (no return value: nobody waits for it);
//   * the workgroup with the highest blockIdx of each way is that way's SUB-COLLECTOR: it polls the way's arrival counter
//     (s_sleep between polls) until every other member has arrived, selects the k best of the way's segment, writes them
//     to the root table, re-arms the way's counters and arrives at the root;
//   * the sub-collector with the highest blockIdx is the ROOT: it waits for the other sub-collectors, selects the k best
//     of the root table into p.topk_out and re-arms the root counter and the bound (so the next launch on this scratch
//     needs no memset).
// 64 small selections run in parallel where one workgroup used to chew through every published key (~24 k keys after a
// sampled bound: a 60 us serial tail on a 2.5 ms scan).  Only collectors ever wait, and only for workgroups that never
// wait themselves (or, the root, for sub-collectors), so whatever order the dispatcher picks this cannot deadlock: at
// worst 64 of the 2048 workgroup slots idle while the rest drain.
// Counters live one per 128-byte line (same-address atomics retire one at a time at the memory side).  Keys are written
// with agent-scope stores (sc1: through to the memory side, where a collector on another XCD -- whose L2 is not coherent
// with this one -- will read them) and the writer only waits for their completion (vmcnt) before it arrives: an
// agent-scope release FENCE would write back the whole L2, which is full of the scan's own result stores (measured
// +0.16 ms per launch with ~13 k publishing workgroups; acq_rel tickets on every workgroup: +0.7 ms).  A collector does
// one agent-scope acquire (L2 invalidate) after its last arrival.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kTopkWays = 64, kTopkLine = 32;  // ways; u32 units per 128-byte control line
// control block (u32 units): line w < 64: {arrivals, candidate count} of way w; line 64: {arrivals at the root}
__device__ __forceinline__ uint32_t* topk_way_arrivals(const ScanParams& p, uint32_t way) { return p.topk_ctl + way * kTopkLine; }
__device__ __forceinline__ uint32_t* topk_way_count(const ScanParams& p, uint32_t way) { return p.topk_ctl + way * kTopkLine + 1; }

__device__ __forceinline__ void topk_arrive(uint32_t* counter)
{
    (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wavefront-uniform wait until *counter == expect, then acquire.  Bounded by wall time (100 MHz constant clock): a lost
// arrival -- a bug -- must not hang the device for good; no scan that fits in HBM runs anywhere near two minutes.
__device__ __forceinline__ void topk_await(const uint32_t* counter, uint32_t expect)
{
    const uint64_t t0 = wall_clock64();
    while (uniform(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != expect) {
        if (wall_clock64() - t0 > 120ull * 100000000ull) break;
        __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Final selection over `n` candidate keys by ONE workgroup of kWavesPerBlock wavefronts: each keeps a sorted k-list over its
// stripe of the candidates (WaveTopK), the lists meet in LDS and wavefront 0 merges them; returns (in wavefront 0) the
// merged list.  kRows 64-key rows per trip are loaded before the first is offered: the loop is bound by load latency.
__device__ __forceinline__ void topk_select(const uint64_t* __restrict__ keys, uint32_t n, uint32_t k, uint64_t (*lists)[kWave], uint32_t wave,
                                            uint32_t lane, WaveTopK& best)
{
    constexpr uint32_t kThreads = kWave * kWavesPerBlock;
    const uint32_t used = min((uint32_t)kWavesPerBlock, (n + kWave - 1) / kWave);  // wavefronts that see any key at all
    best.init();
    constexpr uint32_t kRows = 8;
    uint64_t limit = ~0ull;  // this list's worst key once it is full
    for (uint32_t base = wave * kWave; base < n; base += kRows * kThreads) {
        uint64_t row[kRows];
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint32_t i = base + r * kThreads + lane;
            row[r] = i < n ? keys[i] : ~0ull;
        }
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r)
            if (best.offer(row[r], row[r] != ~0ull, k, lane, limit)) limit = best.worst(k);
    }
    if (used > 1) {  // (workgroup-uniform)
        __syncthreads();  // (the lists array may still be read by wavefront 0 from an earlier use)
        lists[wave][lane] = best.key;
        __syncthreads();
    }
    if (wave == 0)
        for (uint32_t w = 1; w < used; ++w)
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t x = lists[w][j];
                if (x >= best.worst(k)) break;  // the lists are sorted: nothing further in this one can enter
                best.insert(x, lane);
            }
}

// `bound_seen` = the launch-wide pruning bound as this workgroup last saw it: stale is merely conservative, and
// re-reading it here would put one more memory round trip at the end of every workgroup.
__device__ __forceinline__ void topk_block_publish(const ScanParams& p, WaveTopK& best, uint64_t (*lds_topk)[kWave], uint32_t wave, uint32_t lane,
                                                   uint64_t bound_seen)
{
    const uint32_t grid = gridDim.x, way = blockIdx.x % kTopkWays;
    const uint32_t n_ways = min(kTopkWays, grid);
    const bool is_sub = blockIdx.x + n_ways >= grid, is_root = blockIdx.x + 1 == grid;  // workgroup-uniform
    // most workgroups of a launch with a tight bound end with four empty lists: one barrier and they are gone
    if (__syncthreads_or(__ballot(best.key != ~0ull) != 0)) {
        lds_topk[wave][lane] = best.key;
        __syncthreads();
        if (wave == 0) {
            for (uint32_t w = 1; w < kWavesPerBlock; ++w)
                for (uint32_t j = 0; j < p.topk_k; ++j) {
                    const uint64_t x = lds_topk[w][j];  // wavefront-uniform address: a broadcast read
                    if (x >= best.worst(p.topk_k)) break;
                    best.insert(x, lane);
                }
            const bool keep = lane < p.topk_k && best.key != ~0ull && best.key <= bound_seen;
            const uint64_t m = __ballot(keep);
            if (m) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(topk_way_count(p, way), (uint32_t)__popcll(m));
                base = uniform(base);
                uint64_t* seg = p.topk_cand + (size_t)way * p.topk_seg_cap;
                if (keep) __hip_atomic_store(seg + base + __popcll(m & ((1ull << lane) - 1)), best.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the key stores have reached the memory side
            }
        }
    }
    if (!is_sub) {
        if (wave == 0 && lane == 0) topk_arrive(topk_way_arrivals(p, way));
        return;
    }
    // ---- sub-collector of `way` ----
    if (wave == 0) topk_await(topk_way_arrivals(p, way), grid / kTopkWays + (way < grid % kTopkWays ? 1u : 0u) - 1u);
    __syncthreads();
    const uint32_t n = __hip_atomic_load(topk_way_count(p, way), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    topk_select(p.topk_cand + (size_t)way * p.topk_seg_cap, n, p.topk_k, lds_topk, wave, lane, best);
    if (wave == 0) {
        __hip_atomic_store(p.topk_root + way * kWave + lane, lane < p.topk_k ? best.key : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 2) topk_way_arrivals(p, way)[lane] = 0;  // re-arm {arrivals, count}: every member of the way has arrived
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!is_root && lane == 0) topk_arrive(p.topk_ctl + kTopkWays * kTopkLine);
    }
    if (!is_root) return;
    // ---- root ----
    if (wave == 0) topk_await(p.topk_ctl + kTopkWays * kTopkLine, n_ways - 1);
    __syncthreads();
    topk_select(p.topk_root, n_ways * kWave, p.topk_k, lds_topk, wave, lane, best);
    if (wave == 0) {
        if (lane < p.topk_k) p.topk_out[lane] = best.key;
        // The scans admit keys strictly below their limit and the sample's lists are discarded, so the bound handed to
        // the main scan after the SAMPLE pass of a top-k call (topk_core()) is kth + 1: the sample's k-th best candidate
        // may BE the corpus' k-th best and has to be found again (keys are unique: `key < kth + 1` is `key <= kth`).
        const uint64_t kth = best.worst(p.topk_k);
        if (lane == 0) {
            p.topk_ctl[kTopkWays * kTopkLine] = 0;
            *p.topk_bound = (p.topk_bound_from_result && kth != ~0ull) ? kth + 1 : ~0ull;
        }
    }
}

// Launch-wide pruning bound: once ANY wavefront holds k keys, its worst key bounds the global k-th best from above
// (and topk_core() seeds it with the k-th best of a sample before the scan starts).  A wavefront whose list just
// changed publishes its worst key with a 64-bit atomic min.
__device__ __forceinline__ void topk_list_changed(const ScanParams& p, const WaveTopK& best, uint32_t lane, uint64_t& limit)
{
    const uint64_t w = best.worst(p.topk_k);
    if (w < limit) {  // limit <= the last bound this wavefront saw: only then can the global bound improve
        // fire-and-forget, and invisible to the compiler's vmcnt bookkeeping for the same reason as topk_refresh_bound: a
        // conditionally issued vector-memory op inside the tile loop makes every chunk wait pessimistic (an extra op in
        // flight can only make a counted wait longer, never shorter: loads still return in order among themselves)
        if (lane == 0) asm volatile("global_atomic_umin_x2 %0, %1, off" ::"v"(p.topk_bound), "v"(w) : "memory");
        limit = w;
    }
}
// Re-read the launch-wide bound (every few tiles) and fold it into the scalar limit.  The load is issued AND waited for
// inside one asm statement, deliberately: as a compiler-visible load riding across tiles it sat in the middle of the chunk
// prefetch stream, and because it is issued conditionally the compiler's vmcnt bookkeeping had to assume the worst at
// every loop merge point -- its waits for "my chunk" degenerated into drains and the one-chunk-ahead prefetch was lost
// (rocprofv3 PMC, Indel scan in top-k mode: same instruction counts as the plain scan, +30 % SQ_WAIT_INST_ANY,
// 1.32 -> 1.50 ms).  A synchronous read costs one memory round trip per 8 tiles instead.  sc1: read at the memory side
// (the bound is updated by atomics from every XCD and the XCDs' L2s are not coherent with each other).
__device__ __forceinline__ void topk_refresh_bound(const ScanParams& p, uint64_t& limit)
{
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p.topk_bound) : "memory");
    const uint64_t b = uniform64(v);
    limit = b < limit ? b : limit;
}

// ---------------------------------------------------------------------------------------------------
// the scan kernel
// ---------------------------------------------------------------------------------------------------
// PM row of one symbol: W consecutive words in LDS (ds_read_b32 / ds_read_b64 / ds_read_b128)
// (kRowWords: the table's row pitch in Words -- more than W when a narrow state reads the low words of a wider table)
template <class Word, int W, int kRowWords = W>
__device__ __forceinline__ void load_pm(Word (&dst)[W], const Word* lds_pm, uint32_t ch)
{
    const Word* row = lds_pm + ch * kRowWords;
#pragma unroll
    for (int w = 0; w < W; ++w) dst[w] = row[w];
}

// 16 columns in groups of kGroup symbols.  The LDS reads of group g+1 are issued BEFORE the recurrence of group g
// (pinned with sched_barrier, otherwise the scheduler sinks them back next to their first use), so their latency
// -- including the 2-4 way bank conflicts of 64 random slots -- hides behind the VALU work of the current group.
template <class State, int J0 = 0, int J1 = kChunk, int kRowWords = State::kWords>
__device__ __forceinline__ void process_chunk_full(State& st, const typename State::Word* lds_pm, const uint4& c)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    constexpr int kGroup = W == 1 ? 4 : (W == 2 ? 2 : 1);
    constexpr int kGroups = (J1 - J0 + kGroup - 1) / kGroup;  // the last group may be partial (cutoff scans stop a chunk at column 4 or 6)
    const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
    Word cur[kGroup][W], nxt[kGroup][W];
#pragma unroll
    for (int j = 0; j < kGroup; ++j)
        if (J0 + j < J1) load_pm<Word, W, kRowWords>(cur[j], lds_pm, (dw[(J0 + j) / 4] >> (8 * ((J0 + j) % 4))) & 0xFFu);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        if (g + 1 < kGroups) {
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const int n = J0 + (g + 1) * kGroup + j;
                if (n < J1) load_pm<Word, W, kRowWords>(nxt[j], lds_pm, (dw[n / 4] >> (8 * (n % 4))) & 0xFFu);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            if (J0 + g * kGroup + j < J1) st.step(cur[j]);
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
#pragma unroll
            for (int w = 0; w < W; ++w) cur[j][w] = nxt[j][w];
    }
}

// A tile's last, partial chunk: `rem` (< 16, wavefront-uniform) columns.  Single-word states run the full chunk's code shape --
// table rows gathered a group of 4 symbols ahead at COMPILE-TIME byte positions (the bytes behind a candidate's end are zero
// padding, i.e. valid rows) -- and only the recurrence step is guarded by the column count.  (The first version walked the chunk
// with a running byte shift and one exposed LDS round trip per column: three v_alignbit_b32 + a shift per column, all half-rate
// on gfx950.)  Multi-word states keep the compact loop: sixteen more copies of their column would not pay.
template <class State>
__device__ __forceinline__ void process_chunk_tail(State& st, const typename State::Word* lds_pm, uint4 c, uint32_t rem)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    if constexpr (W == 1) {
        constexpr int kGroup = 4, kGroups = kChunk / kGroup;
        const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
        Word cur[kGroup][W], nxt[kGroup][W];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) load_pm<Word, W>(cur[j], lds_pm, (dw[0] >> (8 * j)) & 0xFFu);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            if ((uint32_t)(g * kGroup) >= rem) break;  // wavefront-uniform
            if (g + 1 < kGroups) {
#pragma unroll
                for (int j = 0; j < kGroup; ++j) load_pm<Word, W>(nxt[j], lds_pm, (dw[g + 1] >> (8 * j)) & 0xFFu);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
                if ((uint32_t)(g * kGroup + j) < rem) st.step(cur[j]);
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
#pragma unroll
                for (int w = 0; w < W; ++w) cur[j][w] = nxt[j][w];
        }
    } else {
        for (uint32_t j = 0; j < rem; ++j) {  // rem is wavefront-uniform (tile length)
            Word x[W];
            load_pm<Word, W>(x, lds_pm, c.x & 0xFFu);
            st.step(x);
            c.x = __builtin_amdgcn_alignbit(c.y, c.x, 8);
            c.y = __builtin_amdgcn_alignbit(c.z, c.y, 8);
            c.z = __builtin_amdgcn_alignbit(c.w, c.z, 8);
            c.w >>= 8;
        }
    }
}

// position of the k-th (0-based) set bit of m (m has more than k set bits)
__device__ __forceinline__ uint32_t nth_set_bit(uint64_t m, uint32_t k)
{
    uint32_t pos = 0;
    uint32_t lo = (uint32_t)m, c = (uint32_t)__popc(lo);
    uint32_t w = lo;
    if (k >= c) {
        k -= c;
        pos = 32;
        w = (uint32_t)(m >> 32);
    }
#pragma unroll
    for (uint32_t half = 16; half >= 1; half >>= 1) {
        const uint32_t part = w & ((1u << half) - 1u);
        c = (uint32_t)__popc(part);
        if (k >= c) {
            k -= c;
            pos += half;
            w >>= half;
        } else {
            w = part;
        }
    }
    return pos;
}

// A dense tile of listed lanes (rf_scan.hip lane_list_pack_kernel: 16-byte entries (tile, lane mask lo, hi, listed lanes in front), first[j] = the entry that holds lane
// 64 j): which (tile, lane) does wavefront lane `lane` of dense tile j take?  The 64 entries from first[j] on hold all 64 (an entry holds at least one): one coalesced
// load, the lane's own entry by a binary search across the LANES (the running sums ascend with the lane), its candidate as the n-th set bit of that entry's mask.
// Idle lanes of the last dense tile shadow the last listed lane (defined bytes; `have` false).
struct DenseLane {
    uint32_t tile, lane_in_tile;
    bool have;
};
__device__ __forceinline__ DenseLane dense_lane_source(const uint4* __restrict__ list, uint32_t entries, uint32_t total, const uint32_t* __restrict__ first_of, uint32_t j,
                                                       uint32_t lane)
{
    DenseLane s;
    const uint32_t g = j * kWave + lane;
    s.have = g < total;
    const uint32_t gg = s.have ? g : total - 1;
    const uint32_t e0 = uniform(first_of[j]);
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
    const uint32_t mlo = (uint32_t)__shfl((int)ent.y, (int)lo, kWave), mhi = (uint32_t)__shfl((int)ent.z, (int)lo, kWave), w = (uint32_t)__shfl((int)ent.w, (int)lo, kWave);
    s.tile = (uint32_t)__shfl((int)ent.x, (int)lo, kWave);
    s.lane_in_tile = nth_set_bit(((uint64_t)mhi << 32) | mlo, gg - w);
    return s;
}

struct TileView {
    const uint4* src;  // wavefront-uniform base of the tile payload
    uint32_t len, slot0;
};
template <bool kUniform>
__device__ __forceinline__ TileView load_tile(const ScanParams& p, uint32_t t)
{
    TileView v;
    if (!kUniform) {
        // t is wavefront-uniform and the descriptors are read-only for the whole launch: read them through
        // the constant address space so they become scalar s_load_dwordx4 (no VGPRs, no vmcnt traffic)
        typedef const __attribute__((address_space(4))) uint32_t* cptr;
        cptr td = (cptr)(uintptr_t)(p.tiles + t);
        const uint32_t off_lo = td[0], off_hi = td[1];
        v.len = td[2];
        v.slot0 = td[3];
        v.src = reinterpret_cast<const uint4*>(p.data + (((uint64_t)off_hi << 32) | off_lo));
    } else {  // single-length corpus: tile t is at t * tile_bytes, no descriptor traffic at all
        v.len = p.uniform_len;
        v.slot0 = t * kWave;
        v.src = reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes);
    }
    return v;
}


// ScanParams::xcd_deal: the workgroup's place in the deal of tiles.  Workgroups go to the 8 XCDs round-robin, so workgroup w
// takes the place (w % 8) * (grid / 8) + w / 8 and consecutive tiles are walked by workgroups of one XCD, at about the same time:
// with the tiles ordered by origin (rf_api.hip tiles_by_origin) their result stores meet in that XCD's L2.
__device__ __forceinline__ uint32_t dealt_workgroup(const ScanParams& p)
{
    return (p.xcd_deal && !(gridDim.x & 7)) ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
}

}  // namespace rf
