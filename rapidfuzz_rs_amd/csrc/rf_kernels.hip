// rf_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the one-vs-many scan.
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
#include <algorithm>
#include <cstdlib>

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
template <int W>
struct LevState {
    using Word = uint64_t;
    static constexpr int kWords = W;
    uint64_t vp[W], vn[W];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            vp[w] = ~0ull;  // levenshtein.rs:454-455
            vn[w] = 0;
        }
    }
    __device__ __forceinline__ void step(const uint64_t (&pm_row)[W])
    {
        uint32_t hp_c = 1, hn_c = 0;  // levenshtein.rs:824-825
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint64_t x = pm_row[w];
            if (w > 0) x |= hn_c;                            // :847
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);    // (sum ^ vp) | x
            const uint64_t d0 = e | n;                       // :848
            const uint64_t hn = e & p;                       // == d0 & vp because vp & vn == 0   (:852)
            const uint64_t hp = lut3<T_OR_NOR>(n, d0, p);    // vn | ~(d0 | vp)                  (:851)
            const uint64_t hps = w == 0 ? shl1_const<1>(hp) : shl1_var(hp, hp_c);  // :865-866
            const uint64_t hns = w == 0 ? shl1_const<0>(hn) : shl1_var(hn, hn_c);
            if (w + 1 < W) {                                 // :857-858
                hp_c = (uint32_t)(hp >> 63);
                hn_c = (uint32_t)(hn >> 63);
            }
            vn[w] = hps & d0;                                // :869
            vp[w] = lut3<T_OR_NOR>(hns, hps, d0);            // hn | ~(d0 | hp)                  (:868)
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
            const uint64_t d0 = e | n;
            const uint64_t hn = e & p;
            const uint64_t hp = lut3<T_OR_NOR>(n, d0, p);
            const uint64_t hps = shl1_var(hp, hp_c);
            const uint64_t hns = shl1_var(hn, hn_c);
            hp_c = (uint32_t)(hp >> 63);
            hn_c = (uint32_t)(hn >> 63);
            vn[w] = hps & d0;
            vp[w] = lut3<T_OR_NOR>(hns, hps, d0);
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
    // as result().  True = this lane can no longer end at or below `raw_cutoff`.
    // A second bound comes from the diagonal through (len1, len2): values never decrease along a diagonal of the
    // Levenshtein matrix, so D[len1][len2] >= D[j + len1 - len2][j] -- the same popcount identity with a shorter row
    // mask.  For equal lengths that is the distance between the two j-prefixes, which for unrelated strings grows by
    // almost 1 per column: nearly every wavefront of a random corpus is past a small cutoff after 8 columns.
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ bool hopeless(uint32_t len1, uint32_t j, uint32_t len2, uint32_t raw_cutoff) const
    {
        const int32_t last_row = (int32_t)result(len1, j) - (int32_t)(len2 - j);
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;  // <= len1 because j <= len2
        const int32_t diag = i > 0 ? (int32_t)result((uint32_t)i, j) : 0;
        return max(last_row, diag) > (int32_t)raw_cutoff;
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
            const uint64_t t = ~d0[w] & pm_j;                              // candidates for a transposition
            const uint64_t tr = (w == 0 ? shl1_const<0>(t) : shl1_var(t, tr_c)) & pm_old[w];  // osa.rs:180
            if (w + 1 < W) tr_c = (uint32_t)(t >> 63);                      // ((~d0_last) & pm_last) >> 63 for the next word
            uint64_t x = pm_j;
            if (w > 0) x |= hn_c;                                           // osa.rs:182
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);
            const uint64_t d = e | n | tr;                                  // osa.rs:183
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
    // the vertical-delta identity and the last-row bound hold for the OSA matrix as well (unit steps)
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ bool hopeless(uint32_t len1, uint32_t j, uint32_t len2, uint32_t raw_cutoff) const
    {
        return (int32_t)result(len1, j) - (int32_t)(len2 - j) > (int32_t)raw_cutoff;
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
    static constexpr bool kCanPrune = false;
    __device__ __forceinline__ bool hopeless(uint32_t, uint32_t, uint32_t, uint32_t) const { return false; }
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
        const uint32_t d0 = e | vn;
        const uint32_t hn = e & vp;
        const uint32_t hp = lut3w<T_OR_NOR>(vn, d0, vp);
        const uint32_t hps = (hp << 1) | 1u;
        const uint32_t hns = hn << 1;
        vn = hps & d0;
        vp = lut3w<T_OR_NOR>(hns, hps, d0);
    }
    static constexpr bool kCanPrune = true;
    __device__ __forceinline__ bool hopeless(uint32_t len1, uint32_t j, uint32_t len2, uint32_t raw_cutoff) const
    {
        const int32_t last_row = (int32_t)result(len1, j) - (int32_t)(len2 - j);
        const int32_t i = (int32_t)j + (int32_t)len1 - (int32_t)len2;  // the diagonal bound, see LevState::hopeless
        const int32_t diag = i > 0 ? (int32_t)result((uint32_t)i, j) : 0;
        return max(last_row, diag) > (int32_t)raw_cutoff;
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
    static constexpr bool kCanPrune = false;
    __device__ __forceinline__ bool hopeless(uint32_t, uint32_t, uint32_t, uint32_t) const { return false; }
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
// so the host folds metric, weights and op into a few coefficients (rf_api.hip plan()) and the kernels do one
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
        const uint32_t up_lo = __shfl_up((uint32_t)key, 1), up_hi = __shfl_up((uint32_t)(key >> 32), 1);
        const uint64_t up = ((uint64_t)up_hi << 32) | up_lo;
        key = lane > pos ? up : (lane == pos ? x : key);
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

// A workgroup's merged list leaves the kernel through ONE launch-wide candidate buffer: only keys at or below the
// pruning bound can still be in the answer (the bound is some full list's worst key, so the true k-th best is <= it),
// and those are appended behind an atomic counter.  On a corpus in no particular order a few dozen keys survive per
// launch, so the final selection (topk_final_kernel) is one small workgroup instead of staged reductions over
// grid x k keys.  Worst case (scores improving along the corpus) everything is appended: capacity is grid x k.
__device__ __forceinline__ void topk_publish(const ScanParams& p, const WaveTopK& best, uint32_t lane)
{
    const uint64_t bound = __hip_atomic_load(p.topk_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool keep = lane < p.topk_k && best.key != ~0ull && best.key <= bound;
    const uint64_t m = __ballot(keep);
    if (m == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(p.topk_count, (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (keep) p.topk_cand[base + __popcll(m & ((1ull << lane) - 1))] = best.key;
}

// Launch-wide pruning bound: once ANY wavefront holds k keys, its worst key bounds the global k-th best from above
// (and topk_core() seeds it with the k-th best of a sample before the scan starts).  A wavefront whose list just
// changed publishes its worst key with a 64-bit atomic min.
__device__ __forceinline__ void topk_list_changed(const ScanParams& p, const WaveTopK& best, uint32_t lane, uint64_t& limit)
{
    const uint64_t w = best.worst(p.topk_k);
    if (w < limit) {  // limit <= the last bound this wavefront saw: only then can the global bound improve
        if (lane == 0) atomicMin((unsigned long long*)p.topk_bound, (unsigned long long)w);
        limit = w;
    }
}
// `bound_inflight` is the raw result of a load issued at the end of the previous tile: it is folded into the
// (scalar) limit here, one tile later, and the next fetch is issued -- stale by a tile (merely conservative), never
// waited for.
__device__ __forceinline__ void topk_refresh_bound(const ScanParams& p, uint64_t& bound_inflight, uint64_t& limit)
{
    const uint64_t b = uniform64(bound_inflight);
    limit = b < limit ? b : limit;
    bound_inflight = __hip_atomic_load(p.topk_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// the scan kernel
// ---------------------------------------------------------------------------------------------------
// PM row of one symbol: W consecutive words in LDS (ds_read_b32 / ds_read_b64 / ds_read_b128)
template <class Word, int W>
__device__ __forceinline__ void load_pm(Word (&dst)[W], const Word* lds_pm, uint32_t ch)
{
    const Word* row = lds_pm + ch * W;
#pragma unroll
    for (int w = 0; w < W; ++w) dst[w] = row[w];
}

// 16 columns in groups of kGroup symbols.  The LDS reads of group g+1 are issued BEFORE the recurrence of group g
// (pinned with sched_barrier, otherwise the scheduler sinks them back next to their first use), so their latency
// -- including the 2-4 way bank conflicts of 64 random slots -- hides behind the VALU work of the current group.
template <class State, int J0 = 0, int J1 = kChunk>
__device__ __forceinline__ void process_chunk_full(State& st, const typename State::Word* lds_pm, const uint4& c)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    constexpr int kGroup = W == 1 ? 4 : (W == 2 ? 2 : 1);
    constexpr int kGroups = (J1 - J0) / kGroup;
    const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
    Word cur[kGroup][W], nxt[kGroup][W];
#pragma unroll
    for (int j = 0; j < kGroup; ++j) load_pm<Word, W>(cur[j], lds_pm, (dw[(J0 + j) / 4] >> (8 * ((J0 + j) % 4))) & 0xFFu);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        if (g + 1 < kGroups) {
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const int n = J0 + (g + 1) * kGroup + j;
                load_pm<Word, W>(nxt[j], lds_pm, (dw[n / 4] >> (8 * (n % 4))) & 0xFFu);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < kGroup; ++j) st.step(cur[j]);
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
#pragma unroll
            for (int w = 0; w < W; ++w) cur[j][w] = nxt[j][w];
    }
}

template <class State>
__device__ __forceinline__ void process_chunk_tail(State& st, const typename State::Word* lds_pm, uint4 c, uint32_t rem)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
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

template <class State, bool kUniform>
__device__ __forceinline__ void scan_body(const ScanParams& p, typename State::Word* lds_pm, uint64_t (*lds_topk)[kWave])
{
    constexpr int W = State::kWords;
    // stage the PM table; the 32-bit states keep the low half of each (single-word) entry
    // (the corpus stores renamed symbols sigma(c), see rf_corpus: row c of the table goes to row sigma(c))
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock)
        lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (typename State::Word)p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    const bool early = State::kCanPrune && p.early != 0;
    WaveTopK best;
    best.init();
    // offers are filtered by `limit` = min(launch-wide pruning bound as last seen, own list's worst key), a scalar.
    // This body is at its VGPR budget (cutoff state + early-out), so the bound is re-read only every 8th tile and
    // consumed at once instead of riding in registers across a tile like stream_body's.
    uint64_t limit = ~0ull;
    uint32_t tiles_done = 0;

    // Each wavefront walks its tiles as one continuous stream of 16-column chunks.  The load of the NEXT
    // chunk (the next 16 columns of this tile, or the first 16 of the wavefront's next tile) is always issued
    // before the current chunk is processed, so exactly one 1 KiB request per wavefront is in flight and the
    // wait before each chunk is a counted vmcnt(1), never a drain.
    // With a cutoff (`early`) the bet is the opposite: after 16 columns nearly every wavefront of a random
    // corpus is past the cutoff, so the prefetch goes to the NEXT TILE and a surviving wavefront fetches its
    // own next chunk on demand -- a dead tile costs 16 of its 64+ bytes per candidate in HBM traffic.
    uint32_t t = p.tile_begin + (blockIdx.x * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        TileView cur_tile = load_tile<kUniform>(p, t);
        uint4 cur = load_chunk(cur_tile.src + lane);  // the packed buffer carries one chunk of tail padding: always readable

        while (true) {
            const uint32_t t_next = t + stride;
            const bool has_next = t_next < p.tile_end;
            const TileView next_tile = load_tile<kUniform>(p, has_next ? t_next : t);

            const uint32_t len2 = cur_tile.len;
            const uint32_t slot = cur_tile.slot0 + lane;
            uint32_t idx = slot;
            if (!kUniform) idx = p.orig[slot];  // issued early; consumed after the columns

            State st;
            st.init();
            const uint32_t nch = (len2 + kChunk - 1) / kChunk;
            bool dead = false;
            uint4 ahead = make_uint4(0, 0, 0, 0);
            if (early || nch == 0) ahead = load_chunk(next_tile.src + lane);
            for (uint32_t c = 0; c < nch; ++c) {
                uint4 nxt = ahead;
                if (!early) {
                    const uint4* nsrc = (c + 1 < nch) ? cur_tile.src + (size_t)(c + 1) * kWave : next_tile.src;
                    nxt = load_chunk(nsrc + lane);
                }
                const uint32_t cols = len2 - c * kChunk;
                if (cols >= kChunk) {
                    if (early && c == 0) {
                        // first chance to stop: after 8 columns a random candidate is already ~6 edits off
                        process_chunk_full<State, 0, kChunk / 2>(st, lds_pm, cur);
                        if (__ballot(!st.hopeless(p.len1, kChunk / 2, len2, p.raw_cutoff)) == 0) {
                            dead = true;
                            break;
                        }
                        process_chunk_full<State, kChunk / 2, kChunk>(st, lds_pm, cur);
                    } else {
                        process_chunk_full<State>(st, lds_pm, cur);
                    }
                } else {
                    process_chunk_tail<State>(st, lds_pm, cur, cols);
                }
                if (early) {
                    const uint32_t j = min(len2, (c + 1) * kChunk);
                    if (__ballot(!st.hopeless(p.len1, j, len2, p.raw_cutoff)) == 0) {
                        dead = true;  // the whole wavefront is beyond the cutoff: stop reading this tile
                        break;
                    }
                    if (c + 1 < nch) nxt = load_chunk(cur_tile.src + (size_t)(c + 1) * kWave + lane);
                }
                cur = nxt;
            }
            if (dead || nch == 0) cur = ahead;

            const bool valid = kUniform ? slot < p.n : idx != kPad;
            const uint32_t raw = st.result(p.len1, len2);
            if (p.out && valid) {
                if (dead)
                    reinterpret_cast<uint32_t*>(p.out)[idx] = RF_NONE_U32;  // early only runs for u32 distance output
                else
                    emit_usize(p, raw, len2, idx);
            }
            if (topk && !dead) {
                bool keep;
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);
                if ((tiles_done++ & 7u) == 0) {
                    const uint64_t b = uniform64(__hip_atomic_load(p.topk_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    limit = b < limit ? b : limit;
                }
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);
            }

            if (!has_next) break;
            t = t_next;
            cur_tile = next_tile;
        }
    }

    if (topk) {  // 4 wavefront lists -> one list per workgroup -> global; the final merge is topk_merge_kernel
        lds_topk[wave][lane] = best.key;
        __syncthreads();
        if (wave == 0) {
            for (uint32_t w = 1; w < kWavesPerBlock; ++w)
                for (uint32_t j = 0; j < p.topk_k; ++j) {
                    const uint64_t x = lds_topk[w][j];  // wavefront-uniform address: a broadcast read
                    if (x < best.worst(p.topk_k)) best.insert(x, lane);
                }
            topk_publish(p, best, lane);
        }
    }
}

// Two entry points over the same body: the single-word kernels are pinned to 8 wavefronts per SIMD (otherwise the
// scalar state of the tile loop pushes them to 96 SGPRs = 7 resident workgroups per CU); W >= 2 keeps the
// compiler's own register budget (forcing 8 would spill VGPRs).
template <class State, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void scan_kernel(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    scan_body<State, kUniform>(p, lds_pm, lds_topk);
}
template <class State, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void scan_kernel_occ8(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    scan_body<State, kUniform>(p, lds_pm, lds_topk);
}

// ---------------------------------------------------------------------------------------------------
// The no-cutoff scan of the single-word states as a stream: every wavefront sees its tiles as ONE sequence of
// 16-column chunks and keeps kDepth chunk loads (1 KiB each) in flight ahead of the chunk it is working on, across
// tile boundaries.  A FETCH cursor runs kDepth chunks ahead of the PROCESS cursor; both walk (tile, chunk) pairs, and
// a zero-length tile counts as one (unused) chunk so the two stay in lock-step.  Past the wavefront's last tile the
// fetch cursor parks on its last valid chunk (a cached re-read).  Compared with scan_body (which also carries the
// cutoff early-out) this loop has about half the scalar/branch instructions per chunk.
// ---------------------------------------------------------------------------------------------------
template <class State, bool kUniform, int kDepth>
__device__ __forceinline__ void stream_body(const ScanParams& p, typename State::Word* lds_pm, uint64_t (*lds_topk)[kWave])
{
    constexpr int W = State::kWords;
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock)
        lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (typename State::Word)p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    WaveTopK best;
    best.init();
    // offers are filtered by `limit` = min(launch-wide pruning bound as last seen, own list's worst key), a scalar;
    // `bound` is the bound fetch in flight (topk_refresh_bound)
    // (the first value is waited for once: it is the sampled bound, and without it the first tile of EVERY wavefront
    // would insert 64 keys and then hit the one bound word with an atomic -- 32768 serialized device-scope atomics)
    uint64_t bound = topk ? __hip_atomic_load(p.topk_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
    uint64_t limit = uniform64(bound);

    uint32_t t = p.tile_begin + (blockIdx.x * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        // fetch cursor
        uint32_t ft = t, fc = 0;
        TileView fv = load_tile<kUniform>(p, ft);
        uint32_t fn = max(1u, (fv.len + kChunk - 1) / kChunk);
        auto fetch = [&]() {
            const uint4 v = load_chunk(fv.src + (size_t)fc * kWave + lane);
            if (++fc == fn) {
                const uint32_t nt = ft + stride;
                if (nt < p.tile_end) {
                    ft = nt;
                    fv = load_tile<kUniform>(p, ft);
                    fn = max(1u, (fv.len + kChunk - 1) / kChunk);
                    fc = 0;
                } else {
                    fc = fn - 1;
                }
            }
            return v;
        };
        // Ring of kDepth + 1 chunk buffers with STATIC names: the loop below is unrolled over the ring phase, so
        // no buffer is ever copied (a copy of a register with a load in flight would force a wait for that load)
        // and the compiler's vmcnt bookkeeping stays exact.
        uint4 buf[kDepth + 1];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) buf[d] = fetch();

        // process cursor
        TileView cur_tile = load_tile<kUniform>(p, t);
        uint32_t c = 0;
        uint32_t idx = cur_tile.slot0 + lane;
        if (!kUniform) idx = p.orig[idx];
        State st;
        st.init();
        bool done = false;
        auto step = [&](const uint4& use, uint4& refill) {
            refill = fetch();
            const uint32_t len2 = cur_tile.len;
            const uint32_t nch = (len2 + kChunk - 1) / kChunk;
            const uint32_t cols = len2 - c * kChunk;
            if (cols >= kChunk)
                process_chunk_full<State>(st, lds_pm, use);
            else if (nch)
                process_chunk_tail<State>(st, lds_pm, use, cols);
            if (++c < max(1u, nch)) return;

            // tile finished
            const uint32_t slot = cur_tile.slot0 + lane;
            const bool valid = kUniform ? slot < p.n : idx != kPad;
            const uint32_t raw = st.result(p.len1, len2);
            if (p.out && valid) emit_usize(p, raw, len2, idx);
            if (topk) {
                bool keep;
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);
                topk_refresh_bound(p, bound, limit);
            }
            t += stride;
            if (t >= p.tile_end) {
                done = true;
                return;
            }
            cur_tile = load_tile<kUniform>(p, t);
            c = 0;
            idx = cur_tile.slot0 + lane;
            if (!kUniform) idx = p.orig[idx];
            st.init();
        };
        while (!done) {
#pragma unroll
            for (int ph = 0; ph <= kDepth; ++ph) {
                step(buf[ph], buf[(ph + kDepth) % (kDepth + 1)]);
                if (done) break;
            }
        }
    }

    if (topk) {
        lds_topk[wave][lane] = best.key;
        __syncthreads();
        if (wave == 0) {
            for (uint32_t w = 1; w < kWavesPerBlock; ++w)
                for (uint32_t j = 0; j < p.topk_k; ++j) {
                    const uint64_t x = lds_topk[w][j];
                    if (x < best.worst(p.topk_k)) best.insert(x, lane);
                }
            topk_publish(p, best, lane);
        }
    }
}
template <class State, bool kUniform, int kDepth>
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void stream_kernel_occ8(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    stream_body<State, kUniform, kDepth>(p, lds_pm, lds_topk);
}

// ---------------------------------------------------------------------------------------------------
// Many queries x one corpus: Q pattern-match tables sit side by side in LDS and every 16-column
// chunk a wavefront loads from HBM is run through Q recurrences before the next chunk is touched, so the candidate
// bytes are read ONCE for Q queries -- the arithmetic intensity per HBM byte rises Q-fold, which is what the
// HBM-bound kernels (LCS / Indel / the 32-bit forms) need.  out is [Q][n], original candidate order per query.
// ---------------------------------------------------------------------------------------------------
template <class State, int Q, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void scan_multi_kernel(const ScanParams p)
{
    using Word = typename State::Word;
    static_assert(State::kWords == 1, "multi-query kernels are single-word");
    __shared__ Word lds_pm[Q][256];
    for (int i = threadIdx.x; i < Q * 256; i += kWave * kWavesPerBlock) {
        const int q = i / 256, c = i % 256;
        lds_pm[q][p.sigma[c]] = (Word)p.multi_pm[q][c];  // single-word tables: row stride 1
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    uint32_t t = blockIdx.x * kWavesPerBlock + wave;
    if (t >= p.n_tiles) return;
    TileView cur_tile = load_tile<kUniform>(p, t);
    uint4 cur = load_chunk(cur_tile.src + lane);

    while (true) {
        const uint32_t t_next = t + stride;
        const bool has_next = t_next < p.n_tiles;
        const TileView next_tile = load_tile<kUniform>(p, has_next ? t_next : t);
        const uint32_t len2 = cur_tile.len;
        const uint32_t slot = cur_tile.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];

        State st[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) st[q].init();
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        for (uint32_t c = 0; c < nch; ++c) {
            const uint4* nsrc = (c + 1 < nch) ? cur_tile.src + (size_t)(c + 1) * kWave : next_tile.src;
            const uint4 nxt = load_chunk(nsrc + lane);
            const uint32_t cols = len2 - c * kChunk;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (cols >= kChunk)
                    process_chunk_full<State>(st[q], lds_pm[q], cur);
                else
                    process_chunk_tail<State>(st[q], lds_pm[q], cur, cols);
            }
            cur = nxt;
        }
        if (nch == 0) cur = load_chunk(next_tile.src + lane);

        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t raw = st[q].result(p.multi_len1[q], len2);
                char* out = reinterpret_cast<char*>(p.out) + (size_t)q * p.n * (p.out_f64 ? sizeof(double) : sizeof(uint32_t));
                emit_usize(p, raw, len2, idx, out, p.multi_len1[q]);
            }
        }
        if (!has_next) break;
        t = t_next;
        cur_tile = next_tile;
    }
}

template <class State, int Q>
static hipError_t launch_multi_q(const ScanParams& p, hipStream_t stream, int grid)
{
    const dim3 g(grid), b(kWave * kWavesPerBlock);
    if (p.tiles)
        hipLaunchKernelGGL((scan_multi_kernel<State, Q, false>), g, b, 0, stream, p);
    else
        hipLaunchKernelGGL((scan_multi_kernel<State, Q, true>), g, b, 0, stream, p);
    return hipGetLastError();
}
template <class State>
static hipError_t launch_multi_state(const ScanParams& p, hipStream_t stream, int grid)
{
    switch (p.multi_q) {
    case 2: return launch_multi_q<State, 2>(p, stream, grid);
    case 4: return launch_multi_q<State, 4>(p, stream, grid);
    default: return hipErrorInvalidValue;
    }
}
// raw: RAW_LEV or RAW_LCS; all queries single-word; `narrow` = every query <= 32 symbols
hipError_t launch_scan_multi(RawKind raw, bool narrow, const ScanParams& p, hipStream_t stream)
{
    if (p.n_tiles == 0) return hipSuccess;
    const int grid = scan_grid(p.n_tiles);
    if (raw == RAW_LEV) return narrow ? launch_multi_state<Lev32State>(p, stream, grid) : launch_multi_state<LevState<1>>(p, stream, grid);
    if (raw == RAW_LCS) return narrow ? launch_multi_state<Lcs32State>(p, stream, grid) : launch_multi_state<LcsState<1>>(p, stream, grid);
    return hipErrorInvalidValue;
}

// Final selection: `count` candidate keys (device counter, or an immediate for the post-all-gather merge) -> the k
// smallest, ascending, ~0 = empty.  One workgroup of 16 wavefronts: each keeps a sorted k-list over its stripe of
// the candidates (WaveTopK), the lists meet in LDS and wavefront 0 merges them.  Before it exits the kernel re-arms
// the launch-wide state (counter = 0, bound = ~0) so the next top-k call on this scratch needs no memset.
constexpr int kFinalThreads = 256;
__global__ __launch_bounds__(kFinalThreads) void topk_final_kernel(const uint64_t* __restrict__ keys, uint32_t* count_ptr, uint32_t count_imm,
                                                                   uint32_t k, uint64_t* __restrict__ out, uint64_t* bound_ptr,
                                                                   bool bound_from_result)
{
    constexpr uint32_t kWaves = kFinalThreads / kWave;
    __shared__ uint64_t lists[kWaves][kWave];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = uniform(threadIdx.x / kWave);
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    const uint32_t used = min(kWaves, (n + kWave - 1) / kWave);  // wavefronts that see any key at all
    WaveTopK best;
    best.init();
    // kRows 64-key rows per trip, all loaded before the first is offered: the loop is bound by load latency, not work
    constexpr uint32_t kRows = 8;
    uint64_t limit = ~0ull;  // this list's worst key once it is full
    for (uint32_t base = wave * kWave; base < n; base += kRows * kFinalThreads) {
        uint64_t row[kRows];
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint32_t i = base + r * kFinalThreads + lane;
            row[r] = i < n ? keys[i] : ~0ull;
        }
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r)
            if (best.offer(row[r], row[r] != ~0ull, k, lane, limit)) limit = best.worst(k);
    }
    if (used > 1) {
        lists[wave][lane] = best.key;
        __syncthreads();
    }
    if (wave == 0) {
        for (uint32_t w = 1; w < used; ++w)
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t x = lists[w][j];
                if (x >= best.worst(k)) break;  // the lists are sorted: nothing further in this one can enter
                best.insert(x, lane);
            }
        if (lane < k) out[lane] = best.key;
        // re-arm for the next launch on this scratch.  After the SAMPLE pass of a top-k call the bound becomes the
        // sample's k-th best key: the k-th best of a subset bounds the k-th best of the whole corpus from above.
        const uint64_t kth = best.worst(k);
        if (lane == 0) {
            if (count_ptr) *count_ptr = 0;
            if (bound_ptr) *bound_ptr = bound_from_result ? kth : ~0ull;
        }
    }
}

hipError_t launch_topk_final(const uint64_t* keys, uint32_t* count_ptr, uint32_t count_imm, uint32_t k, uint64_t* out, uint64_t* bound_ptr,
                             bool bound_from_result, hipStream_t stream)
{
    hipLaunchKernelGGL(topk_final_kernel, dim3(1), dim3(kFinalThreads), 0, stream, keys, count_ptr, count_imm, k, out, bound_ptr,
                       bound_from_result);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Generalized weights (levenshtein.rs:212-259 generalized_wagner_fischer, reached from _distance_with_pm :1328-1330
// for every weight table that is neither (f,f,f) nor (f,f,>=2f)): the O(len1 * len2) row DP, one candidate per lane.
//   new[i+1] = s1[i] == ch2 ? old[i] : min(new[i] + del, old[i] + sub, old[i+1] + ins)
// The row (len1 + 1 u32 per lane) lives in LDS as [i][lane] -- conflict-free, 256 B per row entry and wavefront -- so
// the workgroup has as many wavefronts as fit (plan(): wf_waves).  The query, renamed like the corpus, is rebuilt
// from the PM table into LDS and read back 4 symbols at a time with a wavefront-uniform (broadcast) address.
// A completeness path (~10 VALU + 2.25 LDS operations per cell); the reference's common-affix stripping and minimum-
// edits test (:286-309) change nothing in the value and are not replayed.
// ---------------------------------------------------------------------------------------------------
template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void wf_kernel(const ScanParams p)
{
    extern __shared__ uint32_t lds_wf[];
    const uint32_t len1 = p.len1;
    const uint32_t qwords = (len1 + 3) / 4 + 1;  // query bytes, 4 per word, one word of slack
    uint8_t* lds_q = reinterpret_cast<uint8_t*>(lds_wf);
    for (uint32_t i = threadIdx.x; i < qwords; i += blockDim.x) lds_wf[i] = 0;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) {  // PM row c, bit i  <=>  s1[i] == c
        const uint8_t stored = p.sigma[c];
        for (uint32_t w = 0; w * 64 < len1; ++w) {
            uint64_t bits = p.pm[(size_t)c * p.words + w];
            while (bits) {
                lds_q[64 * w + (__ffsll((unsigned long long)bits) - 1)] = stored;
                bits &= bits - 1;
            }
        }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t waves = blockDim.x / kWave;
    uint32_t* row = lds_wf + qwords + (size_t)wave * (len1 + 1) * kWave + lane;  // row[i * kWave] = cache[i] of this lane
    const uint32_t ins = p.w_ins, del = p.w_del, sub = p.w_sub;

    for (uint32_t t = blockIdx.x * waves + wave; t < p.n_tiles; t += gridDim.x * waves) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        for (uint32_t i = 0; i <= len1; ++i) row[i * kWave] = i * del;  // :219-221
        uint32_t top = 0;  // cache[0] = j * ins, the same in every lane
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        for (uint32_t c = 0; c < nch; ++c) {
            const uint4 data = load_chunk(tv.src + (size_t)c * kWave + lane);
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            for (uint32_t j = 0; j < cols; ++j) {
                const uint32_t word = j < 4 ? data.x : (j < 8 ? data.y : (j < 12 ? data.z : data.w));
                const uint32_t ch2 = (word >> (8 * (j & 3))) & 0xFFu;
                uint32_t diag = top;  // old[i]
                top += ins;           // :226
                uint32_t left = top;  // new[i]
                for (uint32_t i = 0; i < len1; i += 4) {
                    const uint32_t q4 = lds_wf[i / 4];  // wavefront-uniform address: one broadcast read for 4 symbols
                    const uint32_t lim = min(4u, len1 - i);
                    for (uint32_t k = 0; k < lim; ++k) {
                        const uint32_t up = row[(i + k + 1) * kWave];  // old[i+1]
                        const uint32_t x = ((q4 >> (8 * k)) & 0xFFu) == ch2 ? diag : min(min(left + del, diag + sub), up + ins);
                        row[(i + k + 1) * kWave] = x;
                        diag = up;
                        left = x;
                    }
                }
            }
        }
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
            const uint32_t dist = len1 ? row[len1 * kWave] : top;
            // _maximum, levenshtein.rs:263-277 (not affine in the lengths for a general table)
            const uint32_t max_dist = len1 * del + len2 * ins;
            const uint32_t alt = len1 >= len2 ? len2 * sub + (len1 - len2) * del : len1 * sub + (len2 - len1) * ins;
            TileFin f;
            f.max = min(max_dist, alt);
            f.d0 = 0;
            f.v0 = p.fin_flip ? f.max : 0;  // similarity = maximum - distance (fin_vR = -1), distance = raw (fin_vR = +1)
            emit_fin(p, f, dist, idx, p.out);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Patterns longer than 512 symbols (the reference's hyrroe2003_block / lcs_blockwise territory,
// levenshtein.rs:769-1019, lcs_seq.rs:267-341): the pattern is cut into groups of 8 words (512 rows).  A
// wavefront sweeps the candidate once per group with that group's 16 bit-vectors in registers; the horizontal
// deltas crossing the group boundary (2 bits per column and lane for Levenshtein, the adder carry for LCS) wait in
// a chunk-interleaved HBM scratch strip between sweeps.  PM words come straight from global memory (the table of a
// long pattern does not fit LDS; it is L2-resident).  Throughput path for completeness, not for the roofline.
// ---------------------------------------------------------------------------------------------------
constexpr int kLongGroup = 8;

template <bool kLcs, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void long_kernel(const ScanParams p)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;  // global wavefront id: owns one scratch strip
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t words_pad = p.long_words_pad;
    const uint32_t groups = words_pad / kLongGroup;
    uint32_t* strip = p.long_scratch + (size_t)gw * p.long_chunks_max * kWave;
    __shared__ uint8_t lds_unrename[256];  // stored symbol -> original symbol (the PM table stays in global memory)
    lds_unrename[p.sigma[threadIdx.x & 255]] = (uint8_t)(threadIdx.x & 255);
    __syncthreads();

    for (uint32_t t = gw; t < p.n_tiles; t += stride) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        int32_t acc = 0;

        for (uint32_t g = 0; g < groups; ++g) {
            LevState<kLongGroup> lev;
            LcsState<kLongGroup> lcs;
            if (kLcs)
                lcs.init();
            else
                lev.init();
            for (uint32_t c = 0; c < nch; ++c) {
                uint4 data = tv.src[(size_t)c * kWave + lane];
                // carries entering word 0 of this group for the 16 columns of the chunk:
                // bits 0..15 = hp (or the LCS adder carry), bits 16..31 = hn
                uint32_t cin = g == 0 ? (kLcs ? 0u : 0x0000FFFFu) : strip[(size_t)c * kWave + lane];
                uint32_t cout = 0;
                const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
                for (uint32_t j = 0; j < cols; ++j) {
                    const uint32_t ch = lds_unrename[data.x & 0xFFu];
                    const uint64_t* row = p.pm + (size_t)ch * words_pad + (size_t)g * kLongGroup;
                    uint64_t x[kLongGroup];
#pragma unroll
                    for (int w = 0; w < kLongGroup; ++w) x[w] = row[w];
                    if (kLcs) {
                        uint32_t carry = (cin >> j) & 1u;
                        lcs.step_carry(x, carry);
                        cout |= carry << j;
                    } else {
                        uint32_t hp_c = (cin >> j) & 1u, hn_c = (cin >> (16 + j)) & 1u;
                        lev.step_carry(x, hp_c, hn_c);
                        cout |= (hp_c << j) | (hn_c << (16 + j));
                    }
                    data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                    data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                    data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                    data.w >>= 8;
                }
                if (g + 1 < groups) strip[(size_t)c * kWave + lane] = cout;
            }
            acc += kLcs ? (int32_t)lcs.result(0, 0) : lev.delta_sum(p.len1, g * kLongGroup);
        }
        const uint32_t raw = kLcs ? (uint32_t)acc : (uint32_t)((int32_t)len2 + acc);
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) emit_usize(p, raw, len2, idx);
    }
}

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
    uint64_t p_flag, t_flag, hits;
    uint32_t tacc;  // T bits of the 32 columns currently being processed
};

// The sliding window mask (jaro.rs:168,176,185) is wavefront-uniform; the asm pins its recurrence
// bm = (bm << 1) | (j < bound) to the scalar ALU as two 32-bit halves (hipcc otherwise migrates it to VGPRs next to
// the per-lane flags).  SCC-based: the low bit is free after the shift, so adding the compare's carry sets it.
__device__ __forceinline__ void window_next(uint32_t& lo, uint32_t& hi, uint32_t j, uint32_t bound)
{
    uint32_t tmp;
    asm("s_lshr_b32 %2, %0, 31\n\t"
        "s_lshl_b32 %1, %1, 1\n\t"
        "s_or_b32 %1, %1, %2\n\t"
        "s_lshl_b32 %0, %0, 1\n\t"
        "s_cmp_lt_u32 %3, %4\n\t"
        "s_addc_u32 %0, %0, 0"
        : "+s"(lo), "+s"(hi), "=&s"(tmp)
        : "s"(j), "s"(bound)
        : "scc");
}

template <bool kFull>
__device__ __forceinline__ void jaro_flag_chunk(JaroWordState& st, const uint64_t* lds_pm0, const uint4& c, uint32_t j0, uint32_t cols,
                                                uint32_t bound, uint32_t& bm_lo_io, uint32_t& bm_hi_io)
{
    const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
    uint32_t bm_lo = uniform(bm_lo_io), bm_hi = uniform(bm_hi_io);  // (re)pin to SGPRs for the asm recurrence
    bound = uniform(bound);
    uint64_t cur[4], nxt[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) cur[b] = lds_pm0[(dw[0] >> (8 * b)) & 0xFFu];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) nxt[b] = lds_pm0[(dw[g + 1] >> (8 * b)) & 0xFFu];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t j = j0 + g * 4 + b;
            if (kFull || (uint32_t)(g * 4 + b) < cols) {
                const uint64_t pm_j = lut3<T_AND_ANDN>(cur[b], ((uint64_t)bm_hi << 32) | bm_lo, st.p_flag);  // PM & window & ~P
                const uint64_t below = pm_j - 1;
                st.p_flag = lut3<T_OR_ANDN_B>(st.p_flag, pm_j, below);  // P |= blsi(pm_j)
                st.tacc |= pm_j != 0 ? (1u << (j & 31)) : 0u;           // jaro.rs:174 / :183
                window_next(bm_lo, bm_hi, uniform(j), bound);  // :176 / :185
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) cur[b] = nxt[b];
    }
    bm_lo_io = bm_lo;
    bm_hi_io = bm_hi;
}

template <bool kFull>
__device__ __forceinline__ void jaro_transpose_chunk(JaroWordState& st, const uint64_t* lds_pm0, const uint4& c, uint32_t j0, uint32_t cols)
{
    const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
    const uint32_t thalf = (j0 & 32) ? (uint32_t)(st.t_flag >> 32) : (uint32_t)st.t_flag;
    uint64_t cur[4], nxt[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) cur[b] = lds_pm0[(dw[0] >> (8 * b)) & 0xFFu];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) nxt[b] = lds_pm0[(dw[g + 1] >> (8 * b)) & 0xFFu];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t j = j0 + g * 4 + b;
            if (kFull || (uint32_t)(g * 4 + b) < cols) {
                const uint32_t f32 = (uint32_t)__builtin_amdgcn_sbfe((int)thalf, j & 31, 1);  // all ones iff T bit j
                const uint64_t f = ((uint64_t)f32 << 32) | f32;
                const uint64_t below = st.p_flag - 1;
                const uint64_t m = lut3<T_ANDN_AND>(st.p_flag, below, f);  // lowest remaining pattern flag, if flagged
                st.hits = lut3<T_OR_AND>(st.hits, cur[b], m);              // match iff PM[text char] has that bit
                st.p_flag = lut3<T_AND_ORN>(st.p_flag, below, f);          // consume it, if flagged
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) cur[b] = nxt[b];
    }
}

template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void jaro_word_kernel(const ScanParams p)
{
    const uint32_t W = p.words;  // PM row stride; only block 0 is read on this path (jaro.rs:172, pm.get(0, ..))
    extern __shared__ uint64_t lds_pm0[];  // 256 entries: block 0 of every row
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds_pm0[p.sigma[i]] = p.pm[(size_t)i * W];  // renamed rows
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t q4 = p.query_head;  // first four query bytes, little endian (Winkler prefix)

    for (uint32_t t = p.tile_begin + blockIdx.x * kWavesPerBlock + wave; t < p.tile_end; t += stride) {
        const TileView tv = load_tile<kUniform>(p, t);
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
        uint4 cur = tv.src[lane];
        JaroRaw r;
        r.eq11 = (cur.x & 0xFFu) == (q4 & 0xFFu);
        {  // Winkler prefix: equal leading bytes among the first min(4, len1_orig, len2_orig), jaro_winkler.rs:118-123
            const uint32_t lim = min(4u, min(len1_orig, len2_orig));
            const uint32_t diff = cur.x ^ q4;
            const uint32_t first_diff = diff ? (uint32_t)(__ffs(diff) - 1) / 8 : 4u;
            r.prefix = min(first_diff, lim);
        }

        JaroWordState st;
        st.p_flag = st.t_flag = st.hits = 0;
        st.tacc = 0;
        const uint64_t bm0 = mask_lsb64(bound + 1);
        uint32_t bm_lo = uniform((uint32_t)bm0), bm_hi = uniform((uint32_t)(bm0 >> 32));
        for (uint32_t k = 0; k < nch; ++k) {  // pass 1
            const uint4 nxt = tv.src[(size_t)(k + 1 < nch ? k + 1 : 0) * kWave + lane];  // next chunk, then chunk 0 again
            const uint32_t cols = len2 - k * kChunk;
            if (cols >= (uint32_t)kChunk)
                jaro_flag_chunk<true>(st, lds_pm0, cur, k * kChunk, kChunk, bound, bm_lo, bm_hi);
            else
                jaro_flag_chunk<false>(st, lds_pm0, cur, k * kChunk, cols, bound, bm_lo, bm_hi);
            if ((k & 1) || k + 1 == nch) {  // 32 columns (or the tail) done: bank their T bits
                st.t_flag |= (uint64_t)st.tacc << ((k & 2) ? 32 : 0);
                st.tacc = 0;
            }
            cur = nxt;
        }
        r.common = __popcll(st.p_flag);
        for (uint32_t k = 0; k < nch; ++k) {  // pass 2
            uint4 nxt = cur;
            if (k + 1 < nch) nxt = tv.src[(size_t)(k + 1) * kWave + lane];
            const uint32_t cols = len2 - k * kChunk;
            if (cols >= (uint32_t)kChunk)
                jaro_transpose_chunk<true>(st, lds_pm0, cur, k * kChunk, kChunk);
            else
                jaro_transpose_chunk<false>(st, lds_pm0, cur, k * kChunk, cols);
            cur = nxt;
        }
        r.transpositions = r.common - __popcll(st.hits);

        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
            bool keep;
            const double v = f64_metric_value(p, len2_orig, r, &keep);
            reinterpret_cast<double*>(p.out)[idx] = keep ? v : __longlong_as_double(0x7FF8000000000000ll);
        }
    }
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

    for (uint32_t t = p.tile_begin + blockIdx.x * kWavesPerBlock + wave; t < p.tile_end; t += stride) {
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
        for (uint32_t c = 0; c < nch; ++c) {
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
// dispatch
// ---------------------------------------------------------------------------------------------------
int scan_max_grid()
{
    // 8 workgroups (32 waves) are resident per CU; launching 4x that lets early finishers be replaced and
    // measured +5% over an exactly-resident grid (profiles/grid_sweep_r01.txt).  RF_SCAN_BLOCKS_PER_CU overrides.
    static const int per_cu = [] {
        const char* e = getenv("RF_SCAN_BLOCKS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 32;
    }();
    return 256 * per_cu;
}

template <class State>
static hipError_t launch_state(const ScanParams& p, hipStream_t stream, int grid)
{
    const dim3 g(grid), b(kWave * kWavesPerBlock);
    if constexpr (State::kWords == 1) {
        // no cutoff early-out to serve: the leaner stream loop.  Ring depth 1 measured best (2 and 3 were 1-3% slower
        // on every metric: these kernels are issue-bound, not latency-bound); RF_STREAM=0 selects scan_body for A/B.
        static const bool use_stream = [] { const char* e = getenv("RF_STREAM"); return !e || atoi(e) != 0; }();
        if (!p.early && use_stream) {
            if (p.tiles)
                hipLaunchKernelGGL((stream_kernel_occ8<State, false, 1>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((stream_kernel_occ8<State, true, 1>), g, b, 0, stream, p);
            return hipGetLastError();
        }
        if (p.tiles)
            hipLaunchKernelGGL((scan_kernel_occ8<State, false>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((scan_kernel_occ8<State, true>), g, b, 0, stream, p);
    } else {
        if (p.tiles)
            hipLaunchKernelGGL((scan_kernel<State, false>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((scan_kernel<State, true>), g, b, 0, stream, p);
    }
    return hipGetLastError();
}

template <template <int> class StateT>
static hipError_t launch_words(const ScanParams& p, hipStream_t stream, int grid)
{
    switch (p.words) {
    case 1: return launch_state<StateT<1>>(p, stream, grid);
    case 2: return launch_state<StateT<2>>(p, stream, grid);
    case 3: return launch_state<StateT<3>>(p, stream, grid);
    case 4: return launch_state<StateT<4>>(p, stream, grid);
    case 5: return launch_state<StateT<5>>(p, stream, grid);
    case 6: return launch_state<StateT<6>>(p, stream, grid);
    case 7: return launch_state<StateT<7>>(p, stream, grid);
    case 8: return launch_state<StateT<8>>(p, stream, grid);
    default: return hipErrorInvalidValue;
    }
}

int scan_grid(uint32_t n_tiles)
{
    return (int)std::min<uint32_t>((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (uint32_t)scan_max_grid());
}

hipError_t launch_scan(RawKind raw, const ScanParams& p, hipStream_t stream, int* grid_used)
{
    if (p.n_tiles == 0) return hipSuccess;
    const uint32_t launch_tiles = p.tile_end > p.tile_begin ? (p.tile_end - p.tile_begin + p.tile_step - 1) / p.tile_step : 0;
    const int grid = p.long_words_pad ? (int)p.long_grid : std::max(1, scan_grid(launch_tiles));
    if (grid_used) *grid_used = grid;
    if (p.prefill_none && p.out) {  // candidates outside the cutoff's length window (plan()): None without being read
        const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)p.out, (int)RF_NONE_U32, p.n, stream);
        if (e != hipSuccess) return e;
    }
    if (p.long_words_pad && (raw == RAW_LEV || raw == RAW_LCS)) {
        const dim3 g(grid), b(kWave * kWavesPerBlock);
        if (raw == RAW_LCS) {
            if (p.tiles)
                hipLaunchKernelGGL((long_kernel<true, false>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((long_kernel<true, true>), g, b, 0, stream, p);
        } else {
            if (p.tiles)
                hipLaunchKernelGGL((long_kernel<false, false>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((long_kernel<false, true>), g, b, 0, stream, p);
        }
        return hipGetLastError();
    }
    if (raw == RAW_WF) {
        const size_t lds = ((size_t)(p.len1 + 3) / 4 + 1) * 4 + (size_t)p.wf_waves * (p.len1 + 1) * kWave * 4;
        const dim3 g(std::max(1, scan_grid(p.n_tiles))), b(kWave * p.wf_waves);
        auto k = p.tiles ? wf_kernel<false> : wf_kernel<true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, g, b, lds, stream, p);
        return hipGetLastError();
    }
    switch (raw) {
    case RAW_LEV: return p.len1 <= 32 ? launch_state<Lev32State>(p, stream, grid) : launch_words<LevState>(p, stream, grid);
    case RAW_LCS: return p.len1 <= 32 ? launch_state<Lcs32State>(p, stream, grid) : launch_words<LcsState>(p, stream, grid);
    case RAW_OSA: return launch_words<OsaState>(p, stream, grid);
    case RAW_JARO: {
        // tiles [tile_begin, jaro_split) take the single-word path, [jaro_split, tile_end) the multi-word path
        // (tiles ascend by length, and the single-word condition holds for a length prefix)
        const dim3 b(kWave * kWavesPerBlock);
        ScanParams q = p;
        q.tile_begin = 0;
        q.tile_end = p.jaro_split;
        if (q.tile_end > q.tile_begin) {
            const dim3 g(scan_grid(q.tile_end - q.tile_begin));
            if (p.tiles)
                hipLaunchKernelGGL(jaro_word_kernel<false>, g, b, 256 * sizeof(uint64_t), stream, q);
            else
                hipLaunchKernelGGL(jaro_word_kernel<true>, g, b, 256 * sizeof(uint64_t), stream, q);
        }
        q.tile_begin = p.jaro_split;
        q.tile_end = p.n_tiles;
        if (q.tile_end > q.tile_begin) {
            const dim3 g(scan_grid(q.tile_end - q.tile_begin));
            const size_t lds = ((size_t)256 * p.words + p.words + 1) * sizeof(uint64_t);
            if (p.tiles)
                hipLaunchKernelGGL(jaro_block_kernel<false>, g, b, lds, stream, q);
            else
                hipLaunchKernelGGL(jaro_block_kernel<true>, g, b, lds, stream, q);
        }
        return hipGetLastError();
    }
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rf
