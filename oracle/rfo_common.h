/*
 * rfo_common.h -- shared pieces of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * This directory is a plain-C restatement of rapidfuzz-rs v0.5.0's one-vs-many path for `u8`
 * elements.  It exists only to CHECK the HIP kernels (tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py).  Nothing under rapidfuzz_rs_amd/ may include, link or call it.
 *
 * Parity status: the Rust reference cannot be built here (no cargo/rustc), so the oracle is pinned
 * against every known-answer test the reference's own test modules hold for this path
 * (tests/test_oracle_known_answers.py) and against an independent textbook DP.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef RFO_COMMON_H
#define RFO_COMMON_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RFO_USIZE_MAX ((size_t)-1)

/* a borrowed byte string: the restatement of a `DoubleEndedIterator + Clone` over u8 */
typedef struct {
    const uint8_t *p;
    size_t len;
} rfo_str;

/* Option<usize> / Option<f64> (src/common.rs:3-86 passes cutoffs around as Option) */
typedef struct {
    int has;
    size_t v;
} rfo_opt_usize;
typedef struct {
    int has;
    double v;
} rfo_opt_f64;

static inline rfo_opt_usize rfo_some_u(size_t v) { rfo_opt_usize o = {1, v}; return o; }
static inline rfo_opt_usize rfo_none_u(void) { rfo_opt_usize o = {0, 0}; return o; }
static inline rfo_opt_f64 rfo_some_f(double v) { rfo_opt_f64 o = {1, v}; return o; }
static inline rfo_opt_f64 rfo_none_f(void) { rfo_opt_f64 o = {0, 0.0}; return o; }

static inline size_t rfo_min(size_t a, size_t b) { return a < b ? a : b; }
static inline size_t rfo_max(size_t a, size_t b) { return a > b ? a : b; }
static inline size_t rfo_abs_diff(size_t a, size_t b) { return a > b ? a - b : b - a; }

/* ---- src/details/intrinsics.rs:1-45 ---- */
static inline size_t rfo_ceil_div(size_t a, size_t d) { return a / d + (a % d != 0); } /* :1-3 */
static inline uint64_t rfo_shr64(uint64_t a, size_t s) { return s < 64 ? a >> s : 0; }  /* :5-11 */
static inline uint64_t rfo_bit_mask_lsb(size_t n)                                       /* :28-34 */
{
    uint64_t mask = ~(uint64_t)0;
    if (n < 64) mask += (uint64_t)1 << n;
    return mask;
}
static inline uint64_t rfo_blsi(uint64_t v) { return v & (0 - v); } /* :35-37 */
static inline int rfo_popcount64(uint64_t v) { return __builtin_popcountll(v); }
static inline int rfo_ctz64(uint64_t v) { return __builtin_ctzll(v); }

/* ---- BlockPatternMatchVector, byte branch only:
 *      src/details/pattern_match_vector.rs:194-321 + BitMatrix src/details/matrix.rs:3-42.
 *      bits[c * block_count + b] bit (i % 64) set  <=>  query[64*b + i] == c              ---- */
typedef struct {
    size_t block_count;
    uint64_t *bits; /* 256 * block_count, row-major (matrix.rs:32-36) */
} rfo_pm;

static inline int rfo_pm_init(rfo_pm *pm, const uint8_t *s1, size_t len1)
{
    pm->block_count = rfo_ceil_div(len1, 64); /* pattern_match_vector.rs:203-211 */
    size_t n = 256 * (pm->block_count ? pm->block_count : 1);
    pm->bits = (uint64_t *)calloc(n, sizeof(uint64_t));
    if (!pm->bits) return -1;
    uint64_t mask = 1; /* :213-224: mask rotates, block = i / 64 */
    for (size_t i = 0; i < len1; ++i) {
        pm->bits[(size_t)s1[i] * pm->block_count + i / 64] |= mask;
        mask = (mask << 1) | (mask >> 63);
    }
    return 0;
}
static inline void rfo_pm_free(rfo_pm *pm)
{
    free(pm->bits);
    pm->bits = NULL;
}
/* pattern_match_vector.rs:284-316 (u8 => Hash::UNSIGNED(value<=255) => extended_ascii) */
static inline uint64_t rfo_pm_get(const rfo_pm *pm, size_t block, uint8_t ch)
{
    return pm->bits[(size_t)ch * pm->block_count + block];
}

/* ---- src/details/common.rs:39-108 ---- */
static inline size_t rfo_common_prefix(rfo_str a, rfo_str b) /* :39-50 */
{
    size_t n = rfo_min(a.len, b.len), i = 0;
    while (i < n && a.p[i] == b.p[i]) ++i;
    return i;
}
static inline size_t rfo_common_suffix(rfo_str a, rfo_str b) /* :52-64 */
{
    size_t n = rfo_min(a.len, b.len), i = 0;
    while (i < n && a.p[a.len - 1 - i] == b.p[b.len - 1 - i]) ++i;
    return i;
}
typedef struct {
    rfo_str s1, s2;
    size_t prefix_len, suffix_len;
} rfo_affix;
static inline rfo_affix rfo_remove_common_affix(rfo_str s1, rfo_str s2) /* :79-108: suffix first */
{
    rfo_affix r;
    r.suffix_len = rfo_common_suffix(s1, s2);
    s1.len -= r.suffix_len;
    s2.len -= r.suffix_len;
    r.prefix_len = rfo_common_prefix(s1, s2);
    r.s1.p = s1.p + r.prefix_len;
    r.s1.len = s1.len - r.prefix_len;
    r.s2.p = s2.p + r.prefix_len;
    r.s2.len = s2.len - r.prefix_len;
    return r;
}
static inline int rfo_str_eq(rfo_str a, rfo_str b)
{
    return a.len == b.len && (a.len == 0 || memcmp(a.p, b.p, a.len) == 0);
}

/* src/details/common.rs:4-7 */
static inline double rfo_norm_sim_to_norm_dist(double score_cutoff)
{
    double imprecision = 0.00001;
    double v = 1.0 - score_cutoff + imprecision;
    return v < 1.0 ? v : 1.0; /* f64::min */
}

/* ---- weights (src/distance/levenshtein.rs:128-148) ---- */
typedef struct {
    size_t insertion_cost, deletion_cost, substitution_cost;
} rfo_weights;

/* kernels (implemented in rfo_levenshtein.c / rfo_lcs.c / rfo_jaro.c) */
size_t rfo_lev_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, const rfo_weights *w,
                                size_t score_cutoff, size_t score_hint);
size_t rfo_lev_distance_without_pm(rfo_str s1, rfo_str s2, const rfo_weights *w, size_t score_cutoff,
                                   size_t score_hint);
size_t rfo_lev_maximum(size_t len1, size_t len2, const rfo_weights *w);

size_t rfo_lcs_similarity_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, size_t score_cutoff);
size_t rfo_lcs_similarity_without_pm(rfo_str s1, rfo_str s2, size_t score_cutoff);
size_t rfo_indel_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, size_t score_cutoff);

size_t rfo_osa_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2);
size_t rfo_osa_distance_without_pm(rfo_str s1, rfo_str s2);

double rfo_jaro_similarity_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, double score_cutoff);
double rfo_jaro_similarity_without_pm(rfo_str s1, rfo_str s2, double score_cutoff);
double rfo_jw_similarity_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, double prefix_weight,
                                 double score_cutoff);
double rfo_jw_similarity_without_pm(rfo_str s1, rfo_str s2, double prefix_weight, double score_cutoff);

/* which internal kernel the last rfo_lev_* call on this thread ended in (test instrumentation so the
 * path-targeted known-answer tests can assert they really hit mbleven / small band / block) */
enum {
    RFO_PATH_NONE = 0,
    RFO_PATH_EQ,
    RFO_PATH_LENDIFF,
    RFO_PATH_EMPTY,
    RFO_PATH_HYRROE2003,
    RFO_PATH_SMALL_BAND,
    RFO_PATH_BLOCK,
    RFO_PATH_MBLEVEN,
    RFO_PATH_WAGNER_FISCHER,
    RFO_PATH_AFFIX
};
extern __thread int rfo_last_path;

#endif
