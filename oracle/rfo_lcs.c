/*
 * rfo_lcs.c -- CPU ORACLE (test infrastructure only): restatement of src/distance/lcs_seq.rs and the
 * kernel-level part of src/distance/indel.rs (v0.5.0) for u8 elements.  See rfo_common.h.
 */
#include "rfo_common.h"

/* src/distance/lcs_seq.rs:113-133 LCS_SEQ_MBLEVEN2018_MATRIX (0x1 = DELETE, 0x2 = INSERT) */
static const uint8_t LCS_MBLEVEN[14][6] = {
    {0x00, 0x00, 0x00, 0x00, 0x00, 0x00}, /* max edit distance 1, len_diff 0 (does not occur) */
    {0x01, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 1 */
    {0x09, 0x06, 0x00, 0x00, 0x00, 0x00}, /* max edit distance 2, len_diff 0 */
    {0x01, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 1 */
    {0x05, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 2 */
    {0x09, 0x06, 0x00, 0x00, 0x00, 0x00}, /* max edit distance 3, len_diff 0 */
    {0x25, 0x19, 0x16, 0x00, 0x00, 0x00}, /*                      len_diff 1 */
    {0x05, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 2 */
    {0x15, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 3 */
    {0x96, 0x66, 0x5A, 0x99, 0x69, 0xA5}, /* max edit distance 4, len_diff 0 */
    {0x25, 0x19, 0x16, 0x00, 0x00, 0x00}, /*                      len_diff 1 */
    {0x65, 0x56, 0x95, 0x59, 0x00, 0x00}, /*                      len_diff 2 */
    {0x15, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 3 */
    {0x55, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 4 */
};

/* lcs_seq.rs:135-197 mbleven2018 */
static size_t lcs_mbleven2018(rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    if (s1.len < s2.len) return lcs_mbleven2018(s2, s1, score_cutoff);

    size_t len_diff = s1.len - s2.len;
    size_t max_misses = s1.len + s2.len - 2 * score_cutoff;
    size_t ops_index = (max_misses + max_misses * max_misses) / 2 + len_diff - 1;
    const uint8_t *possible_ops = LCS_MBLEVEN[ops_index];
    size_t max_len = 0;

    for (int k = 0; k < 6; ++k) {
        uint8_t ops = possible_ops[k];
        size_t i1 = 0, i2 = 0, cur_len = 0;
        if (ops == 0) break;

        while (i1 < s1.len && i2 < s2.len) {
            if (s1.p[i1] == s2.p[i2]) {
                cur_len += 1;
                ++i1;
                ++i2;
            } else {
                if (ops == 0) break;
                if (ops & 1)
                    ++i1;
                else if (ops & 2)
                    ++i2;
                ops >>= 2;
            }
        }
        max_len = rfo_max(max_len, cur_len);
    }
    return max_len;
}

/* lcs_seq.rs:199-261 lcs_unroll<N, 0>: the manual 3-way unrolling only orders the per-word calls
 * 0..N-1, so a plain loop over words is the same computation. */
static size_t lcs_unroll(const rfo_pm *pm, size_t n_words, rfo_str s2, size_t score_cutoff)
{
    uint64_t s[8];
    for (size_t w = 0; w < n_words; ++w) s[w] = ~(uint64_t)0;

    for (size_t i = 0; i < s2.len; ++i) {
        uint8_t ch2 = s2.p[i];
        uint64_t carry = 0;
        for (size_t word = 0; word < n_words; ++word) {
            uint64_t matches = rfo_pm_get(pm, word, ch2);
            uint64_t u = s[word] & matches;
            /* carrying_add (intrinsics.rs:22-26) */
            uint64_t a = s[word] + u;
            uint64_t b = a < s[word];
            uint64_t x = a + carry;
            uint64_t d = x < a;
            carry = b | d;
            s[word] = x | (s[word] - u);
        }
    }
    size_t sim = 0;
    for (size_t w = 0; w < n_words; ++w) sim += (size_t)rfo_popcount64(~s[w]);
    return sim >= score_cutoff ? sim : 0;
}

/* instrumentation (tests only): how often the last lcs_blockwise call on this thread moved its band's last block with a row index that is a multiple of 64 --
 * the precondition of quirk Q8 below (rfo_last_lcs_q8_edges) */
__thread unsigned rfo_q8_edges = 0;

/* lcs_seq.rs:267-341 lcs_blockwise<0> */
static size_t lcs_blockwise(const rfo_pm *pm, size_t len1, rfo_str s2, size_t score_cutoff)
{
    const size_t word_size = 64;
    size_t len2 = s2.len;
    size_t words = pm->block_count;
    uint64_t *s = (uint64_t *)malloc((words ? words : 1) * sizeof(uint64_t));
    for (size_t w = 0; w < words; ++w) s[w] = ~(uint64_t)0;

    size_t band_width_left = len1 - score_cutoff;
    size_t band_width_right = len2 - score_cutoff;

    size_t first_block = 0;
    size_t last_block = rfo_min(words, rfo_ceil_div(band_width_left + 1, word_size));
    rfo_q8_edges = 0;

    for (size_t row = 0; row < len2; ++row) {
        uint8_t ch2 = s2.p[row];
        uint64_t carry = 0;
        for (size_t word = first_block; word < last_block; ++word) {
            uint64_t matches = rfo_pm_get(pm, word, ch2);
            uint64_t u = s[word] & matches;
            uint64_t a = s[word] + u;
            uint64_t b = a < s[word];
            uint64_t x = a + carry;
            uint64_t d = x < a;
            carry = b | d;
            s[word] = x | (s[word] - u);
        }
        if (row > band_width_right) first_block = (row - band_width_right) / word_size;
        /* (quirk Q8: the next row needs block (row + 1 + band_width_left) / 64 + 1; the reference's ceil_div is one short when that
           index is a multiple of 64 -- restated as it stands, tests/test_oracle_vs_textbook.py) */
        if (row + 1 + band_width_left <= len1) {
            last_block = rfo_ceil_div(row + 1 + band_width_left, word_size);
            /* (instrumentation only: pattern bit row + 1 + band_width_left lives in block (row + 1 + band_width_left) / 64 = last_block when the index
               is a multiple of 64 -- one past what the next row will walk) */
            if ((row + 1 + band_width_left) % word_size == 0 && last_block < words) ++rfo_q8_edges;
        }
    }

    size_t sim = 0;
    for (size_t w = 0; w < words; ++w) sim += (size_t)rfo_popcount64(~s[w]);
    free(s);
    return sim >= score_cutoff ? sim : 0;
}

/* lcs_seq.rs:343-409 longest_common_subsequence_with_pm */
static size_t longest_common_subsequence_with_pm(const rfo_pm *pm, size_t len1, rfo_str s2, size_t score_cutoff)
{
    const size_t word_size = 64;
    size_t words = pm->block_count;
    size_t band_width_left = len1 - score_cutoff;
    size_t band_width_right = s2.len - score_cutoff;
    size_t full_band = band_width_left + 1 + band_width_right;
    size_t full_band_words = rfo_min(words, full_band / word_size + 2);

    if (full_band_words < words) return lcs_blockwise(pm, len1, s2, score_cutoff);

    size_t n = rfo_ceil_div(len1, word_size);
    if (n == 0) return 0;
    if (n <= 8) return lcs_unroll(pm, n, s2, score_cutoff);
    return lcs_blockwise(pm, len1, s2, score_cutoff);
}

/* lcs_seq.rs:411-437 longest_common_subsequence_without_pm */
static size_t longest_common_subsequence_without_pm(rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    if (s1.len == 0) return 0;
    rfo_pm pm; /* PatternMatchVector (<=64) and BlockPatternMatchVector agree for u8 */
    rfo_pm_init(&pm, s1.p, s1.len);
    size_t r = longest_common_subsequence_with_pm(&pm, s1.len, s2, score_cutoff);
    rfo_pm_free(&pm);
    return r;
}

/* lcs_seq.rs:439-486 similarity_with_pm */
size_t rfo_lcs_similarity_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    size_t len1 = s1.len, len2 = s2.len;
    rfo_q8_edges = 0; /* (instrumentation: a call that never reaches lcs_blockwise reports none) */
    if (score_cutoff > len1 || score_cutoff > len2) return 0;

    size_t max_misses = len1 + len2 - 2 * score_cutoff;
    if (max_misses == 0 || (max_misses == 1 && len1 == len2)) return rfo_str_eq(s1, s2) ? len1 : 0;
    if (max_misses < rfo_abs_diff(len1, len2)) return 0;

    if (max_misses >= 5) return longest_common_subsequence_with_pm(pm, len1, s2, score_cutoff);

    rfo_affix affix = rfo_remove_common_affix(s1, s2);
    size_t lcs_sim = affix.prefix_len + affix.suffix_len;
    if (affix.s1.len != 0 && affix.s2.len != 0) {
        size_t adjusted_cutoff = score_cutoff >= lcs_sim ? score_cutoff - lcs_sim : 0;
        lcs_sim += lcs_mbleven2018(affix.s1, affix.s2, adjusted_cutoff);
    }
    return lcs_sim;
}

/* lcs_seq.rs:488-544 similarity_without_pm */
size_t rfo_lcs_similarity_without_pm(rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    if (s1.len < s2.len) return rfo_lcs_similarity_without_pm(s2, s1, score_cutoff);
    size_t len1 = s1.len, len2 = s2.len;
    if (score_cutoff > len1 || score_cutoff > len2) return 0;

    size_t max_misses = len1 + len2 - 2 * score_cutoff;
    if (max_misses == 0 || (max_misses == 1 && len1 == len2)) return rfo_str_eq(s1, s2) ? len1 : 0;
    if (max_misses < rfo_abs_diff(len1, len2)) return 0;

    rfo_affix affix = rfo_remove_common_affix(s1, s2);
    size_t lcs_sim = affix.prefix_len + affix.suffix_len;
    if (affix.s1.len != 0 && affix.s2.len != 0) {
        size_t adjusted_cutoff = score_cutoff >= lcs_sim ? score_cutoff - lcs_sim : 0;
        if (max_misses < 5)
            lcs_sim += lcs_mbleven2018(affix.s1, affix.s2, adjusted_cutoff);
        else
            lcs_sim += longest_common_subsequence_without_pm(affix.s1, affix.s2, adjusted_cutoff);
    }
    return lcs_sim;
}

/* src/distance/indel.rs:287-310 distance_with_pm */
size_t rfo_indel_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    size_t maximum = s1.len + s2.len;
    size_t lcs_cutoff = maximum / 2 >= score_cutoff ? maximum / 2 - score_cutoff : 0;
    size_t lcs_sim = rfo_lcs_similarity_with_pm(pm, s1, s2, lcs_cutoff);
    return maximum - 2 * lcs_sim;
}
