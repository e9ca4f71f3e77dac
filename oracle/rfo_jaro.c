/*
 * rfo_jaro.c -- CPU ORACLE (test infrastructure only): restatement of src/distance/jaro.rs and
 * src/distance/jaro_winkler.rs (v0.5.0) for u8 elements.  See rfo_common.h.
 * Build with -ffp-contract=off: Rust never fuses a*b+c, so neither may we.
 */
#include "rfo_common.h"

/* jaro.rs:106-119 calculate_similarity */
static double calculate_similarity(size_t p_len, size_t t_len, size_t common_chars, size_t transposition)
{
    transposition /= 2;
    double sim = 0.0;
    sim += (double)common_chars / (double)p_len;
    sim += (double)common_chars / (double)t_len;
    sim += ((double)common_chars - (double)transposition) / (double)common_chars;
    return sim / 3.0;
}

/* jaro.rs:122-131 length_filter */
static int length_filter(size_t p_len, size_t t_len, double score_cutoff)
{
    if (t_len == 0 || p_len == 0) return 0;
    double min_len = (double)rfo_min(p_len, t_len);
    double sim = min_len / (double)p_len + min_len / (double)t_len + 1.0;
    sim /= 3.0;
    return sim >= score_cutoff;
}

/* jaro.rs:134-145 common_char_filter */
static int common_char_filter(size_t p_len, size_t t_len, size_t common_chars, double score_cutoff)
{
    if (common_chars == 0) return 0;
    double sim = 0.0;
    sim += (double)common_chars / (double)p_len;
    sim += (double)common_chars / (double)t_len;
    sim += 1.0;
    sim /= 3.0;
    return sim >= score_cutoff;
}

/* jaro.rs:147-190 flag_similar_characters_word */
static void flag_similar_characters_word(const rfo_pm *pm, rfo_str s2, size_t bound, uint64_t *p_flag_out,
                                         uint64_t *t_flag_out)
{
    uint64_t p_flag = 0, t_flag = 0;
    uint64_t bound_mask = rfo_bit_mask_lsb(bound + 1);
    size_t j = 0;
    size_t head = rfo_min(bound, s2.len);
    for (; j < head; ++j) {
        uint64_t pm_j = rfo_pm_get(pm, 0, s2.p[j]) & bound_mask & ~p_flag;
        p_flag |= rfo_blsi(pm_j);
        t_flag |= (uint64_t)(pm_j != 0) << j;
        bound_mask = (bound_mask << 1) | 1;
    }
    for (; j < s2.len; ++j) {
        uint64_t pm_j = rfo_pm_get(pm, 0, s2.p[j]) & bound_mask & ~p_flag;
        p_flag |= rfo_blsi(pm_j);
        t_flag |= (uint64_t)(pm_j != 0) << j;
        bound_mask <<= 1;
    }
    *p_flag_out = p_flag;
    *t_flag_out = t_flag;
}

typedef struct { /* jaro.rs:99-104 */
    size_t words, empty_words;
    uint64_t last_mask, first_mask;
} search_bound_mask;

/* jaro.rs:192-284 flag_similar_characters_step.  The 4-way unrolled `is_ascii` loop at :233-265 tests
 * words in the same order with the same first-hit rule as the scalar loop at :267-276, so one scalar
 * loop reproduces both. */
static void flag_similar_characters_step(const rfo_pm *pm, uint8_t t_j, uint64_t *p_flag, uint64_t *t_flag,
                                         size_t j, const search_bound_mask *bm)
{
    size_t j_word = j / 64, j_pos = j % 64;
    size_t word = bm->empty_words;
    size_t last_word = word + bm->words;

    if (bm->words == 1) {
        uint64_t pm_j = rfo_pm_get(pm, word, t_j) & bm->last_mask & bm->first_mask & ~p_flag[word];
        p_flag[word] |= rfo_blsi(pm_j);
        t_flag[j_word] |= (uint64_t)(pm_j != 0) << j_pos;
        return;
    }
    if (bm->first_mask != 0) {
        uint64_t pm_j = rfo_pm_get(pm, word, t_j) & bm->first_mask & ~p_flag[word];
        if (pm_j != 0) {
            p_flag[word] |= rfo_blsi(pm_j);
            t_flag[j_word] |= (uint64_t)1 << j_pos;
            return;
        }
        word += 1;
    }
    while (word < last_word - 1) {
        uint64_t pm_j = rfo_pm_get(pm, word, t_j) & ~p_flag[word];
        if (pm_j != 0) {
            p_flag[word] |= rfo_blsi(pm_j);
            t_flag[j_word] |= (uint64_t)1 << j_pos;
            return;
        }
        word += 1;
    }
    if (bm->last_mask != 0) {
        uint64_t pm_j = rfo_pm_get(pm, word, t_j) & bm->last_mask & ~p_flag[word];
        p_flag[word] |= rfo_blsi(pm_j);
        t_flag[j_word] |= (uint64_t)(pm_j != 0) << j_pos;
    }
}

/* jaro.rs:286-337 flag_similar_characters_block */
static void flag_similar_characters_block(const rfo_pm *pm, size_t len1, rfo_str s2, size_t bound, uint64_t *p_flag,
                                          uint64_t *t_flag)
{
    size_t start_range = rfo_min(bound + 1, len1);
    search_bound_mask bm;
    bm.words = 1 + start_range / 64;
    bm.empty_words = 0;
    bm.last_mask = ((uint64_t)1 << (start_range % 64)) - 1;
    bm.first_mask = ~(uint64_t)0;

    for (size_t j = 0; j < s2.len; ++j) {
        flag_similar_characters_step(pm, s2.p[j], p_flag, t_flag, j, &bm);

        if (j + bound + 1 < len1) {
            bm.last_mask = (bm.last_mask << 1) | 1;
            if (j + bound + 2 < len1 && bm.last_mask == ~(uint64_t)0) {
                bm.last_mask = 0;
                bm.words += 1;
            }
        }
        if (j >= bound) {
            bm.first_mask <<= 1;
            if (bm.first_mask == 0) {
                bm.first_mask = ~(uint64_t)0;
                bm.words -= 1;
                bm.empty_words += 1;
            }
        }
    }
}

/* jaro.rs:339-368 count_transpositions_word */
static size_t count_transpositions_word(const rfo_pm *pm, rfo_str s2, uint64_t p_flag, uint64_t t_flag)
{
    size_t transpositions = 0;
    size_t it = 0; /* iterator position of `s2` (nth(k) yields s2[it + k], then it += k + 1) */
    while (t_flag != 0) {
        uint64_t pattern_flag_mask = rfo_blsi(p_flag);
        size_t s2_index = (size_t)rfo_ctz64(t_flag);
        uint8_t ch2 = s2.p[it + s2_index];
        it += s2_index + 1;
        transpositions += (rfo_pm_get(pm, 0, ch2) & pattern_flag_mask) == 0;
        t_flag = (t_flag >> 1) >> s2_index;
        p_flag ^= pattern_flag_mask;
    }
    return transpositions;
}

/* jaro.rs:370-420 count_transpositions_block */
static size_t count_transpositions_block(const rfo_pm *pm, rfo_str s2, const uint64_t *p_flags,
                                         const uint64_t *t_flags, size_t flagged_chars)
{
    size_t text_word = 0, pattern_word = 0;
    uint64_t t_flag = t_flags[text_word];
    uint64_t p_flag = p_flags[pattern_word];
    size_t transpositions = 0;
    size_t s2_pos = 0;
    size_t it = 0;

    while (flagged_chars != 0) {
        while (t_flag == 0) {
            text_word += 1;
            if (s2_pos < 64) it += 64 - s2_pos; /* s2.nth(64 - 1 - s2_pos) */
            t_flag = t_flags[text_word];
            s2_pos = 0;
        }
        while (t_flag != 0) {
            while (p_flag == 0) {
                pattern_word += 1;
                p_flag = p_flags[pattern_word];
            }
            uint64_t pattern_flag_mask = rfo_blsi(p_flag);
            size_t s2_index = (size_t)rfo_ctz64(t_flag);
            uint8_t ch2 = s2.p[it + s2_index];
            it += s2_index + 1;
            s2_pos += s2_index + 1;
            transpositions += (rfo_pm_get(pm, pattern_word, ch2) & pattern_flag_mask) == 0;
            t_flag = (t_flag >> 1) >> s2_index;
            p_flag ^= pattern_flag_mask;
            flagged_chars -= 1;
        }
    }
    return transpositions;
}

static size_t popcount_words(const uint64_t *w, size_t n)
{
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += (size_t)rfo_popcount64(w[i]);
    return c;
}

/* shared tail of jaro.rs:422-514 (without_pm, prefix already stripped) and :516-598 (with_pm):
 * flag + filter + transpositions, given the window-truncated lengths */
static int jaro_core(const rfo_pm *pm, size_t len1, rfo_str s2, size_t bound, size_t len1_orig, size_t len2_orig,
                     double score_cutoff, size_t *common_chars, size_t *transpositions)
{
    size_t len2 = s2.len;
    if (len1 == 0 || len2 == 0) return 1;
    if (len1 <= 64 && len2 <= 64) {
        uint64_t p_flag, t_flag;
        flag_similar_characters_word(pm, s2, bound, &p_flag, &t_flag);
        *common_chars += (size_t)rfo_popcount64(p_flag);
        if (!common_char_filter(len1_orig, len2_orig, *common_chars, score_cutoff)) return 0;
        *transpositions = count_transpositions_word(pm, s2, p_flag, t_flag);
    } else {
        size_t pw = rfo_ceil_div(len1, 64), tw = rfo_ceil_div(len2, 64);
        uint64_t *p_flag = (uint64_t *)calloc(pw + 1, sizeof(uint64_t));
        uint64_t *t_flag = (uint64_t *)calloc(tw + 1, sizeof(uint64_t));
        flag_similar_characters_block(pm, len1, s2, bound, p_flag, t_flag);
        /* FlaggedCharsMultiword::count_common_chars jaro.rs:78-97 */
        size_t flagged_chars = pw < tw ? popcount_words(p_flag, pw) : popcount_words(t_flag, tw);
        *common_chars += flagged_chars;
        int ok = common_char_filter(len1_orig, len2_orig, *common_chars, score_cutoff);
        if (ok) *transpositions = count_transpositions_block(pm, s2, p_flag, t_flag, flagged_chars);
        free(p_flag);
        free(t_flag);
        if (!ok) return 0;
    }
    return 1;
}

/* jaro.rs:516-598 similarity_with_pm */
double rfo_jaro_similarity_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, double score_cutoff)
{
    size_t len1 = s1.len, len2 = s2.len;
    size_t len1_orig = len1, len2_orig = len2;

    if (score_cutoff > 1.0) return 0.0;
    if (len1_orig == 0 && len2_orig == 0) return 1.0;
    if (!length_filter(len1_orig, len2_orig, score_cutoff)) return 0.0;
    if (len1_orig == 1 && len2_orig == 1) return s1.p[0] == s2.p[0] ? 1.0 : 0.0;

    size_t bound;
    if (len2 > len1) {
        bound = len2 / 2 - 1;
        if (len2 > len1 + bound) len2 = len1 + bound;
    } else {
        bound = len1 / 2 - 1;
        if (len1 > len2 + bound) len1 = len2 + bound;
    }
    rfo_str s2_win = {s2.p, len2};

    size_t common_chars = 0, transpositions = 0;
    if (!jaro_core(pm, len1, s2_win, bound, len1_orig, len2_orig, score_cutoff, &common_chars, &transpositions))
        return 0.0;
    return calculate_similarity(len1_orig, len2_orig, common_chars, transpositions);
}

/* jaro.rs:422-514 similarity_without_pm */
double rfo_jaro_similarity_without_pm(rfo_str s1, rfo_str s2, double score_cutoff)
{
    size_t len1 = s1.len, len2 = s2.len;
    size_t len1_orig = len1, len2_orig = len2;

    if (score_cutoff > 1.0) return 0.0;
    if (len1_orig == 0 && len2_orig == 0) return 1.0;
    if (!length_filter(len1_orig, len2_orig, score_cutoff)) return 0.0;
    if (len1_orig == 1 && len2_orig == 1) return s1.p[0] == s2.p[0] ? 1.0 : 0.0;

    size_t bound;
    if (len2 > len1) {
        bound = len2 / 2 - 1;
        if (len2 > len1 + bound) len2 = len1 + bound;
    } else {
        bound = len1 / 2 - 1;
        if (len1 > len2 + bound) len1 = len2 + bound;
    }
    rfo_str s1_win = {s1.p, len1}, s2_win = {s2.p, len2};

    /* common prefix never includes transpositions (:474-479) */
    size_t common_chars = rfo_common_prefix(s1_win, s2_win);
    rfo_str s1_it = {s1_win.p + common_chars, len1 - common_chars};
    rfo_str s2_it = {s2_win.p + common_chars, len2 - common_chars};
    size_t transpositions = 0;

    if (s1_it.len != 0 && s2_it.len != 0) {
        rfo_pm pm;
        rfo_pm_init(&pm, s1_it.p, s1_it.len);
        int ok = jaro_core(&pm, s1_it.len, s2_it, bound, len1_orig, len2_orig, score_cutoff, &common_chars,
                           &transpositions);
        rfo_pm_free(&pm);
        if (!ok) return 0.0;
    }
    return calculate_similarity(len1_orig, len2_orig, common_chars, transpositions);
}

/* jaro_winkler.rs:62-101 (without_pm) and :103-141 (with_pm) share everything but the jaro call */
static size_t jw_prefix(rfo_str s1, rfo_str s2)
{
    size_t n = rfo_min(rfo_min(s1.len, s2.len), 4), prefix = 0;
    while (prefix < n && s1.p[prefix] == s2.p[prefix]) ++prefix;
    return prefix;
}
static double jw_jaro_cutoff(size_t prefix, double prefix_weight, double score_cutoff)
{
    double jaro_score_cutoff = score_cutoff;
    if (jaro_score_cutoff > 0.7) {
        double prefix_sim = (double)prefix * prefix_weight;
        if (prefix_sim >= 1.0) {
            jaro_score_cutoff = 0.7;
        } else {
            double v = (prefix_sim - jaro_score_cutoff) / (prefix_sim - 1.0);
            jaro_score_cutoff = 0.7 > v ? 0.7 : v; /* 0.7_f64.max(v) */
        }
    }
    return jaro_score_cutoff;
}

double rfo_jw_similarity_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, double prefix_weight, double score_cutoff)
{
    size_t prefix = jw_prefix(s1, s2);
    double sim = rfo_jaro_similarity_with_pm(pm, s1, s2, jw_jaro_cutoff(prefix, prefix_weight, score_cutoff));
    if (sim > 0.7) sim += (double)prefix * prefix_weight * (1.0 - sim);
    return sim;
}

double rfo_jw_similarity_without_pm(rfo_str s1, rfo_str s2, double prefix_weight, double score_cutoff)
{
    size_t prefix = jw_prefix(s1, s2);
    double sim = rfo_jaro_similarity_without_pm(s1, s2, jw_jaro_cutoff(prefix, prefix_weight, score_cutoff));
    if (sim > 0.7) sim += (double)prefix * prefix_weight * (1.0 - sim);
    return sim;
}
