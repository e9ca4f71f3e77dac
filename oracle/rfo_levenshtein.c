/*
 * rfo_levenshtein.c -- CPU ORACLE (test infrastructure only): restatement of
 * src/distance/levenshtein.rs (v0.5.0) for u8 elements.  See rfo_common.h for the rules.
 */
#include "rfo_common.h"

__thread int rfo_last_path = RFO_PATH_NONE;

/* src/distance/levenshtein.rs:212-259 generalized_wagner_fischer */
static size_t generalized_wagner_fischer(rfo_str s1, rfo_str s2, const rfo_weights *w)
{
    size_t cache_size = s1.len + 1;
    size_t *cache = (size_t *)malloc(cache_size * sizeof(size_t));
    for (size_t i = 0; i < cache_size; ++i) cache[i] = i * w->deletion_cost;

    for (size_t j = 0; j < s2.len; ++j) {
        uint8_t ch2 = s2.p[j];
        size_t temp = cache[0];
        cache[0] += w->insertion_cost;
        for (size_t i = 0; i < s1.len; ++i) {
            if (s1.p[i] != ch2) {
                temp = rfo_min(cache[i] + w->deletion_cost, temp + w->substitution_cost);
                temp = rfo_min(temp, cache[i + 1] + w->insertion_cost);
            }
            /* mem::swap(cur_cache, &mut temp) with cur_cache = &cache[i + 1] */
            size_t t = cache[i + 1];
            cache[i + 1] = temp;
            temp = t;
        }
    }
    size_t r = cache[cache_size - 1];
    free(cache);
    rfo_last_path = RFO_PATH_WAGNER_FISCHER;
    return r;
}

/* :263-277 _maximum */
size_t rfo_lev_maximum(size_t len1, size_t len2, const rfo_weights *w)
{
    size_t max_dist = len1 * w->deletion_cost + len2 * w->insertion_cost;
    if (len1 >= len2)
        return rfo_min(max_dist, len2 * w->substitution_cost + (len1 - len2) * w->deletion_cost);
    return rfo_min(max_dist, len1 * w->substitution_cost + (len2 - len1) * w->insertion_cost);
}

/* :279-284 _min_distance */
static size_t min_distance(size_t len1, size_t len2, const rfo_weights *w)
{
    ptrdiff_t a = ((ptrdiff_t)len1 - (ptrdiff_t)len2) * (ptrdiff_t)w->deletion_cost;
    ptrdiff_t b = ((ptrdiff_t)len2 - (ptrdiff_t)len1) * (ptrdiff_t)w->insertion_cost;
    return (size_t)(a > b ? a : b);
}

/* :286-309 generalized_distance */
static size_t generalized_distance(rfo_str s1, rfo_str s2, const rfo_weights *w, size_t score_cutoff)
{
    size_t min_edits = min_distance(s1.len, s2.len, w);
    if (min_edits > score_cutoff) {
        rfo_last_path = RFO_PATH_LENDIFF;
        return RFO_USIZE_MAX;
    }
    rfo_affix affix = rfo_remove_common_affix(s1, s2);
    return generalized_wagner_fischer(affix.s1, affix.s2, w);
}

/* :324-337 LEVENSHTEIN_MBLEVEN2018_MATRIX (01 = DELETE, 10 = INSERT, 11 = SUBSTITUTE) */
static const uint8_t LEV_MBLEVEN[9][7] = {
    {0x03, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00}, /* max edit distance 1, len_diff 0 */
    {0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 1 */
    {0x0F, 0x09, 0x06, 0x00, 0x00, 0x00, 0x00}, /* max edit distance 2, len_diff 0 */
    {0x0D, 0x07, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 1 */
    {0x05, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 2 */
    {0x3F, 0x27, 0x2D, 0x39, 0x36, 0x1E, 0x1B}, /* max edit distance 3, len_diff 0 */
    {0x3D, 0x37, 0x1F, 0x25, 0x19, 0x16, 0x00}, /*                      len_diff 1 */
    {0x35, 0x1D, 0x17, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 2 */
    {0x15, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00}, /*                      len_diff 3 */
};

/* :339-427 mbleven2018.  The Rust code walks two iterators with one look-ahead item each
 * (cur1/cur2); i1/i2 below are the indices of those current items, and "iter.count()" at :422 is
 * what is left AFTER the current item. */
static size_t lev_mbleven2018(rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    if (s1.len < s2.len) return lev_mbleven2018(s2, s1, score_cutoff);
    rfo_last_path = RFO_PATH_MBLEVEN;

    size_t len_diff = s1.len - s2.len;
    if (score_cutoff == 1) return (len_diff == 1 || s1.len != 1) ? RFO_USIZE_MAX : 1;

    size_t ops_index = (score_cutoff + score_cutoff * score_cutoff) / 2 + len_diff - 1;
    const uint8_t *possible_ops = LEV_MBLEVEN[ops_index];
    size_t dist = score_cutoff + 1;

    for (int k = 0; k < 7; ++k) {
        uint8_t ops = possible_ops[k];
        size_t i1 = 0, i2 = 0, cur_dist = 0;
        if (ops == 0) break;

        for (;;) {
            int has1 = i1 < s1.len, has2 = i2 < s2.len;
            if (has1 && has2) {
                if (s1.p[i1] == s2.p[i2]) {
                    ++i1;
                    ++i2;
                } else {
                    cur_dist += 1;
                    if (ops == 0) break;
                    if (ops & 1) ++i1;
                    if (ops & 2) ++i2;
                    ops >>= 2;
                }
            } else if (has1) {
                cur_dist += 1;
                ++i1;
            } else if (has2) {
                cur_dist += 1;
                ++i2;
            } else {
                break;
            }
        }
        /* :422 cur_dist += iter_s1.count() + iter_s2.count() */
        cur_dist += (i1 < s1.len ? s1.len - i1 - 1 : 0) + (i2 < s2.len ? s2.len - i2 - 1 : 0);
        dist = rfo_min(dist, cur_dist);
    }
    return dist;
}

/* :435-507 hyrroe2003 (RECORD_MATRIX = RECORD_BIT_ROW = 0) */
static size_t hyrroe2003(const rfo_pm *pm, size_t len1, rfo_str s2, size_t score_cutoff)
{
    uint64_t vp = ~(uint64_t)0, vn = 0;
    size_t dist = len1;
    uint64_t mask = (uint64_t)1 << (len1 - 1);

    for (size_t i = 0; i < s2.len; ++i) {
        uint64_t x = rfo_pm_get(pm, 0, s2.p[i]);
        uint64_t d0 = (((x & vp) + vp) ^ vp) | x | vn;
        uint64_t hp = vn | ~(d0 | vp);
        uint64_t hn = d0 & vp;
        dist += (hp & mask) != 0;
        dist -= (hn & mask) != 0;
        hp = (hp << 1) | 1;
        hn <<= 1;
        vp = hn | ~(d0 | hp);
        vn = hp & d0;
    }
    rfo_last_path = RFO_PATH_HYRROE2003;
    return dist <= score_cutoff ? dist : RFO_USIZE_MAX;
}

/* :509-617 hyrroe2003_small_band_with_pm */
static size_t hyrroe2003_small_band_with_pm(const rfo_pm *pm, size_t len1, rfo_str s2, size_t score_cutoff)
{
    uint64_t vp = ~(uint64_t)0 << (64 - score_cutoff - 1);
    uint64_t vn = 0;
    size_t words = pm->block_count;
    size_t curr_dist = score_cutoff;
    const uint64_t diagonal_mask = (uint64_t)1 << 63;
    uint64_t horizontal_mask = (uint64_t)1 << 62;
    ptrdiff_t start_pos = (ptrdiff_t)score_cutoff + 1 - 64;
    size_t len2 = s2.len;
    size_t break_score =
        (size_t)((ptrdiff_t)score_cutoff + (ptrdiff_t)len2 - ((ptrdiff_t)len1 - (ptrdiff_t)score_cutoff));
    size_t j = 0;
    rfo_last_path = RFO_PATH_SMALL_BAND;

    size_t first = len1 > score_cutoff ? rfo_min(len1 - score_cutoff, len2) : 0;
    for (; j < len2; ++j) {
        uint8_t ch2 = s2.p[j];
        uint64_t pm_j;
        if (start_pos < 0) {
            pm_j = rfo_pm_get(pm, 0, ch2) << (-start_pos);
        } else {
            size_t word = (size_t)start_pos / 64, word_pos = (size_t)start_pos % 64;
            pm_j = rfo_pm_get(pm, word, ch2) >> word_pos;
            if (word + 1 < words && word_pos != 0) pm_j |= rfo_pm_get(pm, word + 1, ch2) << (64 - word_pos);
        }
        uint64_t x = pm_j;
        uint64_t d0 = (((x & vp) + vp) ^ vp) | x | vn;
        uint64_t hp = vn | ~(d0 | vp);
        uint64_t hn = d0 & vp;

        if (j < first) { /* first loop :539-574: walk the diagonal */
            curr_dist += (d0 & diagonal_mask) == 0;
        } else { /* second loop :577-614: walk the last row */
            curr_dist += (hp & horizontal_mask) != 0;
            curr_dist -= (hn & horizontal_mask) != 0;
            horizontal_mask >>= 1;
        }
        if (curr_dist > break_score) return RFO_USIZE_MAX;

        vp = hn | ~((d0 >> 1) | hp);
        vn = (d0 >> 1) & hp;
        start_pos += 1;
    }
    return curr_dist;
}

/* :619-767 hyrroe2003_small_band_without_pm (RECORD_MATRIX = 0).  The HybridGrowingHashmap of
 * (isize, u64) collapses to its extended_ascii[256] array for u8 keys. */
static size_t hyrroe2003_small_band_without_pm(rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    size_t len1 = s1.len, len2 = s2.len;
    uint64_t vp = ~(uint64_t)0 << (64 - score_cutoff - 1);
    uint64_t vn = 0;
    size_t dist = score_cutoff;
    const uint64_t diagonal_mask = (uint64_t)1 << 63;
    uint64_t horizontal_mask = (uint64_t)1 << 62;
    size_t break_score =
        (size_t)((ptrdiff_t)score_cutoff + (ptrdiff_t)len2 - ((ptrdiff_t)len1 - (ptrdiff_t)score_cutoff));
    struct {
        ptrdiff_t pos;
        uint64_t bits;
    } pm[256];
    memset(pm, 0, sizeof(pm));
    rfo_last_path = RFO_PATH_SMALL_BAND;

    ptrdiff_t i = 0 - (ptrdiff_t)score_cutoff;
    size_t i1 = 0, i2 = 0;
    for (; i1 < rfo_min(score_cutoff, len1); ++i1) { /* :673-678 */
        uint8_t ch1 = s1.p[i1];
        pm[ch1].bits = rfo_shr64(pm[ch1].bits, (size_t)(i - pm[ch1].pos)) | ((uint64_t)1 << 63);
        pm[ch1].pos = i;
        i += 1;
    }

    /* :681-719: zip(s1, s2).take(len1 - score_cutoff) */
    size_t n_diag = len1 - score_cutoff;
    for (size_t t = 0; t < n_diag && i1 < len1 && i2 < len2; ++t, ++i1, ++i2) {
        uint8_t ch1 = s1.p[i1], ch2 = s2.p[i2];
        pm[ch1].bits = rfo_shr64(pm[ch1].bits, (size_t)(i - pm[ch1].pos)) | ((uint64_t)1 << 63);
        pm[ch1].pos = i;
        uint64_t pm_j = rfo_shr64(pm[ch2].bits, (size_t)(i - pm[ch2].pos));

        uint64_t x = pm_j;
        uint64_t d0 = (((x & vp) + vp) ^ vp) | x | vn;
        uint64_t hp = vn | ~(d0 | vp);
        uint64_t hn = d0 & vp;
        dist += (d0 & diagonal_mask) == 0;
        if (dist > break_score) return RFO_USIZE_MAX;
        vp = hn | ~((d0 >> 1) | hp);
        vn = (d0 >> 1) & hp;
        i += 1;
    }

    for (; i2 < len2; ++i2) { /* :721-759 */
        uint8_t ch2 = s2.p[i2];
        if (i1 < len1) {
            uint8_t ch1 = s1.p[i1++];
            pm[ch1].bits = rfo_shr64(pm[ch1].bits, (size_t)(i - pm[ch1].pos)) | ((uint64_t)1 << 63);
            pm[ch1].pos = i;
        }
        uint64_t pm_j = rfo_shr64(pm[ch2].bits, (size_t)(i - pm[ch2].pos));

        uint64_t x = pm_j;
        uint64_t d0 = (((x & vp) + vp) ^ vp) | x | vn;
        uint64_t hp = vn | ~(d0 | vp);
        uint64_t hn = d0 & vp;
        dist += (hp & horizontal_mask) != 0;
        dist -= (hn & horizontal_mask) != 0;
        horizontal_mask >>= 1;
        if (dist > break_score) return RFO_USIZE_MAX;
        vp = hn | ~((d0 >> 1) | hp);
        vn = (d0 >> 1) & hp;
        i += 1;
    }
    return dist <= score_cutoff ? dist : RFO_USIZE_MAX;
}

/* :769-1019 hyrroe2003_block (RECORD_MATRIX = RECORD_BIT_ROW = 0, stop_row = -1) */
typedef struct {
    uint64_t vp, vn;
} lev_row;

static size_t hyrroe2003_block(const rfo_pm *pm, size_t len1, rfo_str s2, size_t score_cutoff)
{
    size_t len2 = s2.len;
    rfo_last_path = RFO_PATH_BLOCK;
    if (score_cutoff < rfo_abs_diff(len1, len2)) return RFO_USIZE_MAX;

    const size_t word_size = 64;
    size_t words = pm->block_count;
    lev_row *vecs = (lev_row *)malloc(words * sizeof(lev_row));
    size_t *scores = (size_t *)malloc(words * sizeof(size_t));
    for (size_t x = 0; x < words; ++x) {
        vecs[x].vp = ~(uint64_t)0;
        vecs[x].vn = 0;
        scores[x] = (x + 1) * word_size;
    }
    scores[words - 1] = len1;
    uint64_t last = (uint64_t)1 << ((len1 - 1) % word_size);

    score_cutoff = rfo_min(score_cutoff, rfo_max(len1, len2));
    size_t first_block = 0;
    /* :814-820 */
    size_t last_block =
        rfo_min(words, rfo_ceil_div(rfo_min(score_cutoff, (score_cutoff + len1 - len2) / 2) + 1, word_size)) - 1;
    size_t result = RFO_USIZE_MAX;
    int early = 0;

#define GET_ROW_NUM(word) ((word) + 1 == words ? len1 - 1 : ((word) + 1) * word_size - 1)
/* advance_block closure :838-875 */
#define ADVANCE_BLOCK(word)                                               \
    do {                                                                  \
        uint64_t pm_j = rfo_pm_get(pm, (word), ch2);                      \
        uint64_t vn = vecs[(word)].vn, vp = vecs[(word)].vp;              \
        uint64_t x = pm_j | (uint64_t)hn_carry;                           \
        uint64_t d0 = (((x & vp) + vp) ^ vp) | x | vn;                    \
        uint64_t hp = vn | ~(d0 | vp);                                    \
        uint64_t hn = d0 & vp;                                            \
        int hp_carry_temp = hp_carry, hn_carry_temp = hn_carry;           \
        if ((word) < words - 1) {                                         \
            hp_carry = (hp >> 63) != 0;                                   \
            hn_carry = (hn >> 63) != 0;                                   \
        } else {                                                          \
            hp_carry = (hp & last) != 0;                                  \
            hn_carry = (hn & last) != 0;                                  \
        }                                                                 \
        hp = (hp << 1) | (uint64_t)hp_carry_temp;                         \
        hn = (hn << 1) | (uint64_t)hn_carry_temp;                         \
        vecs[(word)].vp = hn | ~(d0 | hp);                                \
        vecs[(word)].vn = hp & d0;                                        \
    } while (0)

    for (size_t row = 0; row < len2; ++row) {
        uint8_t ch2 = s2.p[row];
        int hp_carry = 1, hn_carry = 0;

        for (size_t word = first_block; word <= last_block; ++word) { /* :885-895 */
            ADVANCE_BLOCK(word);
            scores[word] += (size_t)hp_carry;
            scores[word] -= (size_t)hn_carry;
        }

        { /* :897-904 */
            ptrdiff_t a = (ptrdiff_t)len2 - (ptrdiff_t)row - 1;
            ptrdiff_t b = (ptrdiff_t)len1 - (ptrdiff_t)((1 + last_block) * word_size - 1) - 1;
            ptrdiff_t c = (ptrdiff_t)scores[last_block] + (a > b ? a : b);
            ptrdiff_t sc = (ptrdiff_t)score_cutoff;
            score_cutoff = (size_t)(sc < c ? sc : c);
        }

        /* :912-934 band adjustment: last_block */
        if (last_block + 1 < words &&
            (ptrdiff_t)GET_ROW_NUM(last_block) <= (ptrdiff_t)score_cutoff + 2 * (ptrdiff_t)word_size + (ptrdiff_t)row +
                                                      (ptrdiff_t)len1 - (ptrdiff_t)scores[last_block] - 2 -
                                                      (ptrdiff_t)len2) {
            last_block += 1;
            vecs[last_block].vp = ~(uint64_t)0;
            vecs[last_block].vn = 0;
            size_t chars_in_block = (last_block + 1 == words) ? (len1 - 1) % word_size + 1 : 64;
            scores[last_block] = scores[last_block - 1] + chars_in_block - (size_t)hp_carry + (size_t)hn_carry;
            ADVANCE_BLOCK(last_block);
            scores[last_block] += (size_t)hp_carry;
            scores[last_block] -= (size_t)hn_carry;
        }

        /* :936-960.  `while last_block >= first_block` with usize: when the loop would step below
         * zero the Rust code underflows (panic in debug); it cannot, because block 0 ... see note
         * below -- we use a signed copy and treat "< first_block" as band-empty. */
        ptrdiff_t lb = (ptrdiff_t)last_block;
        while (lb >= (ptrdiff_t)first_block) {
            int in_band_cond1 = scores[lb] < score_cutoff + word_size;
            int in_band_cond2 = (ptrdiff_t)GET_ROW_NUM((size_t)lb) <=
                                (ptrdiff_t)score_cutoff + 2 * (ptrdiff_t)word_size + (ptrdiff_t)row + (ptrdiff_t)len1 +
                                    1 - (ptrdiff_t)scores[lb] - 2 - (ptrdiff_t)len2;
            if (in_band_cond1 && in_band_cond2) break;
            lb -= 1;
        }
        if (lb < (ptrdiff_t)first_block) { /* :982-985 (also covers the would-be underflow at block 0) */
            early = 1;
            break;
        }
        last_block = (size_t)lb;

        /* :963-979 band adjustment: first_block */
        while (first_block <= last_block) {
            int in_band_cond1 = scores[first_block] < score_cutoff + word_size;
            int in_band_cond2 = (ptrdiff_t)GET_ROW_NUM(first_block) >=
                                (ptrdiff_t)scores[first_block] + (ptrdiff_t)len1 + (ptrdiff_t)row -
                                    (ptrdiff_t)score_cutoff - (ptrdiff_t)len2;
            if (in_band_cond1 && in_band_cond2) break;
            first_block += 1;
        }
        if (last_block < first_block) { /* :982-985 */
            early = 1;
            break;
        }
    }
#undef ADVANCE_BLOCK
#undef GET_ROW_NUM

    if (!early) { /* :1012-1018 */
        size_t dist = scores[words - 1];
        result = dist <= score_cutoff ? dist : RFO_USIZE_MAX;
    }
    free(vecs);
    free(scores);
    return result;
}

/* :1021-1102 uniform_distance_with_pm */
static size_t uniform_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, size_t score_cutoff,
                                       size_t score_hint)
{
    size_t len1 = s1.len, len2 = s2.len;
    score_cutoff = rfo_min(score_cutoff, rfo_max(len1, len2));
    score_hint = rfo_max(score_hint, 31);

    if (score_cutoff == 0) {
        rfo_last_path = RFO_PATH_EQ;
        return rfo_str_eq(s1, s2) ? 0 : RFO_USIZE_MAX;
    }
    if (score_cutoff < rfo_abs_diff(len1, len2)) {
        rfo_last_path = RFO_PATH_LENDIFF;
        return RFO_USIZE_MAX;
    }
    if (len1 == 0 || len2 == 0) {
        rfo_last_path = RFO_PATH_EMPTY;
        return len1 + len2;
    }

    if (score_cutoff >= 4) {
        size_t full_band = rfo_min(len1, 2 * score_cutoff + 1);
        if (len1 <= 64) return hyrroe2003(pm, len1, s2, score_cutoff);
        if (full_band <= 64) return hyrroe2003_small_band_with_pm(pm, len1, s2, score_cutoff);

        while (score_hint < score_cutoff) { /* :1069-1088 */
            full_band = rfo_min(len1, 2 * score_hint + 1);
            size_t score = full_band <= 64 ? hyrroe2003_small_band_with_pm(pm, len1, s2, score_hint)
                                           : hyrroe2003_block(pm, len1, s2, score_hint);
            if (score <= score_hint) return score;
            if (RFO_USIZE_MAX / 2 < score_hint) break;
            score_hint *= 2;
        }
        return hyrroe2003_block(pm, len1, s2, score_cutoff);
    }

    rfo_affix affix = rfo_remove_common_affix(s1, s2);
    if (affix.s1.len == 0 || affix.s2.len == 0) {
        rfo_last_path = RFO_PATH_AFFIX;
        return affix.s1.len + affix.s2.len;
    }
    return lev_mbleven2018(affix.s1, affix.s2, score_cutoff);
}

/* :1104-1222 uniform_distance_without_pm */
static size_t uniform_distance_without_pm(rfo_str s1, rfo_str s2, size_t score_cutoff, size_t score_hint)
{
    if (s1.len < s2.len) return uniform_distance_without_pm(s2, s1, score_cutoff, score_hint);
    size_t len1 = s1.len, len2 = s2.len;

    score_cutoff = rfo_min(score_cutoff, rfo_max(len1, len2));
    score_hint = rfo_max(score_hint, 31);

    if (score_cutoff == 0) {
        rfo_last_path = RFO_PATH_EQ;
        return rfo_str_eq(s1, s2) ? 0 : RFO_USIZE_MAX;
    }
    if (score_cutoff < rfo_abs_diff(len1, len2)) {
        rfo_last_path = RFO_PATH_LENDIFF;
        return RFO_USIZE_MAX;
    }

    rfo_affix affix = rfo_remove_common_affix(s1, s2);
    if (affix.s1.len == 0 || affix.s2.len == 0) {
        rfo_last_path = RFO_PATH_AFFIX;
        return affix.s1.len + affix.s2.len;
    }
    if (score_cutoff < 4) return lev_mbleven2018(affix.s1, affix.s2, score_cutoff);

    size_t full_band = rfo_min(affix.s1.len, 2 * score_cutoff + 1);
    size_t res;

    if (affix.s2.len <= 64) { /* :1151-1163: the SHORTER string becomes the pattern */
        rfo_pm pm;
        rfo_pm_init(&pm, affix.s2.p, affix.s2.len);
        res = hyrroe2003(&pm, affix.s2.len, affix.s1, score_cutoff);
        rfo_pm_free(&pm);
    } else if (full_band <= 64) {
        res = hyrroe2003_small_band_without_pm(affix.s1, affix.s2, score_cutoff);
    } else {
        rfo_pm pm;
        rfo_pm_init(&pm, affix.s1.p, affix.s1.len);
        int done = 0;
        res = RFO_USIZE_MAX;
        while (score_hint < score_cutoff) {
            full_band = rfo_min(affix.s1.len, 2 * score_hint + 1);
            size_t score = full_band <= 64
                               ? hyrroe2003_small_band_with_pm(&pm, affix.s1.len, affix.s2, score_hint)
                               : hyrroe2003_block(&pm, affix.s1.len, affix.s2, score_hint);
            if (score <= score_hint) {
                res = score;
                done = 1;
                break;
            }
            if (RFO_USIZE_MAX / 2 < score_hint) break;
            score_hint *= 2;
        }
        if (!done) res = hyrroe2003_block(&pm, affix.s1.len, affix.s2, score_cutoff);
        rfo_pm_free(&pm);
    }
    return res;
}

/* indel::IndividualComparator{}._distance (src/distance/indel.rs:60-105), needed by :1268 */
static size_t indel_individual_distance(rfo_str s1, rfo_str s2, size_t score_cutoff)
{
    size_t maximum = s1.len + s2.len;
    size_t lcs_cutoff = maximum / 2 >= score_cutoff ? maximum / 2 - score_cutoff : 0;
    /* lcs_seq::IndividualComparator._similarity (lcs_seq.rs:553-569) ignores the hint */
    size_t lcs_sim = rfo_lcs_similarity_without_pm(s1, s2, lcs_cutoff);
    return maximum - 2 * lcs_sim;
}

/* :1224-1282 _distance_without_pm */
size_t rfo_lev_distance_without_pm(rfo_str s1, rfo_str s2, const rfo_weights *w, size_t score_cutoff,
                                   size_t score_hint)
{
    if (s1.len * s2.len < 90) return generalized_distance(s1, s2, w, score_cutoff);

    if (w->insertion_cost == w->deletion_cost) {
        if (w->insertion_cost == 0) return 0;
        if (w->insertion_cost == w->substitution_cost) {
            size_t new_cutoff = rfo_ceil_div(score_cutoff, w->insertion_cost);
            size_t new_hint = rfo_ceil_div(score_hint, w->insertion_cost);
            size_t dist = uniform_distance_without_pm(s1, s2, new_cutoff, new_hint);
            dist *= w->insertion_cost; /* wrapping like release-mode Rust for the usize::MAX sentinel */
            return dist;
        } else if (w->substitution_cost >= w->insertion_cost + w->deletion_cost) {
            size_t new_cutoff = rfo_ceil_div(score_cutoff, w->insertion_cost);
            size_t dist = indel_individual_distance(s1, s2, new_cutoff);
            dist *= w->insertion_cost;
            return dist;
        }
    }
    return generalized_distance(s1, s2, w, score_cutoff);
}

/* :1285-1331 _distance_with_pm */
size_t rfo_lev_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2, const rfo_weights *w,
                                size_t score_cutoff, size_t score_hint)
{
    if (w->insertion_cost == w->deletion_cost) {
        if (w->insertion_cost == 0) return 0;
        if (w->insertion_cost == w->substitution_cost) {
            size_t new_cutoff = rfo_ceil_div(score_cutoff, w->insertion_cost);
            size_t new_hint = rfo_ceil_div(score_hint, w->insertion_cost);
            size_t dist = uniform_distance_with_pm(pm, s1, s2, new_cutoff, new_hint);
            dist *= w->insertion_cost;
            return dist;
        } else if (w->substitution_cost >= w->insertion_cost + w->deletion_cost) {
            size_t new_cutoff = rfo_ceil_div(score_cutoff, w->insertion_cost);
            size_t dist = rfo_indel_distance_with_pm(pm, s1, s2, new_cutoff);
            dist *= w->insertion_cost;
            return dist;
        }
    }
    return generalized_distance(s1, s2, w, score_cutoff);
}
