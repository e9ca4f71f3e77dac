/*
 * rfo_osa.c -- CPU ORACLE (test infrastructure only): restatement of src/distance/osa.rs (v0.5.0), the
 * Optimal String Alignment distance, for u8 elements.  See rfo_common.h for the rules.
 */
#include "rfo_common.h"

/* osa.rs:60-117 hyrroe2003 */
static size_t osa_hyrroe2003(const rfo_pm *pm, size_t len1, rfo_str s2)
{
    uint64_t vp = ~(uint64_t)0, vn = 0, d0 = 0, pm_j_old = 0;
    size_t curr_dist = len1;
    uint64_t mask = (uint64_t)1 << (len1 - 1);
    for (size_t i = 0; i < s2.len; ++i) {
        uint64_t pm_j = rfo_pm_get(pm, 0, s2.p[i]);
        uint64_t tr = (((~d0) & pm_j) << 1) & pm_j_old;
        d0 = (((pm_j & vp) + vp) ^ vp) | pm_j | vn;
        d0 |= tr;
        uint64_t hp = vn | ~(d0 | vp);
        uint64_t hn = d0 & vp;
        curr_dist += (hp & mask) != 0;
        curr_dist -= (hn & mask) != 0;
        hp = (hp << 1) | 1;
        hn <<= 1;
        vp = hn | ~(d0 | hp);
        vn = hp & d0;
        pm_j_old = pm_j;
    }
    return curr_dist;
}

typedef struct { /* osa.rs:119-136 */
    uint64_t vp, vn, d0, pm;
} osa_row;

/* osa.rs:138-226 hyrroe2003_block */
static size_t osa_hyrroe2003_block(const rfo_pm *pm, size_t len1, rfo_str s2)
{
    const size_t word_size = 64;
    size_t words = pm->block_count;
    uint64_t last = (uint64_t)1 << ((len1 - 1) % word_size);
    size_t curr_dist = len1;
    osa_row *old_vecs = (osa_row *)malloc((words + 1) * sizeof(osa_row));
    osa_row *new_vecs = (osa_row *)malloc((words + 1) * sizeof(osa_row));
    for (size_t w = 0; w <= words; ++w) {
        osa_row r = {~(uint64_t)0, 0, 0, 0};
        old_vecs[w] = new_vecs[w] = r;
    }
    for (size_t i = 0; i < s2.len; ++i) {
        uint8_t ch2 = s2.p[i];
        uint64_t hp_carry = 1, hn_carry = 0;
        for (size_t word = 0; word < words; ++word) {
            uint64_t vn = old_vecs[word + 1].vn, vp = old_vecs[word + 1].vp;
            uint64_t d0 = old_vecs[word + 1].d0;
            uint64_t d0_last = old_vecs[word].d0;
            uint64_t pm_j_old = old_vecs[word + 1].pm;
            uint64_t pm_last = new_vecs[word].pm;

            uint64_t pm_j = rfo_pm_get(pm, word, ch2);
            uint64_t x = pm_j;
            uint64_t tr = ((((~d0) & x) << 1) | (((~d0_last) & pm_last) >> 63)) & pm_j_old;

            x |= hn_carry;
            d0 = (((x & vp) + vp) ^ vp) | x | vn | tr;

            uint64_t hp = vn | ~(d0 | vp);
            uint64_t hn = d0 & vp;
            if (word == words - 1) {
                curr_dist += (hp & last) != 0;
                curr_dist -= (hn & last) != 0;
            }
            uint64_t hp_carry_temp = hp_carry;
            hp_carry = hp >> 63;
            hp = (hp << 1) | hp_carry_temp;
            uint64_t hn_carry_temp = hn_carry;
            hn_carry = hn >> 63;
            hn = (hn << 1) | hn_carry_temp;

            new_vecs[word + 1].vp = hn | ~(d0 | hp);
            new_vecs[word + 1].vn = hp & d0;
            new_vecs[word + 1].d0 = d0;
            new_vecs[word + 1].pm = pm_j;
        }
        osa_row *t = new_vecs;
        new_vecs = old_vecs;
        old_vecs = t;
    }
    free(old_vecs);
    free(new_vecs);
    return curr_dist;
}

/* osa.rs:228-268 IndividualComparator::_distance (cutoff and hint are unused there) */
size_t rfo_osa_distance_without_pm(rfo_str s1, rfo_str s2)
{
    if (s1.len < s2.len) return rfo_osa_distance_without_pm(s2, s1);
    rfo_affix affix = rfo_remove_common_affix(s1, s2);
    if (affix.s1.len == 0) return affix.s2.len;
    rfo_pm pm;
    rfo_pm_init(&pm, affix.s1.p, affix.s1.len);
    size_t r = affix.s1.len <= 64 ? osa_hyrroe2003(&pm, affix.s1.len, affix.s2) : osa_hyrroe2003_block(&pm, affix.s1.len, affix.s2);
    rfo_pm_free(&pm);
    return r;
}

/* osa.rs:431-461 BatchComparator::_distance */
size_t rfo_osa_distance_with_pm(const rfo_pm *pm, rfo_str s1, rfo_str s2)
{
    if (s1.len == 0) return s2.len;
    if (s2.len == 0) return s1.len;
    if (s1.len <= 64) return osa_hyrroe2003(pm, s1.len, s2);
    return osa_hyrroe2003_block(pm, s1.len, s2);
}
