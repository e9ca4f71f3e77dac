/*
 * rfo_api.c -- CPU ORACLE (test infrastructure only): the metric-framework layer
 * (src/details/distance.rs MetricUsize / Metricf64 default methods, src/common.rs cutoff wrappers) and
 * the public entry points (free functions + BatchComparator) of levenshtein / indel / lcs_seq / jaro /
 * jaro_winkler / fuzz, flattened into a C ABI that tests/ and bench.py's cpu_baseline leg drive through
 * ctypes.  See rfo_common.h for the rules.  usize arithmetic wraps like a release-mode Rust build.
 */
#include "rfo_common.h"
#include "rfo_oracle.h"
#include <math.h>
#include <pthread.h>

/* ------------------------------------------------------------------------------------------------
 * MetricUsize (src/details/distance.rs:154-275): a comparator overrides exactly one of
 * _distance / _similarity; the other and both normalized forms are the trait defaults below.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mu mu;
struct mu {
    size_t (*maximum)(const mu *, size_t, size_t);
    size_t (*distance)(const mu *, rfo_str, rfo_str, rfo_opt_usize, rfo_opt_usize);
    size_t (*similarity)(const mu *, rfo_str, rfo_str, rfo_opt_usize, rfo_opt_usize);
    const rfo_pm *pm; /* NULL for the IndividualComparator flavours */
    rfo_weights w;
};

/* details/distance.rs:157-179 default _distance */
static size_t mu_default_distance(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_usize cutoff, rfo_opt_usize hint)
{
    size_t maximum = m->maximum(m, s1.len, s2.len);
    rfo_opt_usize cs = cutoff, hs = hint;
    if (cs.has) cs.v = maximum >= cs.v ? maximum - cs.v : 0;
    if (hs.has) hs.v = maximum >= hs.v ? maximum - hs.v : 0;
    size_t sim = m->similarity(m, s1, s2, cs, hs);
    return maximum - sim;
}

/* details/distance.rs:181-211 default _similarity */
static size_t mu_default_similarity(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_usize cutoff, rfo_opt_usize hint)
{
    size_t maximum = m->maximum(m, s1.len, s2.len);
    if (cutoff.has) {
        if (cutoff.v > maximum) return maximum;
        if (hint.has) hint.v = rfo_min(hint.v, cutoff.v);
    }
    rfo_opt_usize cd = cutoff, hd = hint;
    if (cd.has) cd.v = maximum - cd.v;
    if (hd.has) hd.v = maximum - hd.v;
    size_t dist = m->distance(m, s1, s2, cd, hd);
    return maximum - dist;
}

/* details/distance.rs:213-251 _normalized_distance */
static double mu_normalized_distance(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_f64 cutoff, rfo_opt_f64 hint)
{
    size_t maximum = m->maximum(m, s1.len, s2.len);
    rfo_opt_usize cd = rfo_none_u(), hd = rfo_none_u();
    if (cutoff.has) {
        double c = cutoff.v < 0.0 ? 0.0 : (cutoff.v > 1.0 ? 1.0 : cutoff.v); /* clamp(0.0, 1.0) */
        cd = rfo_some_u((size_t)ceil((double)maximum * c));
    }
    if (hint.has) {
        double c = hint.v < 0.0 ? 0.0 : (hint.v > 1.0 ? 1.0 : hint.v);
        hd = rfo_some_u((size_t)ceil((double)maximum * c));
    }
    size_t dist = m->distance(m, s1, s2, cd, hd);
    return maximum == 0 ? 0.0 : (double)dist / (double)maximum;
}

/* details/distance.rs:253-274 _normalized_similarity */
static double mu_normalized_similarity(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_f64 cutoff, rfo_opt_f64 hint)
{
    rfo_opt_f64 cs = cutoff, hs = hint;
    if (cs.has) cs.v = rfo_norm_sim_to_norm_dist(cs.v);
    if (hs.has) hs.v = rfo_norm_sim_to_norm_dist(hs.v);
    return 1.0 - mu_normalized_distance(m, s1, s2, cs, hs);
}

/* -- levenshtein: IndividualComparator :1333-1367, BatchComparatorImpl :1587-1623 -- */
static size_t lev_maximum(const mu *m, size_t l1, size_t l2) { return rfo_lev_maximum(l1, l2, &m->w); }
static size_t lev_distance(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_usize c, rfo_opt_usize h)
{
    size_t cutoff = c.has ? c.v : RFO_USIZE_MAX, hint = h.has ? h.v : RFO_USIZE_MAX;
    if (m->pm) return rfo_lev_distance_with_pm(m->pm, s1, s2, &m->w, cutoff, hint);
    return rfo_lev_distance_without_pm(s1, s2, &m->w, cutoff, hint);
}
/* -- lcs_seq: IndividualComparator :546-570, BatchComparator :772-793 -- */
static size_t lcs_maximum(const mu *m, size_t l1, size_t l2) { (void)m; return rfo_max(l1, l2); }
static size_t lcs_similarity(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_usize c, rfo_opt_usize h)
{
    (void)h;
    size_t cutoff = c.has ? c.v : 0;
    if (m->pm) return rfo_lcs_similarity_with_pm(m->pm, s1, s2, cutoff);
    return rfo_lcs_similarity_without_pm(s1, s2, cutoff);
}
/* -- indel: IndividualComparator indel.rs:60-105, BatchComparator :327-368 -- */
static size_t indel_maximum(const mu *m, size_t l1, size_t l2) { (void)m; return l1 + l2; }
static size_t indel_distance(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_usize c, rfo_opt_usize h)
{
    (void)h; /* the lcs hint is computed (:90-94, :352-356) but lcs_seq ignores it */
    size_t score_cutoff = c.has ? c.v : RFO_USIZE_MAX;
    size_t maximum = s1.len + s2.len;
    size_t lcs_cutoff = maximum / 2 >= score_cutoff ? maximum / 2 - score_cutoff : 0;
    size_t lcs_sim = m->pm ? rfo_lcs_similarity_with_pm(m->pm, s1, s2, lcs_cutoff)
                           : rfo_lcs_similarity_without_pm(s1, s2, lcs_cutoff);
    return maximum - 2 * lcs_sim;
}

/* -- osa: IndividualComparator osa.rs:228-268, BatchComparator :431-461 (both ignore cutoff and hint) -- */
static size_t osa_distance(const mu *m, rfo_str s1, rfo_str s2, rfo_opt_usize c, rfo_opt_usize h)
{
    (void)c;
    (void)h;
    return m->pm ? rfo_osa_distance_with_pm(m->pm, s1, s2) : rfo_osa_distance_without_pm(s1, s2);
}

static void mu_make(mu *m, int metric, const rfo_pm *pm, const rfo_weights *w)
{
    static const rfo_weights unit = {1, 1, 1};
    m->pm = pm;
    m->w = w ? *w : unit;
    switch (metric) {
    case RFO_LEVENSHTEIN:
        m->maximum = lev_maximum;
        m->distance = lev_distance;
        m->similarity = mu_default_similarity;
        break;
    case RFO_OSA:
        m->maximum = lcs_maximum; /* len1.max(len2), osa.rs:231-233, :432-434 */
        m->distance = osa_distance;
        m->similarity = mu_default_similarity;
        break;
    case RFO_INDEL:
        m->maximum = indel_maximum;
        m->distance = indel_distance;
        m->similarity = mu_default_similarity;
        break;
    default: /* RFO_LCS_SEQ */
        m->maximum = lcs_maximum;
        m->distance = mu_default_distance;
        m->similarity = lcs_similarity;
        break;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Metricf64 (src/details/distance.rs:277-385) for jaro / jaro_winkler: _similarity is overridden.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int winkler;
    double prefix_weight;
    const rfo_pm *pm;
} mf;

static double mf_similarity(const mf *m, rfo_str s1, rfo_str s2, rfo_opt_f64 cutoff)
{
    double c = cutoff.has ? cutoff.v : 0.0; /* jaro.rs:622, :820; jaro_winkler.rs:166, :391-399 */
    if (m->winkler)
        return m->pm ? rfo_jw_similarity_with_pm(m->pm, s1, s2, m->prefix_weight, c)
                     : rfo_jw_similarity_without_pm(s1, s2, m->prefix_weight, c);
    return m->pm ? rfo_jaro_similarity_with_pm(m->pm, s1, s2, c) : rfo_jaro_similarity_without_pm(s1, s2, c);
}
/* details/distance.rs:280-302 (maximum is 1.0 for both metrics) */
static double mf_distance(const mf *m, rfo_str s1, rfo_str s2, rfo_opt_f64 cutoff)
{
    double maximum = 1.0;
    rfo_opt_f64 cs = cutoff;
    if (cs.has) cs.v = maximum >= cs.v ? maximum - cs.v : 0.0;
    return maximum - mf_similarity(m, s1, s2, cs);
}
/* :336-361 */
static double mf_normalized_distance(const mf *m, rfo_str s1, rfo_str s2, rfo_opt_f64 cutoff)
{
    double maximum = 1.0;
    rfo_opt_f64 cd = cutoff;
    if (cd.has) cd.v = maximum * cd.v;
    double dist = mf_distance(m, s1, s2, cd);
    return maximum > 0.0 ? dist / maximum : 0.0;
}
/* :363-384 */
static double mf_normalized_similarity(const mf *m, rfo_str s1, rfo_str s2, rfo_opt_f64 cutoff)
{
    rfo_opt_f64 cs = cutoff;
    if (cs.has) cs.v = rfo_norm_sim_to_norm_dist(cs.v);
    return 1.0 - mf_normalized_distance(m, s1, s2, cs);
}

/* ------------------------------------------------------------------------------------------------
 * public entry points
 * ---------------------------------------------------------------------------------------------- */
struct rfo_batch {
    int metric;
    uint8_t *s1;
    size_t len1;
    rfo_pm pm;
};

rfo_batch *rfo_batch_new(int metric, const uint8_t *s1, size_t len1)
{
    rfo_batch *b = (rfo_batch *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->metric = metric;
    b->len1 = len1;
    b->s1 = (uint8_t *)malloc(len1 ? len1 : 1);
    if (len1) memcpy(b->s1, s1, len1);
    rfo_pm_init(&b->pm, b->s1, len1);
    return b;
}
void rfo_batch_free(rfo_batch *b)
{
    if (!b) return;
    rfo_pm_free(&b->pm);
    free(b->s1);
    free(b);
}
const uint64_t *rfo_batch_pm(const rfo_batch *b, size_t *block_count)
{
    *block_count = b->pm.block_count;
    return b->pm.bits;
}

/* score() of DistanceCutoff / SimilarityCutoff (src/common.rs:28-30, :43-45, :68-70, :83-85) */
static int score_usize(int op, const rfo_call_args *a, size_t raw, size_t *out)
{
    *out = raw;
    if (!a->has_cutoff) return 1;
    return op == RFO_OP_DISTANCE ? raw <= a->cutoff_usize : raw >= a->cutoff_usize;
}
static int score_f64(int op, const rfo_call_args *a, double raw, double *out)
{
    *out = raw;
    if (!a->has_cutoff) return 1;
    int dist_like = (op == RFO_OP_DISTANCE || op == RFO_OP_NORMALIZED_DISTANCE);
    return dist_like ? raw <= a->cutoff_f64 : raw >= a->cutoff_f64;
}

static int call_usize(int metric, const rfo_pm *pm, int op, rfo_str s1, rfo_str s2, const rfo_call_args *a,
                      size_t *out)
{
    mu m;
    mu_make(&m, metric, pm, (const rfo_weights *)&a->weights);
    rfo_opt_usize c = a->has_cutoff ? rfo_some_u(a->cutoff_usize) : rfo_none_u();
    rfo_opt_usize h = a->has_hint ? rfo_some_u(a->hint_usize) : rfo_none_u();
    size_t raw = op == RFO_OP_DISTANCE ? m.distance(&m, s1, s2, c, h) : m.similarity(&m, s1, s2, c, h);
    return score_usize(op, a, raw, out);
}

static int call_f64(int metric, const rfo_pm *pm, int op, rfo_str s1, rfo_str s2, const rfo_call_args *a, double *out)
{
    rfo_opt_f64 c = a->has_cutoff ? rfo_some_f(a->cutoff_f64) : rfo_none_f();
    rfo_opt_f64 h = a->has_hint ? rfo_some_f(a->hint_f64) : rfo_none_f();
    double raw;
    if (metric == RFO_JARO || metric == RFO_JARO_WINKLER) {
        mf m = {metric == RFO_JARO_WINKLER, a->prefix_weight, pm};
        switch (op) {
        case RFO_OP_DISTANCE: raw = mf_distance(&m, s1, s2, c); break;
        case RFO_OP_SIMILARITY: raw = mf_similarity(&m, s1, s2, c); break;
        case RFO_OP_NORMALIZED_DISTANCE: raw = mf_normalized_distance(&m, s1, s2, c); break;
        default: raw = mf_normalized_similarity(&m, s1, s2, c); break;
        }
        return score_f64(op, a, raw, out);
    }
    if (metric == RFO_FUZZ_RATIO) {
        /* fuzz::ratio_with_args (src/fuzz.rs:60-85) = indel IndividualComparator normalized similarity;
         * RatioBatchComparator::similarity_with_args (src/fuzz.rs:127-149) calls
         * self.scorer.scorer._normalized_similarity, i.e. the INNER lcs_seq::BatchComparator
         * (quirk Q1 in SURVEY.md App. C): LCS / max(len1, len2). */
        mu m;
        mu_make(&m, pm ? RFO_LCS_SEQ : RFO_INDEL, pm, NULL);
        raw = mu_normalized_similarity(&m, s1, s2, c, h);
        return score_f64(RFO_OP_NORMALIZED_SIMILARITY, a, raw, out);
    }
    mu m;
    mu_make(&m, metric, pm, (const rfo_weights *)&a->weights);
    raw = op == RFO_OP_NORMALIZED_DISTANCE ? mu_normalized_distance(&m, s1, s2, c, h)
                                           : mu_normalized_similarity(&m, s1, s2, c, h);
    return score_f64(op, a, raw, out);
}

int rfo_batch_usize(const rfo_batch *b, int op, const uint8_t *s2, size_t len2, const rfo_call_args *a, size_t *out)
{
    rfo_str s1 = {b->s1, b->len1}, t = {s2, len2};
    return call_usize(b->metric, &b->pm, op, s1, t, a, out);
}
int rfo_batch_f64(const rfo_batch *b, int op, const uint8_t *s2, size_t len2, const rfo_call_args *a, double *out)
{
    rfo_str s1 = {b->s1, b->len1}, t = {s2, len2};
    return call_f64(b->metric, &b->pm, op, s1, t, a, out);
}
int rfo_free_usize(int metric, int op, const uint8_t *s1, size_t len1, const uint8_t *s2, size_t len2,
                   const rfo_call_args *a, size_t *out)
{
    rfo_str a1 = {s1, len1}, a2 = {s2, len2};
    return call_usize(metric, NULL, op, a1, a2, a, out);
}
int rfo_free_f64(int metric, int op, const uint8_t *s1, size_t len1, const uint8_t *s2, size_t len2,
                 const rfo_call_args *a, double *out)
{
    rfo_str a1 = {s1, len1}, a2 = {s2, len2};
    return call_f64(metric, NULL, op, a1, a2, a, out);
}
int rfo_last_lev_path(void) { return rfo_last_path; }
extern __thread unsigned rfo_q8_edges;
unsigned rfo_last_lcs_q8_edges(void) { return rfo_q8_edges; }

/* ---- one-vs-many loops (what a user of the reference writes around BatchComparator, cf.
 *      rapidfuzz-benches/benches/bench_levenshtein.rs:51-60), optionally split over threads ---- */
typedef struct {
    const rfo_batch *b;
    int op, is_f64;
    const uint8_t *bytes;
    const uint64_t *offsets; /* n + 1 entries, or NULL for fixed-stride rows */
    size_t stride, fixed_len;
    size_t begin, end;
    const rfo_call_args *a;
    uint64_t *out_u;
    double *out_f;
} many_job;

static void *many_worker(void *p)
{
    many_job *j = (many_job *)p;
    for (size_t i = j->begin; i < j->end; ++i) {
        const uint8_t *s2;
        size_t len2;
        if (j->offsets) {
            s2 = j->bytes + j->offsets[i];
            len2 = (size_t)(j->offsets[i + 1] - j->offsets[i]);
        } else {
            s2 = j->bytes + i * j->stride;
            len2 = j->fixed_len;
        }
        if (j->is_f64) {
            double v;
            int some = rfo_batch_f64(j->b, j->op, s2, len2, j->a, &v);
            j->out_f[i] = some ? v : NAN;
        } else {
            size_t v;
            int some = rfo_batch_usize(j->b, j->op, s2, len2, j->a, &v);
            j->out_u[i] = some ? (uint64_t)v : UINT64_MAX;
        }
    }
    return NULL;
}

static void run_many(many_job *proto, size_t n, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    if (nthreads == 1) {
        proto->begin = 0;
        proto->end = n;
        many_worker(proto);
        return;
    }
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    many_job *jobs = (many_job *)malloc((size_t)nthreads * sizeof(many_job));
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = *proto;
        jobs[t].begin = n * (size_t)t / (size_t)nthreads;
        jobs[t].end = n * (size_t)(t + 1) / (size_t)nthreads;
        pthread_create(&th[t], NULL, many_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

void rfo_batch_many_usize(const rfo_batch *b, int op, const uint8_t *bytes, const uint64_t *offsets, size_t n,
                          const rfo_call_args *a, uint64_t *out, int nthreads)
{
    many_job j = {b, op, 0, bytes, offsets, 0, 0, 0, 0, a, out, NULL};
    run_many(&j, n, nthreads);
}
void rfo_batch_many_f64(const rfo_batch *b, int op, const uint8_t *bytes, const uint64_t *offsets, size_t n,
                        const rfo_call_args *a, double *out, int nthreads)
{
    many_job j = {b, op, 1, bytes, offsets, 0, 0, 0, 0, a, NULL, out};
    run_many(&j, n, nthreads);
}
void rfo_batch_rows_usize(const rfo_batch *b, int op, const uint8_t *rows, size_t n, size_t len, size_t stride,
                          const rfo_call_args *a, uint64_t *out, int nthreads)
{
    many_job j = {b, op, 0, rows, NULL, stride, len, 0, 0, a, out, NULL};
    run_many(&j, n, nthreads);
}
void rfo_batch_rows_f64(const rfo_batch *b, int op, const uint8_t *rows, size_t n, size_t len, size_t stride,
                        const rfo_call_args *a, double *out, int nthreads)
{
    many_job j = {b, op, 1, rows, NULL, stride, len, 0, 0, a, NULL, out};
    run_many(&j, n, nthreads);
}
