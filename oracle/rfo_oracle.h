/*
 * rfo_oracle.h -- C ABI of the CPU ORACLE (test infrastructure only; see rfo_common.h).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load librf_oracle.so.
 * The product (rapidfuzz_rs_amd/, include/rfgpu.h) never does.
 */
#ifndef RFO_ORACLE_H
#define RFO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RFO_LEVENSHTEIN = 0, RFO_INDEL = 1, RFO_LCS_SEQ = 2, RFO_JARO = 3, RFO_JARO_WINKLER = 4, RFO_FUZZ_RATIO = 5, RFO_OSA = 6 };
enum { RFO_OP_DISTANCE = 0, RFO_OP_SIMILARITY = 1, RFO_OP_NORMALIZED_DISTANCE = 2, RFO_OP_NORMALIZED_SIMILARITY = 3 };

/* flattened `Args` builders (levenshtein.rs:86-126, jaro_winkler.rs:25-62, lcs_seq.rs / indel.rs / jaro.rs /
 * fuzz.rs equivalents).  has_cutoff == 0 is NoScoreCutoff. */
typedef struct rfo_call_args {
    int has_cutoff, has_hint;
    size_t cutoff_usize, hint_usize; /* Args<usize, _> */
    double cutoff_f64, hint_f64;     /* Args<f64, _> */
    struct {
        size_t insertion_cost, deletion_cost, substitution_cost;
    } weights;            /* levenshtein only; {1,1,1} = WeightTable::default() */
    double prefix_weight; /* jaro_winkler only; default 0.1 */
} rfo_call_args;

typedef struct rfo_batch rfo_batch;

/* <metric>::BatchComparator::new */
rfo_batch *rfo_batch_new(int metric, const uint8_t *s1, size_t len1);
void rfo_batch_free(rfo_batch *b);
/* the cached BlockPatternMatchVector: 256 x block_count u64, row-major [c * block_count + b] */
const uint64_t *rfo_batch_pm(const rfo_batch *b, size_t *block_count);

/* BatchComparator::{distance,similarity}_with_args for usize metrics; returns 1 = Some(*out), 0 = None */
int rfo_batch_usize(const rfo_batch *b, int op, const uint8_t *s2, size_t len2, const rfo_call_args *a, size_t *out);
/* BatchComparator::{normalized_*}_with_args for usize metrics, all four ops for jaro / jaro_winkler,
 * RatioBatchComparator::similarity_with_args for RFO_FUZZ_RATIO */
int rfo_batch_f64(const rfo_batch *b, int op, const uint8_t *s2, size_t len2, const rfo_call_args *a, double *out);
/* the free functions <metric>::{distance,similarity,...}_with_args / fuzz::ratio_with_args */
int rfo_free_usize(int metric, int op, const uint8_t *s1, size_t len1, const uint8_t *s2, size_t len2,
                   const rfo_call_args *a, size_t *out);
int rfo_free_f64(int metric, int op, const uint8_t *s1, size_t len1, const uint8_t *s2, size_t len2,
                 const rfo_call_args *a, double *out);
/* instrumentation: which levenshtein kernel the last call on this thread ended in (RFO_PATH_* in rfo_common.h) */
int rfo_last_lev_path(void);
/* instrumentation: in the last banded multi-word LCS call on this thread (lcs_seq.rs:297-331), how many rows moved the band's last block with
 * ceil_div(row + 1 + band_width_left, 64) at a multiple of 64 while a block was left -- the precondition of quirk Q8 (0 = the call cannot have lost a match) */
unsigned rfo_last_lcs_q8_edges(void);

/* the user loop `for c in corpus { scorer.f(c) }`; None is UINT64_MAX / NaN; nthreads splits the
 * candidates into contiguous ranges (1 = what the single-threaded reference does) */
void rfo_batch_many_usize(const rfo_batch *b, int op, const uint8_t *bytes, const uint64_t *offsets, size_t n,
                          const rfo_call_args *a, uint64_t *out, int nthreads);
void rfo_batch_many_f64(const rfo_batch *b, int op, const uint8_t *bytes, const uint64_t *offsets, size_t n,
                        const rfo_call_args *a, double *out, int nthreads);
void rfo_batch_rows_usize(const rfo_batch *b, int op, const uint8_t *rows, size_t n, size_t len, size_t stride,
                          const rfo_call_args *a, uint64_t *out, int nthreads);
void rfo_batch_rows_f64(const rfo_batch *b, int op, const uint8_t *rows, size_t n, size_t len, size_t stride,
                        const rfo_call_args *a, double *out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
