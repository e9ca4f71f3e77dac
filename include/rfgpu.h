/*
 * rfgpu.h -- C ABI of the MI355X (gfx950) one-vs-many fuzzy matching engine.
 *
 * Drop-in boundary for the batch path of rapidfuzz-rs v0.5.0 (`rapidfuzz::distance::*::BatchComparator`
 * and `rapidfuzz::fuzz::RatioBatchComparator`).  The reference has no FFI of its own -- its boundary is
 * the generic Rust API -- so each entry point below names the reference item it replaces
 * (file:line relative to the reference root).  INTEGRATION.md shows the Rust `extern "C"` block and the
 * safe wrapper a maintainer would put behind the original signatures.
 *
 * Plain pointers and sizes only; no torch / HIP types (a `hipStream_t` travels as `void*`).
 * Elements are `u8` (the `HashableChar` case `u8 -> Hash::UNSIGNED`, src/details/common.rs:34) or, through the *_u32
 * entry points, `char` / u32.
 *
 * Error model: metric evaluation never fails in the reference (no `Result` on this path; "above the
 * cutoff" is `None`, src/common.rs:43-45).  rf_status reports ENGINE failures only (bad argument, HIP
 * error, shape the device kernels do not cover).  There is NO CPU fallback: a shape that has no kernel
 * returns RF_ERR_UNSUPPORTED instead of silently computing on the host.
 *
 * Threading: handles are immutable after creation and may be shared between host threads (the
 * reference's comparators are Send + Sync and take &self); each call uses the stream it is given.
 *
 * Environment variables.  The library reads the following variables ONCE per process (first use).  Every one of them selects
 * between kernels / layouts / launch shapes that return IDENTICAL results -- they exist for A/B measurements and for the
 * forced-path test runs (tests/multitile_check.py) -- and none is needed in production; the default is what ships.
 * There is no variable that changes a result: the measurement switch that does (RF_EXP_NOHBM) is compiled only into
 * -DRF_EXPERIMENTS builds (tools/build_stream_variant.sh), never into librfgpu.so (tests/test_abi.py checks the binary).
 *   name                          default   meaning
 *   RF_SCAN_BLOCKS_PER_CU         32        workgroups per CU of short-running launches (cutoff scans, band kernel, long queries)
 *   RF_SCAN_BLOCKS_PER_CU_FULL    256       most workgroups per CU of full (no-cutoff) scans
 *   RF_SCAN_TILES_PER_WAVE        5         tiles per wavefront the grid of a full scan aims at (below the cap above; at least 32 workgroups per CU)
 *   RF_STREAM                     1         0: scan_body instead of the streaming loop of the compiled scans
 *   RF_ASM_STREAM                 1         0: compiled scans instead of the whole-kernel asm scans (rf_stream_asm.hip)
 *   RF_ASM_CHUNK                  1         0: compiled chunk instead of the hybrid asm chunks (in-scan top-k, Jaro)
 *   RF_ASM_BLOCK                  1         0: compiled multi-word scan instead of the asm multi-word scan (queries of 65..512 symbols)
 *   RF_ASM_BAND                   1         0: the compiled column everywhere in the small-band kernel instead of the asm block for full 8-column diagonal runs (rf_band_asm.inc)
 *   RF_EARLY_STATIC / RF_EARLY_LEAN / RF_NARROW_LOOK / RF_HEAD_TWO_PASS / RF_HEAD_LOOK_PASS
 *                                 1         0: the older form of the cutoff scans' first look
 *   RF_LANE_COMPACT               1         0: the second pass of the head-plane cutoff scans walks the surviving TILES (64 lanes each, round 5) instead of the
 *                                           surviving candidates gathered 64 to a wavefront (rf_sparse.hip; rf_filter_* then always takes its general path)
 *   RF_FIRST_CHECK                0 (auto)  4..16: column of the cutoff scans' first look
 *   RF_HEAD8_MIN                  16384     fewest tiles for which a corpus gets an 8-symbol head plane (0: never)
 *   RF_HEAD6                      1         0: no 6-bit head plane (single-length corpora of < 64 distinct symbols stream 6 instead of 8 bytes
 *                                           per candidate through the band prefilter; costs 6 more bytes per candidate of HBM)
 *   RF_BAND_FILTER                -1 (auto) 0 / 1: band prefilter of the head-plane scans never / whenever applicable
 *   RF_NO_BAND                    unset     set: multi-word scan + early-out instead of the band kernel
 *   RF_NORM_BAND                  1         0: normalized_* of a long-query Levenshtein scan under an f64 cutoff that leaves <= 31 raw edits runs the compiled f64
 *                                           early-out scan instead of the small-band kernel under the implied raw cutoff + a normalizing pass
 *   RF_HINT_LISTS                 1         0: a credible score_hint scan of a single-length corpus marks, sums, sizes on the host, copies and re-scans what its band
 *                                           pass left (round 5) instead of letting that pass list it for a scan over the list (rf_sparse.hip sparse_words_kernel)
 *   RF_HINT_TRUST                 1         0: every score_hint scan of a large corpus samples it first (a launch and a host synchronization); 1: not while the
 *                                           corpus' last hinted scans resolved >= 70 % of it (every 16th call looks again)
 *   RF_BAND_DEFER                 1         0: the small-band scan of a single-length corpus runs every tile to its end on 64 lanes instead of handing tiles with few
 *                                           lanes left to a dense second pass (rf_band.hip launch_band)
 *   RF_BAND_RUNS                  1         0: the small-band scan of a length-bucketed corpus is one launch over its tiles; 1: its long runs of one length are walked
 *                                           as single-length corpora of their own (with the hand-over above) while that pays, the rest by the tiles kernel
 *   RF_BAND_RUN_MIN_TILES         32768     fewest tiles of one length for such a run (its launch sequence costs ~70 us whatever its size)
 *   RF_BAND_DEFER_AT              0 (auto)  column (a multiple of 16) at which a tile may be handed over; auto: k + 8 rounded up, 16..64
 *   RF_BAND_DEFER_MAX             44        most lanes still within the band for a tile to be handed over
 *   RF_BAND_DEFER_AFTER           0         such tiles a launch runs in place before it starts handing over (counted in one device word; 0: no count)
 *   RF_BAND_DEFER_ADAPT           1         0: every such launch hands over; 1: by what the stream's last hand-over launch listed (the plain kernel where that saved
 *                                           less than a quarter of the columns, looking again every 16th launch)
 *   RF_TILE_ORDER                 2         0..3: how a length-bucketed corpus' results reach original order
 *   RF_UNSCATTER_MIN              1048576   fewest candidates for the slot-ordered temporary + gather pass
 *   RF_GATHER_WINDOWS             1         0: gather_results_kernel instead of the window gather
 *   RF_NORM_TWO_STEP              1         0: normalized ops of the multi-word Levenshtein scans (queries of 65 .. 512 symbols) in the compiled f64 scan instead of
 *                                           the u32 asm scan + one normalizing pass (+ 4 bytes per candidate kept per length-bucketed corpus: the lengths in original order)
 *   RF_GATHER_XCD                 1         0: the window gather's spans go to workgroups round-robin instead of one eighth of them per XCD
 *   RF_GATHER_OFF16               1         0: the window gather reads orig[] (4 bytes per slot) instead of its 2-byte window offsets (+ 2 bytes per slot kept per corpus)
 *   RF_GATHER_SPAN / RF_GATHER_UNROLL   16384 / 8   window gather tuning
 *   RF_TOPK_VIA_SCORES            1         top-k (k <= 64) as scan + one pass over the scores: 0 never, 1 multi-word Levenshtein, 2 every shape with an asm scan
 *   RF_TOPK_SAMPLE                1024      tiles of the in-scan top-k's bound sample (0: no sample pass)
 *   RF_JARO_PRIV                  0         1: Jaro asm kernel gathers from a conflict-free copy of the pattern table (corpora of <= 64 symbols; measured: no gain)
 *   RF_WF_REG                     1         0: LDS rows instead of register rows for generalized weights, queries <= 64
 *   RF_TRANSLATE_DIRECT           1         0: staged translation of u32 overflow symbols
 *   RF_NO_RENAME / RF_NO_MIXED_TILES   unset   set at PACK time: no symbol renaming / every length padded to whole tiles
 *   RF_RUN_MIN_TILES              256       fewest tiles of one length for which a small-cutoff scan of a length-bucketed corpus walks
 *                                           that length run as a single-length view (head plane, band prefilter)
 *   RF_JOINT_MAX_TILES            16384     a length-bucketed corpus of at most this many exact tiles is scanned in ONE launch (exact and
 *                                           mixed tiles together; the launch is the cost there); 0: always two launches
 *   RF_SCRATCH_CACHE_MB           1024      bound on the per-call scratch the library keeps parked between calls (its own stream-ordered
 *                                           allocator, rf_scratch.hip); 0: every block is released as soon as the work behind it is done
 *   RF_PACK6                      1         0: no 6-bit copy of the payload (corpora of < 64 distinct symbols -- < 63 for a single-length corpus whose length is not a
 *                                           multiple of 16 -- keep one, + 75 % of the payload in HBM, built by the first Indel / LCS / fuzz::ratio scan without an
 *                                           early-out cutoff: that scan then streams 12 instead of 16 bytes per 16 symbols -- rf_stream_asm.hip
 *                                           stream_lcs6[n]_uniform_kernel, and stream_lcs6[n]_tiles_kernel for length-bucketed corpora, u32 results)
 *   RF_PACK6_MIN_TILES            16384     fewest tiles of a corpus for which that copy is made
 *   RF_DEVICE_PACK_MIN            65536     fewest candidates for which rf_corpus_pack does its per-candidate work on the device (upload of the raw bytes + offsets, length
 *                                           histogram, stable radix sort by length, one scatter kernel per destination tile: rf_pack_ragged.hip) instead of the host's
 *                                           counting sort + byte scatter + one upload of the packed image; 0: always the host packer.  The same layout either way, byte
 *                                           for byte (rf_corpus_layout_host is the specification; candidates beyond 65535 symbols always take the host packer)
 *   RF_HINT_MIN_TILES             1024      fewest tiles of a corpus for which a per-candidate Levenshtein scan of a query beyond 64 symbols honours
 *                                           score_hint (pass under max(hint, 31), then only what it left unresolved: rf_hint.hip); 4294967295: never
 *   RF_HINT_SAMPLE_MIN_TILES      16384     fewest tiles for which such a call first runs the hint pass over every (tiles / 512)-th tile and drops the hint when fewer
 *                                           than 70 % of the candidates are within it (a wrong hint then costs ~3 % instead of up to + 55 %); 4294967295: never sample
 *   RF_STREAM_KEEP                1         0: rf_stream_many_* allocates and frees its pinned / device buffer sets per call instead of keeping them
 *   RF_STREAM_THREADS             16        host threads that read a corpus file's payload (rf_stream_many_*, rf_corpus_load)
 *   RF_PACK_TIMING / RF_SELECT_DEBUG / RF_TRACE_PLAN / RF_STREAM_TIMING   unset   set: phase timings / selection statistics / one line per
 *                                           rf_many_* call naming the path its plan took / the phases of a streamed scan, on stderr
 */
#ifndef RFGPU_H
#define RFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rf_status;
enum {
    RF_OK = 0,
    RF_ERR_INVALID_ARG = 1,
    RF_ERR_HIP = 2,         /* a HIP runtime call failed; rf_last_error() has the text */
    RF_ERR_UNSUPPORTED = 3, /* no device kernel for this shape (never falls back to the CPU, never approximates) */
    RF_ERR_NO_DEVICE = 4,
    RF_ERR_OOM = 5
};

/* the metric modules of src/distance.rs:1-10 that have a bit-parallel batch path, + src/fuzz.rs */
typedef enum rf_metric {
    RF_LEVENSHTEIN = 0,  /* src/distance/levenshtein.rs */
    RF_INDEL = 1,        /* src/distance/indel.rs */
    RF_LCS_SEQ = 2,      /* src/distance/lcs_seq.rs */
    RF_JARO = 3,         /* src/distance/jaro.rs */
    RF_JARO_WINKLER = 4, /* src/distance/jaro_winkler.rs */
    RF_FUZZ_RATIO = 5,   /* src/fuzz.rs RatioBatchComparator */
    RF_OSA = 6           /* src/distance/osa.rs (widening beyond the north-star path) */
} rf_metric;

/* the four methods every BatchComparator has (e.g. levenshtein.rs:1660-1817) */
typedef enum rf_op {
    RF_OP_DISTANCE = 0,
    RF_OP_SIMILARITY = 1,
    RF_OP_NORMALIZED_DISTANCE = 2,
    RF_OP_NORMALIZED_SIMILARITY = 3
} rf_op;

typedef enum rf_mem { RF_MEM_HOST = 0, RF_MEM_DEVICE = 1 } rf_mem;

#define RF_NO_CUTOFF UINT64_MAX   /* Args::default(): NoScoreCutoff (src/common.rs:3-30) */
#define RF_NONE_U32 0xFFFFFFFFu   /* Option::None in a u32 result slot */
/* Option::None in an f64 result slot is a quiet NaN (test with isnan) */

/*
 * Flattened `Args` builders: levenshtein.rs:86-126 (score_cutoff, score_hint, weights),
 * jaro_winkler.rs:25-62 (prefix_weight), and the identical score_cutoff/score_hint pairs of
 * lcs_seq / indel / jaro / fuzz.  Use rf_args_default() and then set fields.
 *   cutoff_usize : RF_NO_CUTOFF = NoScoreCutoff, else WithScoreCutoff(v) for the usize-valued ops
 *   cutoff_f64   : NaN = NoScoreCutoff, else WithScoreCutoff(v) for the f64-valued ops
 *   score_hint_* : results never depend on a hint (levenshtein.rs:2153-2160); in the reference it steers the CPU band search
 *                  (levenshtein.rs:1069-1088: a band of `hint`, doubled until the distance fits).  Two callers honour it:
 *                  rf_many_u32, Levenshtein RF_OP_DISTANCE with one common weight (the path that reads the hint in the reference), a
 *                  query beyond 64 symbols, max(hint, 31) below the cutoff and below the longest string: the corpus is first scanned
 *                  under the cutoff max(hint, 31) (the one-word band kernel, or the banded multi-word scans), then only the candidates
 *                  that left unresolved are gathered into dense tiles and scanned under the caller's own cutoff (rf_hint.hip) -- a
 *                  corpus of near-duplicates costs the band pass, an unrelated one the full scan + a few percent (corpora of >= 16384 tiles sample the
 *                  band pass first and drop a hint that is wrong for more than 30 % of the candidates).  Such a call may synchronize `stream`
 *                  once or twice between the passes (RF_MEM_DEVICE too); on a single-length corpus whose last hinted scans resolved >= 70 % of
 *                  it, a hint of <= 31 runs without any: the band pass lists what it leaves and the caller's scan walks the list.
 *                  rf_topk_u32 with RF_OP_DISTANCE and no cutoff: the scan first runs under the
 *                  cutoff `hint` (a cutoff scan costs a fraction of a full one) and the k best are final if k candidates pass,
 *                  otherwise the hint doubles (past a quarter of the longest possible distance the plain scan runs).
 *                  Every other call ignores it.
 *
 * Three places where the device deliberately does NOT reproduce what release-mode rapidfuzz 0.5.0 returns (all tested,
 * tests/test_gpu_parity.py, all also in DESIGN.md section 3):
 *   Q7  For len1 > 64 and an explicit score_hint with 2 * max(hint, 31) < len1 - len2, the reference's hint-doubling loop
 *       (levenshtein.rs:1069-1088) calls the small-band kernel without the |len1 - len2| guard hyrroe2003_block has, and
 *       its result then DEPENDS on the hint (an upstream defect, reproduced by the oracle).  The device's hinted passes are exact and it
 *       returns the exact distance -- what the reference returns for every other hint.
 *   Q2  levenshtein similarity_with_args above its cutoff evaluates `maximum - usize::MAX` (details/distance.rs:209-210:
 *       a panic in debug builds, a wrapped value in release builds).  The device returns None.
 *   Q8  lcs_seq / indel / fuzz::ratio with a query of more than 64 symbols under a cutoff: the reference's banded multi-word LCS
 *       (lcs_seq.rs:297-331) moves the band's last block with ceil_div(row + 1 + band_width_left, 64) and so leaves out, for one
 *       row, the block that starts at bit row + 1 + band_width_left when that index is a multiple of 64.  A pair aligned along the
 *       band's edge there loses a match: the reference reports a similarity below the true LCS, or None for a pair that IS within
 *       the cutoff (an upstream defect, reproduced by the oracle: tests/golden/q8_lcs_band_pair.json).  The device walks every
 *       block and returns the exact value -- what the reference returns without the cutoff.
 */
typedef struct rf_args {
    uint64_t cutoff_usize;
    uint64_t score_hint_usize;
    double cutoff_f64;
    double score_hint_f64;
    uint64_t insertion_cost, deletion_cost, substitution_cost; /* WeightTable, levenshtein only */
    double prefix_weight;                                       /* jaro_winkler only (default 0.1) */
    uint32_t flags;                                             /* RF_FLAG_* */
    uint32_t reserved;
} rf_args;

/* RatioBatchComparator: compute the documented Indel ratio 1 - indel/(len1+len2) instead of reproducing
 * src/fuzz.rs:141, which normalises through the inner lcs_seq comparator (LCS/max(len1,len2)). */
#define RF_FLAG_RATIO_INDEL_NORMALIZATION 0x1u

/* rf_many_u32 / rf_many_f64 on a length-bucketed corpus: write the results in SLOT order -- out[s] belongs to candidate rf_corpus_slot_index()[s], `out` has
 * rf_corpus_slot_count() entries (>= n; slots without a candidate hold unspecified values) -- instead of original order.  The packed corpus keeps candidates grouped
 * by length, so original order costs a length-bucketed corpus either scattered stores or a gather pass over every result (a quarter of an HBM-bound scan:
 * profiles/ragged_indel_r05.txt); a caller that keeps the slot map once (record linkage joins on an id anyway) skips it.  No reference analogue: the reference's
 * loop has no order but the caller's.  A single-length corpus' slots are its indices (the flag changes nothing); for a u32 query with overflow-class symbols
 * rf_many_* refuses the flag with RF_ERR_UNSUPPORTED for such a query; every other entry point (rf_many_multi_*, rf_topk_*, rf_stream_many_*, rf_one_*, rf_filter_*) ignores it.  Values are those of the default call: tests/test_gpu_filter.py permutes and compares. */
#define RF_FLAG_SLOT_ORDER 0x2u

void rf_args_default(rf_args *a);

typedef struct rf_comparator rf_comparator; /* = <metric>::BatchComparator<u8> */
typedef struct rf_corpus rf_corpus;         /* a device-resident, length-bucketed packed candidate set */

/* text of the last failure on the calling thread ("" if none) */
const char *rf_last_error(void);
/* number of visible HIP devices (0 if none / no driver) */
int rf_device_count(void);

/* ---- comparator -------------------------------------------------------------------------------
 * rf_comparator_new = BatchComparator::new: copies the query and builds its BlockPatternMatchVector
 * (256 x ceil(len/64) u64, row-major [c * blocks + b]; src/details/pattern_match_vector.rs:203-224).
 *   levenshtein.rs:1645-1657, lcs_seq.rs:800-812, indel.rs:375-383, jaro.rs:830-842,
 *   jaro_winkler.rs:413-425, fuzz.rs:102-113 */
rf_status rf_comparator_new(rf_metric metric, const uint8_t *s1, size_t len1, rf_comparator **out);
/* The same over `char` / u32 elements (`BatchComparator::new(s.chars())`; the reference keeps non-ASCII symbols in the
 * hashed half of its table, pattern_match_vector.rs:5-65, :228-260).  Such a comparator is searched in corpora packed
 * with rf_corpus_pack_u32 (and in byte corpora if every query symbol is <= 255); a byte comparator may equally be
 * searched in a u32 corpus, its bytes being the code points 0..255.  rf_comparator_pm returns NULL for it: its
 * table is built per corpus, in that corpus' symbol ids. */
rf_status rf_comparator_new_u32(rf_metric metric, const uint32_t *s1, size_t len1, rf_comparator **out);
/* #[derive(Clone)] (levenshtein.rs:1635): deep copy */
rf_status rf_comparator_clone(const rf_comparator *c, rf_comparator **out);
void rf_comparator_free(rf_comparator *c);
rf_metric rf_comparator_metric(const rf_comparator *c);
size_t rf_comparator_query_len(const rf_comparator *c);
/* the host copy of the PM table; *block_count = ceil(len/64) */
const uint64_t *rf_comparator_pm(const rf_comparator *c, size_t *block_count);

/* ---- corpus -----------------------------------------------------------------------------------
 * The candidates a user would feed one by one to `scorer.distance(candidate)` (the loop in
 * rapidfuzz-benches/benches/bench_levenshtein.rs:51-60), packed once and kept in HBM.
 * Layout: candidates are grouped by exact length into tiles of 64 (one candidate per wavefront lane); what is left
 * over of each length (fewer than 64) is pooled, sorted by length, into mixed tiles whose lanes carry their own lengths,
 * so a corpus of few, long, all-different-length candidates packs to its payload instead of 64 lanes per length.
 * Inside a tile the 16-byte chunk k of lane r sits at tile_base + (k*64 + r)*16, so a wavefront's
 * `global_load_dwordx4` of "my chunk k" is one contiguous 1 KiB read.  Symbols are stored renamed by a
 * per-corpus permutation (frequency rank), which the kernels undo when they stage the PM table; results always
 * come back in the ORIGINAL candidate order and never depend on the renaming.
 *
 * rf_corpus_pack: ragged host input, candidate i = bytes[offsets[i] .. offsets[i+1]) (n+1 offsets).  From 65536 candidates on the input is uploaded as it is
 *   and length-bucketed on the device (100 M candidates of <= 64 symbols: host arrays to a scannable corpus in a fraction of a second, profiles/pack_r06.txt).
 * rf_corpus_pack_rows_device: n rows of `len` bytes already in device memory at d_rows + i*stride.
 * Inputs are borrowed for the duration of the call only.  n < 2^32 - 1 per corpus. */
rf_status rf_corpus_pack(const uint8_t *bytes, const uint64_t *offsets, size_t n, int device, rf_corpus **out);
rf_status rf_corpus_pack_rows_device(const void *d_rows, size_t n, size_t len, size_t stride, int device,
                                     void *stream, rf_corpus **out);
void rf_corpus_free(rf_corpus *c);
/* The same layout computed on the host only (no device needed): for tools and for the CPU-side tests of the
 * packer.  Arrays are malloc'ed; release with rf_host_layout_free. */
typedef struct rf_host_layout {
    uint8_t *packed;       /* packed_bytes: tile payloads (+ one zero chunk row of tail padding) */
    uint64_t *tile_off;    /* n_tiles: byte offset of each tile's payload */
    uint32_t *tile_len;    /* n_tiles: candidate length of each tile (ascending) */
    uint32_t *tile_slot0;  /* n_tiles: first slot of each tile */
    uint32_t *orig;        /* n_slots: slot -> original index, 0xFFFFFFFF = padding lane (empty if identity) */
    uint64_t packed_bytes, n_slots;
    uint32_t n_tiles, identity;
    uint8_t sigma[256];    /* symbol renaming: the payload stores sigma[c] for candidate byte c (a permutation that
                              spreads this corpus' frequent symbols over distinct LDS banks) */
    uint32_t n_exact;      /* tiles [0, n_exact) hold 64 candidates of one length each (whole multiples of 64 per length) */
    uint32_t n_mixed;      /* the leftovers of every length, sorted by length, share n_mixed MIXED payload blocks of 64 lanes;
                              tiles [n_exact, n_tiles) are the one-length views of those blocks (one per distinct length in a
                              block, same tile_off, orig = 0xFFFFFFFF for the lanes of other lengths) */
} rf_host_layout;
rf_status rf_corpus_layout_host(const uint8_t *bytes, const uint64_t *offsets, size_t n, rf_host_layout *out);
void rf_host_layout_free(rf_host_layout *l);
size_t rf_corpus_count(const rf_corpus *c);          /* n */
uint64_t rf_corpus_payload_bytes(const rf_corpus *c); /* sum of candidate lengths */
uint64_t rf_corpus_device_bytes(const rf_corpus *c);  /* HBM held by the packed form */
int rf_corpus_device(const rf_corpus *c);
/* Slots (RF_FLAG_SLOT_ORDER, and the storage order rf_filter_* may report in): rf_corpus_slot_count = entries of a slot-ordered result vector (n for a
 * single-length corpus; 64 per tile otherwise); rf_corpus_slot_index writes that many u32 to `out` (host or device memory): the original candidate index of every
 * slot, 0xFFFFFFFF for a slot that holds no candidate.  Synchronous. */
size_t rf_corpus_slot_count(const rf_corpus *c);
rf_status rf_corpus_slot_index(const rf_corpus *c, uint32_t *out, rf_mem out_mem);
/* Candidates over `char` / u32 elements: elems[offsets[i] .. offsets[i+1]) is candidate i (0xFFFFFFFF is reserved).
 * The corpus stores one byte per element -- the element's id in this corpus' own alphabet (the 254 most frequent
 * symbols; all rarer ones share one overflow id).  Every metric on this path only asks whether a candidate symbol
 * EQUALS a query symbol, so that byte image is exact for every query made of alphabet symbols and of symbols the
 * corpus does not contain at all.  A corpus with overflow symbols also keeps its raw symbol stream on the device
 * (2 more bytes per symbol inside the Basic Multilingual Plane, else 4); a query that contains overflow symbols is then served from a per-call byte image
 * translated from it (query symbol -> query-local id, anything else -> 0): still exact, one extra pass over the
 * stream per call (rf_stream_many_* ships each segment's slice of the raw stream with it, for those queries only).
 * Query symbols the corpus does not store share one never-matching id; refused with RF_ERR_UNSUPPORTED only: such a
 * query with more than 254 distinct symbols that the corpus DOES store (8-bit ids cannot tell them apart).
 * rf_corpus_alphabet_size: symbols with an id of their own (256 for a byte corpus), and how many share the
 * overflow id. */
rf_status rf_corpus_pack_u32(const uint32_t *elems, const uint64_t *offsets, size_t n, int device, rf_corpus **out);
size_t rf_corpus_alphabet_size(const rf_corpus *c, size_t *overflow_symbols);

/* ---- corpus files, corpora larger than HBM (no reference analogue) -----------------------
 * rf_corpus_save / rf_corpus_load: the packed form (header, length table, tile descriptors, slot -> original
 * index, alphabet of a u32 corpus, page-aligned payload) written once and mapped back without re-packing.
 * rf_stream_many_*: out[i] = scorer.<op>_with_args(candidate_i, &args) over a corpus FILE that need not fit in
 * HBM: the file is scanned in tile ranges of at most `segment_bytes` of payload (0 = 512 MiB) through three pinned-host +
 * device buffer sets (kept per process between calls: 3 x segment_bytes of pinned memory and of HBM; RF_STREAM_KEEP=0 frees them
 * per call), the read + upload of one segment overlapping the scan of the previous one and -- single-length corpora -- the copy of
 * the one before's results to the host: 47 GB/s of payload on a 64 GB file, link ceiling 57 (profiles/stream_r04.txt).  `out` is HOST memory
 * with room for `out_capacity` entries; the file's own candidate count n (rf_corpus_file_count) decides how many are
 * written, and a file holding more than out_capacity is refused with RF_ERR_INVALID_ARG before anything is written
 * (the result vector itself does live on the device during the pass).  Values, None encoding and errors are those of
 * rf_many_u32 / rf_many_f64.  Files are validated on load (section bounds, tile offsets, slot map): an inconsistent
 * file is RF_ERR_INVALID_ARG, never an out-of-bounds access. */
rf_status rf_corpus_save(const rf_corpus *c, const char *path);
rf_status rf_corpus_load(const char *path, int device, rf_corpus **out);
rf_status rf_corpus_file_count(const char *path, size_t *n);
rf_status rf_stream_many_u32(const rf_comparator *c, const char *path, rf_op op, const rf_args *args, uint32_t *out,
                             size_t out_capacity, uint64_t segment_bytes, int device);
rf_status rf_stream_many_f64(const rf_comparator *c, const char *path, rf_op op, const rf_args *args, double *out,
                             size_t out_capacity, uint64_t segment_bytes, int device);
/* Gives back what the library keeps per PROCESS between calls (no reference analogue; a long-lived service calls it after a burst):
 * the streamed scans' three pinned-host + device buffer sets (unless a streamed scan is running: its sets stay) and every block of
 * the stream-ordered scratch allocator whose last use has completed.  What a corpus keeps (acceleration structures, per-stream
 * temporaries) goes with rf_corpus_free.  Always RF_OK; later calls re-allocate what they need. */
rf_status rf_release_caches(void);

/* ---- one-vs-many ------------------------------------------------------------------------------
 * out[i] = scorer.<op>_with_args(candidate_i, &args) for every candidate, original order.
 *
 * rf_many_u32: the usize-valued methods of levenshtein / indel / lcs_seq / osa
 *   (distance_with_args levenshtein.rs:1750-1777, similarity_with_args :1790-1817; indel.rs:464-521;
 *    lcs_seq.rs:893-949).  RF_NONE_U32 = None.
 * rf_many_f64: normalized_* of those metrics (levenshtein.rs:1670-1737 ...), all four methods of
 *   jaro / jaro_winkler (jaro.rs:845-977, jaro_winkler.rs:428-575) and
 *   RatioBatchComparator::similarity_with_args (fuzz.rs:127-149; pass RF_OP_SIMILARITY).  NaN = None.
 *
 * out_mem says whether `out` is host or device memory.  With RF_MEM_DEVICE the call only enqueues work
 * on `stream` (hipStream_t, NULL = default stream); with RF_MEM_HOST it returns after the copy. */
rf_status rf_many_u32(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args,
                      uint32_t *out, rf_mem out_mem, void *stream);
rf_status rf_many_f64(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args,
                      double *out, rf_mem out_mem, void *stream);

/* The reference's per-candidate methods themselves -- `scorer.<op>_with_args(s2, &args)` for ONE candidate
 * (levenshtein.rs:1740-1817 and siblings): a one-candidate corpus through the same kernels, result on the host.
 * Convenience for drop-in call sites; a loop over candidates belongs in rf_many_*.  Returns RF_OK and sets *is_some
 * to 0 where the reference returns None. */
rf_status rf_one_u32(const rf_comparator *c, const uint8_t *s2, size_t len2, rf_op op, const rf_args *args, int device,
                     uint32_t *out, int *is_some);
rf_status rf_one_f64(const rf_comparator *c, const uint8_t *s2, size_t len2, rf_op op, const rf_args *args, int device,
                     double *out, int *is_some);

/* ---- thresholded one-vs-many: only the candidates within the cutoff ---------------------------------------------
 * The reference returns Option<T> per candidate (src/common.rs:18-46, :83-85) and the caller of a dedup / record-linkage loop keeps the Somes:
 *     corpus.iter().enumerate().filter_map(|(i, c)| scorer.<op>_with_args(c, &args).map(|v| (i, v)))
 * rf_filter_u32 / rf_filter_f64 are that filter_map: the (index, score) pairs of every candidate whose result is not None, instead of an n-entry vector that is
 * nearly all None (which is 0.4 of the 1.0 GB a cutoff-3 scan of 100 M candidates moves, and all of what crosses PCIe afterwards).  Metrics, ops, Args and values
 * are exactly those of rf_many_u32 / rf_many_f64; without a cutoff every candidate qualifies.
 *   index_base   added to every index (shards of one logical corpus)
 *   capacity     entries `out_index` / `out_score` have room for (either may be NULL when capacity is 0: a pure count)
 *   *out_count   (host) the number of candidates that passed -- ALWAYS the true number.  If it exceeds `capacity` the arrays hold `capacity` of the qualifying
 *                pairs (valid, in the requested order among themselves, not necessarily the first ones) and the caller repeats the call with room for
 *                *out_count: nothing is ever dropped silently.
 *   order        RF_FILTER_BY_INDEX: ascending index.  RF_FILTER_BY_SCORE: best first (ascending for the distance ops, descending for the similarity ops),
 *                ties by ascending index.  RF_FILTER_ANY: whatever the scan produces (storage order of the packed corpus: cheapest for length-bucketed corpora).
 *   out_mem      where out_index / out_score live.  The call synchronizes `stream` either way (the count comes back to the host).
 * How: cutoff scans that go through the head plane (Levenshtein / OSA, query <= 64, at most ~5 edits allowed, single-length corpora of >= 16384 tiles) hand the
 * surviving candidates of their first pass straight to a compaction -- no dense vector exists at any point (rf_sparse.hip); every other shape scans into a
 * device temporary (slot order for length-bucketed corpora: no gather pass) and compacts that (rf_filter.hip: order preserving, no atomics). */
typedef enum rf_filter_order { RF_FILTER_BY_INDEX = 0, RF_FILTER_BY_SCORE = 1, RF_FILTER_ANY = 2 } rf_filter_order;
rf_status rf_filter_u32(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args, uint64_t index_base, uint64_t capacity,
                        uint64_t *out_index, uint32_t *out_score, uint64_t *out_count, rf_mem out_mem, rf_filter_order order, void *stream);
rf_status rf_filter_f64(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args, uint64_t index_base, uint64_t capacity,
                        uint64_t *out_index, double *out_score, uint64_t *out_count, rf_mem out_mem, rf_filter_order order, void *stream);

/* ---- many queries x one corpus ------------------------------------------------------------------
 * The reference's user loop one level up: `for q in queries { let scorer = BatchComparator::new(q);
 * for c in corpus { scorer.<op>_with_args(c, &args) } }`.  out is row-major [q][n]: row j is exactly what
 * rf_many_u32 / rf_many_f64 writes for cs[j] (same op, same args for every query).  All comparators must
 * be usable with `op` the way rf_many_* requires.  Neighbouring comparators of one metric whose queries
 * are <= 64 symbols (levenshtein / indel / lcs_seq / fuzz ratio) are evaluated 4 (or 2) at a time by one
 * kernel that reads each candidate once per group; everything else falls back to one launch per query.
 * Memory/stream conventions as rf_many_*. */
rf_status rf_many_multi_u32(const rf_comparator *const *cs, uint32_t q, const rf_corpus *corpus, rf_op op,
                            const rf_args *args, uint32_t *out, rf_mem out_mem, void *stream);
rf_status rf_many_multi_f64(const rf_comparator *const *cs, uint32_t q, const rf_corpus *corpus, rf_op op,
                            const rf_args *args, double *out, rf_mem out_mem, void *stream);

/* ---- top-k ------------------------------------------------------------------------------------
 * The reference has no extract/top-k API; this is the engine's own reduction over the scores above,
 * defined as: evaluate every candidate with `op`/`args`, drop None, order by
 *   (score ascending for RF_OP_DISTANCE / descending for RF_OP_SIMILARITY, index ascending)
 * and keep the first k.  index = index_base + original candidate index, so shards of one logical
 * corpus produce globally comparable entries.  Outputs are HOST arrays of k entries (any k >= 1; up to 64 the lists are
 * kept inside the scan, beyond that the selection path of rf_topk_f64 is used);
 * *out_count <= k.  If out_all is not NULL the same pass also writes every candidate's score there (as
 * rf_many_u32 would; it stays on this GPU).  With a Levenshtein distance cutoff a wavefront stops reading a tile
 * as soon as all of its 64 candidates are provably beyond the cutoff.
 * rf_topk_merge_u32 merges `lists` such results (e.g. after an all-gather across GPUs). */
rf_status rf_topk_u32(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args, uint32_t k,
                      uint64_t index_base, uint32_t *out_score, uint64_t *out_index, uint32_t *out_count,
                      uint32_t *out_all, rf_mem out_all_mem, void *stream);
/* The same for f64-valued scores -- jaro / jaro_winkler / fuzz ratio, and normalized_* of the usize metrics -- and for
 * any k: every candidate is scored into a device vector (out_all if given) and the k best are SELECTED from it exactly
 * (radix selection on order-preserving keys; ties broken by index; None never selected).  Ascending for RF_OP_DISTANCE /
 * RF_OP_NORMALIZED_DISTANCE, descending for the similarity ops.  rf_topk_u32 takes the same path when k > 64 (one list
 * entry per wavefront lane is the limit of the in-scan lists), so k is bounded by the candidate count only. */
rf_status rf_topk_f64(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args, uint64_t k,
                      uint64_t index_base, double *out_score, uint64_t *out_index, uint64_t *out_count,
                      double *out_all, rf_mem out_all_mem, void *stream);
/* Fully asynchronous variant for pipelines that stay on the device (e.g. an RCCL all-gather right after):
 * writes k 64-bit keys to DEVICE memory, best first, empty entries = UINT64_MAX.
 *   key = (score << 32) | (index_base + index)             for RF_OP_DISTANCE   (ascending = best first)
 *   key = (~score << 32) | (index_base + index)            for RF_OP_SIMILARITY
 * so keys from different shards of one corpus merge by a plain unsigned sort.  index_base + n must fit 32 bits. */
rf_status rf_topk_keys_device(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args, uint32_t k,
                              uint32_t index_base, uint64_t *d_keys_out, uint32_t *out_all, rf_mem out_all_mem,
                              void *stream);
/* Device-side merge of n such keys (e.g. the all-gathered lists of all ranks) into the k smallest, best first;
 * asynchronous on `stream`.  d_out may not alias d_keys. */
rf_status rf_topk_merge_keys_device(const uint64_t *d_keys, uint32_t n, uint32_t k, uint64_t *d_out, int device,
                                    void *stream);
/* The whole exchange for a host that drives RCCL itself (C, C++, Rust): ncclAllGather of every rank's k keys
 * (d_local_keys, as written by rf_topk_keys_device) into d_all_keys (world * k entries, rank order) over the caller's
 * communicator, then the device-side merge into d_merged (k entries) -- both enqueued on `stream`, nothing synchronizes.
 * `nccl_comm` is the caller's ncclComm_t; RCCL is not a link dependency of this library: ncclAllGather is resolved at
 * run time in the RCCL instance already loaded in the process (the one that created the communicator).
 * RF_ERR_UNSUPPORTED when no RCCL can be found.  The per-candidate outputs never move: only k * 8 bytes per rank do. */
rf_status rf_topk_allgather_merge(const uint64_t *d_local_keys, uint32_t k, void *nccl_comm, uint32_t world,
                                  uint64_t *d_all_keys, uint64_t *d_merged, int device, void *stream);
rf_status rf_topk_merge_u32(rf_op op, const uint32_t *scores, const uint64_t *indices, const uint32_t *counts,
                            uint32_t lists, uint32_t k, uint32_t *out_score, uint64_t *out_index,
                            uint32_t *out_count);

/* ---- top-k entries: the device-resident exchange format for EVERY top-k the engine offers ---------------------------
 * rf_topk_keys_device packs (u32 score, 32-bit global index) into 8 bytes: k <= 64, usize-valued metrics, at most 2^32
 * candidates in the logical corpus.  An rf_topk_entry is the general form (a score and a 64-bit global index
 * per entry): 16 bytes that merge across shards by a plain unsigned comparison of (key, index), for u32 AND f64 scores,
 * any k, any index_base.
 *   key   : order-preserving image of the score, smaller = better --
 *           u32 scores: the score (RF_OP_DISTANCE) or 0xFFFFFFFF - score (RF_OP_SIMILARITY);
 *           f64 scores: the IEEE bits with the sign bit flipped (negative values: all bits flipped), complemented for the
 *           descending ops (rf_topk_entry_score_* below invert the map)
 *   index : index_base + original candidate index
 * Empty entries (fewer than k candidates passed): key = index = UINT64_MAX.
 * rf_topk_entries_device fills k entries in DEVICE memory, best first.  usize-valued metrics with k <= 64 take the in-scan
 * lists and are fully asynchronous on `stream`; every other shape (f64 scores, k > 64, queries beyond 512 symbols, general
 * weight tables) scores all candidates and selects (rf_topk_f64's path), which synchronizes the stream once.
 * rf_topk_merge_entries_device: the k best of n entries (e.g. all ranks' lists after an all-gather), asynchronous.
 * rf_topk_allgather_merge_entries: ncclAllGather of every rank's k entries over the caller's communicator + that merge
 * (RCCL resolved at run time like rf_topk_allgather_merge).  rf_topk_merge_entries: the same merge on host arrays. */
typedef struct rf_topk_entry {
    uint64_t key;
    uint64_t index;
} rf_topk_entry;
rf_status rf_topk_entries_device(const rf_comparator *c, const rf_corpus *corpus, rf_op op, const rf_args *args, uint64_t k,
                                 uint64_t index_base, rf_topk_entry *d_entries_out, void *stream);
rf_status rf_topk_merge_entries_device(const rf_topk_entry *d_entries, uint64_t n, uint64_t k, rf_topk_entry *d_out, int device,
                                       void *stream);
rf_status rf_topk_allgather_merge_entries(const rf_topk_entry *d_local, uint64_t k, void *nccl_comm, uint32_t world,
                                          rf_topk_entry *d_all, rf_topk_entry *d_merged, int device, void *stream);
rf_status rf_topk_merge_entries(const rf_topk_entry *entries, uint64_t n, uint64_t k, rf_topk_entry *out);
/* the score behind an entry's key (`descending` = the op was a similarity op) */
uint32_t rf_topk_entry_score_u32(uint64_t key, int descending);
double rf_topk_entry_score_f64(uint64_t key, int descending);

/* ---- measurement aid (not part of the drop-in surface) ----------------------------------------------------------
 * The single-word scans are bound by VALU issue, not HBM.  rf_probe_issue_rate runs the library's own
 * recurrence column for (metric, query_len) on register-resident pattern words -- no HBM, no LDS, no tile loop -- with
 * blocks_per_cu (0 = 8) 256-thread workgroups per CU and reports wavefront-columns per nanosecond, which for
 * 64-symbol candidates is the Gpairs/s no scan of that recurrence can exceed on this device.  bench.py reports it as
 * roofline.issue_bound, measured in the same process as the scan.  mode 0: pattern words from registers (the pure
 * recurrence); mode 1: the scans' own chunk code -- byte extraction and the pattern-table gather from LDS over 62 random
 * symbols -- still without HBM traffic or a tile loop; mode 2 (Levenshtein, 33..64 query symbols): the hand-scheduled asm
 * chunk of the headline kernel (rf_lev_asm.hip) alone, the same way.  Synchronous; uses the default stream. */
rf_status rf_probe_issue_rate(rf_metric metric, uint32_t query_len, uint32_t mode, int device, uint32_t blocks_per_cu,
                              double *wave_columns_per_ns);
/* The core clock the device runs at WHILE the caller's other streams are busy: one wavefront on a stream of its own sleeps a
 * known number of core cycles (s_sleep) for about `micros` microseconds between two readings of the constant-rate counter.
 * The issue ceiling above is measured with idle HBM (2.39 GHz on MI355X); a scan that streams HBM runs the same cycle count at
 * 2.05-2.16 GHz (power management), which is the whole gap between the two -- bench.py samples the clock beside back-to-back
 * scans and reports the ceiling at that clock.  ghz_sleep: from the s_sleep count; ghz_counter: from s_memtime (a constant-rate
 * counter on some parts: reported for reference).  Synchronizes only its own stream. */
rf_status rf_probe_core_clock(int device, uint32_t micros, double *ghz_sleep, double *ghz_counter);

#ifdef __cplusplus
}
#endif
#endif /* RFGPU_H */
