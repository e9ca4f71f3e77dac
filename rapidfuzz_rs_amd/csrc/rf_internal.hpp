// rf_internal.hpp -- shared between the host side of the C ABI (rf_api*.hip, rf_host.hpp) and the gfx950 kernels
// (rf_scan.hip, rf_long.hip, rf_jaro.hip, rf_pack.hip; device helpers in rf_device.hpp).  Product code: never includes or links anything from oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/rfgpu.h"

namespace rf {

constexpr int kWave = 64;          // CDNA4 wavefront: one candidate per lane
constexpr int kChunk = 16;         // bytes per lane per global_load_dwordx4
constexpr int kWavesPerBlock = 4;  // 256-thread workgroups, one wave per SIMD
constexpr int kMaxWords = 8;       // query <= 512 symbols keeps VP/VN register-resident
constexpr uint32_t kPad = 0xFFFFFFFFu;
constexpr int kMaxMulti = 4;       // queries fused into one scan_multi_kernel launch

// One tile = 64 candidates of identical length `len`, stored chunk-interleaved:
// byte b of lane r lives at data_off + ((b / 16) * 64 + r) * 16 + (b % 16).
struct TileDesc {
    uint64_t data_off;  // byte offset of the tile payload in the packed buffer
    uint32_t len;       // candidate length (same for every lane of the tile)
    uint32_t slot0;     // index of lane 0 in the slot arrays (orig[])
};

// One MIXED tile: 64 leftover candidates of neighbouring lengths sharing one payload block sized for the longest
// (rf_api.hip HostLayout).  Seen by scan_kernel_mixed with a per-lane length; every other kernel sees it through
// ordinary one-length TileDesc views.
struct MixedDesc {
    uint64_t data_off;          // the shared payload block
    uint32_t max_len, min_len;  // over the real lanes
    uint32_t slot0;             // first entry of this tile in mixed_len / mixed_orig (64 per tile)
    uint32_t pad;
};

enum RawKind : uint32_t {
    RAW_LEV = 0,   // uniform Levenshtein distance (Myers/Hyyro)
    RAW_LCS = 1,   // LCS length (Hyyro)
    RAW_JARO = 2,  // Jaro flags + transpositions
    RAW_WF = 4,    // generalized-weights Levenshtein, Wagner-Fischer rows in LDS
    RAW_OSA = 3    // optimal string alignment distance (Hyyro + transposition term)
};

// how the raw per-candidate primitive becomes the reference's return value
enum Finish : uint32_t {
    FIN_LEV = 0,    // dist = d * factor,             maximum = lev _maximum(len1, len2, weights)
    FIN_LCS = 1,    // dist = max(len1,len2) - l,     maximum = max(len1, len2)
    FIN_INDEL = 2,  // dist = len1+len2 - 2l,         maximum = len1 + len2
    FIN_JARO = 3,
    FIN_JW = 4,
    FIN_LEV_INDEL = 5,  // levenshtein with weights (f, f, >= 2f): dist = (len1+len2-2l)*f, maximum = lev _maximum
    FIN_LEV_GENERAL = 6 // any other weight table: dist = raw (Wagner-Fischer), maximum = lev _maximum (not affine)
};

struct ScanParams {
    const uint8_t* data;
    const TileDesc* tiles;  // nullptr: every tile has length uniform_len and sits at t * uniform_tile_bytes
    const uint32_t* orig;  // slot -> original index (kPad for padding lanes); nullptr = identity
    const MixedDesc* mixed;        // scan_kernel_mixed: the mixed section, tiles [tile_begin, tile_end) of it
    const uint32_t* mixed_len;     // per lane: candidate length
    const uint32_t* mixed_orig;    // per lane: original index, kPad = no candidate
    const uint64_t* pm;    // device PM table, 256 x words, row-major [c * words + w], indexed by ORIGINAL symbol
    const uint8_t* sigma;  // device uint8[256]: original symbol -> the symbol stored in the packed corpus
    const uint8_t* heads8; // small-cutoff scans: the candidates' first 8 symbols, tile t at t * 512 B (rf_pack.hip); nullptr = none
    // the same 8 symbols at 6 bits each (single-length corpora that store fewer than 64 distinct symbols): tile PAIR q at q * 768 B as
    // three dword planes of 64 lanes -- lane l holds candidates 2l, 2l + 1 of the pair: A[31:0] | A[47:32] + B[15:0] << 16 | B[47:16]
    // (rf_pack.hip head6_plane_kernel; head_filter_kernel reads 6 instead of 8 bytes per candidate).  nullptr = none
    const uint32_t* heads6;
    // the whole payload of a single-length corpus at 6 bits per symbol: 16 symbols in 12 bytes, chunk k of lane r of tile t at ((t * nch + k) * 64 + r) * 12
    // (rf_pack.hip pack6_kernel; corpora that store fewer than 64 distinct symbols; a length that is not a whole number of chunks is filled up with the code 63,
    // which the corpus then must not store: max_stored_sym < 63, and the scans zero that table row).  The single-word LCS scans stream it instead of `data`.  nullptr = none
    const uint32_t* data6;
    // A LENGTH RUN of a length-bucketed corpus seen as a single-length corpus (rf_api_scan.hip launch_scan_runs): tiles == nullptr, data /
    // heads8 point at the run's first tile, tile indices and idx = t * 64 + lane are relative to it, and run_orig[idx] is the
    // candidate's original index (kPad = padding lane).  `out` is pre-filled with None: dead tiles store nothing, survivors go
    // through run_orig.  nullptr everywhere else.
    const uint32_t* run_orig;
    void* out;             // uint32_t* or double*
    uint32_t n_tiles;
    uint32_t n;            // number of real candidates
    uint32_t uniform_len;
    uint32_t uniform_tile_bytes;
    uint32_t len1;
    uint32_t words;
    uint32_t finish;       // Finish
    uint32_t op;           // rf_op
    uint32_t out_f64;      // 1: out is double*, 0: uint32_t*
    uint32_t has_cutoff;
    uint32_t cutoff_u32;   // usize cutoff clipped to u32 (valid when has_cutoff and !out_f64)
    uint32_t factor;       // common weight factor (levenshtein.rs:1307-1327)
    uint32_t w_ins, w_del, w_sub;  // for _maximum (levenshtein.rs:263-277)
    uint32_t tile_begin, tile_end;  // tile range of this launch (jaro kernels; the cutoff length window of the scans)
    uint32_t n_exact;               // tiles below this index are exact-length tiles; above: one-length views of mixed tiles
    uint32_t zero_begin[2], zero_end[2];  // the (at most two) runs of zero-length tiles in tile order: one per ascending section (plan())
    uint32_t mixed_begin, mixed_end;  // the mixed tiles (indices into `mixed`) a Levenshtein / LCS / OSA scan has to visit
    uint32_t joint_begin, joint_end;  // scan_kernel_mixed: exact tiles to walk in the same launch, before the mixed ones (small corpora)
    const double* jaro_tab;         // jaro kernels: device tables [65][33] of (c - h) / c, then [130][65] of c / len2 (rf_api.hip jaro_device_table); nullptr = compute
    double jaro_need;               // jaro kernels: the similarity a candidate must reach to pass the cutoff; < 0 = no early-out
    uint32_t wf_query[16];          // wf_reg_kernel: the (renamed) query bytes, 4 per word, for queries of <= 64 symbols
    uint32_t wf_global;             // wf_kernel: the DP row lives in long_scratch (global) instead of LDS: queries beyond ~590 symbols
    uint32_t wf_waves;              // wavefronts per workgroup of wf_kernel (LDS rows per wavefront: (len1 + 1) * 256 B)
    uint32_t tile_step;             // >= 1: visit every tile_step-th tile of the range (the top-k bound sample)
    uint32_t head_need, head_k;     // head-plane cutoff scans: >= head_need of the first 8 symbols must have a partner within head_k positions (0 = filter off)
    // head_filter_kernel's product: the tiles a cutoff scan still has to walk ([0] = their number, then the tiles); tile_list_buf is
    // the scratch the launcher may use for it (n_tiles + 1 words), tile_list / tile_list_count what early_lean_kernel reads
    uint32_t* tile_list_buf;
    const uint32_t* tile_list;
    const uint32_t* tile_list_count;
    // LANE COMPACTION of the head-plane cutoff scans (round 6, rf_sparse.hip): head_filter_kernel attaches to every tile it lists the 64-bit mask of the LANES that
    // passed its tests (every other candidate gets its None there and then), the pack kernel numbers the survivors, and sparse_lean_kernel scans dense tiles of 64
    // survivors each, every lane reading its own candidate's chunk rows -- a corpus in which 2 % of the candidates share the query's head keeps 73 % of its TILES
    // alive and 2 % of its lanes.  lane_list = 1: tile_list_buf is large enough for the 16-byte entries (corpus_tile_list); 0 = round 5's tile list.
    uint32_t lane_list;
    const uint32_t* lane_first;  // sparse_lean_kernel: per dense tile j the packed entry that holds survivor 64 j (lane_list_pack_kernel writes it)
    // ... and where the survivors' results go when the caller wants no dense vector (rf_filter_*): lane_val[g] (u32 or f64 by out_f64; None = beyond the cutoff) and
    // lane_idx[g] = candidate index for survivor g < lane_cap (the number of survivors -- tile_list_buf[1] -- may exceed lane_cap: the host then takes another road)
    void* lane_val;
    uint32_t* lane_idx;
    uint32_t lane_cap;
    uint32_t exp_flags;             // measurement switches (bit 0: RF_EXP_NOHBM on the head-plane scans)
    uint32_t slot_store;            // 1: `orig` is the slot -> slot identity of the gather path (run_many: results into a slot-ordered temporary), so a kernel may
                                    // store lane l of a tile at out[slot0 + l] without reading it -- padding lanes included: the temporary has a slot for them and
                                    // the gather never looks there.  A hint: kernels that do not know it read `orig` as always (stream_body and the asm tiles kernels honour it)
    uint32_t xcd_deal;              // 1: workgroup w takes the tiles of virtual workgroup (w % 8) * (grid / 8) + w / 8: consecutive tiles stay on one XCD (one L2)
    uint32_t prefill_none;          // tiles outside the range are all None: out is pre-filled with RF_NONE_U32
    uint32_t prefill_window;        // 1: pre-fill ONLY the slots of the tiles [tile_begin, tile_end) (slot-ordered results whose reader looks nowhere else: rf_filter_*)
    uint32_t jaro_split;   // first EXACT tile that needs the multi-word jaro path (n_exact = none)
    uint32_t jaro_split2;  // the same for the one-length views of the mixed section, tiles [n_exact, n_tiles)
    uint32_t jaro_long;    // 1: some string exceeds 512 symbols: the multi-word tiles run jaro_long_kernel (flags in long_scratch)
    uint32_t query_head;   // first four query bytes, little endian, zero padded (Winkler prefix)
    uint32_t max_stored_sym;  // largest stored symbol of the corpus payload, exact (0xFFFFFFFF = not known): < 64 lets the Jaro asm kernel use its conflict-free table
    double cutoff_f64;
    double prefix_weight;
    // finishing coefficients (see "Finishing" in rf_device.hpp): value = vS*S + vM*Mx + vR*raw, dist / maximum likewise
    int32_t fin_vS, fin_vM, fin_vR, fin_dS, fin_dM, fin_dR, fin_mS, fin_mM;
    uint32_t fin_flip, fin_cflip;
    // many queries x one corpus (scan_multi_kernel): Q single-word tables, out is [Q][n]
    uint32_t multi_q;
    uint32_t multi_len1[kMaxMulti];
    const uint64_t* multi_pm[kMaxMulti];
    // patterns longer than kMaxWords * 64 symbols (long_kernel): PM rows padded to long_words_pad (multiple of 8)
    uint32_t long_words_pad;  // 0 = register-resident kernels
    uint32_t long_chunks_max; // scratch strip length per wavefront, in 16-column chunks
    uint32_t long_grid;       // workgroups (bounded by the scratch budget)
    uint32_t* long_scratch;   // [grid * 4][long_chunks_max][64]
    // value-preserving early-out under a distance cutoff (levenshtein, u32 distance output / top-k)
    uint32_t early;
    uint32_t first_check;           // column of the first early-out look inside a tile's first chunk: 4, 6 or 8 (plan())
    uint32_t narrow_look;           // early_lean_kernel: run the columns before the first look on 32-bit words (set by the launcher)
    // band_kernel (rf_band.hip): long query, raw distance cutoff band_k with 2 * band_k + 1 <= 64; band = 1 selects it
    uint32_t band, band_k;
    uint32_t band_asm;              // 1: full eight-column diagonal runs as one asm block (rf_band_asm.inc; set by the launcher)
    uint32_t band_defer_at, band_defer_max, band_defer_after;  // != 0 (set by the launcher): a tile with <= band_defer_max lanes within break_score at column band_defer_at is listed for band_sparse_kernel, once the launch has seen band_defer_after of them
    uint32_t* band_defer_seen;      // that count (the last words of the stream's tile-list buffer; band_sparse_kernel zeroes it again)
    uint32_t* band_report;          // pinned host words (or nullptr): band_sparse_kernel leaves [0] tiles listed, [1] lanes listed, [2] tiles of the launch, [3] 1, [4] band_defer_at, [5] band_defer_max, [6] the launch's candidate length; [8..10]: sparse_words_kernel's (lanes listed, candidates, 1)
    uint32_t band_list;             // 1 (score_hint, first pass): the band launch lists every tile that holds lanes it answered None, with their mask (tile_list_buf), for launch_sparse_words
    // the multi-word asm scans (rf_stream_asm.hip, tools/gen_stream_asm.py BlockKind): raw distances above trim_k1 - 1 need not be exact (they must come out above
    // it), which narrows the Ukkonen band the kernels trim their word-columns to; 0 = no bound beyond max(len1, len2)
    uint32_t trim_k1;
    // top-k mode (topk_k != 0): no per-candidate output, one k-entry key list per workgroup
    uint32_t topk_k;       // <= 64
    uint32_t topk_desc;    // 1: larger score is better (similarity)
    uint32_t key_index_base;  // added to the local index inside the key (rf_topk_keys_device)
    uint64_t* topk_bound;  // one u64, initialised to ~0: launch-wide upper bound on the k-th best key
    uint64_t* topk_cand;   // candidate keys: 64 way segments of topk_seg_cap keys; key = (score or ~score) << 32 | (key_index_base + local index)
    uint32_t topk_seg_cap; // keys per way segment (>= workgroups per way x 64)
    uint32_t* topk_ctl;    // control block: 65 lines of 128 bytes (per-way {arrivals, candidate count}, root arrivals) -- rf_device.hpp
    uint64_t* topk_root;   // 64 x 64 keys: the sub-collectors' selections
    uint64_t* topk_out;    // k keys, best first, ~0 = empty: the launch's result
    uint32_t topk_bound_from_result;  // sample pass: leave (k-th best key + 1) in *topk_bound for the main scan
};

// kernel launchers (rf_scan.hip, rf_long.hip, rf_jaro.hip, rf_pack.hip)
hipError_t launch_scan(RawKind raw, const ScanParams& p, hipStream_t stream, int* grid_used);
hipError_t launch_osa1_asm(const ScanParams& p, hipStream_t stream, int grid);   // the same around the OSA column (OsaState<1>)
hipError_t launch_lev32_asm(const ScanParams& p, hipStream_t stream, int grid);  // the same for queries of <= 32 symbols (Lev32State)
bool stream_asm_serves(const ScanParams& p);  // rf_stream_asm.hip: whole-kernel asm scans (u32 results, single word, no early-out / top-k)
hipError_t launch_stream_asm(int kind, const ScanParams& p, hipStream_t stream, int grid);  // kind: 0 Lev64, 1 Lev32, 2 OSA; no zero-length tile in [tile_begin, tile_end)
hipError_t launch_lev1_asm(const ScanParams& p, hipStream_t stream, int grid);  // rf_lev_asm.hip: single-word Levenshtein, single-length corpus, no early-out
void launch_lev1_asm_probe(dim3 g, dim3 b, uint32_t* out, int iters, uint32_t seed);  // rf_lev_asm.hip: the asm chunk alone (rf_probe_issue_rate mode 2)
// rf_hint.hip: what a score_hint pass left unresolved, gathered into dense tiles (rf_api_scan.hip run_many_hinted)
size_t hint_scan_temp_bytes(uint32_t n_tiles);
hipError_t launch_hint_sample(const ScanParams& p, const uint32_t* out, uint32_t tile_begin, uint32_t tile_end, uint32_t step, uint32_t* acc, hipStream_t st);
hipError_t launch_hint_mark(const ScanParams& p, uint32_t* out, uint32_t raw_cutoff, uint32_t zero_value, uint64_t* mask, uint32_t* count, uint32_t* prefix, void* temp,
                            size_t temp_bytes, const uint32_t* run_first, uint32_t R, uint32_t* run_prefix, hipStream_t st);
hipError_t launch_hint_gather(const ScanParams& p, const uint32_t* run_first, uint32_t R, const uint32_t* run_prefix, const uint32_t* prefix, const uint64_t* mask,
                              const uint32_t* run_tile_base, const uint64_t* run_data_base, const uint32_t* run_len, uint32_t n_tiles2, uint8_t* data2, TileDesc* tiles2,
                              uint32_t* orig2, hipStream_t st);
hipError_t launch_band(const ScanParams& p, hipStream_t stream);  // rf_band.hip: exact tiles [tile_begin, tile_end)
bool band_list_geometry(const ScanParams& p, uint32_t** packed_at, uint32_t** first_at);  // rf_band.hip: where a p.band_list launch leaves its packed list / first[] in p.tile_list_buf
// rf_sparse.hip: the lane compaction of the head-plane cutoff scans (ScanParams::lane_list)
hipError_t launch_sparse_lean(int state_kind, const ScanParams& p, hipStream_t stream);  // state_kind: 0 LevState<1>, 1 Lev32State, 2 OsaState<1>; p.tile_list = the packed 16-byte entries
hipError_t launch_sparse_words(const ScanParams& p, hipStream_t stream);  // rf_sparse.hip: the multi-word Levenshtein scan (p.words = 2..8, single-length corpus) over the listed lanes
bool head_two_pass_applies(RawKind raw, const ScanParams& p);  // rf_scan.hip: will launch_scan take head_filter_kernel + a second pass for this launch?
hipError_t launch_lane_list_pack(uint32_t* buf, uint32_t G, uint32_t cap, uint32_t* first_of, hipStream_t stream);  // rf_scan.hip lane_list_pack_kernel over G segments of `cap` entries
hipError_t launch_scan_mixed(RawKind raw, const ScanParams& p, hipStream_t stream);  // p.mixed / tile_begin / tile_end: the mixed section
hipError_t launch_long(RawKind raw, const ScanParams& p, hipStream_t stream, int grid);
hipError_t launch_wf(const ScanParams& p, hipStream_t stream);
hipError_t launch_jaro(const ScanParams& p, hipStream_t stream);
hipError_t launch_scan_multi(RawKind raw, bool narrow, const ScanParams& p, hipStream_t stream);
hipError_t launch_topk_final(const uint64_t* keys, uint32_t count, uint32_t k, uint64_t* out, hipStream_t stream);
hipError_t launch_pack_rows(const uint8_t* rows, size_t n, uint32_t len, size_t stride, uint8_t* packed,
                            uint32_t n_tiles, const uint8_t* sigma, hipStream_t stream);
hipError_t launch_translate(const void* raw, uint32_t raw_elem, uint64_t n_bytes, const uint32_t* keys, const uint8_t* vals, uint32_t cap, uint8_t* out,
                            hipStream_t stream);
// rf_pack_ragged.hip: the device half of rf_corpus_pack (ragged host input)
hipError_t launch_ragged_lengths(const uint64_t* offsets, uint32_t n, uint32_t max_len_allowed, uint32_t* keys, uint32_t* vals, unsigned long long* counts, uint32_t* status,
                                 hipStream_t st);  // status[0] = longest length, status[1]: bit 0 = offsets decrease, bit 1 = a candidate longer than max_len_allowed
hipError_t launch_ragged_byte_hist(const uint8_t* bytes, uint64_t first, uint64_t total, uint64_t stride, unsigned long long* hist, hipStream_t st);
size_t ragged_sort_temp_bytes(uint32_t n);
hipError_t launch_tiles_by_origin(const uint32_t* orig, const TileDesc* tiles, uint32_t z, uint32_t n_exact, uint32_t* keys, uint32_t* vals, uint32_t* keys2, uint32_t* vals2,
                                  void* temp, size_t temp_bytes, TileDesc* out, hipStream_t st);
hipError_t launch_ragged_sort(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, uint32_t bits, void* temp, size_t temp_bytes,
                              hipStream_t st);
hipError_t launch_ragged_scatter_tiles(const uint8_t* bytes, uint64_t first, const uint64_t* offsets, const uint32_t* sorted_idx, const TileDesc* tiles, uint32_t uniform_len, uint32_t n_exact,
                                       const uint32_t* by_len, const uint32_t* g_start, const uint32_t* g_slot0, const uint32_t* g_in_exact, const uint8_t* sigma, uint8_t* packed,
                                       uint32_t* orig, hipStream_t st);
hipError_t launch_ragged_scatter_mixed(const uint8_t* bytes, uint64_t first, const uint64_t* offsets, const uint32_t* sorted_idx, const MixedDesc* mixed, uint32_t n_mixed, uint32_t pool_n,
                                       const uint32_t* pool_len, const uint32_t* pool_spos, const uint32_t* pool_vslot0, const uint8_t* sigma, uint8_t* packed, uint32_t* orig,
                                       uint32_t* mixed_orig, hipStream_t st);
int scan_max_grid();
// results of a ragged corpus in original order without scattered stores (rf_pack.hip): slot -> slot / candidate -> slot maps, and the gather
hipError_t launch_slot_maps(const uint32_t* orig, uint32_t n_slots, uint32_t* slot_of, uint32_t* ident, hipStream_t stream);
hipError_t launch_head8_plane(const uint8_t* data, uint32_t n_tiles, uint32_t tile_bytes, uint8_t* heads, hipStream_t stream);  // rf_pack.hip: the candidates' first 8 symbols
hipError_t launch_max_byte(const uint8_t* data, uint64_t bytes, uint32_t* out, hipStream_t stream);  // rf_pack.hip: largest stored symbol (*out must start at 0)
hipError_t launch_head6_plane(const uint8_t* heads8, uint32_t n_tiles, uint32_t* heads6, hipStream_t stream);
hipError_t launch_pack6(const uint8_t* data, uint32_t n_tiles, uint32_t len, uint32_t* data6, hipStream_t stream);  // rf_pack.hip: the payload at 6 bits per symbol
hipError_t launch_head8_plane_tiles(const uint8_t* data, const TileDesc* tiles, uint32_t n_tiles, uint8_t* heads, hipStream_t stream);  // the same over tile descriptors
// the coalesced gather (rf_pack.hip "window_gather_kernel"): windows of kGatherWindow original indices, at most kMaxGatherRuns runs
constexpr uint32_t kGatherWindow = 4096;
constexpr uint32_t kMaxGatherRuns = 512;
constexpr uint32_t kGatherOff16Mod = 8192;  // the 2-byte slot offsets (launch_slot_off16): original index mod this -- one period holds a workgroup's span
hipError_t launch_run_starts(const uint32_t* orig, uint32_t n_slots, uint32_t* list, uint32_t cap, uint32_t* count, hipStream_t stream);
hipError_t launch_window_table(const uint32_t* orig, const uint32_t* runs, uint32_t n_runs, uint32_t n_rows, uint32_t* table, hipStream_t stream);
hipError_t launch_window_gather(const void* tmp, const uint32_t* orig, const uint16_t* off16, const uint32_t* table, uint32_t n_runs, uint32_t n_rows, void* out, uint32_t n,
                                bool f64, hipStream_t stream);  // off16 (launch_slot_off16) or, nullptr, orig
// u32 distances -> the f64 a normalized op returns (rf_pack.hip normalize_kernel: emit_fin's arithmetic with one maximum)
hipError_t launch_normalize(const uint32_t* dist, const uint32_t* len_of, uint32_t uniform_len, double* out, uint32_t n, uint32_t len1, int32_t fin_mS, int32_t fin_mM, uint32_t op,
                            uint32_t has_cutoff, double cutoff, hipStream_t stream);  // len_of: the candidates' lengths in original order, or nullptr = uniform_len
hipError_t launch_len_of(const TileDesc* tiles, uint32_t n_tiles, const uint32_t* orig, uint32_t* len_of, hipStream_t stream);
hipError_t launch_slot_off16(const uint32_t* orig, uint32_t n_slots, uint16_t* off16, hipStream_t stream);
hipError_t launch_gather_results(const void* tmp, const uint32_t* slot_of, void* out, uint32_t n, bool f64, hipStream_t stream);
// exact selection over a device score vector (rf_select.hip)
hipError_t launch_select_minmax(const void* s, bool f64, uint32_t n, bool desc, void* ctl, hipStream_t st);
hipError_t launch_select_hist(const void* s, bool f64, uint32_t n, bool desc, uint64_t prefix_mask, uint64_t prefix, uint32_t shift, uint32_t bits,
                              unsigned long long* hist, hipStream_t st);
uint32_t select_blocks(uint32_t n);
hipError_t launch_topk_scores(const ScanParams& p, const uint32_t* scores, uint32_t n, hipStream_t st);  // rf_select.hip: top-k (k <= 64) of a u32 score vector in one pass; p: the topk_* fields
hipError_t launch_keys_to_entries(const uint64_t* keys, uint32_t k, uint64_t index_base, rf_topk_entry* out, hipStream_t st);  // rf_select.hip
hipError_t launch_merge_entries(const rf_topk_entry* in, uint32_t n, uint32_t k, rf_topk_entry* out, hipStream_t st);
hipError_t launch_select_count(const void* s, bool f64, uint32_t n, bool desc, uint64_t T, uint32_t* cnt_less, uint32_t* cnt_eq, hipStream_t st);
hipError_t launch_select_emit(const void* s, bool f64, uint32_t n, bool desc, uint64_t T, const uint32_t* off_less, const uint32_t* off_eq, uint32_t n_less,
                              uint32_t need_eq, void* out_key, uint32_t* out_idx, hipStream_t st);
hipError_t launch_core_clock(uint64_t* d_out, uint32_t sleeps, hipStream_t stream);  // rf_probe.hip
hipError_t launch_probe(RawKind raw, uint32_t len1, uint32_t mode, int blocks_per_cu, int iters, double* wave_columns_per_ns);  // rf_probe.hip
hipError_t launch_histogram_rows(const uint8_t* rows, size_t n, uint32_t len, size_t stride, unsigned long long* hist,
                                 hipStream_t stream);
// rf_filter.hip: order-preserving compaction of a result vector into (index, score) pairs, and their ordering (rf_api_filter.hip)
uint32_t filter_segments(uint32_t m_bound);
size_t filter_scan_temp_bytes(uint32_t n_seg);
hipError_t launch_filter_compact(const void* val, bool f64, const uint32_t* map, uint32_t map_from, uint32_t m_bound, const uint32_t* m_dev, uint32_t* seg, void* temp,
                                 size_t temp_bytes, uint32_t capacity, uint32_t* out_idx, void* out_val, hipStream_t st);
hipError_t launch_filter_small(const void* val, bool f64, const uint32_t* map, uint32_t m_bound, const uint32_t* m_dev, bool by_score, bool desc, uint32_t capacity,
                               uint64_t index_base, uint64_t* out_index, void* out_val, uint32_t* res, uint32_t seq, const uint32_t* aux_dev, hipStream_t st);  // everything in one workgroup when the entries are few
hipError_t launch_filter_report(const uint32_t* a_dev, const uint32_t* aux_dev, uint32_t* res, uint32_t seq, hipStream_t st);
uint32_t filter_small_max();
size_t filter_select_work_bytes();
hipError_t launch_filter_select(const void* val, bool f64, const uint32_t* map, uint32_t m_bound, const uint32_t* m_dev, bool by_score, bool desc, uint32_t capacity,
                                uint64_t index_base, uint64_t* out_index, void* out_val, void* ws, uint32_t* res, uint32_t seq, const uint32_t* aux_dev, uint32_t expected, hipStream_t st);
size_t filter_sort_temp_bytes(uint32_t count);
hipError_t launch_filter_sort_by_index(const uint32_t* idx_in, const void* val_in, bool f64, uint32_t count, uint32_t* idx_out, void* val_out, void* temp, size_t temp_bytes,
                                       hipStream_t st);
hipError_t launch_filter_sort_by_score(const uint32_t* idx_in, const void* val_in, bool f64, bool desc, uint32_t count, void* key_in, void* key_out, uint32_t* idx_out,
                                       void* temp, size_t temp_bytes, hipStream_t st);
hipError_t launch_filter_finish(const uint32_t* idx, const void* val, const void* key, bool f64, bool desc, uint32_t count, const uint32_t* count_dev, uint64_t index_base,
                                uint64_t* out_index, void* out_val, hipStream_t st);
int scan_grid(uint32_t n_tiles);       // the grid of short-running launches over n_tiles tiles
int scan_grid_full(uint32_t n_tiles);  // the grid of full (no-cutoff) scans; >= scan_grid

void set_error(const std::string& msg);

}  // namespace rf
