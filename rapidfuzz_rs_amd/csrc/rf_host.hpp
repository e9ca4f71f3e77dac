// rf_host.hpp -- what the translation units of the C ABI share (round 4: rf_api.hip was one 3 500-line file):
//   rf_api.hip        error text, comparators, lowering of u32 queries, corpus packing and layout, probes
//   rf_api_scan.hip   plan(): Args x metric x op -> kernel parameters; rf_many_* / rf_one_* / rf_many_multi_* and what they launch
//   rf_api_topk.hip   top-k: in-scan lists, selection, 8-byte keys and 16-byte entries, the RCCL exchange
//   rf_api_files.hip  corpus files (save / load / validate) and streamed scans
// Host code only; nothing here is visible outside librfgpu.so.
#pragma once
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <atomic>
#include <unordered_map>
#include <unordered_set>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "rf_internal.hpp"

namespace rf {


#define RF_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                              \
            return _e == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP;                                \
        }                                                                                              \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (dev < 0) return;  // nothing to select (e.g. an empty corpus: the call never touches a device)
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace rf

using namespace rf;

struct rf_comparator {
    rf_metric metric;
    std::vector<uint8_t> s1;
    std::vector<uint64_t> pm;  // 256 x words, row-major [c * words + w]; words = max(1, block_count)
    size_t block_count = 0;
    size_t words = 1;
    mutable std::mutex mu;
    mutable std::map<int, uint64_t*> d_pm;  // lazily uploaded per device
    // u32 ("char") queries: the symbols, and one byte-level comparator per wide corpus searched (see resolve())
    bool wide = false;
    std::vector<uint32_t> s1w;
    // (shared_ptr: a call keeps the lowered comparator it runs on alive even if another host thread's call evicts it from
    // this bounded cache meanwhile -- handles may be shared between threads)
    mutable std::map<uint64_t, std::shared_ptr<rf_comparator>> lowered;
};

struct rf_corpus {
    int device = 0;
    size_t n = 0;
    uint64_t payload_bytes = 0;
    uint64_t device_bytes = 0;
    uint64_t data_bytes = 0;     // packed tile payloads + the tail pad chunk row
    bool borrowed = false;       // a view (stream segment, translated image): owns none of its device buffers
    bool no_prefill = false;     // stream segments: the driver pre-fills the whole result vector once
    uint8_t* d_data = nullptr;
    TileDesc* d_tiles = nullptr;
    TileDesc* d_tiles_by_origin = nullptr;  // the same descriptors with the non-empty exact tiles ordered by their first candidate's original index (tiles_by_origin())
    uint32_t* d_orig = nullptr;  // nullptr = identity (single length bucket, original order)
    size_t n_slots = 0;          // entries of d_orig: 64 per tile (exact tiles, then the views)
    // large ragged corpora return their results through a slot-ordered temporary + one gather (rf_pack.hip): built on first use
    mutable uint8_t* d_heads8 = nullptr;       // head plane: the first 8 symbols of every candidate (small-cutoff scans; built on first use)
    mutable uint32_t* d_heads6 = nullptr;      // the same at 6 bits per symbol (single-length corpora of < 64 distinct symbols; ScanParams::heads6)
    mutable bool heads6_tried = false;
    mutable uint32_t* d_data6 = nullptr;       // the payload at 6 bits per symbol (single-length corpora of < 64 distinct symbols; ScanParams::data6); built on first use
    mutable bool data6_tried = false;
    mutable uint32_t max_stored_sym = 0xFFFFFFFFu;  // largest stored symbol of the payload, exact; 0xFFFFFFFF = not computed yet (corpus_max_stored_symbol)
    mutable uint32_t* d_slot_of = nullptr;     // candidate -> its slot
    mutable uint32_t* d_slot_ident = nullptr;  // slot -> slot, kPad on padding lanes (stands in for d_orig in such a launch)
    mutable uint32_t* d_len_of = nullptr;      // candidate -> its length (original order): the normalizing pass of run_many's two-step path; built on first use
    mutable uint16_t* d_slot_off16 = nullptr;  // slot -> original index mod kGatherOff16Mod (0xFFFF: padding): what the window gather reads instead of d_orig
    mutable uint32_t* d_window_table = nullptr;  // the coalesced gather's table (rf_pack.hip window_table_kernel): gather_rows x gather_runs
    mutable uint32_t gather_runs = 0, gather_rows = 0;
    uint32_t n_tiles = 0;        // exact tiles, then the virtual (one-length) views of the mixed section
    uint32_t n_exact = 0;        // tiles [0, n_exact) are exact-length tiles; [n_exact, n_tiles) virtual views (HostLayout)
    // the mixed section as the Levenshtein / LCS / OSA scans see it: one tile of 64 leftovers with per-lane lengths
    uint32_t n_mixed = 0;
    MixedDesc* d_mixed = nullptr;
    uint32_t* d_mixed_len = nullptr;   // 64 per mixed tile
    uint32_t* d_mixed_orig = nullptr;  // 64 per mixed tile, kPad = no candidate
    std::vector<MixedDesc> mixed;      // host copy (length windows of cutoff runs)
    uint32_t max_len = 0;
    bool uniform = false;        // single length bucket: no descriptors, tile t at t * tile_bytes(uniform_len)
    uint32_t uniform_len = 0;
    std::vector<uint32_t> lengths;  // the distinct candidate lengths (host copy, ascending)
    std::vector<uint32_t> length_first_tile;  // first tile of each distinct length
    uint8_t sigma[256];           // symbol renaming: the packed corpus stores sigma[c] for candidate byte c
    float sym_freq[256] = {0};    // relative frequency of candidate byte c (from the histogram sigma is made of; all zero = unknown)
    uint8_t* d_sigma = nullptr;   // device copy
    // top-k scratch, one per stream the corpus has been searched on: [candidate keys by way | root table | bound | counters].  The
    // kernels leave bound/counters re-armed, so a top-k call is two launches (one under a tight cutoff) and no
    // allocation or memset (topk_core()).
    struct TopkScratch {
        uint64_t* cand = nullptr;  // (also the base of the allocation)
        uint64_t* root = nullptr;
        uint64_t* bound = nullptr;
        uint32_t* ctl = nullptr;
        uint32_t seg_cap = 0;
        hipEvent_t done = nullptr;   // recorded behind the last call's launches on this scratch (what a take-over waits for)
        uint32_t* scores = nullptr;  // score vector of the scan + one-pass top-k (rf_api_topk.hip), kept like the rest; own allocation
        size_t scores_cap = 0;       // in candidates
    };
    mutable std::mutex scratch_mu;
    mutable std::map<hipStream_t, TopkScratch> topk_scratch;
    mutable std::vector<hipStream_t> topk_lru;  // the streams of topk_scratch, most recently used first (rf_api_topk.hip: at most 8 hold a scratch)
    // A top-k call is two launches that hand state to each other through the scratch (sample scan -> bound -> scan, each
    // selecting in its last workgroup and re-arming it).  Host threads sharing a stream must not interleave those sequences: the enqueue
    // section of topk_core() runs under this lock (kernels of one stream then execute in enqueue order).
    mutable std::mutex topk_enqueue_mu;
    // The gather path's slot-ordered temporary (run_many), one per stream the corpus has been scanned on that way, kept for the
    // corpus' lifetime: a stream-ordered hipMallocAsync / hipFreeAsync pair per call made the SUBMISSION of such a step wait for the
    // previous step (measured: 560 us per call to submit a 575 us step; 11 us with the buffer kept).  A call's scan + gather are
    // enqueued under gather_enqueue_mu: host threads sharing a stream must not interleave two uses of the same buffer.
    struct GatherTmp {
        hipStream_t stream;
        void* ptr;
        size_t bytes;
        hipEvent_t done;  // recorded behind the gather that last read the buffer; the next use makes its stream wait for it (free on the stream the
                          // buffer was used on, and what orders a NEW stream that was handed a destroyed stream's handle value: ADVICE r4)
    };
    mutable std::vector<GatherTmp> gather_tmp;
    mutable std::mutex gather_enqueue_mu;
    // head_filter_kernel's tile lists (rf_scan.hip), one per stream such a cutoff scan has run on: 8 bytes per tile.  The filter
    // pass and the scan that walks its list are enqueued under filter_enqueue_mu (host threads sharing a stream).
    struct TileList {
        hipStream_t stream;
        uint32_t* ptr;
        hipEvent_t done;  // recorded behind the last scan that walked the list (corpus_tile_list_done): a take-over waits for THIS, never
                          // for the stream handle -- its owner may have destroyed the stream long ago, and the runtime crashes on a dead handle
        volatile uint32_t* band_report = nullptr;  // pinned, 64 bytes: what this stream's last hand-over launch of the small-band scan listed (rf_band.hip band_sparse_kernel
                                                   // writes it; run_many reads it, without waiting, to choose the next launch's form); nullptr = no pinned memory
        uint32_t band_plain_calls = 0;             // launches since that choice last fell on the plain kernel
    };
    mutable std::vector<TileList> tile_lists;
    mutable std::mutex filter_enqueue_mu;
    mutable std::atomic<uint32_t> filter_last_survivors{0};  // rf_filter_*: how many candidates the last lane-compacted call's first pass left (few: one workgroup finishes the call)
    mutable std::atomic<uint32_t> hint_trust{0};  // score_hint scans (run_many_hinted): consecutive hinted calls whose first pass resolved >= 70 % of this corpus -- while it counts, the sample (a launch and a host synchronization) is taken on every 16th call only
    // u32 ("char") corpora: the stored byte is the symbol's id in THIS corpus' alphabet.  Ids 0..253 are the 254 most
    // frequent symbols, kOverflowId lumps every rarer symbol together, kAbsentId is never stored (see resolve()).
    bool wide = false;
    uint64_t uid = 0;
    std::unordered_map<uint32_t, uint8_t> alphabet;
    std::unordered_set<uint32_t> overflow;
    // Only when `overflow` is not empty: the u32 symbol behind every packed byte (d_raw[x] belongs to d_data[x];
    // 0xFFFFFFFF in padding).  A query containing overflow symbols is served from a per-call byte image translated from
    // it (Effective below) -- exact, at the price of one extra pass over 4 bytes per symbol.
    void* d_raw = nullptr;
    uint32_t raw_elem = 4;  // bytes per raw symbol: 2 when every symbol of the corpus is <= 0xFFFE (padding 0xFFFF), else 4 (padding 0xFFFFFFFF)
    mutable uint8_t* d_sigma_identity = nullptr;  // for those images (their bytes are query-local ids, not renamed)
    const rf_corpus* parent = nullptr;            // set on such an image: scratch and locks live in the real corpus
};
constexpr uint8_t kOverflowId = 254, kAbsentId = 255;
// RF_FLAG_SLOT_ORDER is honoured by rf_many_u32 / rf_many_f64 ONLY (their `out` is sized by rf_corpus_slot_count); every other entry point writes n-entry rows and must
// never see it.  run_many therefore looks at an INTERNAL bit that only those two entry points (and rf_filter_*, for its own temporary) set; the bit is stripped from
// whatever a caller passes in.
constexpr uint32_t kFlagSlotsInternal = 0x40000000u;
// ... and (rf_filter_* only, with the bit above): the caller reads nothing outside the slots of the cutoff's length window, so nothing there needs its None
constexpr uint32_t kFlagWindowInternal = 0x20000000u;
inline rf_args sanitized_args(const rf_args* a, bool slots_allowed)
{
    rf_args r = *a;
    r.flags &= ~(kFlagSlotsInternal | kFlagWindowInternal);
    if (slots_allowed && (r.flags & RF_FLAG_SLOT_ORDER)) r.flags |= kFlagSlotsInternal;
    return r;
}
__attribute__((visibility("hidden"))) extern std::atomic<uint64_t> g_corpus_uid;

// Symbol renaming.  Every column of every kernel gathers 64 table rows from LDS, one per lane, and LDS bank
// conflicts between DIFFERENT symbols that share a bank (row index mod 32 for 8-byte rows) are the cost of that
// gather -- ASCII classes collide systematically ('A'/'a', digits/'P'..'Y').  Renaming symbols by frequency rank
// gives the 32 most frequent symbols of THIS corpus 32 distinct banks and pairs the rest with them one by one.
// The packed corpus stores sigma(c); the kernels stage PM row c at LDS row sigma(c); nothing else changes.
using ComparatorRef = std::shared_ptr<rf_comparator>;
// What a call actually runs on: the comparator lowered to the corpus' ids and the corpus itself -- or, for a u32 query
// that contains overflow-class symbols, a comparator over QUERY-LOCAL ids (1..r in order of first appearance) and a
// per-call byte image of the corpus translated from its raw symbol stream (translate_kernel: query symbol -> its id,
// anything else -> 0).  The image is a borrowed view (same tiles / slot map) that lives until the object goes out of
// scope; its payload is released in stream order.
extern "C" __attribute__((visibility("hidden"))) void scratch_free(void* p, hipStream_t st);  // rf_scratch.hip
struct Effective {
    const rf_comparator* c = nullptr;
    ComparatorRef hold;  // keeps a lowered comparator alive for the duration of the call
    const rf_corpus* corpus = nullptr;
    std::unique_ptr<rf_corpus> image;
    uint8_t* temp = nullptr;
    hipStream_t stream = nullptr;
    std::vector<uint32_t> keys;
    std::vector<uint8_t> vals;
    ~Effective()
    {
        if (temp) scratch_free(temp, stream);
    }
};
constexpr size_t kTailPad = (size_t)kWave * kChunk;  // one readable chunk row past the last tile
static inline uint64_t tile_bytes(uint32_t len) { return (uint64_t)((len + kChunk - 1) / kChunk) * kWave * kChunk; }
#define RF_HIP_C(expr)                                                                                 \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                              \
            return fail(_e == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP);                          \
        }                                                                                              \
    } while (0)

// Every exported rf_status function is a function-try-block ending in this: no C++ exception may cross the C ABI (a Rust or C caller
// cannot unwind it; host-side std::vector growth is where one would come from).
#define RF_ABI_CATCH                                                                                    \
    catch (const std::bad_alloc&)                                                                       \
    {                                                                                                   \
        rf::set_error("out of host memory");                                                            \
        return RF_ERR_OOM;                                                                              \
    }                                                                                                   \
    catch (const std::exception& ex)                                                                    \
    {                                                                                                   \
        rf::set_error(std::string("internal error: ") + ex.what());                                     \
        return RF_ERR_INVALID_ARG;                                                                      \
    }

// ---- shared between the translation units (definitions: the file named in the comment; they sit inside the files' extern "C" blocks)
#pragma clang diagnostic ignored "-Wreturn-type-c-linkage"
// (hidden: these are internal to librfgpu.so -- only the rf_* entry points of include/rfgpu.h are exported)
#define RF_LOCAL __attribute__((visibility("hidden")))
extern "C" {
// the library's stream-ordered scratch allocator (rf_scratch.hip: why it is not hipMallocAsync).  scratch_free parks the block behind
// everything enqueued on `st` so far; neither call ever waits.
RF_LOCAL hipError_t scratch_alloc(void** out, size_t bytes, hipStream_t st);
RF_LOCAL void scratch_free(void* p, hipStream_t st);
RF_LOCAL void scratch_trim(void);
RF_LOCAL void symbol_frequencies(const uint64_t* hist, float* freq);                                                      // rf_api.hip
RF_LOCAL rf_status resolve(const rf_comparator* c, const rf_corpus* corpus, const rf_comparator** eff, ComparatorRef* hold, bool* overflow_hit = nullptr);  // rf_api.hip
RF_LOCAL rf_status make_effective(const rf_comparator* c_in, const rf_corpus* corpus, hipStream_t st, Effective* e);     // rf_api.hip
RF_LOCAL const double* jaro_device_table(int device);                                                                     // rf_api.hip
RF_LOCAL size_t pm_stride(const rf_comparator* c);                                                                        // rf_api.hip
RF_LOCAL rf_status comparator_device_pm(const rf_comparator* c, int device, const uint64_t** d_out);                      // rf_api.hip
RF_LOCAL std::vector<TileDesc> tiles_by_origin(const std::vector<TileDesc>& tiles, uint32_t n_exact, const uint32_t* orig);  // rf_api.hip
RF_LOCAL rf_status plan(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, bool f64_out, ScanParams* p, RawKind* raw);  // rf_api_scan.hip
RF_LOCAL const uint32_t* corpus_head6_plane(const rf_corpus* corpus, hipStream_t st);                                  // rf_api_scan.hip
RF_LOCAL const uint32_t* corpus_data6(const rf_corpus* corpus, hipStream_t st);                                        // rf_api_scan.hip
RF_LOCAL const uint8_t* corpus_head8_plane(const rf_corpus* corpus, const ScanParams& p, RawKind raw, hipStream_t st);    // rf_api_scan.hip
RF_LOCAL void plan_band_filter(const rf_comparator* c, const rf_corpus* corpus, rf_op op, bool f64_out, ScanParams* p, uint32_t len2);  // rf_api_scan.hip
RF_LOCAL void corpus_tile_list_done(const rf_corpus* corpus, hipStream_t st);                                           // rf_api_scan.hip
RF_LOCAL uint32_t* corpus_tile_list(const rf_corpus* corpus, hipStream_t st);                                             // rf_api_scan.hip
RF_LOCAL void corpus_lane_buffers(const rf_corpus* corpus, ScanParams* p);                                                // rf_api_scan.hip
RF_LOCAL rf_status run_many(const rf_comparator* c_in, const rf_corpus* corpus_in, rf_op op, const rf_args* args, void* out, rf_mem out_mem, void* stream, bool f64_out);  // rf_api_scan.hip
}  // extern "C"
