// rf_api_filter.hip -- rf_filter_u32 / rf_filter_f64: the (index, score) pairs of the candidates within the cutoff, and the slot map of a corpus (round 6).
// Product code: never includes or links anything from oracle/.
//
// reference: Option<T> per candidate (src/common.rs:18-46, :83-85); the user's filter_map over the corpus keeps the Somes (rfgpu.h has the contract).
// Two roads to the same pairs:
//   * the head-plane cutoff scans of a single-length corpus (rf_scan.hip head_filter_kernel, rf_sparse.hip): the first pass marks surviving lanes, the second
//     scans them 64 to a wavefront and leaves value-or-None + candidate index at each survivor's number; the compaction runs over THOSE (a few percent of n at
//     most) -- no n-entry vector is written or read;
//   * everything else: run_many into a device temporary -- in slot order for length-bucketed corpora (RF_FLAG_SLOT_ORDER: no gather pass) -- and the compaction
//     over it with the slot -> original index map.
// Then the order the caller asked for (results of a slot-ordered temporary need a sort by index; by-score is a stable sort on top), the widening to u64 indices,
// and one synchronization that brings the count home.
#include "rf_host.hpp"

extern "C" {

namespace {

struct ScratchSet {  // everything a call allocates, released in stream order on every way out
    hipStream_t st;
    std::vector<void*> blocks;
    ~ScratchSet()
    {
        for (void* b : blocks) scratch_free(b, st);
    }
    hipError_t get(void** p, size_t bytes)
    {
        const hipError_t e = scratch_alloc(p, std::max<size_t>(bytes, 256), st);
        if (e == hipSuccess) blocks.push_back(*p);
        return e;
    }
};

}  // namespace

// The first road.  Returns RF_OK with *took = false when the launch would not go through the lane compaction (the caller then takes the second road).
// On success: lane_val / lane_idx hold the survivors' results, *d_total (device) their number, cap2 the room they had.
static rf_status filter_fast(const rf_comparator* c_in, const rf_corpus* corpus_in, rf_op op, const rf_args* args, bool f64_out, uint64_t capacity, hipStream_t st,
                             ScratchSet& sc, bool* took, void** lane_val, uint32_t** lane_idx, uint32_t** d_total, uint32_t* cap2_out, std::unique_lock<std::mutex>* held)
{
    *took = false;
    static const bool lane_compact = [] { const char* e = getenv("RF_LANE_COMPACT"); return !e || atoi(e) != 0; }();
    if (!lane_compact || !corpus_in->uniform || corpus_in->borrowed || corpus_in->wide || c_in->wide) return RF_OK;
    const rf_comparator* c = c_in;
    const rf_corpus* corpus = corpus_in;
    ScanParams p;
    RawKind raw = RAW_LEV;
    if (const rf_status rs = plan(c, corpus, op, args, f64_out, &p, &raw); rs != RF_OK) return rs;
    if (!p.early || (raw != RAW_LEV && raw != RAW_OSA) || p.words != 1 || p.band || p.long_words_pad) return RF_OK;
    if (const rf_status rs = comparator_device_pm(c, corpus->device, &p.pm); rs != RF_OK) return rs;
    p.heads8 = corpus_head8_plane(corpus, p, raw, st);
    if (!p.heads8) return RF_OK;
    p.heads6 = corpus_head6_plane(corpus, st);
    plan_band_filter(c, corpus, op, f64_out, &p, corpus->uniform_len);
    std::unique_lock<std::mutex> filter_lock(corpus->filter_enqueue_mu);
    p.tile_list_buf = corpus_tile_list(corpus, st);
    corpus_lane_buffers(corpus, &p);
    if (!p.lane_list || !head_two_pass_applies(raw, p)) return RF_OK;
    // room for the survivors of the first pass (NOT the passers: a corpus that shares prefixes with the query has many more survivors than matches); a call whose
    // survivors do not fit takes the second road afterwards -- correct either way
    const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(corpus->n / 8, 4 * capacity), 1u << 16);
    const uint32_t cap2 = (uint32_t)std::min<uint64_t>(corpus->n, want);
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    RF_HIP(sc.get(lane_val, (size_t)cap2 * elem));
    RF_HIP(sc.get((void**)lane_idx, (size_t)cap2 * sizeof(uint32_t)));
    p.lane_val = *lane_val;
    p.lane_idx = *lane_idx;
    *d_total = p.tile_list_buf + 1;  // (the survivors' number, left there by the pack kernel; `held` keeps other host threads of this stream off the buffer until
                                     // everything that reads it has been enqueued)
    p.lane_cap = cap2;
    p.out = nullptr;
    p.prefill_none = 0;
    static const bool trace_plan = getenv("RF_TRACE_PLAN") != nullptr;
    if (trace_plan) std::fprintf(stderr, "[rf plan] filter: lane compaction, first_check=%u head_need=%u head_k=%u room for %u survivors\n", p.first_check, p.head_need, p.head_k, cap2);
    const hipError_t e = launch_scan(raw, p, st, nullptr);
    *held = std::move(filter_lock);
    if (e != hipSuccess) {
        set_error(std::string("filter scan launch: ") + hipGetErrorString(e));
        return RF_ERR_HIP;
    }
    *cap2_out = cap2;
    *took = true;
    return RF_OK;
}

static rf_status run_filter(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint64_t index_base, uint64_t capacity, uint64_t* out_index,
                            void* out_score, uint64_t* out_count, rf_mem out_mem, rf_filter_order order, void* stream, bool f64_out)
{
    if (!c || !corpus || !args || !out_count || (capacity && (!out_index || !out_score)) || (int)order < 0 || (int)order > (int)RF_FILTER_ANY) {
        set_error("rf_filter: null handle / args / count, null output with a non-zero capacity, or unknown order");
        return RF_ERR_INVALID_ARG;
    }
    *out_count = 0;
    {   // argument errors of the scan itself (metric x op x output type) before anything runs
        ScanParams p;
        RawKind raw = RAW_LEV;
        const rf_comparator* ce = nullptr;
        ComparatorRef hold;
        if (resolve(c, corpus, &ce, &hold) == RF_OK)
            if (const rf_status rs = plan(ce, corpus, op, args, f64_out, &p, &raw); rs != RF_OK) return rs;
    }
    if (corpus->n == 0) return RF_OK;
    if (corpus->n >= 0xFFFFFFFFull || capacity > 0xFFFFFFFFull) capacity = std::min<uint64_t>(capacity, 0xFFFFFFFEull);
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    const bool desc = op == RF_OP_SIMILARITY || op == RF_OP_NORMALIZED_SIMILARITY;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(capacity, corpus->n);
    ScratchSet sc{st, {}};
    // the compact pairs before ordering / widening
    uint32_t* d_idx = nullptr;
    void* d_val = nullptr;
    RF_HIP(sc.get((void**)&d_idx, (size_t)cap * sizeof(uint32_t)));
    RF_HIP(sc.get(&d_val, (size_t)cap * elem));

    uint32_t count = 0;
    bool in_index_order = true;
    bool done = false, delivered = false;
    uint64_t* d_index64 = out_index;
    void* d_score = out_score;
    if (out_mem == RF_MEM_HOST && cap) {
        RF_HIP(sc.get((void**)&d_index64, (size_t)cap * sizeof(uint64_t)));
        RF_HIP(sc.get(&d_score, (size_t)cap * elem));
    }
    // ---- the first road
    {
        bool took = false;
        void* lane_val = nullptr;
        uint32_t *lane_idx = nullptr, *d_total = nullptr, cap2 = 0;
        std::unique_lock<std::mutex> held;
        if (const rf_status rs = filter_fast(c, corpus, op, args, f64_out, capacity, st, sc, &took, &lane_val, &lane_idx, &d_total, &cap2, &held); rs != RF_OK) return rs;
        if (took) {
            const uint32_t n_seg = filter_segments(cap2);
            uint32_t* seg = nullptr;
            void* temp = nullptr;
            const size_t temp_bytes = filter_scan_temp_bytes(n_seg);
            RF_HIP(sc.get((void**)&seg, ((size_t)n_seg + 1) * sizeof(uint32_t)));
            RF_HIP(sc.get(&temp, temp_bytes));
            RF_HIP(launch_filter_compact(lane_val, f64_out, lane_idx, 0, cap2, d_total, seg, temp, temp_bytes, cap, d_idx, d_val, st));
            // the survivors arrive in index order: unless the caller wants them by score, the widening runs before the host has seen the count (one
            // synchronization per call instead of two)
            const bool early_finish = order != RF_FILTER_BY_SCORE && cap != 0;
            if (early_finish) RF_HIP(launch_filter_finish(d_idx, d_val, nullptr, f64_out, desc, cap, seg + n_seg, index_base, d_index64, d_score, st));
            uint32_t h[2] = {0, 0};
            RF_HIP(hipMemcpyAsync(&h[0], seg + n_seg, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            RF_HIP(hipMemcpyAsync(&h[1], d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            corpus_tile_list_done(corpus, st);
            held.unlock();
            RF_HIP(hipStreamSynchronize(st));
            if (h[1] <= cap2) {  // every survivor had room: the count is the true one
                count = h[0];
                done = true;
                delivered = early_finish;
            }  // (else: more survivors than room -- the second road)
        }
    }
    // ---- the second road
    if (!done) {
        const bool slots = !corpus->uniform && corpus->d_orig && !corpus->borrowed && corpus->n_slots && !c->wide && !corpus->wide;
        const size_t m = slots ? corpus->n_slots : corpus->n;
        void* d_tmp = nullptr;
        RF_HIP(sc.get(&d_tmp, m * elem));
        rf_args a = *args;
        if (slots) a.flags |= RF_FLAG_SLOT_ORDER;
        else a.flags &= ~RF_FLAG_SLOT_ORDER;
        if (const rf_status rs = run_many(c, corpus, op, &a, d_tmp, RF_MEM_DEVICE, stream, f64_out); rs != RF_OK) return rs;
        const uint32_t n_seg = filter_segments((uint32_t)m);
        uint32_t* seg = nullptr;
        void* temp = nullptr;
        const size_t temp_bytes = filter_scan_temp_bytes(n_seg);
        RF_HIP(sc.get((void**)&seg, ((size_t)n_seg + 1) * sizeof(uint32_t)));
        RF_HIP(sc.get(&temp, temp_bytes));
        RF_HIP(launch_filter_compact(d_tmp, f64_out, slots ? corpus->d_orig : nullptr, slots ? corpus->n_exact * (uint32_t)kWave : 0u, (uint32_t)m, nullptr, seg, temp, temp_bytes,
                                     cap, d_idx, d_val, st));
        in_index_order = !slots;
        const bool early_finish = cap != 0 && (order == RF_FILTER_ANY || (order == RF_FILTER_BY_INDEX && in_index_order));
        if (early_finish) RF_HIP(launch_filter_finish(d_idx, d_val, nullptr, f64_out, desc, cap, seg + n_seg, index_base, d_index64, d_score, st));
        RF_HIP(hipMemcpyAsync(&count, seg + n_seg, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RF_HIP(hipStreamSynchronize(st));
        delivered = early_finish;
    }
    *out_count = count;
    const uint32_t have = std::min(count, cap);
    if (have == 0) return RF_OK;
    if (!delivered) {
        // ---- order, widen
        const bool by_index = !in_index_order && order != RF_FILTER_ANY;
        const bool by_score = order == RF_FILTER_BY_SCORE;
        const uint32_t* idx_now = d_idx;
        const void* val_now = d_val;
        const void* key_now = nullptr;
        if (by_index || by_score) {
            void* temp = nullptr;
            const size_t temp_bytes = filter_sort_temp_bytes(have);
            RF_HIP(sc.get(&temp, temp_bytes));
            if (by_index) {
                uint32_t* idx2 = nullptr;
                void* val2 = nullptr;
                RF_HIP(sc.get((void**)&idx2, (size_t)have * sizeof(uint32_t)));
                RF_HIP(sc.get(&val2, (size_t)have * elem));
                RF_HIP(launch_filter_sort_by_index(idx_now, val_now, f64_out, have, idx2, val2, temp, temp_bytes, st));
                idx_now = idx2, val_now = val2;
            }
            if (by_score) {
                uint32_t* idx3 = nullptr;
                void *key_in = nullptr, *key_out = nullptr;
                RF_HIP(sc.get((void**)&idx3, (size_t)have * sizeof(uint32_t)));
                RF_HIP(sc.get(&key_in, (size_t)have * sizeof(uint64_t)));
                RF_HIP(sc.get(&key_out, (size_t)have * sizeof(uint64_t)));
                RF_HIP(launch_filter_sort_by_score(idx_now, val_now, f64_out, desc, have, key_in, key_out, idx3, temp, temp_bytes, st));
                idx_now = idx3, key_now = key_out;
            }
        }
        RF_HIP(launch_filter_finish(idx_now, val_now, key_now, f64_out, desc, have, nullptr, index_base, d_index64, d_score, st));
    }
    if (out_mem == RF_MEM_HOST) {
        RF_HIP(hipMemcpyAsync(out_index, d_index64, (size_t)have * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        RF_HIP(hipMemcpyAsync(out_score, d_score, (size_t)have * elem, hipMemcpyDeviceToHost, st));
    }
    if (out_mem == RF_MEM_HOST || !delivered) RF_HIP(hipStreamSynchronize(st));
    return RF_OK;
}

rf_status rf_filter_u32(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint64_t index_base, uint64_t capacity, uint64_t* out_index,
                        uint32_t* out_score, uint64_t* out_count, rf_mem out_mem, rf_filter_order order, void* stream)
try {
    return run_filter(c, corpus, op, args, index_base, capacity, out_index, out_score, out_count, out_mem, order, stream, false);
}
RF_ABI_CATCH

rf_status rf_filter_f64(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint64_t index_base, uint64_t capacity, uint64_t* out_index,
                        double* out_score, uint64_t* out_count, rf_mem out_mem, rf_filter_order order, void* stream)
try {
    return run_filter(c, corpus, op, args, index_base, capacity, out_index, out_score, out_count, out_mem, order, stream, true);
}
RF_ABI_CATCH

// ---- the slot map of a corpus (RF_FLAG_SLOT_ORDER)
size_t rf_corpus_slot_count(const rf_corpus* c)
{
    if (!c) return 0;
    return (c->uniform || !c->d_orig) ? c->n : c->n_slots;
}

rf_status rf_corpus_slot_index(const rf_corpus* c, uint32_t* out, rf_mem out_mem)
try {
    if (!c || (!out && rf_corpus_slot_count(c))) {
        set_error("rf_corpus_slot_index: null argument");
        return RF_ERR_INVALID_ARG;
    }
    const size_t m = rf_corpus_slot_count(c);
    if (m == 0) return RF_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    if (c->uniform || !c->d_orig) {
        std::vector<uint32_t> iota(m);
        for (size_t i = 0; i < m; ++i) iota[i] = (uint32_t)i;
        if (out_mem == RF_MEM_HOST)
            std::memcpy(out, iota.data(), m * sizeof(uint32_t));
        else
            RF_HIP(hipMemcpy(out, iota.data(), m * sizeof(uint32_t), hipMemcpyHostToDevice));
        return RF_OK;
    }
    RF_HIP(hipMemcpy(out, c->d_orig, m * sizeof(uint32_t), out_mem == RF_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return RF_OK;
}
RF_ABI_CATCH

}  // extern "C"
