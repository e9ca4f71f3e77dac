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
#include <functional>

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

// What filter_small_kernel reports -- [0] results, [1] entries, [2] flag, [3] the call's sequence number -- lands in PINNED HOST memory the calling thread
// spins on: no device-to-host copy command, no stream synchronization between the kernel's last store and the host seeing the count (measured on the
// configs[4] shape at 100 M: 202 -> ~185 us per call; the count coming home is the one thing a filter call cannot enqueue and forget).  One 64-byte slot per
// host thread, never freed.  nullptr (no pinned memory to be had): the callers copy and synchronize instead.
struct ResultSlot {
    volatile uint32_t* host = nullptr;
    uint32_t seq = 0;
};
ResultSlot& result_slot()
{
    thread_local ResultSlot slot;
    if (!slot.host) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
            std::memset(p, 0, 64);
            slot.host = static_cast<volatile uint32_t*>(p);
        } else {
            (void)hipGetLastError();
        }
    }
    return slot;
}
// filter_select_kernel's workspace: per host thread and device, zeroed once (the kernel leaves it zeroed); rf_filter_* is synchronous per host thread, so one call at a
// time uses it.  nullptr: no memory for it (the callers take the general compaction).
void* select_workspace(int device)
{
    thread_local std::map<int, void*> ws;
    auto it = ws.find(device);
    if (it != ws.end()) return it->second;
    void* p = nullptr;
    if (hipMalloc(&p, filter_select_work_bytes()) != hipSuccess || hipMemset(p, 0, filter_select_work_bytes()) != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ws[device] = p;
    return p;
}

// wait for the kernel that carries `seq`; false = the stream drained (or failed) without it
bool await_slot(ResultSlot& slot, hipStream_t st, uint32_t h[3])
{
    for (uint64_t spin = 0;; ++spin) {
        if (__atomic_load_n(const_cast<uint32_t*>(&slot.host[3]), __ATOMIC_ACQUIRE) == slot.seq) break;
        if ((spin & 0xFFF) == 0xFFF) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) (void)hipGetLastError();
            if (__atomic_load_n(const_cast<uint32_t*>(&slot.host[3]), __ATOMIC_ACQUIRE) == slot.seq) break;
            return false;
        }
    }
    h[0] = slot.host[0], h[1] = slot.host[1], h[2] = slot.host[2];
    return true;
}

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
    const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(corpus->n / 4, 4 * capacity), 1u << 16);
    const uint32_t cap2 = (uint32_t)std::min<uint64_t>(corpus->n, want);
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    RF_HIP(sc.get(lane_val, (size_t)cap2 * (elem + sizeof(uint32_t))));  // (one block: values, then indices -- every scratch block is a lock and an event per call)
    *lane_idx = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(*lane_val) + (size_t)cap2 * elem);
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
    if (corpus->n == 0) {  // argument errors of the scan itself (metric x op x output type) even when there is nothing to scan; otherwise the roads below report them
        ScanParams p;
        RawKind raw = RAW_LEV;
        const rf_comparator* ce = nullptr;
        ComparatorRef hold;
        if (resolve(c, corpus, &ce, &hold) == RF_OK)
            if (const rf_status rs = plan(ce, corpus, op, args, f64_out, &p, &raw); rs != RF_OK) return rs;
        return RF_OK;
    }
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    const bool desc = op == RF_OP_SIMILARITY || op == RF_OP_NORMALIZED_SIMILARITY;
    const bool by_score = order == RF_FILTER_BY_SCORE;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(capacity, corpus->n);
    ScratchSet sc{st, {}};
    // where the caller's arrays are filled: in place (device memory) or in a device copy that goes home at the end
    uint64_t* d_index64 = out_index;
    void* d_score = out_score;
    if (out_mem == RF_MEM_HOST && cap) {
        RF_HIP(sc.get((void**)&d_index64, (size_t)cap * sizeof(uint64_t)));
        RF_HIP(sc.get(&d_score, (size_t)cap * elem));
    }
    // what the last kernel of a road reports: [0] results, [1] entries it saw, [2] 1 = too many for one workgroup, [4] an auxiliary device word -- into the calling
    // thread's pinned slot (the host spins on the sequence number) or, without pinned memory, into device words that are copied home behind a synchronization
    ResultSlot& slot = result_slot();
    uint32_t* d_res = nullptr;
    if (!slot.host) RF_HIP(sc.get((void**)&d_res, 8 * sizeof(uint32_t)));
    auto report_target = [&]() { return slot.host ? (++slot.seq, const_cast<uint32_t*>(slot.host)) : d_res; };
    auto await_report = [&](uint32_t h[5]) -> rf_status {
        if (slot.host) {
            uint32_t h3[3];
            if (!await_slot(slot, st, h3)) {
                set_error("rf_filter: the stream finished without the result kernel's report");
                return RF_ERR_HIP;
            }
            h[0] = h3[0], h[1] = h3[1], h[2] = h3[2], h[4] = slot.host[4];
            return RF_OK;
        }
        RF_HIP(hipMemcpyAsync(h, d_res, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RF_HIP(hipStreamSynchronize(st));
        return RF_OK;
    };
    uint32_t count = 0;
    bool delivered = false;  // the caller's (device) arrays hold the pairs, in the order asked for

    // general compaction of `bound` entries (or *dev of them) -> the (index, value) pairs of the Somes, then whatever the order needs; ends with the report
    uint32_t* d_idx = nullptr;
    void* d_val = nullptr;
    auto general = [&](const void* e_val, const uint32_t* e_map, uint32_t e_map_from, uint32_t e_bound, const uint32_t* e_dev, bool in_index_order, const uint32_t* aux_dev,
                       uint32_t h[5], const std::function<void()>& enqueued) -> rf_status {
        if (!d_idx) {
            RF_HIP(sc.get((void**)&d_idx, (size_t)std::max(cap, 1u) * sizeof(uint32_t)));
            RF_HIP(sc.get(&d_val, (size_t)std::max(cap, 1u) * elem));
        }
        const uint32_t n_seg = filter_segments(e_bound);
        uint32_t* seg = nullptr;
        void* temp = nullptr;
        const size_t temp_bytes = filter_scan_temp_bytes(n_seg);
        RF_HIP(sc.get((void**)&seg, ((size_t)n_seg + 1) * sizeof(uint32_t)));
        RF_HIP(sc.get(&temp, temp_bytes));
        RF_HIP(launch_filter_compact(e_val, f64_out, e_map, e_map_from, e_bound, e_dev, seg, temp, temp_bytes, cap, d_idx, d_val, st));
        const uint32_t* d_count = seg + n_seg;
        const bool need_sort = cap && (by_score || (order == RF_FILTER_BY_INDEX && !in_index_order));
        if (!need_sort) {
            // the widening runs before the host has seen the count
            if (cap) RF_HIP(launch_filter_finish(d_idx, d_val, nullptr, f64_out, desc, cap, d_count, index_base, d_index64, d_score, st));
            uint32_t* target = report_target();  // (its own statement: the sequence number is read after the increment)
            RF_HIP(launch_filter_report(d_count, aux_dev, target, slot.seq, st));
        } else {
            // few results (the usual case under a cutoff): ordered and widened by one workgroup; else ([2] = 1) hipcub's radix sorts below
            uint32_t* target = report_target();
            RF_HIP(launch_filter_small(d_val, f64_out, d_idx, cap, d_count, by_score, desc, cap, index_base, d_index64, d_score, target, slot.seq, aux_dev, st));
        }
        if (enqueued) enqueued();
        if (const rf_status rs = await_report(h); rs != RF_OK) return rs;
        if (!need_sort) {
            count = h[0];
            delivered = true;
            return RF_OK;
        }
        count = h[1];  // (the entries that kernel saw = the Somes the compaction counted)
        if (h[2] == 0) {
            delivered = true;
            return RF_OK;
        }
        const uint32_t have = std::min(count, cap);
        const uint32_t* idx_now = d_idx;
        const void* val_now = d_val;
        const void* key_now = nullptr;
        void* stemp = nullptr;
        const size_t stemp_bytes = filter_sort_temp_bytes(have);
        RF_HIP(sc.get(&stemp, stemp_bytes));
        if (!in_index_order) {  // (by score too: ties go by index, and the sort by score is stable)
            uint32_t* idx2 = nullptr;
            void* val2 = nullptr;
            RF_HIP(sc.get((void**)&idx2, (size_t)have * sizeof(uint32_t)));
            RF_HIP(sc.get(&val2, (size_t)have * elem));
            RF_HIP(launch_filter_sort_by_index(idx_now, val_now, f64_out, have, idx2, val2, stemp, stemp_bytes, st));
            idx_now = idx2, val_now = val2;
        }
        if (by_score) {
            uint32_t* idx3 = nullptr;
            void *key_in = nullptr, *key_out = nullptr;
            RF_HIP(sc.get((void**)&idx3, (size_t)have * sizeof(uint32_t)));
            RF_HIP(sc.get(&key_in, (size_t)have * sizeof(uint64_t)));
            RF_HIP(sc.get(&key_out, (size_t)have * sizeof(uint64_t)));
            RF_HIP(launch_filter_sort_by_score(idx_now, val_now, f64_out, desc, have, key_in, key_out, idx3, stemp, stemp_bytes, st));
            idx_now = idx3, key_now = key_out;
        }
        RF_HIP(launch_filter_finish(idx_now, val_now, key_now, f64_out, desc, have, nullptr, index_base, d_index64, d_score, st));
        if (out_mem == RF_MEM_DEVICE) RF_HIP(hipStreamSynchronize(st));  // (the scratch behind the sorts is released when this call returns)
        delivered = true;
        return RF_OK;
    };

    // ---- the first road: the survivors of the lane compaction (NOT in index order: the first pass deals tiles to its wavefronts round-robin)
    bool second_road = true;
    {
        bool took = false;
        void* lane_val = nullptr;
        uint32_t *lane_idx = nullptr, *d_total = nullptr, cap2 = 0;
        std::unique_lock<std::mutex> held;
        if (const rf_status rs = filter_fast(c, corpus, op, args, f64_out, capacity, st, sc, &took, &lane_val, &lane_idx, &d_total, &cap2, &held); rs != RF_OK) return rs;
        if (took) {
            uint32_t h[5] = {0, 0, 1, 0, 0};
            auto release = [&]() {  // everything that reads the tile-list buffer has been enqueued
                corpus_tile_list_done(corpus, st);
                held.unlock();
            };
            second_road = false;
            void* ws = cap ? select_workspace(corpus->device) : nullptr;
            if (ws) {
                // the Somes among the survivors -- few, under a cutoff -- selected by every workgroup, ordered, widened and counted by the last one to arrive: ONE kernel
                // behind the scan, whatever the number of survivors
                uint32_t* target = report_target();
                RF_HIP(launch_filter_select(lane_val, f64_out, lane_idx, cap2, d_total, by_score, desc, cap, index_base, d_index64, d_score, ws, target, slot.seq, d_total,
                                            corpus->filter_last_survivors.load(std::memory_order_relaxed), st));
                release();
                if (const rf_status rs = await_report(h); rs != RF_OK) return rs;
                corpus->filter_last_survivors.store(h[4], std::memory_order_relaxed);
                if (h[2] == 0) {
                    count = h[0];
                    delivered = true;
                } else if (h[4] <= cap2) {  // more results than one workgroup orders: the general compaction over the survivors (their number is known now)
                    if (const rf_status rs = general(lane_val, lane_idx, 0, h[4], nullptr, false, nullptr, h, nullptr); rs != RF_OK) return rs;
                } else {
                    second_road = true;  // more survivors than room
                }
            } else {
                if (const rf_status rs = general(lane_val, lane_idx, 0, cap2, d_total, false, d_total, h, release); rs != RF_OK) return rs;
                corpus->filter_last_survivors.store(h[4], std::memory_order_relaxed);
                if (h[4] > cap2) {  // more survivors than room: what was compacted is a part of them
                    second_road = true;
                    delivered = false;
                }
            }
        }
    }
    // ---- the second road: the scan into a device vector -- in slot order for a length-bucketed corpus: no gather pass
    if (second_road) {
        const bool slots = !corpus->uniform && corpus->d_orig && !corpus->borrowed && corpus->n_slots && !c->wide && !corpus->wide;
        const size_t m = slots ? corpus->n_slots : corpus->n;
        void* d_tmp = nullptr;
        RF_HIP(sc.get(&d_tmp, m * elem));
        rf_args a = *args;
        a.flags &= ~(kFlagSlotsInternal | kFlagWindowInternal);
        // A cutoff's length window (plan(): the tiles outside [tile_begin, tile_end) are None by their length alone) is all this call has to look at in a slot-ordered
        // temporary -- slot = 64 x tile + lane for exact tiles and views alike: neither the pre-fill nor the compaction touches the rest (ragged [1, 64] under a
        // cutoff of 3 edits: 4 of 64 lengths)
        size_t w0 = 0, w1 = m;
        if (slots) {
            a.flags |= kFlagSlotsInternal;  // (run_many's own switch: rf_host.hpp)
            ScanParams pw;
            RawKind raww = RAW_LEV;
            if (plan(c, corpus, op, &a, f64_out, &pw, &raww) == RF_OK && pw.prefill_none) {
                a.flags |= kFlagWindowInternal;
                w0 = (size_t)pw.tile_begin * kWave, w1 = (size_t)pw.tile_end * kWave;
            }
        }
        if (const rf_status rs = run_many(c, corpus, op, &a, d_tmp, RF_MEM_DEVICE, stream, f64_out); rs != RF_OK) return rs;
        uint32_t h[5] = {0, 0, 1, 0, 0};
        const uint32_t exact_slots = corpus->n_exact * (uint32_t)kWave;
        if (const rf_status rs = general(static_cast<const uint8_t*>(d_tmp) + w0 * elem, slots ? corpus->d_orig + w0 : nullptr,
                                         slots ? (uint32_t)(exact_slots > w0 ? exact_slots - w0 : 0) : 0u, (uint32_t)(w1 - w0), nullptr, !slots, nullptr, h, nullptr);
            rs != RF_OK)
            return rs;
    }
    *out_count = count;
    const uint32_t have = std::min(count, cap);
    if (have && out_mem == RF_MEM_HOST) {
        RF_HIP(hipMemcpyAsync(out_index, d_index64, (size_t)have * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        RF_HIP(hipMemcpyAsync(out_score, d_score, (size_t)have * elem, hipMemcpyDeviceToHost, st));
        RF_HIP(hipStreamSynchronize(st));
    }
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
