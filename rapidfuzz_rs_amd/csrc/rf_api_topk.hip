// rf_api_topk.hip -- top-k: in-scan lists, exact selection, 8-byte keys and 16-byte entries, the exchange on a raw ncclComm_t (split out of rf_api.hip in round 4; rf_host.hpp has the shared declarations).
// Product code: never includes or links anything from oracle/.
#include "rf_host.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------------
// top-k
// ---------------------------------------------------------------------------------------------------
// shared by rf_topk_u32 (host results) and rf_topk_keys_device (device keys, fully asynchronous)
static rf_status topk_core(const rf_comparator* c_in, const rf_corpus* corpus_in, rf_op op, const rf_args* args, uint32_t k,
                           uint32_t key_index_base, uint64_t* d_best /*device, k entries*/, uint32_t* out_all,
                           rf_mem out_all_mem, hipStream_t st, bool* desc)
{
    const rf_args args_v = args ? sanitized_args(args, false) : rf_args{};  // (RF_FLAG_SLOT_ORDER is rf_many_*'s alone: the scores here are n-entry vectors)
    if (args) args = &args_v;
    Effective eff;
    if (const rf_status rs = make_effective(c_in, corpus_in, st, &eff); rs != RF_OK) return rs;
    const rf_comparator* c = eff.c;
    const rf_corpus* corpus = eff.corpus;
    const rf_corpus* owner = corpus->parent ? corpus->parent : corpus;  // scratch and locks live in the real corpus
    if (k == 0 || k > (uint32_t)kWave) {
        set_error("top-k: k must be in 1..64 (one list entry per wavefront lane)");
        return k == 0 ? RF_ERR_INVALID_ARG : RF_ERR_UNSUPPORTED;
    }
    if (op != RF_OP_DISTANCE && op != RF_OP_SIMILARITY) {
        set_error("top-k: op must be RF_OP_DISTANCE or RF_OP_SIMILARITY");
        return RF_ERR_INVALID_ARG;
    }
    ScanParams p;
    RawKind raw = RAW_LEV;
    rf_status s = plan(c, corpus, op, args, false, &p, &raw);
    if (s != RF_OK) return s;
    if (raw == RAW_JARO) {
        set_error("top-k: usize-valued metrics only");
        return RF_ERR_INVALID_ARG;
    }
    // (a long query under a small cutoff is planned onto the band kernel, which has no top-k epilogue, and the register-resident
    // scans stop at 8 words: beyond 512 symbols that shape goes the selection way like every other long query -- ADVICE r2)
    if (p.long_words_pad || raw == RAW_WF || (p.band && c->words > (size_t)kMaxWords)) {
        set_error("top-k: queries longer than 512 symbols and general Levenshtein weight tables are served by rf_many_* only");
        return RF_ERR_UNSUPPORTED;
    }
    *desc = op == RF_OP_SIMILARITY;
    s = comparator_device_pm(c, corpus->device, &p.pm);
    if (s != RF_OK) return s;
    // (single-length corpora only: every launch of a top-k call selects its own k best, so the per-length-run launches of
    // launch_scan_runs cannot share one call)
    p.heads8 = corpus->uniform ? corpus_head8_plane(corpus, p, raw, st) : nullptr;
    p.heads6 = p.heads8 ? corpus_head6_plane(corpus, st) : nullptr;
    if (corpus->uniform) plan_band_filter(c, corpus, op, false, &p, corpus->uniform_len);
    // persistent per-(corpus, stream) scratch; capacity = every workgroup publishing a full 64-entry list.  At most kTopkStreams
    // streams hold one: a further stream takes over the least recently used stream's scratch (after that stream's work has drained) --
    // a process that makes a stream per request must not grow the corpus by 2 MB (+ 4 bytes per candidate for the score vector of
    // the multi-word path) per stream it has ever used.  The lookup and the enqueue section are ONE critical section (a takeover
    // between a thread's lookup and its launches would hand its scratch to another stream).
    std::lock_guard<std::mutex> enqueue_lock(owner->topk_enqueue_mu);
    constexpr size_t kTopkStreams = 8;
    rf_corpus::TopkScratch sc;
    {
        std::lock_guard<std::mutex> lock(owner->scratch_mu);
        auto& lru = owner->topk_lru;  // most recently used first
        auto it = owner->topk_scratch.find(st);
        if (it != owner->topk_scratch.end()) {
            sc = it->second;
            lru.erase(std::remove(lru.begin(), lru.end(), st), lru.end());
        } else if (owner->topk_scratch.size() >= kTopkStreams && !lru.empty()) {
            const hipStream_t victim = lru.back();
            lru.pop_back();
            sc = owner->topk_scratch[victim];  // bound and counters are left re-armed by every call
            (void)hipEventSynchronize(sc.done);  // (the event behind its last call -- not the stream handle, whose stream may be gone)
            owner->topk_scratch.erase(victim);
            owner->topk_scratch.emplace(st, sc);
        } else {
            // [64 way segments of candidate keys | root table 64 x 64 keys | bound (u64, own line) | control block 65 x 128 B]
            const size_t ways = 64, per_way = ((size_t)scan_grid_full(corpus->n_tiles) + ways - 1) / ways;  // (the largest grid any top-k launch uses)
            sc.seg_cap = (uint32_t)(per_way * kWave);
            const size_t cand_bytes = ways * sc.seg_cap * sizeof(uint64_t), root_bytes = ways * kWave * sizeof(uint64_t), ctl_bytes = 65 * 128;
            uint8_t* mem = nullptr;
            RF_HIP(hipMalloc((void**)&mem, cand_bytes + root_bytes + 128 + ctl_bytes));
            sc.cand = reinterpret_cast<uint64_t*>(mem);
            sc.root = reinterpret_cast<uint64_t*>(mem + cand_bytes);
            sc.bound = reinterpret_cast<uint64_t*>(mem + cand_bytes + root_bytes);
            sc.ctl = reinterpret_cast<uint32_t*>(mem + cand_bytes + root_bytes + 128);
            hipError_t e0 = hipMemsetAsync(sc.bound, 0xFF, sizeof(uint64_t), st);
            if (e0 == hipSuccess) e0 = hipMemsetAsync(sc.ctl, 0, ctl_bytes, st);
            if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&sc.done, hipEventDisableTiming);
            if (e0 != hipSuccess) {
                (void)hipFree(mem);
                set_error(std::string("top-k scratch: ") + hipGetErrorString(e0));
                return RF_ERR_HIP;
            }
            owner->topk_scratch.emplace(st, sc);
        }
        lru.insert(lru.begin(), st);
    }
    p.topk_bound = sc.bound;
    p.topk_cand = sc.cand;
    p.topk_seg_cap = sc.seg_cap;
    p.topk_ctl = sc.ctl;
    p.topk_root = sc.root;
    p.topk_out = d_best;
    p.topk_k = k;
    p.topk_desc = *desc;
    p.key_index_base = key_index_base;
    // optionally also emit every candidate's score from the same pass (they stay sharded)
    uint32_t* d_all = out_all;
    if (out_all && out_all_mem == RF_MEM_HOST) RF_HIP(scratch_alloc((void**)&d_all, corpus->n * sizeof(uint32_t), st));
    p.out = d_all;
    // Round 4: a top-k as the scan into a score vector + ONE pass over it (rf_select.hip topk_scores_kernel) -- for the shapes whose
    // plain scan is a whole-kernel asm scan WITHOUT a fast in-scan top-k form: Levenshtein over 2..4 words (queries of 65..256 symbols),
    // whose in-scan lists live in the compiled multi-word kernel: configs[2] corpus top-16 2.53 -> 2.74 Gpairs/s.  Single-word shapes keep
    // their in-scan lists (sampled bound, selection inside the launch: +3 % over the plain scan; scan + pass measured +8 %, and +50 % on a
    // 20 M ragged corpus where the pass's fixed ~0.2 ms shows: profiles/topk_via_scores_r04.txt).  The score vector is the caller's
    // `out_all` when there is one, else a buffer kept per (corpus, stream) like the rest of the top-k scratch.
    // RF_TOPK_VIA_SCORES: 0 = never, 1 = default, 2 = every shape with an asm scan (tests).
    static const int via_scores = [] { const char* e = getenv("RF_TOPK_VIA_SCORES"); return e ? atoi(e) : 1; }();
    const bool asm_scan = !p.early && !p.band && !p.long_words_pad && p.tile_step == 1 && ((raw == RAW_LEV && p.words <= 4) || (raw == RAW_OSA && p.words == 1));
    if (asm_scan && (via_scores >= 2 || (via_scores == 1 && raw == RAW_LEV && p.words >= 2))) {
        uint32_t* d_scores = d_all;
        if (!d_scores) {
            std::lock_guard<std::mutex> lock(owner->scratch_mu);
            rf_corpus::TopkScratch& slot = owner->topk_scratch[st];
            if (slot.scores_cap < corpus->n) {
                if (slot.scores) (void)hipFree(slot.scores);  // (synchronizes with the work that used it)
                slot.scores = nullptr;
                slot.scores_cap = 0;
                RF_HIP(hipMalloc((void**)&slot.scores, corpus->n * sizeof(uint32_t)));
                slot.scores_cap = corpus->n;
            }
            d_scores = slot.scores;
        }
        const rf_status rs = run_many(c_in, corpus_in, op, args, d_scores, RF_MEM_DEVICE, st, false);
        hipError_t e = rs == RF_OK ? launch_topk_scores(p, d_scores, (uint32_t)corpus->n, st) : hipSuccess;
        if (rs == RF_OK && e == hipSuccess && out_all && out_all_mem == RF_MEM_HOST) e = hipMemcpyAsync(out_all, d_all, corpus->n * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (out_all && out_all_mem == RF_MEM_HOST) (void)scratch_free(d_all, st);
        if (rs != RF_OK) return rs;
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> lock(owner->scratch_mu);
            (void)hipStreamSynchronize(st);
            auto it = owner->topk_scratch.find(st);
            if (it != owner->topk_scratch.end()) {
                (void)hipFree(it->second.cand);
                if (it->second.scores) (void)hipFree(it->second.scores);
                if (it->second.done) (void)hipEventDestroy(it->second.done);
                owner->topk_scratch.erase(it);
                owner->topk_lru.erase(std::remove(owner->topk_lru.begin(), owner->topk_lru.end(), st), owner->topk_lru.end());
            }
            set_error(std::string("top-k: ") + hipGetErrorString(e));
            return RF_ERR_HIP;
        }
        (void)hipEventRecord(sc.done, st);
        return RF_OK;
    }
    std::unique_lock<std::mutex> filter_lock;
    if (p.heads8) {
        filter_lock = std::unique_lock<std::mutex>(corpus->filter_enqueue_mu);
        p.tile_list_buf = corpus_tile_list(corpus, st);
        corpus_lane_buffers(corpus, &p);
    }
    hipError_t e = hipSuccess;
    // Sample pass: the top-k of ~1000 evenly spaced tiles costs 0.1 % of the scan and its k-th best key is a valid
    // launch-wide bound from the first tile on -- without it every wavefront pays k ln(n_wave / k) list insertions to
    // warm its own list up (the shared bound alone is only as good as the luckiest wavefront's k-th best).  Under a
    // tight cutoff (p.early) the cutoff itself keeps nearly everything out of the lists and the pass is skipped.
    // Each launch selects its own k best in its last workgroup (topk_block_publish): 2 launches, or 1.
    // (RF_TOPK_SAMPLE=<tiles> tunes the sample size, 0 disables the pass: A/B switch)
    static const uint32_t kSampleTiles = [] { const char* e = getenv("RF_TOPK_SAMPLE"); return e ? (uint32_t)atoi(e) : 1024u; }();
    if (kSampleTiles && !p.early && p.tile_end - p.tile_begin >= 8 * kSampleTiles) {
        ScanParams ps = p;
        ps.out = nullptr;
        ps.prefill_none = 0;
        ps.tile_step = (p.tile_end - p.tile_begin) / kSampleTiles;
        ps.topk_bound_from_result = 1;
        e = launch_scan(raw, ps, st, nullptr);
    }
    if (e == hipSuccess) e = launch_scan(raw, p, st, nullptr);
    if (e == hipSuccess && out_all && out_all_mem == RF_MEM_HOST) e = hipMemcpyAsync(out_all, d_all, corpus->n * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (out_all && out_all_mem == RF_MEM_HOST) (void)scratch_free(d_all, st);  // (on the failure paths too: ADVICE r4)
    if (e != hipSuccess) {
        // the scratch may be left half-armed: drop it so the next call starts from a fresh one
        std::lock_guard<std::mutex> lock(owner->scratch_mu);
        (void)hipStreamSynchronize(st);
        (void)hipFree(sc.cand);
        if (sc.scores) (void)hipFree(sc.scores);
        if (sc.done) (void)hipEventDestroy(sc.done);
        owner->topk_scratch.erase(st);
        owner->topk_lru.erase(std::remove(owner->topk_lru.begin(), owner->topk_lru.end(), st), owner->topk_lru.end());
        set_error(std::string("top-k: ") + hipGetErrorString(e));
        return RF_ERR_HIP;
    }
    (void)hipEventRecord(sc.done, st);  // what a later take-over of this scratch / tile list waits for
    if (p.tile_list_buf) corpus_tile_list_done(corpus, st);
    return RF_OK;
}

// ---- the general top-k path: scan to a score vector, then exact selection over it (rf_select.hip) ------------------
// Any k, u32 or f64 scores.  Returns the kk = min(k, #not-None) best (key, index) pairs sorted by (key, index).
static rf_status select_topk(const void* d_scores, bool f64, bool desc, uint32_t n, uint64_t k, hipStream_t st, std::vector<uint64_t>* keys,
                             std::vector<uint32_t>* idx)
{
    keys->clear();
    idx->clear();
    const uint32_t nb = select_blocks(n);
    uint8_t* mem = nullptr;
    const size_t hist_bytes = 2048 * sizeof(unsigned long long), cnt_bytes = (size_t)nb * sizeof(uint32_t);
    RF_HIP(scratch_alloc((void**)&mem, 64 + hist_bytes + 2 * cnt_bytes, st));
    struct Free {
        uint8_t* p;
        hipStream_t st;
        ~Free() { (void)scratch_free(p, st); }
    } free_mem{mem, st};
    unsigned long long* d_hist = reinterpret_cast<unsigned long long*>(mem + 64);
    uint32_t* d_less = reinterpret_cast<uint32_t*>(mem + 64 + hist_bytes);
    uint32_t* d_eq = d_less + nb;
    unsigned long long ctl[3] = {~0ull, 0ull, 0ull};  // min key, max key, valid
    RF_HIP(hipMemcpyAsync(mem, ctl, sizeof(ctl), hipMemcpyHostToDevice, st));
    RF_HIP(launch_select_minmax(d_scores, f64, n, desc, mem, st));
    RF_HIP(hipMemcpyAsync(ctl, mem, sizeof(ctl), hipMemcpyDeviceToHost, st));
    RF_HIP(hipStreamSynchronize(st));
    const uint64_t valid = ctl[2];
    if (valid == 0 || k == 0) return RF_OK;
    const uint64_t kk = std::min<uint64_t>(k, valid);
    // the k-th smallest key value T, 11 bits at a time from the first bit in which the keys differ
    uint64_t prefix = 0, prefix_mask = 0, n_less = 0, rank = kk;  // rank: 1-based among the keys matching the prefix
    const uint64_t diff = ctl[0] ^ ctl[1];
    int hb = diff ? 63 - __builtin_clzll(diff) : -1;  // highest differing bit
    if (hb < 63) {
        prefix_mask = hb < 0 ? ~0ull : ~((2ull << hb) - 1);
        prefix = ctl[0] & prefix_mask;
    }
    if (!f64) prefix_mask &= 0xFFFFFFFFull, prefix &= 0xFFFFFFFFull;
    std::vector<unsigned long long> hist(2048);
    while (hb >= 0) {
        const uint32_t shift = hb + 1 > 11 ? (uint32_t)(hb + 1 - 11) : 0u;
        const uint32_t bits = (uint32_t)(hb + 1) - shift;
        RF_HIP(hipMemsetAsync(d_hist, 0, hist_bytes, st));
        RF_HIP(launch_select_hist(d_scores, f64, n, desc, prefix_mask, prefix, shift, bits, d_hist, st));
        RF_HIP(hipMemcpyAsync(hist.data(), d_hist, hist_bytes, hipMemcpyDeviceToHost, st));
        RF_HIP(hipStreamSynchronize(st));
        uint64_t cum = 0;
        uint32_t d = 0;
        for (; d < (1u << bits); ++d) {
            if (cum + hist[d] >= rank) break;
            cum += hist[d];
        }
        if (d == (1u << bits)) {
            if (getenv("RF_SELECT_DEBUG")) {
                unsigned long long tot = 0;
                for (auto h : hist) tot += h;
                std::fprintf(stderr, "[select] min %llx max %llx valid %llu kk %llu hb %d shift %u bits %u prefix %llx mask %llx rank %llu cum %llu total-in-hist %llu\n",
                             ctl[0], ctl[1], (unsigned long long)valid, (unsigned long long)kk, hb, shift, bits, (unsigned long long)prefix,
                             (unsigned long long)prefix_mask, (unsigned long long)rank, (unsigned long long)cum, tot);
            }
            set_error("top-k selection: inconsistent histogram");
            return RF_ERR_HIP;
        }
        n_less += cum;
        rank -= cum;
        prefix |= (uint64_t)d << shift;
        prefix_mask |= (((1ull << bits) - 1) << shift);
        hb = (int)shift - 1;
    }
    const uint64_t T = prefix;
    const uint32_t need_eq = (uint32_t)(kk - n_less);
    uint8_t* out = nullptr;
    const size_t key_bytes = f64 ? 8 : 4;
    RF_HIP(scratch_alloc((void**)&out, kk * (key_bytes + 4), st));
    Free free_out{out, st};
    uint32_t* d_idx = reinterpret_cast<uint32_t*>(out + kk * key_bytes);
    RF_HIP(launch_select_count(d_scores, f64, n, desc, T, d_less, d_eq, st));
    RF_HIP(launch_select_emit(d_scores, f64, n, desc, T, d_less, d_eq, (uint32_t)n_less, need_eq, out, d_idx, st));
    std::vector<uint8_t> hk(kk * key_bytes);
    std::vector<uint32_t> hi(kk);
    RF_HIP(hipMemcpyAsync(hk.data(), out, hk.size(), hipMemcpyDeviceToHost, st));
    RF_HIP(hipMemcpyAsync(hi.data(), d_idx, kk * 4, hipMemcpyDeviceToHost, st));
    RF_HIP(hipStreamSynchronize(st));
    std::vector<std::pair<uint64_t, uint32_t>> pairs(kk);
    for (uint64_t i = 0; i < kk; ++i)
        pairs[i] = {f64 ? reinterpret_cast<const uint64_t*>(hk.data())[i] : (uint64_t) reinterpret_cast<const uint32_t*>(hk.data())[i], hi[i]};
    std::sort(pairs.begin(), pairs.end());
    keys->resize(kk);
    idx->resize(kk);
    for (uint64_t i = 0; i < kk; ++i) (*keys)[i] = pairs[i].first, (*idx)[i] = pairs[i].second;
    return RF_OK;
}

// scan every candidate into a device score vector (the caller's out_all if it is device memory, a temporary otherwise),
// select, and hand the scores to a host out_all if one was asked for
static rf_status topk_by_selection(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint64_t k, bool f64, void* out_all,
                                   rf_mem out_all_mem, hipStream_t st, std::vector<uint64_t>* keys, std::vector<uint32_t>* idx, bool* desc)
{
    const rf_args args_v = args ? sanitized_args(args, false) : rf_args{};
    if (args) args = &args_v;
    const size_t elem = f64 ? sizeof(double) : sizeof(uint32_t);
    *desc = op == RF_OP_SIMILARITY || op == RF_OP_NORMALIZED_SIMILARITY;
    void* d_scores = out_all;
    const bool temp = !(out_all && out_all_mem == RF_MEM_DEVICE);
    if (temp) RF_HIP(scratch_alloc(&d_scores, corpus->n * elem, st));
    rf_status s = run_many(c, corpus, op, args, d_scores, RF_MEM_DEVICE, st, f64);
    if (s == RF_OK) s = select_topk(d_scores, f64, *desc, (uint32_t)corpus->n, k, st, keys, idx);
    if (s == RF_OK && out_all && out_all_mem == RF_MEM_HOST) {
        hipError_t e = hipMemcpyAsync(out_all, d_scores, corpus->n * elem, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) s = RF_ERR_HIP;
    }
    if (temp) (void)scratch_free(d_scores, st);
    return s;
}

rf_status rf_topk_f64(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint64_t k, uint64_t index_base, double* out_score,
                      uint64_t* out_index, uint64_t* out_count, double* out_all, rf_mem out_all_mem, void* stream)
try {
    if (!out_score || !out_index || !out_count || !c || !corpus || !args) {
        set_error("rf_topk_f64: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    *out_count = 0;
    if (corpus->n == 0 || k == 0) return RF_OK;
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    std::vector<uint64_t> keys;
    std::vector<uint32_t> idx;
    bool desc = false;
    const rf_status s = topk_by_selection(c, corpus, op, args, k, true, out_all, out_all_mem, (hipStream_t)stream, &keys, &idx, &desc);
    if (s != RF_OK) return s;
    for (size_t i = 0; i < keys.size(); ++i) {
        uint64_t b = desc ? ~keys[i] : keys[i];
        b ^= (b >> 63) ? 0x8000000000000000ull : ~0ull;  // undo the order-preserving map of rf_select.hip
        std::memcpy(&out_score[i], &b, sizeof(double));
        out_index[i] = index_base + idx[i];
    }
    *out_count = keys.size();
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_topk_u32(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint32_t k,
                      uint64_t index_base, uint32_t* out_score, uint64_t* out_index, uint32_t* out_count,
                      uint32_t* out_all, rf_mem out_all_mem, void* stream)
try {
    if (!out_score || !out_index || !out_count || !c || !corpus) {
        set_error("rf_topk_u32: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    *out_count = 0;
    if (k == 0) {
        set_error("top-k: k must be at least 1");
        return RF_ERR_INVALID_ARG;
    }
    if (corpus->n == 0) return RF_OK;
    if (k > (uint32_t)kWave) {
        // more than one list entry per wavefront lane: the selection path (any k; also general weight tables, long queries)
        if (op != RF_OP_DISTANCE && op != RF_OP_SIMILARITY) {
            set_error("top-k: op must be RF_OP_DISTANCE or RF_OP_SIMILARITY (normalized_*: rf_topk_f64)");
            return RF_ERR_INVALID_ARG;
        }
        DeviceGuard guard(corpus->device);
        if (!guard.ok) {
            set_error("cannot select the corpus' device");
            return RF_ERR_NO_DEVICE;
        }
        std::vector<uint64_t> keys;
        std::vector<uint32_t> idx;
        bool desc = false;
        const rf_status s = topk_by_selection(c, corpus, op, args, k, false, out_all, out_all_mem, (hipStream_t)stream, &keys, &idx, &desc);
        if (s != RF_OK) return s;
        for (size_t i = 0; i < keys.size(); ++i) {
            out_score[i] = desc ? 0xFFFFFFFEu - (uint32_t)keys[i] : (uint32_t)keys[i];  // KeyOf<uint32_t>::get, rf_select.hip
            out_index[i] = index_base + idx[i];
        }
        *out_count = (uint32_t)keys.size();
        return RF_OK;
    }
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    hipStream_t st = (hipStream_t)stream;
    uint64_t* d_best = nullptr;
    RF_HIP(scratch_alloc((void**)&d_best, (size_t)kWave * sizeof(uint64_t), st));
    bool desc = false;
    std::vector<uint64_t> best(kWave, ~0ull);
    hipError_t e = hipSuccess;
    rf_status s = RF_OK;
    // score_hint (RF_OP_DISTANCE, no cutoff, no per-candidate output): the caller expects the k-th best distance to be <= hint.  The
    // reference uses the hint per pair the same way (levenshtein.rs:1069-1088: a band of `hint`, doubled until the distance fits).
    // Here the scan first runs UNDER THE CUTOFF `hint` -- a cutoff scan costs a fraction of a full one (§5.1) -- and if k candidates
    // pass, they are the k best of the corpus; otherwise the hint doubles, and past a quarter of the longest possible distance the
    // plain scan runs.  The result never depends on the hint.
    bool done = false;
    if (args && op == RF_OP_DISTANCE && args->score_hint_usize != RF_NO_CUTOFF && args->cutoff_usize == RF_NO_CUTOFF && !out_all) {
        const uint64_t longest = std::max<uint64_t>(rf_comparator_query_len(c), corpus->max_len);
        for (uint64_t hint = args->score_hint_usize; hint <= longest / 4; hint = std::max<uint64_t>(1, hint * 2)) {  // (hint <= longest / 4: no overflow for absurd hints)
            rf_args a2 = *args;
            a2.cutoff_usize = hint;
            a2.score_hint_usize = RF_NO_CUTOFF;
            s = topk_core(c, corpus, op, &a2, k, 0, d_best, nullptr, RF_MEM_DEVICE, st, &desc);
            if (s != RF_OK) break;  // (shapes the in-scan lists do not cover: the plain path below sorts that out)
            e = hipMemcpyAsync(best.data(), d_best, k * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) break;
            uint32_t found = 0;
            while (found < k && best[found] != ~0ull) ++found;
            if (found >= k || found >= corpus->n) {
                done = true;
                break;
            }
        }
    }
    if (!done) {
        s = topk_core(c, corpus, op, args, k, 0, d_best, out_all, out_all_mem, st, &desc);
        e = hipSuccess;
        if (s == RF_OK) e = hipMemcpyAsync(best.data(), d_best, k * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    }
    (void)scratch_free(d_best, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (s == RF_ERR_UNSUPPORTED && (op == RF_OP_DISTANCE || op == RF_OP_SIMILARITY)) {
        // shapes the in-scan lists do not cover (queries beyond 512 symbols, general weight tables): score everything, select
        std::vector<uint64_t> keys;
        std::vector<uint32_t> idx;
        s = topk_by_selection(c, corpus, op, args, k, false, out_all, out_all_mem, st, &keys, &idx, &desc);
        if (s != RF_OK) return s;
        for (size_t i = 0; i < keys.size(); ++i) {
            out_score[i] = desc ? 0xFFFFFFFEu - (uint32_t)keys[i] : (uint32_t)keys[i];  // KeyOf<uint32_t>::get, rf_select.hip
            out_index[i] = index_base + idx[i];
        }
        *out_count = (uint32_t)keys.size();
        return RF_OK;
    }
    if (s != RF_OK) return s;
    if (e != hipSuccess) {
        set_error(std::string("top-k: ") + hipGetErrorString(e));
        return RF_ERR_HIP;
    }
    uint32_t m = 0;
    for (; m < k && best[m] != ~0ull; ++m) {
        const uint32_t hi = (uint32_t)(best[m] >> 32);
        out_score[m] = desc ? ~hi : hi;
        out_index[m] = index_base + (uint32_t)best[m];
    }
    *out_count = m;
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_topk_keys_device(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint32_t k,
                              uint32_t index_base, uint64_t* d_keys_out, uint32_t* out_all, rf_mem out_all_mem,
                              void* stream)
try {
    if (!d_keys_out || !c || !corpus) {
        set_error("rf_topk_keys_device: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    if ((uint64_t)index_base + corpus->n > 0xFFFFFFFFull) {
        set_error("rf_topk_keys_device: index_base + n must fit 32 bits (use rf_topk_u32 for larger index spaces)");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (corpus->n == 0) {
        RF_HIP(hipMemsetAsync(d_keys_out, 0xFF, (size_t)k * sizeof(uint64_t), st));
        return RF_OK;
    }
    bool desc = false;
    return topk_core(c, corpus, op, args, k, index_base, d_keys_out, out_all, out_all_mem, st, &desc);
}
RF_ABI_CATCH

rf_status rf_topk_merge_keys_device(const uint64_t* d_keys, uint32_t n, uint32_t k, uint64_t* d_out, int device, void* stream)
try {
    if (!d_keys || !d_out || k == 0 || k > (uint32_t)kWave || n == 0) {
        set_error("rf_topk_merge_keys_device: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return RF_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_topk_final(d_keys, n, k, d_out, st);
    if (e != hipSuccess) {
        set_error(std::string("top-k merge: ") + hipGetErrorString(e));
        return RF_ERR_HIP;
    }
    return RF_OK;
}
RF_ABI_CATCH

// ---------------------------------------------------------------------------------------------------
// The exchange step below the host language: all-gather of the per-shard key lists over RCCL + merge.  RCCL is not a
// link-time dependency of this library: the caller owns the communicator, so the RCCL that created it is already in the
// process, and ncclAllGather is looked up in THAT instance (RTLD_NOLOAD), falling back to the system librccl.
// ---------------------------------------------------------------------------------------------------
namespace {
using nccl_all_gather_fn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
nccl_all_gather_fn find_nccl_all_gather()
{
    static nccl_all_gather_fn fn = [] {
        if (void* f = dlsym(RTLD_DEFAULT, "ncclAllGather")) return (nccl_all_gather_fn)f;
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))
                if (void* f = dlsym(h, "ncclAllGather")) return (nccl_all_gather_fn)f;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL))
                if (void* f = dlsym(h, "ncclAllGather")) return (nccl_all_gather_fn)f;
        return (nccl_all_gather_fn) nullptr;
    }();
    return fn;
}
}  // namespace

rf_status rf_topk_allgather_merge(const uint64_t* d_local_keys, uint32_t k, void* nccl_comm, uint32_t world, uint64_t* d_all_keys,
                                  uint64_t* d_merged, int device, void* stream)
try {
    if (!d_local_keys || !d_all_keys || !d_merged || !nccl_comm || k == 0 || k > (uint32_t)kWave || world == 0) {
        set_error("rf_topk_allgather_merge: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    const nccl_all_gather_fn all_gather = find_nccl_all_gather();
    if (!all_gather) {
        set_error("rf_topk_allgather_merge: no RCCL (ncclAllGather) found in this process or on the library path");
        return RF_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return RF_ERR_NO_DEVICE;
    constexpr int kNcclUint64 = 5;  // ncclDataType_t::ncclUint64 (nccl.h)
    const int rc = all_gather(d_local_keys, d_all_keys, k, kNcclUint64, nccl_comm, (hipStream_t)stream);
    if (rc != 0) {
        set_error("rf_topk_allgather_merge: ncclAllGather failed with ncclResult_t " + std::to_string(rc));
        return RF_ERR_HIP;
    }
    return rf_topk_merge_keys_device(d_all_keys, world * k, k, d_merged, device, stream);
}
RF_ABI_CATCH

// ---------------------------------------------------------------------------------------------------
// top-k entries: 16 bytes {order-preserving key, 64-bit global index} -- the exchange format for every top-k (rfgpu.h)
// ---------------------------------------------------------------------------------------------------
uint32_t rf_topk_entry_score_u32(uint64_t key, int descending) { return descending ? 0xFFFFFFFFu - (uint32_t)key : (uint32_t)key; }
double rf_topk_entry_score_f64(uint64_t key, int descending)
{
    uint64_t b = descending ? ~key : key;
    b ^= (b >> 63) ? 0x8000000000000000ull : ~0ull;  // undo the order-preserving map of rf_select.hip KeyOf<uint64_t>
    double d;
    std::memcpy(&d, &b, sizeof(d));
    return d;
}

rf_status rf_topk_entries_device(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint64_t k, uint64_t index_base,
                                 rf_topk_entry* d_entries_out, void* stream)
try {
    if (!c || !corpus || !args || !d_entries_out || k == 0) {
        set_error("rf_topk_entries_device: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (corpus->n == 0) {
        RF_HIP(hipMemsetAsync(d_entries_out, 0xFF, (size_t)k * sizeof(rf_topk_entry), st));
        return RF_OK;
    }
    const bool usize_metric = c->metric == RF_LEVENSHTEIN || c->metric == RF_INDEL || c->metric == RF_LCS_SEQ || c->metric == RF_OSA;
    const bool f64 = !usize_metric || op == RF_OP_NORMALIZED_DISTANCE || op == RF_OP_NORMALIZED_SIMILARITY;
    if (!f64 && k <= (uint64_t)kWave) {
        // the in-scan lists: keys with the LOCAL index, widened on the device -- nothing synchronizes
        uint64_t* d_keys = nullptr;
        RF_HIP(scratch_alloc((void**)&d_keys, (size_t)kWave * sizeof(uint64_t), st));
        bool desc = false;
        const rf_status s = topk_core(c, corpus, op, args, (uint32_t)k, 0, d_keys, nullptr, RF_MEM_HOST, st, &desc);
        if (s == RF_OK) {
            const hipError_t e = launch_keys_to_entries(d_keys, (uint32_t)k, index_base, d_entries_out, st);
            (void)scratch_free(d_keys, st);
            RF_HIP(e);
            return RF_OK;
        }
        (void)scratch_free(d_keys, st);
        if (s != RF_ERR_UNSUPPORTED) return s;  // (long queries, general weight tables: the selection path below)
    }
    std::vector<uint64_t> keys;
    std::vector<uint32_t> idx;
    bool desc = false;
    const rf_status s = topk_by_selection(c, corpus, op, args, k, f64, nullptr, RF_MEM_HOST, st, &keys, &idx, &desc);
    if (s != RF_OK) return s;
    // (a caller-supplied k far beyond the corpus must not size a host allocation: at most min(k, n) entries exist, the tail of the
    // caller's k-entry buffer is filled with the empty entry on the device -- ADVICE r3)
    const size_t have = keys.size();
    std::vector<rf_topk_entry> host(have);
    for (size_t i = 0; i < have; ++i) {
        // (the u32 selection key of a similarity is 0xFFFFFFFE - score, rf_select.hip: the entry format says 0xFFFFFFFF - score)
        host[i].key = f64 ? keys[i] : (desc ? keys[i] + 1 : keys[i]);
        host[i].index = index_base + idx[i];
    }
    if (have) RF_HIP(hipMemcpyAsync(d_entries_out, host.data(), have * sizeof(rf_topk_entry), hipMemcpyHostToDevice, st));
    if (k > have) RF_HIP(hipMemsetAsync(d_entries_out + have, 0xFF, (size_t)(k - have) * sizeof(rf_topk_entry), st));
    RF_HIP(hipStreamSynchronize(st));  // (`host` dies with this frame)
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_topk_merge_entries_device(const rf_topk_entry* d_entries, uint64_t n, uint64_t k, rf_topk_entry* d_out, int device, void* stream)
try {
    if (!d_entries || !d_out || k == 0 || n == 0 || n > 0x7FFFFFFFull || k > 0x7FFFFFFFull) {
        set_error("rf_topk_merge_entries_device: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return RF_ERR_NO_DEVICE;
    const hipError_t e = launch_merge_entries(d_entries, (uint32_t)n, (uint32_t)k, d_out, (hipStream_t)stream);
    if (e != hipSuccess) {
        set_error(std::string("top-k entry merge: ") + hipGetErrorString(e));
        return RF_ERR_HIP;
    }
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_topk_allgather_merge_entries(const rf_topk_entry* d_local, uint64_t k, void* nccl_comm, uint32_t world, rf_topk_entry* d_all,
                                          rf_topk_entry* d_merged, int device, void* stream)
try {
    if (!d_local || !d_all || !d_merged || !nccl_comm || k == 0 || world == 0) {
        set_error("rf_topk_allgather_merge_entries: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    const nccl_all_gather_fn all_gather = find_nccl_all_gather();
    if (!all_gather) {
        set_error("rf_topk_allgather_merge_entries: no RCCL (ncclAllGather) found in this process or on the library path");
        return RF_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return RF_ERR_NO_DEVICE;
    constexpr int kNcclUint64 = 5;  // ncclDataType_t::ncclUint64 (nccl.h): an entry is two of them
    const int rc = all_gather(d_local, d_all, (size_t)k * 2, kNcclUint64, nccl_comm, (hipStream_t)stream);
    if (rc != 0) {
        set_error("rf_topk_allgather_merge_entries: ncclAllGather failed with ncclResult_t " + std::to_string(rc));
        return RF_ERR_HIP;
    }
    return rf_topk_merge_entries_device(d_all, (uint64_t)world * k, k, d_merged, device, stream);
}
RF_ABI_CATCH

rf_status rf_topk_merge_entries(const rf_topk_entry* entries, uint64_t n, uint64_t k, rf_topk_entry* out)
try {
    if ((n && !entries) || !out || k == 0) {
        set_error("rf_topk_merge_entries: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    std::vector<rf_topk_entry> v;
    for (uint64_t i = 0; i < n; ++i)
        if (entries[i].key != ~0ull || entries[i].index != ~0ull) v.push_back(entries[i]);
    std::sort(v.begin(), v.end(), [](const rf_topk_entry& a, const rf_topk_entry& b) { return a.key != b.key ? a.key < b.key : a.index < b.index; });
    for (uint64_t i = 0; i < k; ++i) out[i] = i < v.size() ? v[i] : rf_topk_entry{~0ull, ~0ull};
    return RF_OK;
}
RF_ABI_CATCH


}  // extern "C"
