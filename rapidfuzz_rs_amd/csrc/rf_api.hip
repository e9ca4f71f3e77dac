// rf_api.hip -- host side of the C ABI declared in include/rfgpu.h.
//
// What runs on the host here is what the reference's host would do around the kernels: copy the query and
// build its BlockPatternMatchVector (src/details/pattern_match_vector.rs:203-224), bucket the candidates by
// length and lay them out for the wavefronts, translate `Args` (levenshtein.rs:1285-1331 weight dispatch)
// into kernel parameters.  No metric is ever evaluated on the host: a shape without a device kernel is
// RF_ERR_UNSUPPORTED.  Product code: never includes or links anything from oracle/.
#include "rf_host.hpp"

namespace rf {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

}  // namespace rf

std::atomic<uint64_t> g_corpus_uid{1};


// Symbol renaming.  Every column of every kernel gathers 64 table rows from LDS, one per lane, and LDS bank
// conflicts between DIFFERENT symbols that share a bank (row index mod 32 for 8-byte rows) are the cost of that
// gather -- ASCII classes collide systematically ('A'/'a', digits/'P'..'Y').  Renaming symbols by frequency rank
// gives the 32 most frequent symbols of THIS corpus 32 distinct banks and pairs the rest with them one by one.
// The packed corpus stores sigma(c); the kernels stage PM row c at LDS row sigma(c); nothing else changes.
void symbol_frequencies(const uint64_t* hist, float* freq)
{
    uint64_t total = 0;
    for (int c = 0; c < 256; ++c) total += hist[c];
    for (int c = 0; c < 256; ++c) freq[c] = total ? (float)((double)hist[c] / (double)total) : 0.0f;
}
static void make_sigma(const uint64_t* hist, uint8_t* sigma)
{
    static const bool disabled = getenv("RF_NO_RENAME") != nullptr;  // tuning / A-B knob
    int order[256];
    for (int i = 0; i < 256; ++i) order[i] = i;
    if (!disabled)
        std::stable_sort(order, order + 256, [&](int a, int b) { return hist[a] > hist[b]; });
    for (int r = 0; r < 256; ++r) sigma[order[r]] = (uint8_t)r;
}

extern "C" {

const char* rf_last_error(void) { return g_last_error.c_str(); }

int rf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rf_args_default(rf_args* a)
{
    if (!a) return;
    std::memset(a, 0, sizeof(*a));
    a->cutoff_usize = RF_NO_CUTOFF;
    a->score_hint_usize = RF_NO_CUTOFF;
    a->cutoff_f64 = std::nan("");
    a->score_hint_f64 = std::nan("");
    a->insertion_cost = a->deletion_cost = a->substitution_cost = 1;  // WeightTable::default(), levenshtein.rs:139-148
    a->prefix_weight = 0.1;                                           // jaro_winkler.rs:36
}

// ---------------------------------------------------------------------------------------------------
// comparator
// ---------------------------------------------------------------------------------------------------
rf_status rf_comparator_new(rf_metric metric, const uint8_t* s1, size_t len1, rf_comparator** out)
try {
    if (!out || (len1 && !s1) || (int)metric < 0 || (int)metric > (int)RF_OSA) {
        set_error("rf_comparator_new: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    rf_comparator* c = new (std::nothrow) rf_comparator();
    if (!c) return RF_ERR_OOM;
    c->metric = metric;
    c->s1.assign(s1, s1 + len1);
    // BlockPatternMatchVector::new + insert (pattern_match_vector.rs:203-224): block = i / 64, the mask
    // rotates left once per position; u8 keys land in extended_ascii[c][block] (:262-265).
    c->block_count = (len1 + 63) / 64;
    c->words = std::max<size_t>(1, c->block_count);
    c->pm.assign(256 * c->words, 0);
    uint64_t mask = 1;
    for (size_t i = 0; i < len1; ++i) {
        c->pm[(size_t)s1[i] * c->words + i / 64] |= mask;
        mask = (mask << 1) | (mask >> 63);
    }
    *out = c;
    return RF_OK;
}
RF_ABI_CATCH

// BatchComparator::new over `char` (or any u32) elements.  The reference hashes non-ASCII symbols into its
// pattern-match table (pattern_match_vector.rs:5-65, :228-260); here the table is built per corpus, in terms of that
// corpus' symbol ids, the first time the comparator meets it (resolve()).
rf_status rf_comparator_new_u32(rf_metric metric, const uint32_t* s1, size_t len1, rf_comparator** out)
try {
    if (!out || (len1 && !s1) || (int)metric < 0 || (int)metric > (int)RF_OSA) {
        set_error("rf_comparator_new_u32: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    rf_comparator* c = new (std::nothrow) rf_comparator();
    if (!c) return RF_ERR_OOM;
    c->metric = metric;
    c->wide = true;
    c->s1w.assign(s1, s1 + len1);
    c->block_count = (len1 + 63) / 64;
    c->words = std::max<size_t>(1, c->block_count);
    *out = c;
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_comparator_clone(const rf_comparator* c, rf_comparator** out)
try {
    if (!c || !out) return RF_ERR_INVALID_ARG;
    if (c->wide) return rf_comparator_new_u32(c->metric, c->s1w.data(), c->s1w.size(), out);
    return rf_comparator_new(c->metric, c->s1.data(), c->s1.size(), out);
}
RF_ABI_CATCH

void rf_comparator_free(rf_comparator* c)
{
    if (!c) return;
    c->lowered.clear();
    for (auto& kv : c->d_pm) {
        DeviceGuard g(kv.first);
        (void)hipFree(kv.second);
    }
    delete c;
}

rf_metric rf_comparator_metric(const rf_comparator* c) { return c->metric; }
size_t rf_comparator_query_len(const rf_comparator* c) { return c->wide ? c->s1w.size() : c->s1.size(); }
const uint64_t* rf_comparator_pm(const rf_comparator* c, size_t* block_count)
{
    if (block_count) *block_count = c->block_count;
    return c->wide ? nullptr : c->pm.data();  // a u32 comparator has one table per corpus alphabet, none of its own
}

// Which byte-level comparator serves (c, corpus).  Byte query x byte corpus: c itself.  Anything involving u32
// symbols is LOWERED to the corpus' alphabet: query symbol -> its id; a symbol the corpus does not contain at all ->
// kAbsentId, an id no candidate byte has, so it can never match (which is all any metric on this path asks of it);
// a symbol the corpus lumped into its overflow class cannot be told apart from the other overflow symbols, so the
// call is refused rather than answered approximately.  Lowered comparators are cached per corpus.
static ComparatorRef own_comparator(rf_comparator* c) { return ComparatorRef(c, [](rf_comparator* p) { rf_comparator_free(p); }); }

rf_status resolve(const rf_comparator* c, const rf_corpus* corpus, const rf_comparator** eff, ComparatorRef* hold, bool* overflow_hit)
{
    if (overflow_hit) *overflow_hit = false;
    if (!c || !corpus) {
        set_error("null handle");
        return RF_ERR_INVALID_ARG;
    }
    if (!c->wide && !corpus->wide) {
        *eff = c;
        return RF_OK;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->lowered.find(corpus->uid);
    if (it != c->lowered.end()) {
        *hold = it->second;
        *eff = hold->get();
        return RF_OK;
    }
    const size_t len = c->wide ? c->s1w.size() : c->s1.size();
    std::vector<uint8_t> ids(len);
    for (size_t i = 0; i < len; ++i) {
        const uint32_t ch = c->wide ? c->s1w[i] : (uint32_t)c->s1[i];
        if (!corpus->wide) {  // u32 query on a byte corpus: bytes are the code points 0..255
            if (ch > 0xFF) {
                set_error("a query symbol above 255 cannot be searched in a byte corpus: pack the corpus with rf_corpus_pack_u32");
                return RF_ERR_UNSUPPORTED;
            }
            ids[i] = (uint8_t)ch;
            continue;
        }
        auto a = corpus->alphabet.find(ch);
        if (a != corpus->alphabet.end()) {
            ids[i] = a->second;
        } else if (corpus->overflow.count(ch)) {
            if (overflow_hit) *overflow_hit = true;  // make_effective() serves it from a translated image if it can
            set_error("the query contains a symbol this corpus stores in its overflow class (more than 254 distinct symbols, "
                      "this one among the rarest), and the corpus carries no raw symbol stream to translate from");
            return RF_ERR_UNSUPPORTED;
        } else {
            ids[i] = kAbsentId;
        }
    }
    rf_comparator* low = nullptr;
    const rf_status s = rf_comparator_new(c->metric, ids.data(), ids.size(), &low);
    if (s != RF_OK) return s;
    if (c->lowered.size() >= 64) c->lowered.erase(c->lowered.begin());  // bounded cache: drop the oldest corpus (calls in flight hold their own reference)
    *hold = own_comparator(low);
    c->lowered[corpus->uid] = *hold;
    *eff = low;
    return RF_OK;
}


rf_status make_effective(const rf_comparator* c_in, const rf_corpus* corpus, hipStream_t st, Effective* e)
{
    bool overflow_hit = false;
    const rf_status rs = resolve(c_in, corpus, &e->c, &e->hold, &overflow_hit);
    e->corpus = corpus;
    e->stream = st;
    if (rs == RF_OK || !overflow_hit || !corpus->d_raw) return rs;

    // query-local ids: 1, 2, ... in order of first appearance.  When the query has at most 255 distinct symbols that numbering does
    // not depend on the corpus and the lowered comparator is kept.  A longer alphabet is still served when the symbols this corpus
    // STORES among them number at most 254: the ones it does not store can never match, so they share the id 255, which the
    // translated image never contains (that lowering belongs to this corpus and is rebuilt per call).
    const size_t len = c_in->wide ? c_in->s1w.size() : c_in->s1.size();
    std::unordered_map<uint32_t, uint8_t> local;
    std::vector<uint8_t> ids(len);
    auto number = [&](bool stored_only) {
        local.clear();
        for (size_t i = 0; i < len; ++i) {
            const uint32_t ch = c_in->wide ? c_in->s1w[i] : (uint32_t)c_in->s1[i];
            if (stored_only && !corpus->alphabet.count(ch) && !corpus->overflow.count(ch)) {
                ids[i] = 255;
                continue;
            }
            auto it = local.find(ch);
            if (it == local.end()) {
                if (local.size() >= (stored_only ? 254u : 255u)) return false;
                it = local.emplace(ch, (uint8_t)(local.size() + 1)).first;
            }
            ids[i] = it->second;
        }
        return true;
    };
    const bool portable = number(false);
    if (!portable && !number(true)) {
        set_error("the query has more than 254 distinct symbols that this corpus stores, and the corpus has an overflow class: "
                  "8-bit ids cannot tell them apart");
        return RF_ERR_UNSUPPORTED;
    }
    if (portable) {
        std::lock_guard<std::mutex> lock(c_in->mu);
        auto it = c_in->lowered.find(~0ull);  // this lowering does not depend on the corpus
        if (it == c_in->lowered.end()) {
            rf_comparator* low = nullptr;
            const rf_status s = rf_comparator_new(c_in->metric, ids.data(), ids.size(), &low);
            if (s != RF_OK) return s;
            it = c_in->lowered.emplace(~0ull, own_comparator(low)).first;
        }
        e->hold = it->second;
        e->c = e->hold.get();
    } else {
        rf_comparator* low = nullptr;
        const rf_status s = rf_comparator_new(c_in->metric, ids.data(), ids.size(), &low);
        if (s != RF_OK) return s;
        e->hold = own_comparator(low);
        e->c = low;
    }
    // (symbol -> id) as an open-addressing table the kernel stages in LDS
    uint32_t cap = 8;
    while (cap < 2 * local.size()) cap *= 2;
    e->keys.assign(cap, 0xFFFFFFFFu);
    e->vals.assign(cap, 0);
    for (const auto& kv : local) {
        if (kv.first == 0xFFFFFFFFu || (corpus->raw_elem == 2 && kv.first >= 0xFFFFu)) continue;  // the raw stream's padding value / not representable in it: never a stored symbol
        uint32_t h = (kv.first * 2654435761u) & (cap - 1);
        while (e->keys[h] != 0xFFFFFFFFu) h = (h + 1) & (cap - 1);
        e->keys[h] = kv.first;
        e->vals[h] = kv.second;
    }
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    {
        std::lock_guard<std::mutex> lock(corpus->scratch_mu);
        if (!corpus->d_sigma_identity) {
            uint8_t ident[256];
            for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
            RF_HIP(hipMalloc((void**)&corpus->d_sigma_identity, 256));
            RF_HIP(hipMemcpy(corpus->d_sigma_identity, ident, 256, hipMemcpyHostToDevice));
        }
    }
    const size_t table_off = (corpus->data_bytes + 15) / 16 * 16;
    RF_HIP(scratch_alloc((void**)&e->temp, table_off + (size_t)cap * 5, st));
    uint32_t* d_keys = reinterpret_cast<uint32_t*>(e->temp + table_off);
    uint8_t* d_vals = reinterpret_cast<uint8_t*>(d_keys + cap);
    RF_HIP(hipMemcpyAsync(d_keys, e->keys.data(), (size_t)cap * 4, hipMemcpyHostToDevice, st));
    RF_HIP(hipMemcpyAsync(d_vals, e->vals.data(), cap, hipMemcpyHostToDevice, st));
    const hipError_t le = launch_translate(corpus->d_raw, corpus->raw_elem, corpus->data_bytes, d_keys, d_vals, cap, e->temp, st);
    if (le != hipSuccess) {
        set_error(std::string("translate: ") + hipGetErrorString(le));
        return RF_ERR_HIP;
    }
    e->image.reset(new (std::nothrow) rf_corpus());
    if (!e->image) return RF_ERR_OOM;
    rf_corpus& v = *e->image;
    v.borrowed = true;
    v.parent = corpus;
    v.no_prefill = corpus->no_prefill;
    v.uid = corpus->uid;
    v.device = corpus->device;
    v.n = corpus->n;
    v.payload_bytes = corpus->payload_bytes;
    v.data_bytes = corpus->data_bytes;
    v.d_data = e->temp;
    v.d_tiles = corpus->d_tiles;
    v.d_orig = corpus->d_orig;
    v.n_tiles = corpus->n_tiles;
    v.n_exact = corpus->n_exact;
    v.n_mixed = corpus->n_mixed;
    v.d_mixed = corpus->d_mixed;
    v.d_mixed_len = corpus->d_mixed_len;
    v.d_mixed_orig = corpus->d_mixed_orig;
    v.mixed = corpus->mixed;
    v.max_len = corpus->max_len;
    v.uniform = corpus->uniform;
    v.uniform_len = corpus->uniform_len;
    v.lengths = corpus->lengths;
    v.length_first_tile = corpus->length_first_tile;
    for (int i = 0; i < 256; ++i) v.sigma[i] = (uint8_t)i;
    v.d_sigma = corpus->d_sigma_identity;
    e->corpus = &v;
    return RF_OK;
}

// (c - h) / c for c = 0..64 common characters and h = 0..32 half-transpositions: the third term of jaro.rs:106-119,
// divided HERE with the host's IEEE divide (the values the reference computes) and looked up by the Jaro kernels'
// no-cutoff epilogue.  One 83 KiB table per device, uploaded on first use and kept for the life of the process.
const double* jaro_device_table(int device)
{
    static std::mutex mu;
    static std::map<int, double*> tabs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = tabs.find(device);
    if (it != tabs.end()) return it->second;
    // behind it (round 4): common / len2 for every candidate length that can take the single-word path, [130][65] -- the kernels
    // used to keep this quotient in a per-wavefront LDS table rebuilt with one division per lane whenever the tile length changed,
    // which by origin (run_many) is every tile
    std::vector<double> h(65 * 33 + 130 * 65);
    for (int c = 0; c <= 64; ++c)
        for (int t = 0; t <= 32; ++t) h[(size_t)c * 33 + t] = c == 0 ? 0.0 : ((double)c - (double)t) / (double)c;
    for (int l = 0; l < 130; ++l)
        for (int c = 0; c <= 64; ++c) h[(size_t)65 * 33 + (size_t)l * 65 + c] = l == 0 ? 0.0 : (double)c / (double)l;
    DeviceGuard guard(device);
    double* d = nullptr;
    if (!guard.ok || hipMalloc((void**)&d, h.size() * sizeof(double)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    tabs[device] = d;
    return d;
}

// row stride (in u64) of the device PM table: the word count, or for patterns beyond the register-resident kernels
// the word count rounded up to whole groups of 8
size_t pm_stride(const rf_comparator* c) { return c->words <= (size_t)kMaxWords ? c->words : (c->words + 7) / 8 * 8; }

rf_status comparator_device_pm(const rf_comparator* c, int device, const uint64_t** d_out)
{
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->d_pm.find(device);
    if (it != c->d_pm.end()) {
        *d_out = it->second;
        return RF_OK;
    }
    uint64_t* d = nullptr;
    const size_t stride = pm_stride(c);
    if (stride == c->words) {
        RF_HIP(hipMalloc(&d, c->pm.size() * sizeof(uint64_t)));
        const hipError_t e = hipMemcpy(d, c->pm.data(), c->pm.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) (void)hipFree(d);
        RF_HIP(e);
    } else {  // long pattern: rows padded with zero words to a whole number of 8-word groups
        std::vector<uint64_t> padded(256 * stride, 0);
        for (size_t ch = 0; ch < 256; ++ch) std::memcpy(&padded[ch * stride], &c->pm[ch * c->words], c->words * sizeof(uint64_t));
        RF_HIP(hipMalloc(&d, padded.size() * sizeof(uint64_t)));
        const hipError_t e = hipMemcpy(d, padded.data(), padded.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) (void)hipFree(d);
        RF_HIP(e);
    }
    c->d_pm[device] = d;
    *d_out = d;
    return RF_OK;
}

// ---------------------------------------------------------------------------------------------------
// corpus
// ---------------------------------------------------------------------------------------------------

// Host-side layout of a ragged candidate set (no device needed): the exact bytes rf_corpus_pack uploads.
//
// Candidates are grouped by exact length.  Every whole multiple of 64 candidates of one length becomes EXACT tiles (64
// lanes, one length, no masks anywhere -- the fast shape).  What is left over of each length (< 64 candidates) goes into a
// pool, sorted by length, and the pool is cut into MIXED tiles: 64 consecutive leftovers with neighbouring lengths share
// one payload block sized for the longest of them.  A mixed tile is visible in two ways:
//   * to the Levenshtein / LCS / OSA scans as ONE tile with a per-lane length (MixedDesc + mixed_len / mixed_orig): the
//     column loop runs to the longest lane and a lane steps only while it still has symbols (scan_kernel_mixed);
//   * to every other kernel (Jaro, generalized weights, long patterns, multi-query, top-k) as one ordinary exact-length
//     TileDesc PER DISTINCT LENGTH in it, all pointing at the shared payload, with an orig[] slice that is kPad for the
//     lanes of other lengths ("virtual tiles": those kernels need no change and pay one pass per distinct length, which
//     is what exact-length tiles cost them before -- but the payload is no longer padded to 64 lanes per length).
// Round 1 padded every length to whole tiles: a corpus of few, long, all-different-length candidates used one lane per
// tile -- 64x the HBM bytes and 64x the scan time of a dense corpus.  Now such a corpus packs to its payload (+ < 2 %).
struct HostLayout {
    std::unique_ptr<uint8_t[]> packed_storage;  // (not a vector: no single-threaded zero fill of a multi-GB buffer)
    uint8_t* packed = nullptr;     // tile payloads + one chunk row of tail padding
    size_t packed_size = 0;
    std::vector<TileDesc> tiles;   // [exact tiles, ascending length | virtual tiles of the mixed section, ascending length]
    uint32_t n_exact = 0;          // tiles [0, n_exact) are exact, [n_exact, size) virtual
    std::vector<uint32_t> orig;    // slot -> original index (kPad = padding lane); empty when identity
    std::vector<MixedDesc> mixed;  // the mixed section as the scans see it
    std::vector<uint32_t> mixed_len, mixed_orig;  // 64 per mixed tile: lane -> length / original index (kPad = no candidate)
    uint64_t payload = 0;
    uint32_t max_len = 0;
    bool identity = true;          // a single length bucket: slot i is candidate i
    uint8_t sigma[256];            // symbol renaming applied to the payload
    float sym_freq[256] = {0};     // relative symbol frequencies (rf_corpus::sym_freq)
};

struct PhaseTimer {  // RF_PACK_TIMING=1 prints where rf_corpus_pack spends its time
    bool on = getenv("RF_PACK_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rf_corpus_pack] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

static rf_status build_layout(const uint8_t* bytes, const uint64_t* offsets, size_t n, HostLayout* L)
{
    PhaseTimer timer;
    if ((n && !offsets) || n >= 0xFFFFFFFFull) {
        set_error("rf_corpus_pack: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    // 1. distinct lengths and their counts (a sorted map: one 4 GiB candidate must not cost a 32 GiB table)
    struct Group {
        uint64_t count = 0, next = 0;
        uint64_t slot0 = 0, off0 = 0, in_exact = 0;  // exact part: first slot, payload offset, candidates in it
        uint64_t pool0 = 0;                           // leftovers: first position in the pool
    };
    std::map<uint32_t, Group> groups;
    uint32_t max_len = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xFFFFFFF0ull) {
            set_error("rf_corpus_pack: offsets must be non-decreasing and candidates shorter than 4 GiB");
            return RF_ERR_INVALID_ARG;
        }
        max_len = std::max(max_len, (uint32_t)(offsets[i + 1] - offsets[i]));
    }
    if (max_len <= (1u << 20)) {  // the usual case: count in a flat table, then keep the non-empty lengths
        std::vector<uint64_t> counts((size_t)max_len + 1, 0);
        for (size_t i = 0; i < n; ++i) counts[offsets[i + 1] - offsets[i]]++;
        for (uint32_t len = 0; len <= max_len; ++len)
            if (counts[len]) groups[len].count = counts[len];
    } else {
        for (size_t i = 0; i < n; ++i) groups[(uint32_t)(offsets[i + 1] - offsets[i])].count++;
    }
    timer.lap("lengths");

    // a single length bucket in original order is addressed arithmetically (tile t at t * tile_bytes), last tile partial
    L->identity = groups.size() <= 1 && tile_bytes(max_len) <= 0xFFFFFFFFull;
    L->max_len = max_len;
    static const bool no_mixed = getenv("RF_NO_MIXED_TILES") != nullptr;  // A/B switch: round-1 layout (every length padded to whole tiles)

    // 2. exact tiles: whole multiples of 64 per length, ascending (everything, for the identity layout)
    uint64_t slots = 0, data_bytes = 0, pool_n = 0;
    for (auto& kv : groups) {
        Group& g = kv.second;
        const bool all = L->identity || no_mixed;
        const uint64_t nt = all ? (g.count + kWave - 1) / kWave : g.count / kWave;
        g.in_exact = all ? g.count : nt * kWave;
        g.slot0 = slots;
        g.off0 = data_bytes;
        for (uint64_t t = 0; t < nt; ++t) {
            L->tiles.push_back(TileDesc{data_bytes, kv.first, (uint32_t)slots});
            slots += kWave;
            data_bytes += tile_bytes(kv.first);
        }
        g.pool0 = pool_n;
        pool_n += g.count - g.in_exact;
    }
    L->n_exact = (uint32_t)L->tiles.size();
    // 3. mixed tiles over the pool (sorted by length because the groups are), and their virtual exact views
    const uint64_t n_mixed = (pool_n + kWave - 1) / kWave;
    std::vector<uint32_t> pool_len(pool_n);
    for (auto& kv : groups)
        for (uint64_t k = 0; k < kv.second.count - kv.second.in_exact; ++k) pool_len[kv.second.pool0 + k] = kv.first;
    std::vector<uint64_t> mixed_off(n_mixed);
    L->mixed_len.assign(n_mixed * kWave, 0);
    L->mixed_orig.assign(n_mixed * kWave, kPad);
    std::vector<std::pair<uint32_t, uint32_t>> vtile_of_len;  // per mixed tile: (length -> virtual tile) as a flat sorted run
    std::vector<uint64_t> vrun0(n_mixed + 1, 0);
    for (uint64_t m = 0; m < n_mixed; ++m) {
        const uint64_t p0 = m * kWave, p1 = std::min<uint64_t>(pool_n, p0 + kWave);
        const uint32_t lo = pool_len[p0], hi = pool_len[p1 - 1];
        mixed_off[m] = data_bytes;
        L->mixed.push_back(MixedDesc{data_bytes, hi, lo, (uint32_t)(m * kWave), 0});
        for (uint64_t q = p0; q < p1; ++q) L->mixed_len[q] = pool_len[q];
        uint32_t prev = 0xFFFFFFFFu;
        for (uint64_t q = p0; q < p1; ++q)
            if (pool_len[q] != prev) {  // one virtual exact tile per distinct length, sharing the payload block
                prev = pool_len[q];
                vtile_of_len.emplace_back(prev, (uint32_t)L->tiles.size());
                L->tiles.push_back(TileDesc{data_bytes, prev, (uint32_t)slots});
                slots += kWave;
            }
        vrun0[m + 1] = vtile_of_len.size();
        data_bytes += tile_bytes(hi);
    }
    if (slots >= 0xFFFFFFFFull) {
        set_error("rf_corpus_pack: too many candidates for one corpus");
        return RF_ERR_INVALID_ARG;
    }

    // 3b. symbol renaming from the byte histogram of the whole payload
    {
        // (any permutation is valid; the frequencies only steer it, so a strided sample of ~256 MiB is enough)
        uint64_t hist[256] = {0};
        const uint64_t first = n ? offsets[0] : 0, total = n ? offsets[n] : 0;
        const uint64_t span = total - first, block = 1u << 16;
        const uint64_t stride = std::max<uint64_t>(1, span / (256ull << 20));
        for (uint64_t b0 = first; b0 < total; b0 += block * stride)
            for (uint64_t b = b0, e = std::min(total, b0 + block); b < e; ++b) hist[bytes[b]]++;
        make_sigma(hist, L->sigma);
        symbol_frequencies(hist, L->sym_freq);
    }
    timer.lap("tiles + symbol histogram");

    // 4. where every candidate goes (k-th of its length, original order preserved) -- sequential, cheap:
    //    place[i] = exact slot, or 0x80000000 | pool position for a leftover
    std::vector<uint32_t> place(n);
    {
        const bool direct = max_len <= (1u << 20);
        std::vector<Group*> by_len;
        if (direct) {
            by_len.assign((size_t)max_len + 1, nullptr);
            for (auto& kv : groups) by_len[kv.first] = &kv.second;
        }
        if (pool_n >= 0x80000000ull) {
            set_error("rf_corpus_pack: too many candidates for one corpus");
            return RF_ERR_INVALID_ARG;
        }
        for (size_t i = 0; i < n; ++i) {
            const uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
            Group& g = direct ? *by_len[len] : groups[len];
            const uint64_t k = g.next++;
            place[i] = k < g.in_exact ? (uint32_t)(g.slot0 + k) : (0x80000000u | (uint32_t)(g.pool0 + (k - g.in_exact)));
            L->payload += len;
        }
    }
    timer.lap("slots");
    // + one readable chunk row: the scan prefetches one row ahead.  The buffer is zero-filled by the worker threads
    // below (first touch in parallel: a single-threaded fill of a multi-GB vector costs as much as the scatter).
    L->packed_storage.reset(new (std::nothrow) uint8_t[data_bytes + kTailPad]);
    if (!L->packed_storage) {
        set_error("rf_corpus_pack: out of host memory");
        return RF_ERR_OOM;
    }
    L->packed = L->packed_storage.get();
    L->packed_size = data_bytes + kTailPad;
    if (!L->identity) L->orig.assign(slots, kPad);

    // 5. scatter the (renamed) bytes into the chunk-interleaved tiles; candidates are independent -> threads
    std::map<uint32_t, std::pair<uint64_t, uint64_t>> base;  // len -> (slot0, off0) of the exact part
    for (auto& kv : groups) base[kv.first] = {kv.second.slot0, kv.second.off0};
    auto worker = [&](size_t lo, size_t hi) {
        uint32_t cached_len = 0xFFFFFFFFu;
        uint64_t c_slot0 = 0, c_off0 = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
            uint8_t* dst;
            if (!(place[i] & 0x80000000u)) {
                if (len != cached_len) {
                    const auto& b = base.find(len)->second;
                    cached_len = len;
                    c_slot0 = b.first;
                    c_off0 = b.second;
                }
                const uint64_t slot = place[i], k = slot - c_slot0;
                if (!L->identity) L->orig[slot] = (uint32_t)i;
                dst = L->packed + c_off0 + (k / kWave) * tile_bytes(len) + (k % kWave) * kChunk;
            } else {
                const uint64_t q = place[i] & 0x7FFFFFFFu, m = q / kWave, lane = q % kWave;
                L->mixed_orig[q] = (uint32_t)i;
                // its virtual tile: the one of this mixed tile with this length
                auto first = vtile_of_len.begin() + vrun0[m], last = vtile_of_len.begin() + vrun0[m + 1];
                auto it = std::lower_bound(first, last, std::make_pair(len, 0u));
                L->orig[(uint64_t)L->tiles[it->second].slot0 + lane] = (uint32_t)i;
                dst = L->packed + mixed_off[m] + lane * kChunk;
            }
            const uint8_t* src = bytes + offsets[i];
            for (uint32_t b = 0; b < len; ++b) dst[(uint64_t)(b / kChunk) * kWave * kChunk + b % kChunk] = L->sigma[src[b]];
        }
    };
    const size_t hw = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 32));
    const size_t nthreads = L->payload < (8u << 20) ? 1 : hw;
    auto run = [&](auto&& fn) {  // fn(t, nthreads)
        if (nthreads == 1) {
            fn(0, 1);
            return;
        }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(fn, t, nthreads);
        for (auto& th : pool) th.join();
    };
    run([&](size_t t, size_t nt) {
        const uint64_t lo = L->packed_size * t / nt, hi = L->packed_size * (t + 1) / nt;
        std::memset(L->packed + lo, 0, hi - lo);
    });
    timer.lap("allocate + zero");
    run([&](size_t t, size_t nt) { worker(n * t / nt, n * (t + 1) / nt); });
    timer.lap("scatter");
    return RF_OK;
}

rf_status rf_corpus_layout_host(const uint8_t* bytes, const uint64_t* offsets, size_t n, rf_host_layout* out)
try {
    if (!out) return RF_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    HostLayout L;
    rf_status s = build_layout(bytes, offsets, n, &L);
    if (s != RF_OK) return s;
    out->n_tiles = (uint32_t)L.tiles.size();
    out->packed_bytes = L.packed_size;
    out->n_slots = L.identity ? 0 : L.orig.size();
    out->identity = L.identity ? 1 : 0;
    out->n_exact = L.n_exact;
    out->n_mixed = (uint32_t)L.mixed.size();
    out->packed = (uint8_t*)std::malloc(std::max<size_t>(1, L.packed_size));
    out->tile_off = (uint64_t*)std::malloc(std::max<size_t>(1, L.tiles.size()) * sizeof(uint64_t));
    out->tile_len = (uint32_t*)std::malloc(std::max<size_t>(1, L.tiles.size()) * sizeof(uint32_t));
    out->tile_slot0 = (uint32_t*)std::malloc(std::max<size_t>(1, L.tiles.size()) * sizeof(uint32_t));
    out->orig = (uint32_t*)std::malloc(std::max<size_t>(1, L.orig.size()) * sizeof(uint32_t));
    if (!out->packed || !out->tile_off || !out->tile_len || !out->tile_slot0 || !out->orig) {
        rf_host_layout_free(out);
        return RF_ERR_OOM;
    }
    std::memcpy(out->packed, L.packed, L.packed_size);
    for (size_t t = 0; t < L.tiles.size(); ++t) {
        out->tile_off[t] = L.tiles[t].data_off;
        out->tile_len[t] = L.tiles[t].len;
        out->tile_slot0[t] = L.tiles[t].slot0;
    }
    if (!L.orig.empty()) std::memcpy(out->orig, L.orig.data(), L.orig.size() * sizeof(uint32_t));
    std::memcpy(out->sigma, L.sigma, 256);
    return RF_OK;
}
RF_ABI_CATCH

void rf_host_layout_free(rf_host_layout* l)
{
    if (!l) return;
    std::free(l->packed);
    std::free(l->tile_off);
    std::free(l->tile_len);
    std::free(l->tile_slot0);
    std::free(l->orig);
    std::memset(l, 0, sizeof(*l));
}


// A second ORDER for the same tiles.  Tiles are stored ascending by length, so the tiles a launch has in flight at any moment
// hold candidates from all over the corpus and their out[orig[slot]] stores never meet in a cache.  But the k-th tile of every
// length holds roughly the same stretch of original indices (a stable counting sort keeps each length's candidates in original
// order): walking the non-empty exact tiles by their first candidate's original index makes concurrently running tiles write one
// compact window of `out`.  Only launches that do not care about tile order use it (full no-cutoff scans of the register-resident
// Levenshtein / LCS / OSA kernels); zero-length tiles and the views keep their positions.
std::vector<TileDesc> tiles_by_origin(const std::vector<TileDesc>& tiles, uint32_t n_exact, const uint32_t* orig)
{
    std::vector<TileDesc> out = tiles;
    uint32_t z = 0;
    while (z < n_exact && tiles[z].len == 0) ++z;
    std::vector<uint64_t> key(n_exact - z);  // (first original index, position): one flat sort, ties keep the storage order
    for (uint32_t i = 0; i < key.size(); ++i) key[i] = (uint64_t)orig[tiles[z + i].slot0] << 32 | (z + i);
    std::sort(key.begin(), key.end());
    for (uint32_t i = 0; i < key.size(); ++i) out[z + i] = tiles[(uint32_t)key[i]];
    return out;
}

static rf_status corpus_from_layout(const HostLayout& L, size_t n, int device, rf_corpus** out)
{
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_corpus_pack: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    rf_corpus* c = new (std::nothrow) rf_corpus();
    if (!c) return RF_ERR_OOM;
    c->uid = g_corpus_uid.fetch_add(1);
    c->device = device;
    c->n = n;
    c->payload_bytes = L.payload;
    c->n_tiles = (uint32_t)L.tiles.size();
    c->n_exact = L.n_exact;
    c->n_mixed = (uint32_t)L.mixed.size();
    c->mixed = L.mixed;
    c->max_len = L.max_len;
    for (size_t t = 0; t < L.tiles.size(); ++t)
        if (c->lengths.empty() || c->lengths.back() != L.tiles[t].len) {
            c->lengths.push_back(L.tiles[t].len);
            c->length_first_tile.push_back((uint32_t)t);
        }
    auto fail = [&](rf_status st) {
        rf_corpus_free(c);
        return st;
    };
    PhaseTimer timer;
    RF_HIP_C(hipMalloc(&c->d_data, L.packed_size));
    RF_HIP_C(hipMemcpy(c->d_data, L.packed, L.packed_size, hipMemcpyHostToDevice));
    timer.lap("hipMalloc + upload");
    c->device_bytes = L.packed_size;
    c->data_bytes = L.packed_size;
    std::memcpy(c->sigma, L.sigma, 256);
    std::memcpy(c->sym_freq, L.sym_freq, sizeof(c->sym_freq));
    RF_HIP_C(hipMalloc(&c->d_sigma, 256));
    RF_HIP_C(hipMemcpy(c->d_sigma, c->sigma, 256, hipMemcpyHostToDevice));
    if (L.identity) {  // one length bucket in original order: tiles are addressed arithmetically
        c->uniform = true;
        c->uniform_len = L.max_len;
    } else {
        RF_HIP_C(hipMalloc(&c->d_tiles, L.tiles.size() * sizeof(TileDesc)));
        RF_HIP_C(hipMemcpy(c->d_tiles, L.tiles.data(), L.tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        RF_HIP_C(hipMalloc(&c->d_orig, L.orig.size() * sizeof(uint32_t)));
        RF_HIP_C(hipMemcpy(c->d_orig, L.orig.data(), L.orig.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->n_slots = L.orig.size();
        if (L.n_exact >= 1024) {  // (small corpora fit the caches whatever the order)
            const std::vector<TileDesc> ordered = tiles_by_origin(L.tiles, L.n_exact, L.orig.data());
            RF_HIP_C(hipMalloc(&c->d_tiles_by_origin, ordered.size() * sizeof(TileDesc)));
            RF_HIP_C(hipMemcpy(c->d_tiles_by_origin, ordered.data(), ordered.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        }
        c->device_bytes += L.tiles.size() * sizeof(TileDesc) + L.orig.size() * sizeof(uint32_t);
    }
    if (c->n_mixed) {
        RF_HIP_C(hipMalloc(&c->d_mixed, L.mixed.size() * sizeof(MixedDesc)));
        RF_HIP_C(hipMemcpy(c->d_mixed, L.mixed.data(), L.mixed.size() * sizeof(MixedDesc), hipMemcpyHostToDevice));
        RF_HIP_C(hipMalloc(&c->d_mixed_len, L.mixed_len.size() * sizeof(uint32_t)));
        RF_HIP_C(hipMemcpy(c->d_mixed_len, L.mixed_len.data(), L.mixed_len.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        RF_HIP_C(hipMalloc(&c->d_mixed_orig, L.mixed_orig.size() * sizeof(uint32_t)));
        RF_HIP_C(hipMemcpy(c->d_mixed_orig, L.mixed_orig.data(), L.mixed_orig.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->device_bytes += L.mixed.size() * sizeof(MixedDesc) + 2 * L.mixed_len.size() * sizeof(uint32_t);
    }
    *out = c;
    return RF_OK;
}

// ---------------------------------------------------------------------------------------------------
// rf_corpus_pack on the device (round 6): raw bytes + offsets go up as they are, the GPU does the per-candidate work (rf_pack_ragged.hip has the scheme);
// what is left here is the part of the layout that is a function of the LENGTH HISTOGRAM alone -- the same arithmetic as build_layout steps 1-3, which stays the
// specification (rf_corpus_layout_host) and which the device-packed corpus is held to byte for byte (tests/test_gpu_filter.py, through rf_corpus_save).
// ---------------------------------------------------------------------------------------------------
namespace {

// pageable host memory -> device at the link's rate: worker threads copy slices into pinned staging buffers (kept per process) and enqueue them on streams of
// their own -- hipMemcpy from pageable memory stages through ONE internal buffer on ONE thread (~12 GB/s here; this: the sum of the threads' memcpy rates, up to the link)
struct Staging {
    std::mutex mu;
    std::vector<void*> bufs;  // pinned, kSlice bytes each
    std::vector<hipStream_t> streams;
    static constexpr size_t kSlice = 8u << 20, kPerThread = 2;
};
Staging& staging()
{
    static Staging* s = new Staging();
    return *s;
}
hipError_t staged_upload(void* d_dst, const void* h_src, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    Staging& S = staging();
    std::lock_guard<std::mutex> lock(S.mu);  // (one staged upload at a time per process: the threads would only fight over the link)
    const size_t threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)16, (bytes + Staging::kSlice - 1) / Staging::kSlice}));
    while (S.streams.size() < threads) {
        hipStream_t st = nullptr;
        if (const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking); e != hipSuccess) return e;
        S.streams.push_back(st);
    }
    while (S.bufs.size() < threads * Staging::kPerThread) {
        void* p = nullptr;
        if (const hipError_t e = hipHostMalloc(&p, Staging::kSlice, hipHostMallocDefault); e != hipSuccess) return e;
        S.bufs.push_back(p);
    }
    int device = 0;
    (void)hipGetDevice(&device);
    const size_t slices = (bytes + Staging::kSlice - 1) / Staging::kSlice;
    std::vector<hipError_t> errs(threads, hipSuccess);
    auto worker = [&](size_t t) {
        (void)hipSetDevice(device);
        hipEvent_t ev[Staging::kPerThread] = {nullptr, nullptr};
        bool used[Staging::kPerThread] = {false, false};
        for (auto& e : ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) errs[t] = hipErrorUnknown;
        size_t turn = 0;
        for (size_t s = t; s < slices && errs[t] == hipSuccess; s += threads, ++turn) {
            const size_t b = turn % Staging::kPerThread;
            if (used[b]) errs[t] = hipEventSynchronize(ev[b]);  // the copy that last read this staging buffer
            if (errs[t] != hipSuccess) break;
            const size_t off = s * Staging::kSlice, m = std::min(Staging::kSlice, bytes - off);
            void* stage = S.bufs[t * Staging::kPerThread + b];
            std::memcpy(stage, static_cast<const uint8_t*>(h_src) + off, m);
            errs[t] = hipMemcpyAsync(static_cast<uint8_t*>(d_dst) + off, stage, m, hipMemcpyHostToDevice, S.streams[t]);
            if (errs[t] == hipSuccess) errs[t] = hipEventRecord(ev[b], S.streams[t]);
            used[b] = true;
        }
        if (errs[t] == hipSuccess) errs[t] = hipStreamSynchronize(S.streams[t]);
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    for (const hipError_t e : errs)
        if (e != hipSuccess) return e;
    return hipSuccess;
}

struct DeviceTemps {  // released on every way out
    std::vector<void*> ptrs;
    ~DeviceTemps()
    {
        for (void* p : ptrs) (void)hipFree(p);
    }
    hipError_t get(void** p, size_t bytes)
    {
        const hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 256));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

}  // namespace

// *declined = true: this corpus is the host packer's (too small to pay for the launches, a candidate beyond 65535 symbols, no room for the temporaries)
static rf_status pack_ragged_device(const uint8_t* bytes, const uint64_t* offsets, size_t n, int device, rf_corpus** out, bool* declined)
{
    *declined = true;
    static const size_t min_n = [] { const char* e = getenv("RF_DEVICE_PACK_MIN"); return e ? (size_t)atoll(e) : (size_t)1 << 16; }();  // 0: never
    constexpr uint32_t kMaxLen = 0xFFFFu;
    if (!min_n || n < min_n || n >= 0x7FFFFFFFull || !offsets) return RF_OK;
    const uint64_t first = offsets[0], total = offsets[n];
    if (total < first || (total > first && !bytes)) return RF_OK;  // (the host packer words the error)
    PhaseTimer timer;
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_corpus_pack: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    DeviceTemps tmp;
    auto soft = [&](hipError_t e) {  // a temporary that cannot be had: the host packer, not an error
        if (e == hipSuccess) return false;
        (void)hipGetLastError();
        return true;
    };
    uint8_t* d_bytes = nullptr;
    uint64_t* d_off = nullptr;
    uint32_t *d_keys = nullptr, *d_vals = nullptr, *d_keys2 = nullptr, *d_vals2 = nullptr, *d_status = nullptr;
    unsigned long long *d_counts = nullptr, *d_hist = nullptr;
    if (soft(tmp.get((void**)&d_bytes, total - first + 16)) || soft(tmp.get((void**)&d_off, (n + 1) * sizeof(uint64_t))) || soft(tmp.get((void**)&d_keys, n * 4)) ||
        soft(tmp.get((void**)&d_vals, n * 4)) || soft(tmp.get((void**)&d_keys2, n * 4)) || soft(tmp.get((void**)&d_vals2, n * 4)) ||
        soft(tmp.get((void**)&d_counts, ((size_t)kMaxLen + 1) * 8)) || soft(tmp.get((void**)&d_hist, 256 * 8)) || soft(tmp.get((void**)&d_status, 8)))
        return RF_OK;
    RF_HIP(hipMemsetAsync(d_counts, 0, ((size_t)kMaxLen + 1) * 8, nullptr));
    RF_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, nullptr));
    RF_HIP(hipMemsetAsync(d_status, 0, 8, nullptr));
    RF_HIP(hipMemsetAsync(d_bytes + (total - first), 0, 16, nullptr));
    RF_HIP(hipStreamSynchronize(nullptr));
    // ---- 1. the input, as it is
    RF_HIP(staged_upload(d_off, offsets, (n + 1) * sizeof(uint64_t)));
    RF_HIP(staged_upload(d_bytes, bytes + first, total - first));
    timer.lap("device: upload bytes+offsets");
    // ---- 2. lengths (sort keys), their histogram, validation; the byte sample behind the symbol renaming
    RF_HIP(launch_ragged_lengths(d_off, (uint32_t)n, kMaxLen, d_keys, d_vals, d_counts, d_status, nullptr));
    const uint64_t span = total - first, block = 1u << 16, stride = std::max<uint64_t>(1, span / (256ull << 20));
    RF_HIP(launch_ragged_byte_hist(d_bytes, 0, span, stride, d_hist, nullptr));
    std::vector<unsigned long long> counts((size_t)kMaxLen + 1);
    uint64_t hist[256];
    uint32_t status[2] = {0, 0};
    RF_HIP(hipMemcpyAsync(status, d_status, 8, hipMemcpyDeviceToHost, nullptr));
    RF_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, nullptr));
    RF_HIP(hipStreamSynchronize(nullptr));
    if (status[1] & 1u) {
        set_error("rf_corpus_pack: offsets must be non-decreasing and candidates shorter than 4 GiB");
        *declined = false;
        return RF_ERR_INVALID_ARG;
    }
    if (status[1] & 2u) return RF_OK;  // a candidate beyond 65535 symbols: the host packer
    const uint32_t max_len = status[0];
    RF_HIP(hipMemcpy(counts.data(), d_counts, ((size_t)max_len + 1) * 8, hipMemcpyDeviceToHost));
    (void)block;
    timer.lap("device: lengths + histograms");
    // ---- 3. the layout that follows from the histogram (build_layout steps 1-3, same arithmetic)
    struct Group {
        uint32_t len;
        uint64_t count, slot0, off0, in_exact, pool0, start;
    };
    std::vector<Group> groups;
    {
        uint64_t at = 0;
        for (uint32_t len = 0; len <= max_len; ++len)
            if (counts[len]) {
                groups.push_back(Group{len, counts[len], 0, 0, 0, 0, at});
                at += counts[len];
            }
    }
    const bool identity = groups.size() <= 1 && tile_bytes(max_len) <= 0xFFFFFFFFull;
    static const bool no_mixed = getenv("RF_NO_MIXED_TILES") != nullptr;
    std::vector<TileDesc> tiles;
    uint64_t slots = 0, data_bytes = 0, pool_n = 0;
    for (Group& g : groups) {
        const bool all = identity || no_mixed;
        const uint64_t nt = all ? (g.count + kWave - 1) / kWave : g.count / kWave;
        g.in_exact = all ? g.count : nt * kWave;
        g.slot0 = slots;
        g.off0 = data_bytes;
        for (uint64_t t = 0; t < nt; ++t) {
            tiles.push_back(TileDesc{data_bytes, g.len, (uint32_t)slots});
            slots += kWave;
            data_bytes += tile_bytes(g.len);
        }
        g.pool0 = pool_n;
        pool_n += g.count - g.in_exact;
    }
    const uint32_t n_exact = (uint32_t)tiles.size();
    const uint64_t n_mixed = (pool_n + kWave - 1) / kWave;
    std::vector<uint32_t> pool_len(n_mixed * kWave, 0u), pool_spos(n_mixed * kWave, 0u), pool_vslot0(n_mixed * kWave, 0u);
    for (const Group& g : groups)
        for (uint64_t k = 0; k < g.count - g.in_exact; ++k) {
            pool_len[g.pool0 + k] = g.len;
            pool_spos[g.pool0 + k] = (uint32_t)(g.start + g.in_exact + k);
        }
    std::vector<MixedDesc> mixed;
    for (uint64_t m = 0; m < n_mixed; ++m) {
        const uint64_t p0 = m * kWave, p1 = std::min<uint64_t>(pool_n, p0 + kWave);
        const uint32_t lo = pool_len[p0], hi = pool_len[p1 - 1];
        mixed.push_back(MixedDesc{data_bytes, hi, lo, (uint32_t)(m * kWave), 0});
        uint32_t prev = 0xFFFFFFFFu, vslot0 = 0;
        for (uint64_t q = p0; q < p1; ++q) {
            if (pool_len[q] != prev) {  // one virtual exact tile per distinct length, sharing the payload block
                prev = pool_len[q];
                vslot0 = (uint32_t)slots;
                tiles.push_back(TileDesc{data_bytes, prev, (uint32_t)slots});
                slots += kWave;
            }
            pool_vslot0[q] = vslot0;
        }
        data_bytes += tile_bytes(hi);
    }
    if (slots >= 0xFFFFFFFFull || pool_n >= 0x80000000ull) {
        set_error("rf_corpus_pack: too many candidates for one corpus");
        *declined = false;
        return RF_ERR_INVALID_ARG;
    }
    timer.lap("device: layout from the histogram");
    // ---- 4. the corpus object and its buffers
    rf_corpus* c = new (std::nothrow) rf_corpus();
    if (!c) return RF_ERR_OOM;
    *declined = false;
    auto fail = [&](rf_status st) {
        rf_corpus_free(c);
        return st;
    };
    c->uid = g_corpus_uid.fetch_add(1);
    c->device = device;
    c->n = n;
    c->payload_bytes = span;
    c->n_tiles = (uint32_t)tiles.size();
    c->n_exact = n_exact;
    c->n_mixed = (uint32_t)mixed.size();
    c->mixed = mixed;
    c->max_len = max_len;
    for (size_t t = 0; t < tiles.size(); ++t)
        if (c->lengths.empty() || c->lengths.back() != tiles[t].len) {
            c->lengths.push_back(tiles[t].len);
            c->length_first_tile.push_back((uint32_t)t);
        }
    make_sigma(hist, c->sigma);
    symbol_frequencies(hist, c->sym_freq);
    const size_t packed_size = data_bytes + kTailPad;
    RF_HIP_C(hipMalloc(&c->d_data, packed_size));
    c->device_bytes = c->data_bytes = packed_size;
    RF_HIP_C(hipMemsetAsync(c->d_data + data_bytes, 0, kTailPad, nullptr));
    RF_HIP_C(hipMalloc(&c->d_sigma, 256));
    RF_HIP_C(hipMemcpyAsync(c->d_sigma, c->sigma, 256, hipMemcpyHostToDevice, nullptr));
    if (identity) {
        c->uniform = true;
        c->uniform_len = max_len;
    } else {
        RF_HIP_C(hipMalloc(&c->d_tiles, tiles.size() * sizeof(TileDesc)));
        RF_HIP_C(hipMemcpyAsync(c->d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(hipMalloc(&c->d_orig, slots * sizeof(uint32_t)));
        RF_HIP_C(hipMemsetAsync(c->d_orig, 0xFF, slots * sizeof(uint32_t), nullptr));  // (the views' lanes of other lengths stay kPad)
        c->n_slots = slots;
        c->device_bytes += tiles.size() * sizeof(TileDesc) + slots * sizeof(uint32_t);
    }
    if (c->n_mixed) {
        RF_HIP_C(hipMalloc(&c->d_mixed, mixed.size() * sizeof(MixedDesc)));
        RF_HIP_C(hipMemcpyAsync(c->d_mixed, mixed.data(), mixed.size() * sizeof(MixedDesc), hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(hipMalloc(&c->d_mixed_len, pool_len.size() * sizeof(uint32_t)));
        RF_HIP_C(hipMemcpyAsync(c->d_mixed_len, pool_len.data(), pool_len.size() * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(hipMalloc(&c->d_mixed_orig, pool_len.size() * sizeof(uint32_t)));
        c->device_bytes += mixed.size() * sizeof(MixedDesc) + 2 * pool_len.size() * sizeof(uint32_t);
    }
    // ---- 5. the order: (length, index) pairs sorted by length over the significant bits, stable
    uint32_t bits = 1;
    while (bits < 32 && (max_len >> bits)) ++bits;
    void* d_sort_temp = nullptr;
    const size_t sort_bytes = ragged_sort_temp_bytes((uint32_t)n);
    RF_HIP_C(tmp.get(&d_sort_temp, sort_bytes));
    RF_HIP_C(launch_ragged_sort(d_keys, d_keys2, d_vals, d_vals2, (uint32_t)n, bits, d_sort_temp, sort_bytes, nullptr));
    // ---- 6. the scatter, by destination tile
    std::vector<uint32_t> by_len((size_t)max_len + 1, 0u), g_start(groups.size()), g_slot0(groups.size()), g_in_exact(groups.size());
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        by_len[groups[gi].len] = (uint32_t)gi;
        g_start[gi] = (uint32_t)groups[gi].start, g_slot0[gi] = (uint32_t)groups[gi].slot0, g_in_exact[gi] = (uint32_t)groups[gi].in_exact;
    }
    uint32_t *d_by_len = nullptr, *d_g = nullptr, *d_pool = nullptr;
    const size_t G = std::max<size_t>(groups.size(), 1), P = pool_len.size();
    RF_HIP_C(tmp.get((void**)&d_by_len, by_len.size() * 4));
    RF_HIP_C(tmp.get((void**)&d_g, 3 * G * 4));
    RF_HIP_C(tmp.get((void**)&d_pool, 2 * std::max<size_t>(P, 1) * 4));
    RF_HIP_C(hipMemcpyAsync(d_by_len, by_len.data(), by_len.size() * 4, hipMemcpyHostToDevice, nullptr));
    if (!groups.empty()) {
        RF_HIP_C(hipMemcpyAsync(d_g, g_start.data(), groups.size() * 4, hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(hipMemcpyAsync(d_g + G, g_slot0.data(), groups.size() * 4, hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(hipMemcpyAsync(d_g + 2 * G, g_in_exact.data(), groups.size() * 4, hipMemcpyHostToDevice, nullptr));
    }
    RF_HIP_C(launch_ragged_scatter_tiles(d_bytes, first, d_off, d_vals2, identity ? nullptr : c->d_tiles, max_len, n_exact, d_by_len, d_g, d_g + G, d_g + 2 * G, c->d_sigma, c->d_data,
                                         c->d_orig, nullptr));
    if (c->n_mixed) {
        RF_HIP_C(hipMemcpyAsync(d_pool, pool_spos.data(), P * 4, hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(hipMemcpyAsync(d_pool + P, pool_vslot0.data(), P * 4, hipMemcpyHostToDevice, nullptr));
        RF_HIP_C(launch_ragged_scatter_mixed(d_bytes, first, d_off, d_vals2, c->d_mixed, c->n_mixed, (uint32_t)pool_n, c->d_mixed_len, d_pool, d_pool + P, c->d_sigma, c->d_data, c->d_orig,
                                             c->d_mixed_orig, nullptr));
    }
    // ---- 7. the second tile order (tiles_by_origin): sorted on the device too (the host's std::sort of 1.5 M keys was most of what was left of the call)
    if (!identity && n_exact >= 1024) {
        uint32_t z = 0;
        while (z < n_exact && tiles[z].len == 0) ++z;
        RF_HIP_C(hipMalloc(&c->d_tiles_by_origin, tiles.size() * sizeof(TileDesc)));
        RF_HIP_C(hipMemcpyAsync(c->d_tiles_by_origin, c->d_tiles, tiles.size() * sizeof(TileDesc), hipMemcpyDeviceToDevice, nullptr));
        // (the candidates' sort buffers are free again: n >= n_exact entries each)
        RF_HIP_C(launch_tiles_by_origin(c->d_orig, c->d_tiles, z, n_exact, d_keys, d_vals, d_keys2, d_vals2, d_sort_temp, sort_bytes, c->d_tiles_by_origin, nullptr));
    }
    RF_HIP_C(hipStreamSynchronize(nullptr));  // the input is only borrowed for the duration of the call; the temporaries go with `tmp`
    timer.lap("device: sort + scatter + tile order");
    *out = c;
    return RF_OK;
}

rf_status rf_corpus_pack(const uint8_t* bytes, const uint64_t* offsets, size_t n, int device, rf_corpus** out)
try {
    if (!out) {
        set_error("rf_corpus_pack: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    {   // large inputs: the per-candidate work on the device (RF_DEVICE_PACK_MIN=<candidates>, default 65536; 0 = always the host packer)
        bool declined = true;
        const rf_status sd = pack_ragged_device(bytes, offsets, n, device, out, &declined);
        if (!declined) return sd;
    }
    HostLayout L;
    const rf_status s = build_layout(bytes, offsets, n, &L);
    if (s != RF_OK) return s;
    return corpus_from_layout(L, n, device, out);
}
RF_ABI_CATCH

// Candidates over `char` (or any u32) elements.  The corpus gets its own alphabet: the 254 most frequent symbols
// become byte ids 0..253 (frequency order, so the LDS rows the kernels touch most sit in distinct banks), every rarer
// symbol becomes kOverflowId, and the id bytes are packed exactly like a byte corpus.  Results are exact for every
// query that contains no overflow symbol (resolve()): candidate symbols outside the query only ever need to be
// "not equal", and the lumped ones still are.
rf_status rf_corpus_pack_u32(const uint32_t* elems, const uint64_t* offsets, size_t n, int device, rf_corpus** out)
try {
    if (!out || (n && !offsets)) {
        set_error("rf_corpus_pack_u32: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    const uint64_t total = n ? offsets[n] : 0;
    if (total && !elems) {
        set_error("rf_corpus_pack_u32: null elements");
        return RF_ERR_INVALID_ARG;
    }
    // histogram: a direct table for the BMP, a hash map above it; one pair per worker thread, merged afterwards
    const size_t nthreads = total < (4u << 20) ? 1 : std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 32));
    auto run = [&](auto&& fn) {  // fn(t): elements [total * t / nthreads, total * (t + 1) / nthreads)
        if (nthreads == 1) {
            fn((size_t)0);
            return;
        }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(fn, t);
        for (auto& th : pool) th.join();
    };
    std::vector<std::vector<uint64_t>> lows(nthreads, std::vector<uint64_t>(0x10000, 0));
    std::vector<std::unordered_map<uint32_t, uint64_t>> highs(nthreads);
    run([&](size_t t) {
        std::vector<uint64_t>& lo = lows[t];
        std::unordered_map<uint32_t, uint64_t>& hi = highs[t];
        for (uint64_t i = total * t / nthreads, e = total * (t + 1) / nthreads; i < e; ++i) {
            const uint32_t ch = elems[i];
            if (ch < 0x10000)
                ++lo[ch];
            else
                ++hi[ch];
        }
    });
    std::vector<uint64_t>& low = lows[0];
    std::unordered_map<uint32_t, uint64_t>& high = highs[0];
    for (size_t t = 1; t < nthreads; ++t) {
        for (uint32_t ch = 0; ch < 0x10000; ++ch) low[ch] += lows[t][ch];
        for (const auto& kv : highs[t]) high[kv.first] += kv.second;
    }
    if (high.count(0xFFFFFFFFu)) {
        set_error("rf_corpus_pack_u32: the symbol 0xFFFFFFFF is reserved");
        return RF_ERR_INVALID_ARG;
    }
    std::vector<std::pair<uint64_t, uint32_t>> syms;  // (count, symbol)
    for (uint32_t ch = 0; ch < 0x10000; ++ch)
        if (low[ch]) syms.emplace_back(low[ch], ch);
    for (const auto& kv : high) syms.emplace_back(kv.second, kv.first);
    std::sort(syms.begin(), syms.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    std::unordered_map<uint32_t, uint8_t> alphabet;
    std::unordered_set<uint32_t> overflow;
    for (size_t r = 0; r < syms.size(); ++r) {
        if (r < (size_t)kOverflowId)
            alphabet.emplace(syms[r].second, (uint8_t)r);
        else
            overflow.insert(syms[r].second);
    }
    std::vector<uint16_t> low_id(0x10000, 0xFFFF);
    for (const auto& kv : alphabet)
        if (kv.first < 0x10000) low_id[kv.first] = kv.second;
    std::unique_ptr<uint8_t[]> ids(new (std::nothrow) uint8_t[std::max<uint64_t>(1, total)]);
    if (!ids) {
        set_error("rf_corpus_pack_u32: out of host memory");
        return RF_ERR_OOM;
    }
    run([&](size_t t) {
        for (uint64_t i = total * t / nthreads, e = total * (t + 1) / nthreads; i < e; ++i) {
            const uint32_t ch = elems[i];
            if (ch < 0x10000) {
                ids[i] = low_id[ch] == 0xFFFF ? kOverflowId : (uint8_t)low_id[ch];
            } else {
                auto a = alphabet.find(ch);
                ids[i] = a == alphabet.end() ? kOverflowId : a->second;
            }
        }
    });
    HostLayout L;
    rf_status s = build_layout(ids.get(), offsets, n, &L);
    if (s != RF_OK) return s;
    rf_corpus* c = nullptr;
    s = corpus_from_layout(L, n, device, &c);
    if (s != RF_OK) return s;
    c->wide = true;
    c->alphabet = std::move(alphabet);
    c->overflow = std::move(overflow);
    if (!c->overflow.empty()) {
        // the symbol behind every packed byte, same chunk-interleaved positions (one worker per range of tiles); two bytes
        // per symbol when the whole corpus lies below 0xFFFF (the Basic Multilingual Plane), four otherwise
        const bool narrow = high.empty() && low[0xFFFF] == 0;
        c->raw_elem = narrow ? 2 : 4;
        const size_t raw_bytes = L.packed_size * c->raw_elem;
        std::unique_ptr<uint8_t[]> raw(new (std::nothrow) uint8_t[raw_bytes]);
        if (!raw) {
            rf_corpus_free(c);
            set_error("rf_corpus_pack_u32: out of host memory");
            return RF_ERR_OOM;
        }
        const size_t n_tiles = L.tiles.size();
        auto fill = [&](auto* dst_base) {
            using Sym = std::remove_pointer_t<decltype(dst_base)>;
            auto tiles_worker = [&, dst_base](size_t t0, size_t t1) {
                for (size_t t = t0; t < t1; ++t) {
                    const TileDesc& td = L.tiles[t];
                    for (uint32_t r = 0; r < (uint32_t)kWave; ++r) {
                        const uint64_t slot = (uint64_t)td.slot0 + r;
                        const uint64_t i = L.identity ? slot : (uint64_t)L.orig[slot];
                        if ((L.identity && i >= n) || (!L.identity && i == kPad)) continue;
                        const uint32_t* src = elems + offsets[i];
                        Sym* dst = dst_base + td.data_off + (uint64_t)r * kChunk;
                        for (uint32_t b = 0; b < td.len; ++b) dst[(uint64_t)(b / kChunk) * kWave * kChunk + b % kChunk] = (Sym)src[b];
                    }
                }
            };
            // padding value everywhere first (in parallel): the virtual tiles of a mixed tile share one payload block, so no
            // worker may blank "its" tile after another one has written its own lanes into the same block
            run([&](size_t t) {
                const uint64_t lo = (uint64_t)L.packed_size * t / nthreads, hi = (uint64_t)L.packed_size * (t + 1) / nthreads;
                std::memset(dst_base + lo, 0xFF, (hi - lo) * sizeof(Sym));
            });
            if (nthreads == 1) {
                tiles_worker(0, n_tiles);
            } else {
                std::vector<std::thread> pool;
                for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(tiles_worker, n_tiles * t / nthreads, n_tiles * (t + 1) / nthreads);
                for (auto& th : pool) th.join();
            }
        };
        if (narrow)
            fill(reinterpret_cast<uint16_t*>(raw.get()));
        else
            fill(reinterpret_cast<uint32_t*>(raw.get()));
        DeviceGuard guard(device);
        hipError_t e = hipMalloc(&c->d_raw, raw_bytes);
        if (e == hipSuccess) e = hipMemcpy(c->d_raw, raw.get(), raw_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rf_corpus_free(c);
            set_error(std::string("rf_corpus_pack_u32: ") + hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP;
        }
        c->device_bytes += raw_bytes;
    }
    *out = c;
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_corpus_pack_rows_device(const void* d_rows, size_t n, size_t len, size_t stride, int device, void* stream,
                                     rf_corpus** out)
try {
    if (!out || (n && len && !d_rows) || n >= 0xFFFFFFFFull - kWave || len > 0xFFFFFFF0ull || stride < len) {
        set_error("rf_corpus_pack_rows_device: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_corpus_pack_rows_device: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    rf_corpus* c = new (std::nothrow) rf_corpus();
    if (!c) return RF_ERR_OOM;
    c->uid = g_corpus_uid.fetch_add(1);
    c->device = device;
    c->n = n;
    c->payload_bytes = (uint64_t)n * len;
    c->max_len = (uint32_t)len;
    c->n_tiles = (uint32_t)((n + kWave - 1) / kWave);
    c->n_exact = c->n_tiles;
    if (n) {
        c->lengths.push_back((uint32_t)len);
        c->length_first_tile.push_back(0);
    }
    auto fail = [&](rf_status s) {
        rf_corpus_free(c);
        return s;
    };
    const uint64_t tb = tile_bytes((uint32_t)len);
    const uint64_t data_bytes = tb * c->n_tiles;
    if (tb > 0xFFFFFFFFull) {
        set_error("rf_corpus_pack_rows_device: rows too long");
        return fail(RF_ERR_INVALID_ARG);
    }
    hipStream_t st = (hipStream_t)stream;
    c->uniform = true;
    c->uniform_len = (uint32_t)len;
    RF_HIP_C(hipMalloc(&c->d_data, data_bytes + kTailPad));
    RF_HIP_C(hipMemsetAsync(c->d_data + data_bytes, 0, kTailPad, st));
    RF_HIP_C(hipMalloc(&c->d_sigma, 256));
    {   // rename permutation from the byte histogram of (at most) the first 4 Mi rows
        unsigned long long* d_hist = nullptr;
        RF_HIP_C(hipMalloc(&d_hist, 256 * sizeof(unsigned long long)));
        hipError_t e = hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), st);
        if (e == hipSuccess) e = launch_histogram_rows((const uint8_t*)d_rows, std::min<size_t>(n, (size_t)4 << 20), (uint32_t)len, stride, d_hist, st);
        uint64_t hist[256] = {0};
        if (e == hipSuccess) e = hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_hist);
        RF_HIP_C(e);
        make_sigma(hist, c->sigma);
        symbol_frequencies(hist, c->sym_freq);
        RF_HIP_C(hipMemcpyAsync(c->d_sigma, c->sigma, 256, hipMemcpyHostToDevice, st));
    }
    if (data_bytes) RF_HIP_C(launch_pack_rows((const uint8_t*)d_rows, n, (uint32_t)len, stride, c->d_data, c->n_tiles, c->d_sigma, st));
    RF_HIP_C(hipStreamSynchronize(st));  // the input is only borrowed for the duration of the call
    c->device_bytes = data_bytes + kTailPad;
    c->data_bytes = data_bytes + kTailPad;
    *out = c;
    return RF_OK;
}
RF_ABI_CATCH

void rf_corpus_free(rf_corpus* c)
{
    if (!c) return;
    if (c->borrowed) {
        delete c;
        return;
    }
    DeviceGuard guard(c->device);
    if (c->d_data) (void)hipFree(c->d_data);
    if (c->d_tiles) (void)hipFree(c->d_tiles);
    if (c->d_tiles_by_origin) (void)hipFree(c->d_tiles_by_origin);
    if (c->d_orig) (void)hipFree(c->d_orig);
    if (c->d_heads8) (void)hipFree(c->d_heads8);
    if (c->d_heads6) (void)hipFree(c->d_heads6);
    if (c->d_data6) (void)hipFree(c->d_data6);
    if (c->d_slot_of) (void)hipFree(c->d_slot_of);
    if (c->d_slot_ident) (void)hipFree(c->d_slot_ident);
    if (c->d_window_table) (void)hipFree(c->d_window_table);
    if (c->d_slot_off16) (void)hipFree(c->d_slot_off16);
    if (c->d_len_of) (void)hipFree(c->d_len_of);
    for (const rf_corpus::GatherTmp& t : c->gather_tmp) {
        if (t.done) (void)hipEventDestroy(t.done);
        (void)hipFree(t.ptr);
    }
    for (const rf_corpus::TileList& t : c->tile_lists) {
        (void)hipFree(t.ptr);
        (void)hipEventDestroy(t.done);
        if (t.band_report) (void)hipHostFree(const_cast<uint32_t*>(t.band_report));
    }
    if (c->d_mixed) (void)hipFree(c->d_mixed);
    if (c->d_mixed_len) (void)hipFree(c->d_mixed_len);
    if (c->d_mixed_orig) (void)hipFree(c->d_mixed_orig);
    if (c->d_sigma) (void)hipFree(c->d_sigma);
    if (c->d_raw) (void)hipFree(c->d_raw);
    if (c->d_sigma_identity) (void)hipFree(c->d_sigma_identity);
    for (auto& kv : c->topk_scratch) {
        (void)hipFree(kv.second.cand);
        if (kv.second.scores) (void)hipFree(kv.second.scores);
        if (kv.second.done) (void)hipEventDestroy(kv.second.done);
    }
    delete c;
}

size_t rf_corpus_count(const rf_corpus* c) { return c->n; }
uint64_t rf_corpus_payload_bytes(const rf_corpus* c) { return c->payload_bytes; }
// the packed corpus + every acceleration structure built beside it so far (head plane, tile lists, the gather path's maps and kept
// temporaries -- DESIGN.md 4): what the handle holds in HBM right now
uint64_t rf_corpus_device_bytes(const rf_corpus* c)
{
    if (!c) return 0;
    uint64_t aux = 0;
    {
        std::lock_guard<std::mutex> lock(c->scratch_mu);
        if (c->d_heads8) aux += ((uint64_t)(c->uniform ? c->n_tiles : c->n_exact) + 1) * kWave * 8;
        if (c->d_heads6) aux += ((uint64_t)(c->n_tiles + 1) / 2 + 1) * 3 * kWave * 4;
        if (c->d_data6) aux += ((c->uniform ? (uint64_t)c->n_tiles * ((c->uniform_len + kChunk - 1) / kChunk) : c->data_bytes / (kWave * kChunk)) + 1) * kWave * 12;
        if (c->d_slot_ident) aux += (uint64_t)c->n_slots * sizeof(uint32_t);
        if (c->d_slot_of) aux += (uint64_t)c->n * sizeof(uint32_t);
        if (c->d_window_table) aux += (uint64_t)c->gather_rows * c->gather_runs * sizeof(uint32_t);
        if (c->d_slot_off16) aux += (uint64_t)c->n_slots * sizeof(uint16_t);
        if (c->d_len_of) aux += (uint64_t)c->n * sizeof(uint32_t);
        for (const auto& kv : c->topk_scratch)  // (candidate ways + root table + bound line + control block, and the score vector if any)
            aux += (uint64_t)64 * kv.second.seg_cap * sizeof(uint64_t) + 64 * kWave * sizeof(uint64_t) + 128 + 65 * 128 + (uint64_t)kv.second.scores_cap * sizeof(uint32_t);
    }
    {
        std::lock_guard<std::mutex> lock(c->gather_enqueue_mu);
        for (const rf_corpus::GatherTmp& t : c->gather_tmp) aux += t.bytes;
    }
    {
        std::lock_guard<std::mutex> lock(c->filter_enqueue_mu);
        aux += (uint64_t)c->tile_lists.size() * (9 * (uint64_t)c->n_tiles + 12 * 16384 + 64) * sizeof(uint32_t);  // (rf_api_scan.hip tile_list_words: the lane lists' 16-byte entries since round 6)
    }
    return c->device_bytes + aux;
}
int rf_corpus_device(const rf_corpus* c) { return c->device; }
size_t rf_corpus_alphabet_size(const rf_corpus* c, size_t* overflow_symbols)
{
    if (overflow_symbols) *overflow_symbols = c->overflow.size();
    return c->wide ? c->alphabet.size() : 256;
}


// Issue-rate probe (rf_probe.hip): the product's own column code on register-resident PM words.
rf_status rf_probe_issue_rate(rf_metric metric, uint32_t query_len, uint32_t mode, int device, uint32_t blocks_per_cu, double* wave_columns_per_ns)
try {
    if (!wave_columns_per_ns) return RF_ERR_INVALID_ARG;
    *wave_columns_per_ns = 0.0;
    RawKind raw;
    switch (metric) {
    case RF_LEVENSHTEIN: raw = RAW_LEV; break;
    case RF_INDEL:
    case RF_LCS_SEQ:
    case RF_FUZZ_RATIO: raw = RAW_LCS; break;
    case RF_OSA: raw = RAW_OSA; break;
    default: set_error("rf_probe_issue_rate: no register-only probe for this metric"); return RF_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_probe_issue_rate: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    const hipError_t e = launch_probe(raw, query_len, mode, blocks_per_cu ? (int)blocks_per_cu : 8, 40000, wave_columns_per_ns);
    if (e == hipErrorInvalidValue) {
        set_error("rf_probe_issue_rate: no probe for this query length / mode");
        return RF_ERR_UNSUPPORTED;
    }
    RF_HIP(e);
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_probe_core_clock(int device, uint32_t micros, double* ghz_sleep, double* ghz_counter)
try {
    if (!ghz_sleep || !ghz_counter || micros == 0) {
        set_error("rf_probe_core_clock: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_probe_core_clock: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    // a stream of its own (high priority: the sampler's one wavefront has to get a slot beside a scan that fills the chip)
    hipStream_t st = nullptr;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    RF_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    uint64_t* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, 16);
    const uint32_t sleeps = (uint32_t)std::max<uint64_t>(1, (uint64_t)micros * 2400 / (127 * 64));  // ~micros at 2.4 GHz
    uint64_t h[2] = {0, 0};
    if (e == hipSuccess) e = launch_core_clock(d, sleeps, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d) (void)hipFree(d);
    (void)hipStreamDestroy(st);
    RF_HIP(e);
    if (h[0] == 0) {
        set_error("rf_probe_core_clock: the sampler did not run");
        return RF_ERR_HIP;
    }
    const double ns = (double)h[0] * 10.0;  // s_memrealtime: 100 MHz
    *ghz_sleep = (double)sleeps * 127.0 * 64.0 / ns;
    *ghz_counter = (double)h[1] / ns;
    return RF_OK;
}
RF_ABI_CATCH

rf_status rf_topk_merge_u32(rf_op op, const uint32_t* scores, const uint64_t* indices, const uint32_t* counts,
                            uint32_t lists, uint32_t k, uint32_t* out_score, uint64_t* out_index, uint32_t* out_count)
try {
    if (!scores || !indices || !counts || !out_score || !out_index || !out_count) return RF_ERR_INVALID_ARG;
    struct E {
        uint32_t s;
        uint64_t i;
    };
    std::vector<E> all;
    for (uint32_t l = 0; l < lists; ++l)
        for (uint32_t j = 0; j < std::min(counts[l], k); ++j) all.push_back(E{scores[(size_t)l * k + j], indices[(size_t)l * k + j]});
    const bool desc = op == RF_OP_SIMILARITY;
    std::sort(all.begin(), all.end(), [desc](const E& a, const E& b) {
        if (a.s != b.s) return desc ? a.s > b.s : a.s < b.s;
        return a.i < b.i;
    });
    const uint32_t m = (uint32_t)std::min<size_t>(all.size(), k);
    for (uint32_t j = 0; j < m; ++j) {
        out_score[j] = all[j].s;
        out_index[j] = all[j].i;
    }
    *out_count = m;
    return RF_OK;
}
RF_ABI_CATCH

}  // extern "C"
