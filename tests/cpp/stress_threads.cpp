// Host-thread stress driver over the C ABI (include/rfgpu.h), meant for a ThreadSanitizer build of the library
// (tools/build_tsan.sh): eight host threads, each on its own HIP stream, run a shuffled mix of every call that fills one of the
// per-corpus caches lazily -- head plane, band-filter tile lists, length-run views, gather temporaries, top-k scratch and score
// vectors, translated images of u32 corpora, lowered comparators -- on three SHARED corpora whose caches are still empty when the
// threads start.  Every result is compared with the same call made single-threaded on separate corpus objects beforehand.
// Exit code 0 = all results equal (TSan's own reports go to stderr and its exit code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <random>
#include <thread>
#include <vector>

#include "rfgpu.h"

#define CHECK(x)                                                                      \
    do {                                                                              \
        rf_status s_ = (x);                                                           \
        if (s_ != RF_OK) {                                                            \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, (int)s_, rf_last_error()); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

static const char kAlnum[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789";

struct Corpora {
    rf_corpus *uniform = nullptr, *ragged = nullptr, *wide = nullptr, *small = nullptr;
};

struct Data {
    std::vector<uint8_t> rows, flat;
    std::vector<uint64_t> row_off, flat_off, wide_off;
    std::vector<uint32_t> wide;
    std::vector<uint8_t> q64, q128, q600;
    std::vector<uint32_t> q_rare;
    std::vector<uint8_t> small;  // 20 000 candidates of 0..700 symbols for the heavy kernels (long patterns, general weights, Jaro blocks)
    std::vector<uint64_t> small_off;
    std::vector<std::vector<uint8_t>> q4;  // four short queries for the fused multi-query kernel
};

static void make_data(Data& d)
{
    std::mt19937_64 rng(12345);
    const size_t n1 = 1100000, len = 64;
    d.q64.resize(64);
    for (auto& b : d.q64) b = (uint8_t)kAlnum[rng() % 62];
    d.q128.resize(128);
    for (auto& b : d.q128) b = (uint8_t)kAlnum[rng() % 62];
    d.q600.resize(600);
    for (auto& b : d.q600) b = (uint8_t)kAlnum[rng() % 4];
    for (int i = 0; i < 4; ++i) {
        d.q4.emplace_back(20 + 11 * i);
        for (auto& b : d.q4.back()) b = (uint8_t)kAlnum[rng() % 62];
    }
    d.small_off.assign(1, 0);
    for (size_t i = 0; i < 20000; ++i) {
        const size_t l = i % 50 == 0 ? 500 + rng() % 200 : rng() % 100;
        for (size_t j = 0; j < l; ++j) d.small.push_back(i % 50 == 0 && j < 600 && rng() % 20 ? d.q600[j] : (uint8_t)kAlnum[rng() % 4]);
        d.small_off.push_back(d.small.size());
    }
    d.rows.resize(n1 * len);
    for (auto& b : d.rows) b = (uint8_t)kAlnum[rng() % 62];
    for (size_t i = 450; i < n1; i += 900) {  // near-duplicates of the query: something for cutoffs and top-k to find
        memcpy(&d.rows[i * len], d.q64.data(), len);
        for (unsigned e = 0; e < rng() % 5; ++e) d.rows[i * len + rng() % len] = '#';
    }
    d.row_off.resize(n1 + 1);
    for (size_t i = 0; i <= n1; ++i) d.row_off[i] = i * len;
    const size_t n2 = 2000000;
    d.flat_off.assign(1, 0);
    for (size_t i = 0; i < n2; ++i) {
        const size_t l = 57 + rng() % 8;
        const size_t at = d.flat.size();
        d.flat.resize(at + l);
        if (i % 997 == 0 && l == 64) {
            memcpy(&d.flat[at], d.q64.data(), 64);
            d.flat[at + rng() % 8] = '~';
        } else {
            for (size_t j = 0; j < l; ++j) d.flat[at + j] = (uint8_t)kAlnum[rng() % 62];
        }
        d.flat_off.push_back(d.flat.size());
    }
    // u32 candidates over ~600 symbols with a long tail: more than 254 distinct symbols -> an overflow class
    d.wide_off.assign(1, 0);
    std::vector<uint32_t> count(700, 0);
    for (size_t i = 0; i < 3000; ++i) {
        const size_t l = rng() % 41;
        for (size_t j = 0; j < l; ++j) {
            const double u = (double)(rng() % 1000000) / 1e6;
            const uint32_t sym = (uint32_t)(600.0 * u * u * u);  // skewed towards small ids
            d.wide.push_back(0x4E00 + sym);
            count[sym]++;
        }
        d.wide_off.push_back(d.wide.size());
    }
    uint32_t rare = 599;
    while (rare > 0 && count[rare] == 0) --rare;  // a symbol that occurs, and rarely: overflow class
    for (int i = 0; i < 20; ++i) d.q_rare.push_back(0x4E00 + (uint32_t)i);
    d.q_rare.push_back(0x4E00 + rare);
    d.q_rare.insert(d.q_rare.begin(), 0x4E00 + rare);
}

static Corpora pack(const Data& d)
{
    Corpora c;
    CHECK(rf_corpus_pack(d.rows.data(), d.row_off.data(), d.row_off.size() - 1, 0, &c.uniform));
    CHECK(rf_corpus_pack(d.flat.data(), d.flat_off.data(), d.flat_off.size() - 1, 0, &c.ragged));
    CHECK(rf_corpus_pack_u32(d.wide.data(), d.wide_off.data(), d.wide_off.size() - 1, 0, &c.wide));
    CHECK(rf_corpus_pack(d.small.data(), d.small_off.data(), d.small_off.size() - 1, 0, &c.small));
    return c;
}

using Result = std::vector<uint64_t>;  // raw 8-byte words of whatever the call returned
using Job = std::function<Result(const Corpora&, void*)>;

static rf_args cutoff_u(uint64_t c)
{
    rf_args a;
    rf_args_default(&a);
    a.cutoff_usize = c;
    return a;
}
static rf_args cutoff_f(double c)
{
    rf_args a;
    rf_args_default(&a);
    a.cutoff_f64 = c;
    return a;
}

static Result many_u32(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, rf_op op, const rf_args& a, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    std::vector<uint32_t> out(rf_corpus_count(corpus) + 1, 0);
    CHECK(rf_many_u32(c, corpus, op, &a, out.data(), RF_MEM_HOST, st));
    rf_comparator_free(c);
    Result r((out.size() + 1) / 2, 0);
    memcpy(r.data(), out.data(), out.size() * 4);
    return r;
}
static Result many_f64(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, rf_op op, const rf_args& a, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    std::vector<double> out(rf_corpus_count(corpus));
    CHECK(rf_many_f64(c, corpus, op, &a, out.data(), RF_MEM_HOST, st));
    rf_comparator_free(c);
    Result r(out.size());
    memcpy(r.data(), out.data(), out.size() * 8);
    return r;
}
static Result topk_u32(rf_comparator* c, const rf_corpus* corpus, const rf_args& a, uint32_t k, void* st)
{
    std::vector<uint32_t> score(k);
    std::vector<uint64_t> index(k);
    uint32_t cnt = 0;
    CHECK(rf_topk_u32(c, corpus, RF_OP_DISTANCE, &a, k, 0, score.data(), index.data(), &cnt, nullptr, RF_MEM_HOST, st));
    Result r;
    for (uint32_t i = 0; i < cnt; ++i) {
        r.push_back(score[i]);
        r.push_back(index[i]);
    }
    return r;
}
static Result topk_bytes(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, const rf_args& a, uint32_t k, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    Result r = topk_u32(c, corpus, a, k, st);
    rf_comparator_free(c);
    return r;
}

static Result topk_f64(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, rf_op op, const rf_args& a, uint64_t k, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    std::vector<double> score(k);
    std::vector<uint64_t> index(k);
    uint64_t cnt = 0;
    CHECK(rf_topk_f64(c, corpus, op, &a, k, 0, score.data(), index.data(), &cnt, nullptr, RF_MEM_HOST, st));
    rf_comparator_free(c);
    Result r;
    for (uint64_t i = 0; i < cnt; ++i) {
        uint64_t bits;
        memcpy(&bits, &score[i], 8);
        r.push_back(bits);
        r.push_back(index[i]);
    }
    return r;
}
static Result entries(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, rf_op op, const rf_args& a, uint64_t k, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    rf_topk_entry* d = nullptr;
    if (hipMalloc((void**)&d, k * sizeof(rf_topk_entry)) != hipSuccess) exit(3);
    CHECK(rf_topk_entries_device(c, corpus, op, &a, k, 1000, d, st));
    Result r(2 * k);
    if (hipMemcpyAsync(r.data(), d, k * sizeof(rf_topk_entry), hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess || hipStreamSynchronize((hipStream_t)st) != hipSuccess) exit(3);
    (void)hipFree(d);
    rf_comparator_free(c);
    return r;
}

// rf_filter_*: the (index, score) pairs of the candidates within the cutoff (round 6: pinned report slot per host thread, per-stream lane lists, slot-ordered temporaries)
static Result filter_u32(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, const rf_args& a, rf_filter_order order, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    std::vector<uint64_t> idx(8192);
    std::vector<uint32_t> val(8192);
    uint64_t cnt = 0;
    CHECK(rf_filter_u32(c, corpus, RF_OP_DISTANCE, &a, 7, idx.size(), idx.data(), val.data(), &cnt, RF_MEM_HOST, order, st));
    rf_comparator_free(c);
    Result r{cnt};
    for (uint64_t i = 0; i < std::min<uint64_t>(cnt, idx.size()); ++i) {
        r.push_back(idx[i]);
        r.push_back(val[i]);
    }
    return r;
}
static Result filter_f64(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, const rf_args& a, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    std::vector<uint64_t> idx(8192);
    std::vector<double> val(8192);
    uint64_t cnt = 0;
    CHECK(rf_filter_f64(c, corpus, RF_OP_SIMILARITY, &a, 0, idx.size(), idx.data(), val.data(), &cnt, RF_MEM_HOST, RF_FILTER_BY_SCORE, st));
    rf_comparator_free(c);
    Result r{cnt};
    for (uint64_t i = 0; i < std::min<uint64_t>(cnt, idx.size()); ++i) {
        uint64_t bits;
        memcpy(&bits, &val[i], 8);
        r.push_back(idx[i]);
        r.push_back(bits);
    }
    return r;
}
static Result many_slots_u32(rf_metric m, const std::vector<uint8_t>& q, const rf_corpus* corpus, rf_args a, void* st)
{
    rf_comparator* c = nullptr;
    CHECK(rf_comparator_new(m, q.data(), q.size(), &c));
    a.flags |= RF_FLAG_SLOT_ORDER;
    const size_t slots = rf_corpus_slot_count(corpus);
    std::vector<uint32_t> out(slots, 0), map(slots, 0);
    CHECK(rf_many_u32(c, corpus, RF_OP_DISTANCE, &a, out.data(), RF_MEM_HOST, st));
    CHECK(rf_corpus_slot_index(corpus, map.data(), RF_MEM_HOST));
    rf_comparator_free(c);
    std::vector<uint32_t> back(rf_corpus_count(corpus) + 1, 0);  // the caller's own permutation: padding slots hold unspecified values and are skipped
    for (size_t s2 = 0; s2 < slots; ++s2)
        if (map[s2] != 0xFFFFFFFFu) back[map[s2]] = out[s2];
    Result r((back.size() + 1) / 2, 0);
    memcpy(r.data(), back.data(), back.size() * 4);
    return r;
}

int main(int argc, char** argv)
{
    bool warm = false, noreuse = false;
    std::vector<size_t> only;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "warm")) warm = true;        // experiment: fill every cache single-threaded first
        if (!strcmp(argv[i], "noreuse")) noreuse = true;  // experiment: no cross-stream reuse in the default memory pool
        if (!strncmp(argv[i], "jobs=", 5)) {                // only these jobs (comma-separated indices)
            for (const char* q = argv[i] + 5; *q;) {
                only.push_back((size_t)strtoul(q, (char**)&q, 10));
                if (*q == ',') ++q;
            }
        }
    }
    if (rf_device_count() < 1) {
        printf("no device: nothing to stress\n");
        return 0;
    }
    Data d;
    make_data(d);
    rf_args none;
    rf_args_default(&none);
    rf_comparator* shared_rare = nullptr;  // ONE u32 comparator shared by every thread: its lowered forms are cached inside it
    CHECK(rf_comparator_new_u32(RF_LEVENSHTEIN, d.q_rare.data(), d.q_rare.size(), &shared_rare));
    std::vector<Job> jobs = {
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q64, c.uniform, RF_OP_DISTANCE, none, st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q64, c.uniform, RF_OP_DISTANCE, cutoff_u(3), st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q64, c.uniform, RF_OP_DISTANCE, cutoff_u(1), st); },
        [&](const Corpora& c, void* st) { return many_f64(RF_LEVENSHTEIN, d.q64, c.uniform, RF_OP_NORMALIZED_SIMILARITY, cutoff_f(0.9), st); },
        [&](const Corpora& c, void* st) { return topk_bytes(RF_LEVENSHTEIN, d.q64, c.uniform, cutoff_u(3), 16, st); },
        [&](const Corpora& c, void* st) { return topk_bytes(RF_LEVENSHTEIN, d.q64, c.uniform, none, 16, st); },
        [&](const Corpora& c, void* st) { return topk_bytes(RF_LEVENSHTEIN, d.q128, c.uniform, none, 8, st); },
        [&](const Corpora& c, void* st) { return many_f64(RF_JARO_WINKLER, d.q64, c.uniform, RF_OP_SIMILARITY, cutoff_f(0.9), st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_OSA, d.q64, c.uniform, RF_OP_DISTANCE, cutoff_u(2), st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q64, c.ragged, RF_OP_DISTANCE, none, st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q64, c.ragged, RF_OP_DISTANCE, cutoff_u(3), st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_INDEL, d.q64, c.ragged, RF_OP_DISTANCE, none, st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_INDEL, d.q64, c.ragged, RF_OP_DISTANCE, cutoff_u(12), st); },
        [&](const Corpora& c, void* st) { return topk_bytes(RF_LEVENSHTEIN, d.q64, c.ragged, cutoff_u(4), 16, st); },
        [&](const Corpora& c, void* st) { return many_f64(RF_JARO_WINKLER, d.q64, c.ragged, RF_OP_SIMILARITY, none, st); },
        [&](const Corpora& c, void* st) {
            std::vector<uint32_t> out(rf_corpus_count(c.wide) + 1, 0);
            CHECK(rf_many_u32(shared_rare, c.wide, RF_OP_DISTANCE, &none, out.data(), RF_MEM_HOST, st));
            Result r((out.size() + 1) / 2, 0);
            memcpy(r.data(), out.data(), out.size() * 4);
            return r;
        },
        [&](const Corpora& c, void* st) { return topk_u32(shared_rare, c.wide, none, 5, st); },
        // 17: four queries fused into one pass
        [&](const Corpora& c, void* st) {
            rf_comparator* cs[4];
            for (int i = 0; i < 4; ++i) CHECK(rf_comparator_new(RF_INDEL, d.q4[i].data(), d.q4[i].size(), &cs[i]));
            const size_t n = rf_corpus_count(c.uniform);
            std::vector<uint32_t> out(4 * n);
            CHECK(rf_many_multi_u32(cs, 4, c.uniform, RF_OP_DISTANCE, &none, out.data(), RF_MEM_HOST, st));
            for (int i = 0; i < 4; ++i) rf_comparator_free(cs[i]);
            Result r(2 * n);
            memcpy(r.data(), out.data(), out.size() * 4);
            return r;
        },
        // 18, 19: the selection path (f64 scores; k beyond the in-scan lists)
        [&](const Corpora& c, void* st) { return topk_f64(RF_JARO_WINKLER, d.q64, c.ragged, RF_OP_SIMILARITY, none, 10, st); },
        [&](const Corpora& c, void* st) { return topk_f64(RF_LEVENSHTEIN, d.q64, c.uniform, RF_OP_NORMALIZED_DISTANCE, none, 100, st); },
        // 20: 16-byte entries on the device
        [&](const Corpora& c, void* st) { return entries(RF_INDEL, d.q64, c.ragged, RF_OP_NORMALIZED_SIMILARITY, none, 70, st); },
        // 21..25: the kernels with per-launch scratch -- general weights (rows in registers / LDS / global), long patterns, the band
        // kernel, multi-word and long Jaro
        [&](const Corpora& c, void* st) {
            rf_args a = none;
            a.insertion_cost = 1, a.deletion_cost = 2, a.substitution_cost = 3;
            return many_u32(RF_LEVENSHTEIN, d.q64, c.small, RF_OP_DISTANCE, a, st);
        },
        [&](const Corpora& c, void* st) {
            rf_args a = none;
            a.insertion_cost = 2, a.deletion_cost = 1, a.substitution_cost = 2;
            return many_u32(RF_LEVENSHTEIN, d.q600, c.small, RF_OP_DISTANCE, a, st);
        },
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q600, c.small, RF_OP_DISTANCE, none, st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_LEVENSHTEIN, d.q600, c.small, RF_OP_DISTANCE, cutoff_u(20), st); },
        [&](const Corpora& c, void* st) { return many_f64(RF_JARO_WINKLER, d.q600, c.small, RF_OP_SIMILARITY, none, st); },
        [&](const Corpora& c, void* st) { return many_u32(RF_OSA, d.q600, c.small, RF_OP_DISTANCE, none, st); },
        // 27: the per-candidate method (a one-candidate corpus per call)
        [&](const Corpora&, void*) {
            rf_comparator* cc = nullptr;
            CHECK(rf_comparator_new(RF_LEVENSHTEIN, d.q64.data(), d.q64.size(), &cc));
            uint32_t v = 0;
            int some = 0;
            CHECK(rf_one_u32(cc, d.q128.data(), d.q128.size(), RF_OP_DISTANCE, &none, 0, &v, &some));
            rf_comparator_free(cc);
            return Result{v, (uint64_t)some};
        },
        // 28: a streamed scan of the ragged corpus' file (kept buffer sets: one scan at a time uses them, the others bring their own)
        [&](const Corpora& c, void*) {
            rf_comparator* cc = nullptr;
            CHECK(rf_comparator_new(RF_LEVENSHTEIN, d.q64.data(), d.q64.size(), &cc));
            std::vector<uint32_t> out(rf_corpus_count(c.ragged) + 1, 0);
            CHECK(rf_stream_many_u32(cc, "/tmp/rf_stress_ragged.rfc", RF_OP_DISTANCE, &none, out.data(), out.size(), 32u << 20, 0));
            rf_comparator_free(cc);
            Result r((out.size() + 1) / 2, 0);
            memcpy(r.data(), out.data(), out.size() * 4);
            return r;
        },
        // 29..34 (round 6): compact results -- the lane-compacted road (uniform corpus, cutoff 3 / 1), the general road (cutoff beyond the head plane, a ragged corpus
        // through its slot-ordered temporary, f64 scores by score) -- and per-candidate results in slot order
        [&](const Corpora& c, void* st) { return filter_u32(RF_LEVENSHTEIN, d.q64, c.uniform, cutoff_u(3), RF_FILTER_BY_INDEX, st); },
        [&](const Corpora& c, void* st) { return filter_u32(RF_OSA, d.q64, c.uniform, cutoff_u(1), RF_FILTER_BY_SCORE, st); },
        [&](const Corpora& c, void* st) { return filter_u32(RF_LEVENSHTEIN, d.q64, c.uniform, cutoff_u(30), RF_FILTER_BY_INDEX, st); },
        [&](const Corpora& c, void* st) { return filter_u32(RF_LEVENSHTEIN, d.q64, c.ragged, cutoff_u(3), RF_FILTER_BY_INDEX, st); },
        [&](const Corpora& c, void* st) { return filter_f64(RF_JARO_WINKLER, d.q64, c.ragged, cutoff_f(0.8), st); },
        [&](const Corpora& c, void* st) { return many_slots_u32(RF_INDEL, d.q64, c.ragged, none, st); },
    };
    std::vector<Result> expect;
    {
        Corpora ref = pack(d);
        CHECK(rf_corpus_save(ref.ragged, "/tmp/rf_stress_ragged.rfc"));
        for (auto& j : jobs) expect.push_back(j(ref, nullptr));
        rf_corpus_free(ref.uniform);
        rf_corpus_free(ref.ragged);
        rf_corpus_free(ref.wide);
        rf_corpus_free(ref.small);
    }
    Corpora shared = pack(d);  // fresh objects: every cache is still empty
    if (noreuse) {
        hipMemPool_t pool;
        int off = 0;
        if (hipDeviceGetDefaultMemPool(&pool, 0) == hipSuccess) {
            (void)hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowOpportunistic, &off);
            (void)hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowInternalDependencies, &off);
            (void)hipMemPoolSetAttribute(pool, hipMemPoolReuseFollowEventDependencies, &off);
        }
    }
    if (warm)
        for (auto& j : jobs) (void)j(shared, nullptr);
    std::atomic<int> bad{0}, ready{0};
    const int kThreads = 8;
    std::vector<std::thread> threads;
    for (int t = 0; t < kThreads; ++t)
        threads.emplace_back([&, t] {
            hipStream_t st = nullptr;
            if (hipSetDevice(0) != hipSuccess || hipStreamCreate(&st) != hipSuccess) {
                bad++;
                return;
            }
            std::vector<size_t> order;
            for (size_t i = 0; i < jobs.size(); ++i)
                if (only.empty() || std::find(only.begin(), only.end(), i) != only.end()) order.push_back(i);
            std::mt19937 rng(t);
            ready++;
            while (ready.load() < kThreads) std::this_thread::yield();
            for (int rep = 0; rep < (only.empty() ? 3 : 12); ++rep) {
                std::shuffle(order.begin(), order.end(), rng);
                for (size_t j : order) {
                    const Result got = jobs[j](shared, st);
                    if (got != expect[j]) {
                        size_t at = 0, differing = 0;
                        for (size_t i = 0; i < std::min(got.size(), expect[j].size()); ++i)
                            if (got[i] != expect[j][i]) {
                                if (!differing) at = i;
                                ++differing;
                            }
                        fprintf(stderr, "thread %d rep %d job %zu: result differs (%zu vs %zu words, %zu differ, first at %zu: %016llx vs %016llx)\n", t, rep, j,
                                got.size(), expect[j].size(), differing, at, at < got.size() ? (unsigned long long)got[at] : 0ull,
                                at < expect[j].size() ? (unsigned long long)expect[j][at] : 0ull);
                        bad++;
                    }
                }
            }
            (void)hipStreamDestroy(st);
        });
    for (auto& th : threads) th.join();
    rf_comparator_free(shared_rare);
    rf_corpus_free(shared.uniform);
    rf_corpus_free(shared.ragged);
    rf_corpus_free(shared.wide);
    rf_corpus_free(shared.small);
    remove("/tmp/rf_stress_ragged.rfc");
    printf("stress_threads: %d threads x 3 x %zu jobs, %d mismatches\n", kThreads, jobs.size(), bad.load());
    return bad.load() ? 1 : 0;
}
