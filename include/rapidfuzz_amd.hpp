// rapidfuzz_amd.hpp -- header-only C++17 facade over the C ABI of rfgpu.h that mirrors the reference's names:
//   rapidfuzz::distance::{levenshtein, indel, lcs_seq, jaro, jaro_winkler}::{Args, BatchComparator, distance, ...}
//   rapidfuzz::fuzz::{ratio, RatioBatchComparator}
// `Option<T>` becomes std::optional<T>; the one-vs-many entry points take a rapidfuzz::Corpus.
// Reference: rapidfuzz-rs v0.5.0, e.g. src/distance/levenshtein.rs:86-148 (Args, WeightTable), :1636-1818
// (BatchComparator), src/fuzz.rs:48-150.  Every score is computed by the HIP kernels behind rfgpu.h.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "rfgpu.h"

namespace rapidfuzz {

struct Error : std::runtime_error {
    rf_status status;
    Error(rf_status s, const char* msg) : std::runtime_error(msg), status(s) {}
};
inline void check(rf_status s)
{
    if (s != RF_OK) throw Error(s, rf_last_error());
}

/// The device-resident candidate set (what the user's `for candidate in corpus` loop walks in the reference).
class Corpus {
public:
    Corpus(const std::vector<std::string_view>& candidates, int device = 0)
    {
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> offsets{0};
        for (auto c : candidates) {
            bytes.insert(bytes.end(), c.begin(), c.end());
            offsets.push_back(bytes.size());
        }
        check(rf_corpus_pack(bytes.data(), offsets.data(), candidates.size(), device, &h_));
    }
    /// candidates as code points (`s.chars()` in the reference): one u32 per element, the corpus keeps its own alphabet
    Corpus(const std::vector<std::u32string_view>& candidates, int device = 0)
    {
        std::vector<uint32_t> elems;
        std::vector<uint64_t> offsets{0};
        for (auto c : candidates) {
            elems.insert(elems.end(), c.begin(), c.end());
            offsets.push_back(elems.size());
        }
        check(rf_corpus_pack_u32(elems.data(), offsets.data(), candidates.size(), device, &h_));
    }
    /// a packed corpus written by save()
    static Corpus load(const std::string& path, int device = 0)
    {
        Corpus c;
        check(rf_corpus_load(path.c_str(), device, &c.h_));
        return c;
    }
    void save(const std::string& path) const { check(rf_corpus_save(h_, path.c_str())); }
    /// n rows of `len` bytes already in device memory
    Corpus(const void* d_rows, size_t n, size_t len, size_t stride, int device = 0, void* stream = nullptr)
    {
        check(rf_corpus_pack_rows_device(d_rows, n, len, stride, device, stream, &h_));
    }
    Corpus(const Corpus&) = delete;
    Corpus& operator=(const Corpus&) = delete;
    Corpus(Corpus&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~Corpus() { rf_corpus_free(h_); }
    size_t size() const { return rf_corpus_count(h_); }
    const rf_corpus* handle() const { return h_; }

private:
    Corpus() = default;
    rf_corpus* h_ = nullptr;
};

namespace detail {

struct WeightTable {  // levenshtein.rs:128-148
    size_t insertion_cost = 1, deletion_cost = 1, substitution_cost = 1;
};

/// `Args::default().score_cutoff(x).score_hint(y).weights(&w).prefix_weight(p)`
template <class T>
struct Args {
    std::optional<T> cutoff, hint;
    WeightTable w;
    double pw = 0.1;
    uint32_t flags = 0;
    Args score_cutoff(T v) const { Args a = *this; a.cutoff = v; return a; }
    Args score_hint(T v) const { Args a = *this; a.hint = v; return a; }
    Args weights(const WeightTable& t) const { Args a = *this; a.w = t; return a; }
    Args prefix_weight(double p) const { Args a = *this; a.pw = p; return a; }
    rf_args to_c() const
    {
        rf_args a;
        rf_args_default(&a);
        if constexpr (std::is_floating_point_v<T>) {
            if (cutoff) a.cutoff_f64 = *cutoff;
            if (hint) a.score_hint_f64 = *hint;
        } else {
            if (cutoff) a.cutoff_usize = *cutoff;
            if (hint) a.score_hint_usize = *hint;
        }
        a.insertion_cost = w.insertion_cost;
        a.deletion_cost = w.deletion_cost;
        a.substitution_cost = w.substitution_cost;
        a.prefix_weight = pw;
        a.flags = flags;
        return a;
    }
};

template <rf_metric M, bool FloatMetric>
class BatchComparator {
public:
    using usize_result = std::conditional_t<FloatMetric, double, size_t>;
    explicit BatchComparator(std::string_view s1)  // BatchComparator::new
    {
        check(rf_comparator_new(M, reinterpret_cast<const uint8_t*>(s1.data()), s1.size(), &h_));
    }
    explicit BatchComparator(std::u32string_view s1)  // BatchComparator::new(s1.chars())
    {
        check(rf_comparator_new_u32(M, reinterpret_cast<const uint32_t*>(s1.data()), s1.size(), &h_));
    }
    BatchComparator(const BatchComparator& o) { check(rf_comparator_clone(o.h_, &h_)); }  // #[derive(Clone)]
    BatchComparator& operator=(const BatchComparator&) = delete;
    ~BatchComparator() { rf_comparator_free(h_); }

    // ---- one-vs-many: out[i] = this-><op>_with_args(candidate_i, args)
    std::vector<std::optional<usize_result>> distance_many(const Corpus& c, const Args<usize_result>& a = {}) const
    {
        return many<usize_result>(c, RF_OP_DISTANCE, a.to_c());
    }
    std::vector<std::optional<usize_result>> similarity_many(const Corpus& c, const Args<usize_result>& a = {}) const
    {
        return many<usize_result>(c, RF_OP_SIMILARITY, a.to_c());
    }
    std::vector<std::optional<double>> normalized_distance_many(const Corpus& c, const Args<double>& a = {}) const
    {
        return many<double>(c, RF_OP_NORMALIZED_DISTANCE, a.to_c());
    }
    std::vector<std::optional<double>> normalized_similarity_many(const Corpus& c, const Args<double>& a = {}) const
    {
        return many<double>(c, RF_OP_NORMALIZED_SIMILARITY, a.to_c());
    }

    // ---- only the candidates within the cutoff: the reference user's `corpus.iter().enumerate().filter_map(|(i, c)| scorer.<op>_with_args(c, &args).map(|v| (i, v)))`
    // as ONE device pass that never builds the n-entry vector of Nones (rf_filter_u32 / rf_filter_f64); pairs in ascending index order (or `order`)
    std::vector<std::pair<uint64_t, usize_result>> distance_filter_many(const Corpus& c, const Args<usize_result>& a = {}, rf_filter_order order = RF_FILTER_BY_INDEX) const
    {
        return filter<usize_result>(c, RF_OP_DISTANCE, a.to_c(), order);
    }
    std::vector<std::pair<uint64_t, usize_result>> similarity_filter_many(const Corpus& c, const Args<usize_result>& a = {}, rf_filter_order order = RF_FILTER_BY_INDEX) const
    {
        return filter<usize_result>(c, RF_OP_SIMILARITY, a.to_c(), order);
    }
    std::vector<std::pair<uint64_t, double>> normalized_similarity_filter_many(const Corpus& c, const Args<double>& a = {}, rf_filter_order order = RF_FILTER_BY_INDEX) const
    {
        return filter<double>(c, RF_OP_NORMALIZED_SIMILARITY, a.to_c(), order);
    }

    // ---- many queries x one corpus: res[j][i] = scorers[j].<op>_with_args(candidate_i, args), one call (rf_many_multi_*)
    static std::vector<std::vector<std::optional<usize_result>>> distance_many_multi(const std::vector<const BatchComparator*>& scorers,
                                                                                      const Corpus& c, const Args<usize_result>& a = {})
    {
        const rf_args ca = a.to_c();
        std::vector<const rf_comparator*> hs;
        for (const BatchComparator* s : scorers) hs.push_back(s->h_);
        std::vector<std::vector<std::optional<usize_result>>> res(scorers.size(), std::vector<std::optional<usize_result>>(c.size()));
        if constexpr (FloatMetric) {
            std::vector<double> out(scorers.size() * c.size());
            check(rf_many_multi_f64(hs.data(), (uint32_t)hs.size(), c.handle(), RF_OP_DISTANCE, &ca, out.data(), RF_MEM_HOST, nullptr));
            for (size_t j = 0; j < scorers.size(); ++j)
                for (size_t i = 0; i < c.size(); ++i)
                    if (!std::isnan(out[j * c.size() + i])) res[j][i] = out[j * c.size() + i];
        } else {
            std::vector<uint32_t> out(scorers.size() * c.size());
            check(rf_many_multi_u32(hs.data(), (uint32_t)hs.size(), c.handle(), RF_OP_DISTANCE, &ca, out.data(), RF_MEM_HOST, nullptr));
            for (size_t j = 0; j < scorers.size(); ++j)
                for (size_t i = 0; i < c.size(); ++i)
                    if (out[j * c.size() + i] != RF_NONE_U32) res[j][i] = out[j * c.size() + i];
        }
        return res;
    }

    // ---- the reference's per-candidate methods (a one-candidate corpus through the same kernels)
    std::optional<usize_result> distance_with_args(std::string_view s2, const Args<usize_result>& a) const
    {
        return distance_many(Corpus({s2}), a)[0];
    }
    std::optional<usize_result> similarity_with_args(std::string_view s2, const Args<usize_result>& a) const
    {
        return similarity_many(Corpus({s2}), a)[0];
    }
    std::optional<double> normalized_distance_with_args(std::string_view s2, const Args<double>& a) const
    {
        return normalized_distance_many(Corpus({s2}), a)[0];
    }
    std::optional<double> normalized_similarity_with_args(std::string_view s2, const Args<double>& a) const
    {
        return normalized_similarity_many(Corpus({s2}), a)[0];
    }
    usize_result distance(std::string_view s2) const { return *distance_with_args(s2, {}); }
    usize_result similarity(std::string_view s2) const { return *similarity_with_args(s2, {}); }
    double normalized_distance(std::string_view s2) const { return *normalized_distance_with_args(s2, {}); }
    double normalized_similarity(std::string_view s2) const { return *normalized_similarity_with_args(s2, {}); }

    const rf_comparator* handle() const { return h_; }

private:
    template <class T>
    std::vector<std::optional<T>> many(const Corpus& c, rf_op op, const rf_args& a) const
    {
        std::vector<std::optional<T>> res(c.size());
        if constexpr (std::is_floating_point_v<T>) {
            std::vector<double> out(c.size());
            check(rf_many_f64(h_, c.handle(), op, &a, out.data(), RF_MEM_HOST, nullptr));
            for (size_t i = 0; i < out.size(); ++i)
                if (!std::isnan(out[i])) res[i] = out[i];
        } else {
            std::vector<uint32_t> out(c.size());
            check(rf_many_u32(h_, c.handle(), op, &a, out.data(), RF_MEM_HOST, nullptr));
            for (size_t i = 0; i < out.size(); ++i)
                if (out[i] != RF_NONE_U32) res[i] = out[i];
        }
        return res;
    }
    template <class T>
    std::vector<std::pair<uint64_t, T>> filter(const Corpus& c, rf_op op, const rf_args& a, rf_filter_order order) const
    {
        uint64_t cap = std::max<uint64_t>(1024, c.size() / 64), n = 0;
        for (;;) {  // (the device reports the true number of matches: at most one repeat)
            std::vector<uint64_t> idx(cap);
            std::vector<std::pair<uint64_t, T>> res;
            if constexpr (std::is_floating_point_v<T>) {
                std::vector<double> val(cap);
                check(rf_filter_f64(h_, c.handle(), op, &a, 0, cap, idx.data(), val.data(), &n, RF_MEM_HOST, order, nullptr));
                if (n > cap) { cap = n; continue; }
                for (uint64_t i = 0; i < n; ++i) res.emplace_back(idx[i], val[i]);
            } else {
                std::vector<uint32_t> val(cap);
                check(rf_filter_u32(h_, c.handle(), op, &a, 0, cap, idx.data(), val.data(), &n, RF_MEM_HOST, order, nullptr));
                if (n > cap) { cap = n; continue; }
                for (uint64_t i = 0; i < n; ++i) res.emplace_back(idx[i], (T)val[i]);
            }
            return res;
        }
    }
    rf_comparator* h_ = nullptr;
};

template <rf_metric M, bool F>
struct Module {
    using WeightTable = detail::WeightTable;
    template <class T>
    using Args = detail::Args<T>;
    using BatchComparator = detail::BatchComparator<M, F>;
    using R = std::conditional_t<F, double, size_t>;
    static R distance(std::string_view s1, std::string_view s2) { return BatchComparator(s1).distance(s2); }
    static std::optional<R> distance_with_args(std::string_view s1, std::string_view s2, const Args<R>& a)
    {
        return BatchComparator(s1).distance_with_args(s2, a);
    }
    static R similarity(std::string_view s1, std::string_view s2) { return BatchComparator(s1).similarity(s2); }
    static std::optional<R> similarity_with_args(std::string_view s1, std::string_view s2, const Args<R>& a)
    {
        return BatchComparator(s1).similarity_with_args(s2, a);
    }
    static double normalized_distance(std::string_view s1, std::string_view s2) { return BatchComparator(s1).normalized_distance(s2); }
    static double normalized_similarity(std::string_view s1, std::string_view s2) { return BatchComparator(s1).normalized_similarity(s2); }
};

}  // namespace detail

namespace distance {
using levenshtein = detail::Module<RF_LEVENSHTEIN, false>;    // src/distance/levenshtein.rs
using indel = detail::Module<RF_INDEL, false>;                // src/distance/indel.rs
using lcs_seq = detail::Module<RF_LCS_SEQ, false>;            // src/distance/lcs_seq.rs
using jaro = detail::Module<RF_JARO, true>;                   // src/distance/jaro.rs
using jaro_winkler = detail::Module<RF_JARO_WINKLER, true>;   // src/distance/jaro_winkler.rs
}  // namespace distance

namespace fuzz {
/// src/fuzz.rs:98-150 (similarity only; reproduces fuzz.rs:141 unless RF_FLAG_RATIO_INDEL_NORMALIZATION is set)
class RatioBatchComparator {
public:
    explicit RatioBatchComparator(std::string_view s1) : c_(s1) {}
    std::vector<std::optional<double>> similarity_many(const Corpus& c, const detail::Args<double>& a = {}) const
    {
        return c_.similarity_many(c, a);
    }
    double similarity(std::string_view s2) const { return *c_.similarity_with_args(s2, {}); }

private:
    detail::BatchComparator<RF_FUZZ_RATIO, true> c_;
};
/// fuzz::ratio (src/fuzz.rs:48-85) = normalized Indel similarity
inline double ratio(std::string_view s1, std::string_view s2) { return distance::indel::normalized_similarity(s1, s2); }
}  // namespace fuzz

}  // namespace rapidfuzz
