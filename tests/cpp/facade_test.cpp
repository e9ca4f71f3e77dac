// Compiles the C++ facade against librfgpu.so.  Without a GPU it checks the handle/PM plumbing and that scoring
// fails loudly; with a GPU (argv[1] == "gpu") it runs the reference's doc examples through the kernels.
#include <cstdio>
#include <cstring>

#include "rapidfuzz_amd.hpp"

using namespace rapidfuzz;

#define EXPECT(c)                                                  \
    do {                                                           \
        if (!(c)) {                                                \
            std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
            return 1;                                              \
        }                                                          \
    } while (0)

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    distance::levenshtein::BatchComparator scorer("kitten");
    size_t blocks = 0;
    const uint64_t* pm = rf_comparator_pm(scorer.handle(), &blocks);
    EXPECT(blocks == 1 && pm['t'] == 0b001100 && pm['k'] == 1);
    distance::levenshtein::BatchComparator copy(scorer);  // Clone
    EXPECT(rf_comparator_query_len(copy.handle()) == 6);
    if (!gpu) {
        if (rf_device_count() == 0) {
            try {
                scorer.distance("sitting");
                return 1;  // must not silently compute anywhere else
            } catch (const Error& e) {
                EXPECT(e.status == RF_ERR_NO_DEVICE);
            }
        }
        std::printf("facade ok (cpu)\n");
        return 0;
    }
    // src/lib.rs:32-71 doc examples
    EXPECT(distance::levenshtein::distance("kitten", "sitting") == 3);
    EXPECT(!distance::levenshtein::distance_with_args("kitten", "sitting", distance::levenshtein::Args<size_t>{}.score_cutoff(2)));
    EXPECT(scorer.distance("kitten") == 0);
    Corpus corpus({"sitting", "mitten", "kitchen", ""});
    auto d = scorer.distance_many(corpus, distance::levenshtein::Args<size_t>{}.score_cutoff(2));
    EXPECT(!d[0] && *d[1] == 1 && *d[2] == 2 && !d[3]);
    {   // the reference user's filter_map over Option<T>: only the candidates within the cutoff, as (index, distance) pairs (rf_filter_u32)
        auto f = scorer.distance_filter_many(corpus, distance::levenshtein::Args<size_t>{}.score_cutoff(2));
        EXPECT(f.size() == 2 && f[0].first == 1 && f[0].second == 1 && f[1].first == 2 && f[1].second == 2);
        auto best = scorer.distance_filter_many(corpus, distance::levenshtein::Args<size_t>{}.score_cutoff(6), RF_FILTER_BY_SCORE);
        EXPECT(best.size() == 4 && best[0].first == 1 && best[3].second == 6);
        auto ns = scorer.normalized_similarity_filter_many(corpus, distance::levenshtein::Args<double>{}.score_cutoff(0.7));
        EXPECT(ns.size() == 2 && ns[0].first == 1 && ns[1].first == 2);
    }
    {
        distance::levenshtein::BatchComparator a("mitten"), b("kitchen"), c("");
        auto m = distance::levenshtein::BatchComparator::distance_many_multi({&scorer, &a, &b, &c}, corpus);
        EXPECT(m.size() == 4 && *m[0][1] == 1 && *m[1][1] == 0 && *m[2][2] == 0 && *m[3][0] == 7 && *m[3][3] == 0);
    }
    {   // `char` elements: levenshtein.rs:2163-2169 ("Иванко" / "Петрунко" = 5), then a save / load round trip
        distance::levenshtein::BatchComparator ivanko(std::u32string_view(U"\u0418\u0432\u0430\u043d\u043a\u043e"));
        Corpus wide(std::vector<std::u32string_view>{U"\u041f\u0435\u0442\u0440\u0443\u043d\u043a\u043e", U"\u0418\u0432\u0430\u043d", U"plain"});
        auto d = ivanko.distance_many(wide);
        EXPECT(*d[0] == 5 && *d[1] == 2 && *d[2] == 6);
        wide.save("/tmp/rf_facade_test.rfc");
        Corpus again = Corpus::load("/tmp/rf_facade_test.rfc");
        auto d2 = ivanko.distance_many(again);
        EXPECT(*d2[0] == 5 && *d2[1] == 2 && *d2[2] == 6);
        std::remove("/tmp/rf_facade_test.rfc");
    }
    EXPECT(distance::indel::distance("lewenstein", "levenshtein") == 3);
    EXPECT(distance::lcs_seq::similarity("lewenstein", "levenshtein") == 9);
    EXPECT(std::fabs(distance::jaro::similarity("james", "robert") - 0.455556) < 1e-4);
    EXPECT(std::fabs(fuzz::ratio("this is a test", "this is a test!") - 28.0 / 29.0) < 1e-12);
    std::printf("facade ok (gpu)\n");
    return 0;
}
