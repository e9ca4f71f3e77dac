// rf_api_scan.hip -- plan(): Args x metric x op -> kernel parameters, and the rf_many_* / rf_one_* / rf_many_multi_* entry points (split out of rf_api.hip in round 4; rf_host.hpp has the shared declarations).
// Product code: never includes or links anything from oracle/.
#include "rf_host.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------------
// one-vs-many
// ---------------------------------------------------------------------------------------------------
rf_status plan(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, bool f64_out,
                      ScanParams* p, RawKind* raw)
{
    if (!c || !corpus || !args) {
        set_error("null handle or args");
        return RF_ERR_INVALID_ARG;
    }
    std::memset(p, 0, sizeof(*p));
    p->max_stored_sym = 0xFFFFFFFFu;  // not known (run_many lowers it)
    p->len1 = (uint32_t)c->s1.size();
    p->words = (uint32_t)pm_stride(c);  // row stride of the device table
    p->op = (uint32_t)op;
    p->out_f64 = f64_out ? 1 : 0;
    p->factor = 1;
    p->w_ins = p->w_del = p->w_sub = 1;
    p->prefix_weight = args->prefix_weight;
    for (size_t i = 0; i < std::min<size_t>(4, c->s1.size()); ++i) p->query_head |= (uint32_t)corpus->sigma[c->s1[i]] << (8 * i);
    p->sigma = corpus->d_sigma;
    p->data = corpus->d_data;
    p->tiles = corpus->uniform ? nullptr : corpus->d_tiles;
    p->uniform_len = corpus->uniform_len;
    p->uniform_tile_bytes = (uint32_t)tile_bytes(corpus->uniform_len);
    p->orig = corpus->d_orig;
    p->n_tiles = corpus->n_tiles;
    p->tile_begin = 0;
    p->tile_end = corpus->n_tiles;
    p->n_exact = corpus->n_exact;
    p->mixed = corpus->d_mixed;  // (nullptr when the corpus has no mixed section, and on views without one)
    p->mixed_len = corpus->d_mixed_len;
    p->mixed_orig = corpus->d_mixed_orig;
    p->mixed_begin = 0;
    p->mixed_end = corpus->d_mixed ? corpus->n_mixed : 0;
    p->tile_step = 1;
    p->n = (uint32_t)corpus->n;
    {   // zero-length tiles (the asm stream kernels are not given them): at most one run per ascending section of the tile order
        int runs = 0;
        for (size_t i = 0; i < corpus->lengths.size() && runs < 2; ++i)
            if (corpus->lengths[i] == 0) {
                p->zero_begin[runs] = corpus->length_first_tile[i];
                p->zero_end[runs] = i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles;
                ++runs;
            }
    }

    const bool usize_metric = c->metric == RF_LEVENSHTEIN || c->metric == RF_INDEL || c->metric == RF_LCS_SEQ || c->metric == RF_OSA;
    const bool norm_op = op == RF_OP_NORMALIZED_DISTANCE || op == RF_OP_NORMALIZED_SIMILARITY;
    if ((int)op < 0 || (int)op > (int)RF_OP_NORMALIZED_SIMILARITY) {
        set_error("unknown rf_op");
        return RF_ERR_INVALID_ARG;
    }
    if (usize_metric && (norm_op != f64_out)) {
        set_error("levenshtein/indel/lcs_seq: distance and similarity are u32-valued (rf_many_u32), normalized_* are "
                  "f64-valued (rf_many_f64)");
        return RF_ERR_INVALID_ARG;
    }
    if (!usize_metric && !f64_out) {
        set_error("jaro / jaro_winkler / fuzz ratio are f64-valued: use rf_many_f64");
        return RF_ERR_INVALID_ARG;
    }

    if (f64_out) {
        p->has_cutoff = std::isnan(args->cutoff_f64) ? 0 : 1;
        p->cutoff_f64 = args->cutoff_f64;
    } else {
        p->has_cutoff = args->cutoff_usize != RF_NO_CUTOFF;
        p->cutoff_u32 = (uint32_t)std::min<uint64_t>(args->cutoff_usize, 0xFFFFFFFFull);
    }

    switch (c->metric) {
    case RF_LEVENSHTEIN: {
        // _distance_with_pm weight dispatch, levenshtein.rs:1285-1331
        const uint64_t ins = args->insertion_cost, del = args->deletion_cost, sub = args->substitution_cost;
        if (ins > 0xFFFF || del > 0xFFFF || sub > 0xFFFF) {
            set_error("levenshtein weights above 65535 are not supported on the device");
            return RF_ERR_UNSUPPORTED;
        }
        p->w_ins = (uint32_t)ins;
        p->w_del = (uint32_t)del;
        p->w_sub = (uint32_t)sub;
        if (ins == del && (ins == 0 || ins == sub)) {  // :1303-1316 (ins == del == 0 -> every distance is 0)
            *raw = RAW_LEV;
            p->finish = FIN_LEV;
            p->factor = (uint32_t)ins;
        } else if (ins == del && sub >= ins + del) {  // :1321-1327: Indel distance times the common factor
            *raw = RAW_LCS;
            p->finish = FIN_LEV_INDEL;
            p->factor = (uint32_t)ins;
        } else {
            // every other table: the generalized Wagner-Fischer row DP (levenshtein.rs:212-259) in LDS
            *raw = RAW_WF;
            p->finish = FIN_LEV_GENERAL;
            const uint64_t row_bytes = ((uint64_t)p->len1 + 1) * kWave * sizeof(uint32_t);
            const uint64_t lds_budget = 150u << 10;  // of the 160 KiB a gfx950 workgroup may hold
            const uint64_t worst = ((uint64_t)p->len1 + corpus->max_len) * std::max(std::max(ins, del), sub);
            if (worst >= (1ull << 31) || p->len1 > 150000) {
                set_error("levenshtein with a general weight table: distances would not fit 31 bits (or the query is beyond 150 000 symbols)");
                return RF_ERR_UNSUPPORTED;
            }
            if (row_bytes + p->len1 + 8 > lds_budget) {
                // the row does not fit LDS (queries beyond ~590 symbols): one global scratch strip per wavefront instead
                p->wf_global = 1;
                p->wf_waves = kWavesPerBlock;
                const uint64_t per_block = row_bytes * kWavesPerBlock, budget = 1ull << 30;
                p->long_grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(budget / per_block, (uint64_t)scan_grid(corpus->n_tiles)));
            } else {
                p->wf_waves = (uint32_t)std::min<uint64_t>(kWavesPerBlock, (lds_budget - p->len1 - 8) / row_bytes);
            }
            for (size_t i = 0; i < std::min<size_t>(64, c->s1.size()); ++i)  // the register-resident kernel compares against these
                p->wf_query[i / 4] |= (uint32_t)corpus->sigma[c->s1[i]] << (8 * (i % 4));
        }
        break;
    }
    case RF_OSA:  // osa.rs:431-461; maximum = max(len1, len2) = levenshtein's at unit weights
        *raw = RAW_OSA;
        p->finish = FIN_LEV;  // (beyond 512 symbols: long_kernel, with the transposition bit carried between word groups)
        break;
    case RF_INDEL:
        *raw = RAW_LCS;
        p->finish = FIN_INDEL;
        break;
    case RF_LCS_SEQ:
        *raw = RAW_LCS;
        p->finish = FIN_LCS;
        break;
    case RF_FUZZ_RATIO:
        // RatioBatchComparator::similarity_with_args, fuzz.rs:127-149: normalized similarity of the inner
        // lcs_seq comparator (quirk Q1), or of Indel when the caller asks for the documented ratio.
        if (op != RF_OP_SIMILARITY && op != RF_OP_NORMALIZED_SIMILARITY) {
            set_error("RatioBatchComparator only has similarity (fuzz.rs:115-149)");
            return RF_ERR_INVALID_ARG;
        }
        *raw = RAW_LCS;
        p->finish = (args->flags & RF_FLAG_RATIO_INDEL_NORMALIZATION) ? FIN_INDEL : FIN_LCS;
        p->op = RF_OP_NORMALIZED_SIMILARITY;
        break;
    case RF_JARO:
    case RF_JARO_WINKLER: {
        *raw = RAW_JARO;
        p->finish = c->metric == RF_JARO ? FIN_JARO : FIN_JW;
        // Early-out under a tight cutoff (the reference's own common_char_filter idea, jaro.rs:134-145, applied while the
        // flags are still being collected): `jaro_need` is the similarity a candidate has to reach.
        p->jaro_need = -1.0;
        if (!p->has_cutoff) p->jaro_tab = jaro_device_table(corpus->device);  // (nullptr on failure: the kernels then divide)
        if (p->has_cutoff && args->prefix_weight >= 0.0 && 4.0 * args->prefix_weight <= 1.0) {
            const double need = (op == RF_OP_SIMILARITY || op == RF_OP_NORMALIZED_SIMILARITY) ? args->cutoff_f64 : 1.0 - args->cutoff_f64;
            if (need >= 0.6 && need <= 1.0) p->jaro_need = need;
        }
        // Single-word path (jaro.rs:574-583) when both strings are <= 64 symbols AFTER the window truncation of
        // jaro.rs:550-565, multi-word path (up to 512 symbols each) otherwise.  Tiles ascend by length TWICE -- the exact
        // tiles, then the one-length views of the mixed section -- and the single-word condition holds for a length prefix
        // of each run, so each section splits at one tile index (jaro_split, jaro_split2): a short leftover behind a long
        // exact tile still takes the single-word kernel.
        const uint64_t len1 = c->s1.size();
        p->jaro_split = corpus->n_exact;
        p->jaro_split2 = corpus->n_tiles;
        for (size_t i = 0; i < corpus->lengths.size(); ++i) {
            uint64_t a = len1, b = corpus->lengths[i];
            if (b > a) {
                const uint64_t bound = b / 2 - 1;
                if (b > a + bound) b = a + bound;
            } else if (a >= 2) {
                const uint64_t bound = a / 2 - 1;
                if (a > b + bound) a = b + bound;
            }
            const bool needs_flags = a != 0 && b != 0;  // otherwise decided by the length filter alone
            const bool word_ok = !needs_flags || (a <= 64 && b <= 64);
            if (word_ok) continue;
            // this run of equal-length tiles [first, end) needs the multi-word path.  (A run may straddle the two sections: the last
            // exact length and the first view can be the same length, and the length table merges them.)
            const uint32_t first = corpus->length_first_tile[i];
            const uint32_t end = i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles;
            if (first < corpus->n_exact) p->jaro_split = std::min(p->jaro_split, first);
            if (end > corpus->n_exact) p->jaro_split2 = std::min(p->jaro_split2, std::max(first, corpus->n_exact));
            if (a > 64 * (uint64_t)kMaxWords || b > 64 * (uint64_t)kMaxWords || c->words > (size_t)kMaxWords)
                p->jaro_long = 1;  // beyond 512 symbols: the flag words move from registers to global scratch strips
        }
        if (p->jaro_long) {
            // per wavefront: P words (len1 / 64 + 1) and T words (max candidate length / 64) for 64 lanes
            p->long_chunks_max = (corpus->max_len + 63) / 64 + 1;
            const uint64_t strip_bytes = ((uint64_t)(p->len1 + 63) / 64 + 1 + p->long_chunks_max) * kWave * sizeof(uint64_t);
            const uint64_t budget = 1ull << 30;
            p->long_grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(budget / (strip_bytes * kWavesPerBlock), (uint64_t)scan_grid(corpus->n_tiles)));
        }
        if (p->jaro_need >= 0.0) {
            // Length window, the reference's length_filter (jaro.rs:122-131) hoisted to the host: with m = min(len1, L)
            // the similarity of a candidate of length L is at most (m/len1 + m/L + 1)/3 (+ the largest Winkler boost);
            // lengths that cannot reach `jaro_need` are never read -- pre-filled with None like the usize metrics'.
            const auto& L = corpus->lengths;
            size_t first = L.size(), last = 0;
            const double boost = c->metric == RF_JARO_WINKLER ? 4.0 * args->prefix_weight : 0.0;
            for (size_t i = 0; i < L.size(); ++i) {
                const double l1 = (double)len1, l2 = (double)L[i], m = std::min(l1, l2);
                double ub = (len1 == 0 && L[i] == 0) ? 1.0 : ((len1 == 0 || L[i] == 0) ? 0.0 : (m / l1 + m / l2 + 1.0) / 3.0);
                ub += boost * (1.0 - ub);
                if (ub + 1e-9 >= p->jaro_need) {
                    first = std::min(first, i);
                    last = i;
                }
            }
            if (first == L.size()) {
                p->tile_begin = p->tile_end = corpus->n_tiles;
            } else {
                p->tile_begin = corpus->length_first_tile[first];
                p->tile_end = last + 1 < L.size() ? corpus->length_first_tile[last + 1] : corpus->n_tiles;
            }
            p->prefill_none = !corpus->no_prefill && (p->tile_begin > 0 || p->tile_end < corpus->n_tiles);
        }
        return RF_OK;  // the PM row stride may exceed kMaxWords: only block 0 is read
    }
    }

    // The device finishes in u32 (results are u32): with a common weight factor f every intermediate is bounded by
    // f * (len1 + max_len) -- refuse what would wrap instead of returning it mod 2^32 (the reference computes in usize).
    if ((uint64_t)std::max<uint32_t>(p->factor, 1) * ((uint64_t)p->len1 + corpus->max_len) > 0xFFFFFFFEull) {
        set_error("weights x string lengths exceed the u32 range of the device results");
        return RF_ERR_UNSUPPORTED;
    }
    {
        // finishing coefficients: dist = dS*S + dM*Mx + dR*raw, maximum = mS*S + mM*Mx (rf_device.hpp "Finishing")
        const int32_t f = (int32_t)p->factor;
        switch (p->finish) {
        case FIN_LEV: p->fin_dS = 0, p->fin_dM = 0, p->fin_dR = f, p->fin_mS = 0, p->fin_mM = f; break;
        case FIN_LCS: p->fin_dS = 0, p->fin_dM = 1, p->fin_dR = -1, p->fin_mS = 0, p->fin_mM = 1; break;
        case FIN_INDEL: p->fin_dS = 1, p->fin_dM = 0, p->fin_dR = -2, p->fin_mS = 1, p->fin_mM = 0; break;
        case FIN_LEV_INDEL: p->fin_dS = f, p->fin_dM = 0, p->fin_dR = -2 * f, p->fin_mS = f, p->fin_mM = 0; break;
        case FIN_LEV_GENERAL: p->fin_dS = 0, p->fin_dM = 0, p->fin_dR = 1, p->fin_mS = 0, p->fin_mM = 0; break;  // maximum: wf_kernel
        default: break;
        }
        if (op == RF_OP_DISTANCE || op == RF_OP_NORMALIZED_DISTANCE) {
            p->fin_vS = p->fin_dS, p->fin_vM = p->fin_dM, p->fin_vR = p->fin_dR;
            p->fin_flip = 0;
            p->fin_cflip = (p->has_cutoff && !f64_out) ? p->cutoff_u32 : 0xFFFFFFFFu;
        } else {  // similarity = maximum - distance (details/distance.rs:209-210)
            p->fin_vS = p->fin_mS - p->fin_dS, p->fin_vM = p->fin_mM - p->fin_dM, p->fin_vR = -p->fin_dR;
            p->fin_flip = 0xFFFFFFFFu;
            p->fin_cflip = ~((p->has_cutoff && !f64_out) ? p->cutoff_u32 : 0u);
        }
    }

    if (*raw == RAW_WF) return RF_OK;
    // Long query + small distance cutoff (the reference's hyrroe2003_small_band_with_pm, levenshtein.rs:509-617, taken when
    // len1 > 64 and 2k + 1 <= 64, :1059-1062): one 64-bit word sliding down the diagonal instead of ceil(len1 / 64) words
    // per column.  k is the cutoff on the RAW distance (the common weight factor divided out).
    static const bool no_band = getenv("RF_NO_BAND") != nullptr;  // A/B switch
    if (!no_band && *raw == RAW_LEV && p->finish == FIN_LEV && p->factor >= 1 && op == RF_OP_DISTANCE && !f64_out && p->has_cutoff && c->words >= 2 &&
        c->words <= 64 && p->cutoff_u32 / p->factor <= 31) {
        p->band = 1;
        p->band_k = p->cutoff_u32 / p->factor;
    }
    // a distance cutoff also narrows the band of the multi-word scans (rf_stream_asm.hip): raw distances beyond it only have to come out beyond it
    if (*raw == RAW_LEV && p->finish == FIN_LEV && p->factor >= 1 && op == RF_OP_DISTANCE && !f64_out && p->has_cutoff) p->trim_k1 = p->cutoff_u32 / p->factor + 1;
    if (c->words > (size_t)kMaxWords && !p->band) {
        // beyond 512 symbols: the multi-sweep kernel (8 words per sweep, carries parked in an HBM scratch strip)
        if (c->words > 0x00FFFFFFu) {
            set_error("query too long");
            return RF_ERR_UNSUPPORTED;
        }
        p->long_words_pad = (uint32_t)pm_stride(c);
        p->long_chunks_max = (corpus->max_len + kChunk - 1) / kChunk;
        const uint64_t strip_bytes = std::max<uint64_t>(1, (uint64_t)p->long_chunks_max * kWave * sizeof(uint32_t)) * (*raw == RAW_OSA ? 2 : 1);
        const uint64_t budget = 256ull << 20;
        const uint64_t waves = std::max<uint64_t>(kWavesPerBlock, budget / strip_bytes);
        p->long_grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(waves / kWavesPerBlock, (uint64_t)scan_grid(corpus->n_tiles)));
        return RF_OK;
    }
    // Value-preserving early-out (the reference applies its cutoffs after the loops, e.g. levenshtein.rs:492-496):
    // the kernels stop reading a tile once no lane can pass the cutoff any more (may_pass() in rf_device.hpp), for
    // every op and output type.  It pays only when the cutoff is tight enough to kill typical candidates early --
    // the early-out loop gives up the streaming prefetch -- so it is switched on by how much normalized distance the
    // cutoff still allows (`slack`; measured on the C2 corpus: Levenshtein wins up to ~0.7, the LCS bound, which only
    // gains one per remaining column, up to ~0.4).
    if (p->has_cutoff && (*raw == RAW_LEV || *raw == RAW_OSA || *raw == RAW_LCS) && !(p->finish == FIN_LEV && p->factor == 0)) {
        const uint64_t S = (uint64_t)p->len1 + corpus->max_len, Mx = std::max<uint64_t>(p->len1, corpus->max_len);
        const double maximum = (double)((int64_t)p->fin_mS * (int64_t)S + (int64_t)p->fin_mM * (int64_t)Mx);
        double slack;  // allowed distance / maximum
        if (f64_out)
            slack = (op == RF_OP_NORMALIZED_DISTANCE) ? p->cutoff_f64 : 1.0 - p->cutoff_f64;
        else if (maximum <= 0.0)
            slack = 1.0;
        else
            slack = (op == RF_OP_DISTANCE) ? (double)p->cutoff_u32 / maximum : 1.0 - (double)p->cutoff_u32 / maximum;
        const double tight = *raw == RAW_LCS ? 0.4 : 0.7;
        if (slack < tight) {
            p->early = 1;
            // where a tile's first chunk takes its first look (scan_body): unrelated strings gain almost one edit per column
            {
                static const int forced = [] { const char* e = getenv("RF_FIRST_CHECK"); return e ? atoi(e) : 0; }();  // A/B switch
                const double raw_allowed = slack * maximum / (double)std::max<uint32_t>(1u, (uint32_t)std::abs(p->fin_dR));
                // measured on the C2 corpus (cutoffs 0..12, looks at 4..16): the best look is the first even column >= cutoff + 3
                // (cutoff 3: 195 -> 213 Gpairs/s, cutoff 0: 197 -> 233, cutoff 8: 125 -> 159)
                const double need = raw_allowed + 3.0;
                p->first_check = need <= 4.0 ? 4u : (need >= 15.0 ? 16u : 2u * (uint32_t)((need + 1.999) / 2.0));
                if (*raw == RAW_LCS) {
                    // the LCS bound loses one per column WITHOUT a match, and random strings still match every third column or so:
                    // Indel cutoff 4 is best looked at in column 10-12 (221 -> 233 Gpairs/s), cutoff 12 not before the chunk's end
                    const double misses = (p->finish == FIN_LCS ? slack * maximum : slack * maximum / 2.0), lcs_need = (misses + 3.0) / 0.55;
                    p->first_check = lcs_need >= 15.0 ? 16u : std::max(4u, 2u * (uint32_t)((lcs_need + 1.999) / 2.0));
                }
                if (forced >= 4 && forced <= 16 && forced % 2 == 0) p->first_check = (uint32_t)forced;
            }
            // Length window: before any byte is read a candidate of length L already has a favourable bound -- distance >=
            // |len1 - L| (the reference's first test, levenshtein.rs:1389-1391), LCS <= min(len1, L).  Lengths whose bound
            // fails the cutoff (same arithmetic as may_pass() on the device) are never read: tiles ascend by length, so
            // the survivors lie in ONE tile range [first passing length, last passing length]; the rest of `out` is
            // pre-filled with None.
            const auto& L = corpus->lengths;
            size_t first = L.size(), last = 0;
            uint32_t pass_lo = 0xFFFFFFFFu, pass_hi = 0;  // the shortest and the longest candidate length that can pass
            for (size_t i = 0; i < L.size(); ++i) {
                const uint32_t len2 = L[i];
                const uint32_t Sv = p->len1 + len2, Mv = std::max(p->len1, len2);
                const uint32_t raw_b = *raw == RAW_LCS ? std::min(p->len1, len2) : (p->len1 > len2 ? p->len1 - len2 : len2 - p->len1);
                bool pass;
                if (!f64_out) {
                    const uint32_t v = (uint32_t)p->fin_vS * Sv + (uint32_t)p->fin_vM * Mv + (uint32_t)p->fin_vR * raw_b;
                    pass = (v ^ p->fin_flip) <= p->fin_cflip;
                } else {
                    const uint32_t dist = (uint32_t)p->fin_dS * Sv + (uint32_t)p->fin_dM * Mv + (uint32_t)p->fin_dR * raw_b;
                    const uint32_t mx = (uint32_t)p->fin_mS * Sv + (uint32_t)p->fin_mM * Mv;
                    const double nd = mx == 0 ? 0.0 : (double)dist / (double)mx;
                    pass = op == RF_OP_NORMALIZED_DISTANCE ? nd <= p->cutoff_f64 : (1.0 - nd) >= p->cutoff_f64;
                }
                if (pass) {
                    first = std::min(first, i);
                    last = i;
                    pass_lo = std::min(pass_lo, len2);
                    pass_hi = std::max(pass_hi, len2);
                }
            }
            // (the length table follows the tile order -- exact tiles ascending, then the one-length views of the mixed
            // section ascending -- so [first, last] may enclose lengths that cannot pass: those tiles are merely read)
            if (first == L.size()) {
                p->tile_begin = p->tile_end = corpus->n_tiles;  // nothing can pass
                p->mixed_begin = p->mixed_end = 0;
            } else {
                p->tile_begin = corpus->length_first_tile[first];
                p->tile_end = last + 1 < L.size() ? corpus->length_first_tile[last + 1] : corpus->n_tiles;
                // mixed tiles ascend by length as well: the ones whose length span meets [pass_lo, pass_hi]
                uint32_t mb = 0, me = p->mixed_end;
                while (mb < me && corpus->mixed[mb].max_len < pass_lo) ++mb;
                while (me > mb && corpus->mixed[me - 1].min_len > pass_hi) --me;
                p->mixed_begin = mb;
                p->mixed_end = me;
            }
            p->prefill_none = !corpus->no_prefill && (p->tile_begin > 0 || p->tile_end < corpus->n_tiles ||
                                                      (corpus->d_mixed && (p->mixed_begin > 0 || p->mixed_end < corpus->n_mixed)));
        }
    }
    return RF_OK;
}

// bytes of per-launch scratch the planned kernels need (0 = none): carry strips of long_kernel, the global DP rows of
// wf_kernel, the flag strips of jaro_long_kernel -- all handed to the kernels through ScanParams::long_scratch
static size_t launch_scratch_bytes(const ScanParams& p, RawKind raw)
{
    const size_t waves = (size_t)p.long_grid * kWavesPerBlock;
    if (p.jaro_long) return waves * (((size_t)p.len1 + 63) / 64 + 1 + p.long_chunks_max) * kWave * sizeof(uint64_t);
    if (p.wf_global) return waves * ((size_t)p.len1 + 1) * kWave * sizeof(uint32_t);
    if (p.long_words_pad) return waves * std::max<uint32_t>(1, p.long_chunks_max) * kWave * sizeof(uint32_t) * (raw == RAW_OSA ? 2 : 1);
    return 0;
}

// RF_PACK_TIMING=1: what the structures a corpus builds on first use cost (stderr, one line each) -- bench.py's accel_build_ms, itemised
struct AccelTimer {
    const char* what;
    bool on = getenv("RF_PACK_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit AccelTimer(const char* w) : what(w) {}
    ~AccelTimer()
    {
        if (on) std::fprintf(stderr, "[rf accel] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// The largest stored symbol of the payload (symbols are stored as their frequency rank: a 62-symbol corpus holds 0 .. 61), computed
// exactly on first use -- one streaming pass -- and kept.  0xFFFFFFFF when it cannot be had.
static uint32_t corpus_max_stored_symbol(const rf_corpus* corpus, hipStream_t st)
{
    if (corpus->borrowed || corpus->wide || !corpus->d_data || corpus->data_bytes < 16) return 0xFFFFFFFFu;
    std::lock_guard<std::mutex> lock(corpus->scratch_mu);
    if (corpus->max_stored_sym == 0xFFFFFFFFu) {
        AccelTimer timer("largest stored symbol");
        uint32_t* d = nullptr;
        uint32_t v = 0;
        if (hipMalloc((void**)&d, sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            return 0xFFFFFFFFu;
        }
        hipError_t e = hipMemsetAsync(d, 0, sizeof(uint32_t), st);
        if (e == hipSuccess) e = launch_max_byte(corpus->d_data, corpus->data_bytes, d, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&v, d, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return 0xFFFFFFFFu;
        }
        corpus->max_stored_sym = v;
    }
    return corpus->max_stored_sym;
}

// Small-cutoff Levenshtein scans of large single-length corpora take their first look from the head plane (rf_pack.hip
// head8_plane_kernel).  Built once per corpus, on the first such scan; RF_HEAD8_MIN=<tiles> moves the threshold (0 = never).
// Failing to allocate it is not an error: the scan then reads the tiles' first chunk rows as before.
const uint8_t* corpus_head8_plane(const rf_corpus* corpus, const ScanParams& p, RawKind raw, hipStream_t st)
{
    static const size_t min_tiles = [] { const char* e = getenv("RF_HEAD8_MIN"); return e ? (size_t)atoll(e) : (size_t)1 << 14; }();
    if (!min_tiles || !p.early || (raw != RAW_LEV && raw != RAW_OSA) || p.words != 1 || p.first_check > 8 || corpus->borrowed)
        return nullptr;
    // single-length corpora: every tile; length-bucketed corpora (round 4): the exact tiles, whose length runs the cutoff scans
    // then walk as single-length corpora of their own (launch_scan_runs)
    const uint32_t plane_tiles = corpus->uniform ? corpus->n_tiles : corpus->n_exact;
    if (plane_tiles < min_tiles || (corpus->uniform ? corpus->uniform_len < (uint32_t)kChunk : (!corpus->d_tiles || !corpus->d_orig || corpus->max_len < (uint32_t)kChunk)))
        return nullptr;
    std::lock_guard<std::mutex> lock(corpus->scratch_mu);
    if (!corpus->d_heads8) {
        AccelTimer timer("head plane (8 symbols)");
        uint8_t* h = nullptr;
        if (hipMalloc((void**)&h, ((size_t)plane_tiles + 1) * kWave * 8) != hipSuccess) {  // (+ one row: head_filter_kernel reads tiles in pairs)
            (void)hipGetLastError();
            return nullptr;
        }
        hipError_t e = corpus->uniform ? launch_head8_plane(corpus->d_data, corpus->n_tiles, (uint32_t)tile_bytes(corpus->uniform_len), h, st)
                                       : launch_head8_plane_tiles(corpus->d_data, corpus->d_tiles, plane_tiles, h, st);
        if (e == hipSuccess) e = hipMemsetAsync(h + (size_t)plane_tiles * kWave * 8, 0, kWave * 8, st);  // the pad row: defined bytes
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // (other streams may use the plane as soon as the lock is released)
        if (e != hipSuccess) {
            (void)hipFree(h);
            return nullptr;
        }
        corpus->d_heads8 = h;
    }
    return corpus->d_heads8;
}

// The head plane at 6 bits per symbol (ScanParams::heads6): single-length corpora whose payload holds no stored symbol above 63 -- the
// band prefilter pass then streams 6 instead of 8 bytes per candidate.  Built from the 8-byte plane on first use, kept beside it
// (+ 6 bytes per candidate); RF_HEAD6=0 switches it off.  nullptr = not to be had.
const uint32_t* corpus_head6_plane(const rf_corpus* corpus, hipStream_t st)
{
    static const bool on = [] { const char* e = getenv("RF_HEAD6"); return !e || atoi(e) != 0; }();
    if (!on || !corpus->uniform || !corpus->d_heads8 || corpus->borrowed) return nullptr;
    if (corpus_max_stored_symbol(corpus, st) >= 64u) return nullptr;
    std::lock_guard<std::mutex> lock(corpus->scratch_mu);
    if (!corpus->d_heads6 && !corpus->heads6_tried) {
        corpus->heads6_tried = true;
        AccelTimer timer("head plane at 6 bits");
        uint32_t* h = nullptr;
        const size_t rows = ((size_t)corpus->n_tiles + 1) / 2 + 1;  // (+ one pair: the pass reads a pair ahead)
        if (hipMalloc((void**)&h, rows * 3 * kWave * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        hipError_t e = launch_head6_plane(corpus->d_heads8, corpus->n_tiles, h, st);
        if (e == hipSuccess) e = hipMemsetAsync(h + (rows - 1) * 3 * kWave, 0, 3 * kWave * sizeof(uint32_t), st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // (other streams may use the plane as soon as the lock is released)
        if (e != hipSuccess) {
            (void)hipFree(h);
            (void)hipGetLastError();
            return nullptr;
        }
        corpus->d_heads6 = h;
    }
    return corpus->d_heads6;
}

// The payload at 6 bits per symbol (ScanParams::data6, rf_pack.hip pack6_kernel): single-length corpora whose payload holds no stored symbol above 63
// (exact: corpus_max_stored_symbol) and at least RF_PACK6_MIN_TILES tiles (default 16384: a scan that does not fill the chip is not HBM-bound).  Built
// from the 8-bit payload on first use by a scan that streams it, kept beside it (+ 75 % of the payload in HBM); RF_PACK6=0 switches it off.
const uint32_t* corpus_data6(const rf_corpus* corpus, hipStream_t st)
{
    static const bool on = [] { const char* e = getenv("RF_PACK6"); return !e || atoi(e) != 0; }();
    static const uint32_t min_tiles = [] { const char* e = getenv("RF_PACK6_MIN_TILES"); return e ? (uint32_t)atoll(e) : 16384u; }();
    if (!on || corpus->borrowed || corpus->n_tiles < min_tiles) return nullptr;
    if (corpus->uniform ? corpus->uniform_len == 0 : (!corpus->d_tiles || corpus->data_bytes % (kWave * kChunk) != 0)) return nullptr;
    // (whole chunks: any 64 codes; the partial last chunk of a single-length corpus is filled up with the code 63, which must then be free; bucketed corpora: the
    // scans shift a partial chunk into place like the 8-bit kernels do, any 64 codes)
    if (corpus_max_stored_symbol(corpus, st) >= ((!corpus->uniform || corpus->uniform_len % kChunk == 0) ? 64u : 63u)) return nullptr;
    std::lock_guard<std::mutex> lock(corpus->scratch_mu);
    if (!corpus->d_data6 && !corpus->data6_tried) {
        corpus->data6_tried = true;
        AccelTimer timer("6-bit payload");
        // chunk rows of 64 lanes: tiles x chunks of a single-length corpus; a bucketed one's whole payload, row by row (the 6-bit image mirrors it at 3/4 of every offset)
        const uint32_t nch = corpus->uniform ? (corpus->uniform_len + kChunk - 1) / kChunk : 1;
        const uint64_t rows = corpus->uniform ? (uint64_t)corpus->n_tiles * nch : corpus->data_bytes / (kWave * kChunk);
        if (rows * kWave >= 0xFFFFFFFFull) return nullptr;
        uint32_t* d = nullptr;
        if (hipMalloc((void**)&d, (rows + 1) * kWave * 12) != hipSuccess) {  // (+ one chunk row: the scans prefetch past the last tile)
            (void)hipGetLastError();
            return nullptr;
        }
        hipError_t e = corpus->uniform ? launch_pack6(corpus->d_data, corpus->n_tiles, corpus->uniform_len, d, st)
                                       : launch_pack6(corpus->d_data, (uint32_t)rows, (uint32_t)kChunk, d, st);  // (rows of one whole chunk each: nothing to fill)
        if (e == hipSuccess) e = hipMemsetAsync(d + rows * kWave * 3, 0, kWave * 12, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // (other streams may use it as soon as the lock is released)
        if (e != hipSuccess) {
            (void)hipFree(d);
            (void)hipGetLastError();
            return nullptr;
        }
        corpus->d_data6 = d;
    }
    return corpus->d_data6;
}

// The BAND PREFILTER of the head-plane cutoff scans (rf_scan.hip early_lean_body has the kernel side and the proof): with at most
// K edits allowed, at least 8 - K of a candidate's first 8 symbols must equal a query symbol within K positions of their own.
// Decides whether a launch uses it: K = the largest raw distance that passes the cutoff (the same arithmetic as may_pass() on
// the device, over every raw value a 64-symbol pair can have) must be <= 3, and by the corpus' symbol frequencies a tile of 64
// random candidates must be unlikely to have a lane that passes the filter -- otherwise (small alphabets, repetitive queries) the
// filter is 30 instructions per tile spent for nothing.  RF_BAND_FILTER=0 / 1 forces it off / on wherever K <= 3.
void plan_band_filter(const rf_comparator* c, const rf_corpus* corpus, rf_op op, bool f64_out, ScanParams* p, uint32_t len2)
{
    p->head_need = 0;
    if (!p->heads8 || !p->early || p->words != 1 || len2 < 8) return;
    static const int forced = [] { const char* e = getenv("RF_BAND_FILTER"); return e ? atoi(e) : -1; }();
    if (forced == 0) return;
    const uint32_t len1 = p->len1;
    const uint32_t Sv = len1 + len2, Mv = std::max(len1, len2);
    int K = -1;
    for (uint32_t raw = 0; raw <= Mv; ++raw) {
        bool pass;
        if (!f64_out) {
            const uint32_t v = (uint32_t)p->fin_vS * Sv + (uint32_t)p->fin_vM * Mv + (uint32_t)p->fin_vR * raw;
            pass = (v ^ p->fin_flip) <= p->fin_cflip;
        } else {
            const uint32_t dist = (uint32_t)p->fin_dS * Sv + (uint32_t)p->fin_dM * Mv + (uint32_t)p->fin_dR * raw;
            const uint32_t mx = (uint32_t)p->fin_mS * Sv + (uint32_t)p->fin_mM * Mv;
            const double nd = mx == 0 ? 0.0 : (double)dist / (double)mx;
            pass = op == RF_OP_NORMALIZED_DISTANCE ? nd <= p->cutoff_f64 : (1.0 - nd) >= p->cutoff_f64;
        }
        if (pass) K = (int)raw;
    }
    if (K < 0 || K > 3) return;
    const uint32_t need = 8u - (uint32_t)K;
    if (forced != 1) {
        // P(symbol i of a random candidate has a partner in the band) from the symbol frequencies, then the distribution of the
        // number of such symbols among 8 (independent positions), then a tile of 64 lanes
        bool known = false;
        for (int ch = 0; ch < 256; ++ch) known = known || corpus->sym_freq[ch] > 0.0f;
        if (!known) return;
        double dist[9] = {1.0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8; ++i) {
            double pi = 0.0;
            for (int j = std::max(0, i - K); j <= i + K && j < (int)c->s1.size(); ++j) {
                bool seen = false;  // (a symbol that occurs twice in the band counts once)
                for (int j2 = std::max(0, i - K); j2 < j; ++j2) seen = seen || c->s1[j2] == c->s1[j];
                if (!seen) pi += corpus->sym_freq[c->s1[j]];
            }
            pi = std::min(1.0, pi);
            for (int m = i + 1; m >= 1; --m) dist[m] = dist[m] * (1.0 - pi) + dist[m - 1] * pi;
            dist[0] *= 1.0 - pi;
        }
        double lane = 0.0;
        for (uint32_t m = need; m <= 8; ++m) lane += dist[m];
        const double tile = 1.0 - std::pow(1.0 - lane, 64.0);
        if (tile > 0.35) return;
    }
    p->head_need = need;
    p->head_k = (uint32_t)K;
}

// words of a tile-list buffer: round 5's layout (packed count, <= 16 K per-wavefront counts and offsets, their segments -- n_tiles + 2 per wavefront of rounding --, the
// packed list) needs 2 n_tiles + 5 x 16384; the lane lists (rf_scan.hip lane_list_pack_kernel: 16-byte entries in the segments and in the packed list) four times the entries
static size_t tile_list_words(uint32_t n_tiles) { return 9 * (size_t)n_tiles + 12 * 16384 + 64; }  // (the last 4 words: launch_band's count of hand-over candidates, zero between launches)  // (+ one word per dense tile: lane_list_pack_kernel's first[])
// the buffer serves the lane compaction (every buffer corpus_tile_list hands out does)
void corpus_lane_buffers(const rf_corpus* corpus, ScanParams* p)
{
    (void)corpus;
    p->lane_list = p->tile_list_buf ? 1u : 0u;
}

// this stream's tile list for head_filter_kernel (the caller holds corpus->filter_enqueue_mu); nullptr = none to be had, the
// scan then filters inside the cutoff kernel
uint32_t* corpus_tile_list(const rf_corpus* corpus, hipStream_t st)
{
    for (size_t i = 0; i < corpus->tile_lists.size(); ++i)
        if (corpus->tile_lists[i].stream == st) {  // most recently used first
            const rf_corpus::TileList hit = corpus->tile_lists[i];
            corpus->tile_lists.erase(corpus->tile_lists.begin() + (long)i);
            corpus->tile_lists.insert(corpus->tile_lists.begin(), hit);
            return hit.ptr;
        }
    if (corpus->tile_lists.size() >= 4) {
        // a fifth stream: the least recently used list changes hands instead of the scan silently falling back to the slower
        // in-kernel filter.  Its old stream's work is waited for first -- rare, and only then.
        rf_corpus::TileList lru = corpus->tile_lists.back();
        corpus->tile_lists.pop_back();
        (void)hipEventSynchronize(lru.done);  // (the event, not the stream handle: that stream may no longer exist)
        lru.stream = st;
        corpus->tile_lists.insert(corpus->tile_lists.begin(), lru);
        return lru.ptr;
    }
    uint32_t* ptr = nullptr;
    if (hipMalloc((void**)&ptr, tile_list_words(corpus->n_tiles) * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemsetAsync(ptr + tile_list_words(corpus->n_tiles) - 4, 0, 4 * sizeof(uint32_t), st) != hipSuccess) (void)hipGetLastError();
    hipEvent_t done = nullptr;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ptr);
        return nullptr;
    }
    corpus->tile_lists.insert(corpus->tile_lists.begin(), {st, ptr, done});
    return ptr;
}
// the scan that walks this stream's list has been enqueued (the caller still holds corpus->filter_enqueue_mu)
void corpus_tile_list_done(const rf_corpus* corpus, hipStream_t st)
{
    for (const rf_corpus::TileList& t : corpus->tile_lists)
        if (t.stream == st) {
            (void)hipEventRecord(t.done, st);
            return;
        }
}

// RF_TILE_ORDER (run_many has what it selects): 0 = never by origin, 1 = by origin without the XCD deal, 2 = default, 3 = also the
// kernels that lose by it
static int tile_order_knob()
{
    static const int v = [] { const char* e = getenv("RF_TILE_ORDER"); return e ? atoi(e) : 2; }();
    return v;
}

// LENGTH-BUCKETED corpora under a small cutoff (round 4; VERDICT r3 missing #1).  The head plane, the band prefilter, the streaming
// first look and the lean cutoff kernel were written for single-length corpora (tile t at t * tile_bytes, slot = index).  The exact
// tiles of ONE length of a bucketed corpus are exactly that -- back to back in the payload, 64 slots per tile -- except that a
// slot's result belongs at out[orig[slot]].  So the tiles of every length inside the cutoff's length window are walked as a
// single-length corpus of their own (ScanParams::run_orig): `out` is pre-filled with None ONCE, dead tiles store nothing (on a
// single-length corpus they cost the filter pass one 8-byte store per lane; here they would be scattered), and the rare surviving
// lane writes through orig[].  Runs too short to pay for three launches (and tiles shorter than a chunk), the one-length views and the
// mixed section keep the general cutoff kernels.  Same values either way: tests/test_gpu_parity.py forces both.
static bool scan_runs_applies(const rf_corpus* corpus, const ScanParams& p, RawKind raw)
{
    return !corpus->uniform && p.heads8 && p.early && p.words == 1 && (raw == RAW_LEV || raw == RAW_OSA) && p.first_check <= 8 && p.tile_step == 1 &&
           p.tiles == corpus->d_tiles && corpus->d_orig && !p.band && !p.long_words_pad;
}
static hipError_t launch_scan_runs(RawKind raw, const ScanParams& p, const rf_comparator* c, const rf_corpus* corpus, rf_op op, bool f64_out, hipStream_t st)
{
    static const uint32_t min_run = [] { const char* e = getenv("RF_RUN_MIN_TILES"); return e ? (uint32_t)atoi(e) : 256u; }();
    hipError_t e = hipSuccess;
    if (p.out && !p.topk_k) {
        const size_t w = p.out_f64 ? 2 : 1, from = p.prefill_window ? (size_t)p.tile_begin * kWave : 0, count = p.prefill_window ? (size_t)(p.tile_end - p.tile_begin) * kWave : (size_t)p.n;
        if (count) e = hipMemsetD32Async((hipDeviceptr_t)(reinterpret_cast<uint32_t*>(p.out) + from * w), (int)RF_NONE_U32, count * w, st);
    }
    const uint32_t ex_begin = std::min(p.tile_begin, corpus->n_exact), ex_end = std::min(p.tile_end, corpus->n_exact);
    uint64_t off = 0;  // payload offset of the current length's first tile (exact tiles lie back to back in length order)
    uint32_t pend_a = 0, pend_b = 0;  // general launches are merged over neighbouring short runs
    auto flush_general = [&]() {
        if (e == hipSuccess && pend_b > pend_a) {
            ScanParams q = p;
            q.tile_begin = pend_a, q.tile_end = pend_b;
            q.mixed = nullptr, q.mixed_begin = q.mixed_end = 0;
            q.prefill_none = 0;
            q.heads8 = nullptr;
            e = launch_scan(raw, q, st, nullptr);
        }
        pend_a = pend_b = 0;
    };
    for (size_t i = 0; i < corpus->lengths.size() && e == hipSuccess; ++i) {
        const uint32_t first = corpus->length_first_tile[i];
        if (first >= corpus->n_exact) break;
        const uint32_t end = std::min(i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles, corpus->n_exact);
        const uint32_t L = corpus->lengths[i];
        const uint32_t a = std::max(first, ex_begin), b = std::min(end, ex_end);
        if (b > a) {
            if (L >= (uint32_t)kChunk && b - a >= min_run && tile_bytes(L) <= 0xFFFFFFFFull) {
                flush_general();
                ScanParams q = p;
                q.tiles = nullptr, q.orig = nullptr;
                q.mixed = nullptr, q.mixed_begin = q.mixed_end = 0;
                q.data = p.data + off + (uint64_t)(a - first) * tile_bytes(L);
                q.heads8 = p.heads8 + (size_t)a * kWave * 8;
                q.uniform_len = L;
                q.uniform_tile_bytes = (uint32_t)tile_bytes(L);
                q.n_tiles = q.n_exact = b - a;
                q.tile_begin = 0, q.tile_end = b - a;
                q.n = (b - a) * (uint32_t)kWave;
                q.run_orig = (p.slot_store ? p.orig : corpus->d_orig) + (size_t)a * kWave;  // (RF_FLAG_SLOT_ORDER: the slot -> slot map, results stay in slot order)
                q.prefill_none = 0;
                q.zero_begin[0] = q.zero_end[0] = q.zero_begin[1] = q.zero_end[1] = 0;
                plan_band_filter(c, corpus, op, f64_out, &q, L);
                e = launch_scan(raw, q, st, nullptr);
            } else {
                if (pend_b != a) flush_general();
                if (pend_b == pend_a) pend_a = a;
                pend_b = b;
            }
        }
        off += (uint64_t)(end - first) * tile_bytes(L);
    }
    flush_general();
    if (e != hipSuccess) return e;
    // what is left of the launch: the one-length views (when the launch walks them) or the mixed section
    ScanParams q = p;
    q.prefill_none = 0;
    q.heads8 = nullptr;
    q.tile_begin = std::max(p.tile_begin, corpus->n_exact);
    q.tile_end = std::max(p.tile_end, q.tile_begin);
    const bool has_mixed = p.mixed && p.mixed_end > p.mixed_begin;
    if (q.tile_end > q.tile_begin || has_mixed) {
        if (has_mixed) q.tile_begin = q.tile_end = corpus->n_exact;  // (launch_scan then runs scan_kernel_mixed over the mixed range alone)
        e = launch_scan(raw, q, st, nullptr);
    }
    return e;
}

// The small-band scan of a LENGTH-BUCKETED corpus (round 6): band_kernel<false> walks tiles, and a tile that holds one candidate near the query runs all its columns on
// 64 lanes.  The hand-over of sparse tiles to a dense second pass (rf_band.hip launch_band) is written for single-length corpora -- tile t at t * tile_bytes -- and the
// exact tiles of ONE length of a bucketed corpus are exactly that, except that a slot's result belongs at out[orig[slot]] (ScanParams::run_orig, as in launch_scan_runs
// above).  So every long run of exact tiles inside the launch's window goes through launch_band as a single-length corpus of its own (its own list, pack and second pass,
// one after the other on the stream's list buffer); short runs, lengths too short to hand anything over and the one-length views keep the tiles kernel.
// (a run's launch sequence -- first pass, list pack, second pass with a dense tile walking all its columns alone -- is ~70 us whatever its size: 13 runs of 770 k
// candidates with 1 % near the query took 0.96 ms where one launch over the tiles takes 0.90; from ~2 M candidates per run on the hand-over wins)
static bool band_run_qualifies(const ScanParams& p, uint32_t L, uint32_t tiles)
{
    static const uint32_t min_run = [] { const char* e = getenv("RF_BAND_RUN_MIN_TILES"); return e ? (uint32_t)atoi(e) : 32768u; }();
    const uint32_t gap = p.len1 > L ? p.len1 - L : L - p.len1;
    return gap <= p.band_k && L >= 80u && tiles >= min_run && tile_bytes(L) <= 0xFFFFFFFFull;  // (80: nothing is handed over below defer_at + 64 columns)
}
// does the launch's window hold a run of exact tiles that launch_band_runs would walk on its own?  (No: one launch over the tiles, as ever.)
static bool band_has_long_run(const ScanParams& p, const rf_corpus* corpus)
{
    const uint32_t xb = std::min(p.tile_begin, corpus->n_exact), xe = std::min(p.tile_end, corpus->n_exact);
    for (size_t i = 0; i < corpus->lengths.size(); ++i) {
        const uint32_t first = corpus->length_first_tile[i];
        if (first >= corpus->n_exact) break;
        const uint32_t end = std::min(i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles, corpus->n_exact);
        const uint32_t a = std::max(first, xb), b = std::min(end, xe);
        if (b > a && band_run_qualifies(p, corpus->lengths[i], b - a)) return true;
    }
    return false;
}
static hipError_t launch_band_runs(RawKind raw, const ScanParams& p, const rf_corpus* corpus, hipStream_t st, bool longest_only)
{
    hipError_t e = hipSuccess;
    auto qualifies = [&](uint32_t L, uint32_t tiles) { return band_run_qualifies(p, L, tiles); };
    size_t longest = (size_t)-1;
    if (longest_only) {
        uint32_t most = 0;
        const uint32_t xb = std::min(p.tile_begin, corpus->n_exact), xe = std::min(p.tile_end, corpus->n_exact);
        for (size_t i = 0; i < corpus->lengths.size(); ++i) {
            const uint32_t first = corpus->length_first_tile[i];
            if (first >= corpus->n_exact) break;
            const uint32_t end = std::min(i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles, corpus->n_exact);
            const uint32_t a = std::max(first, xb), b = std::min(end, xe);
            if (b > a && qualifies(corpus->lengths[i], b - a) && b - a > most) most = b - a, longest = i;
        }
    }
    if (p.prefill_none && p.out) {  // (launch_scan's own pre-fill of the candidates outside the cutoff's length window, once for all the launches below)
        const size_t w = p.out_f64 ? 2 : 1, from = p.prefill_window ? (size_t)p.tile_begin * kWave : 0, count = p.prefill_window ? (size_t)(p.tile_end - p.tile_begin) * kWave : (size_t)p.n;
        if (count) e = hipMemsetD32Async((hipDeviceptr_t)(reinterpret_cast<uint32_t*>(p.out) + from * w), (int)RF_NONE_U32, count * w, st);
    }
    auto plain = [&](uint32_t a, uint32_t b) {  // the tiles kernel over [a, b): no lists
        if (e != hipSuccess || b <= a) return;
        ScanParams q = p;
        q.tile_begin = a, q.tile_end = b;
        q.prefill_none = 0;
        q.lane_list = 0, q.tile_list_buf = nullptr, q.band_defer_seen = nullptr, q.band_report = nullptr;
        e = launch_scan(raw, q, st, nullptr);
    };
    const uint32_t ex_begin = std::min(p.tile_begin, corpus->n_exact), ex_end = std::min(p.tile_end, corpus->n_exact);
    uint64_t off = 0;  // payload offset of the current length's first tile (exact tiles lie back to back in length order)
    uint32_t pend_a = 0, pend_b = 0;
    for (size_t i = 0; i < corpus->lengths.size() && e == hipSuccess; ++i) {
        const uint32_t first = corpus->length_first_tile[i];
        if (first >= corpus->n_exact) break;
        const uint32_t end = std::min(i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles, corpus->n_exact);
        const uint32_t L = corpus->lengths[i];
        const uint32_t a = std::max(first, ex_begin), b = std::min(end, ex_end);
        if (b > a) {
            if (qualifies(L, b - a) && (!longest_only || i == longest)) {
                plain(pend_a, pend_b);
                pend_a = pend_b = 0;
                ScanParams q = p;
                q.tiles = nullptr, q.orig = nullptr;
                q.mixed = nullptr, q.mixed_begin = q.mixed_end = 0;
                q.data = p.data + off + (uint64_t)(a - first) * tile_bytes(L);
                q.uniform_len = L;
                q.uniform_tile_bytes = (uint32_t)tile_bytes(L);
                q.n_tiles = q.n_exact = b - a;
                q.tile_begin = 0, q.tile_end = b - a;
                q.n = (b - a) * (uint32_t)kWave;
                q.run_orig = (p.slot_store ? p.orig : corpus->d_orig) + (size_t)a * kWave;
                q.prefill_none = 0;
                e = launch_scan(raw, q, st, nullptr);
            } else {
                if (pend_b != a) {
                    plain(pend_a, pend_b);
                    pend_a = a;
                }
                pend_b = b;
            }
        }
        off += (uint64_t)(end - first) * tile_bytes(L);
    }
    plain(pend_a, pend_b);
    plain(std::max(p.tile_begin, corpus->n_exact), std::max(p.tile_end, std::max(p.tile_begin, corpus->n_exact)));  // the one-length views of the mixed section
    return e;
}

rf_status run_many(const rf_comparator* c_in, const rf_corpus* corpus_in, rf_op op, const rf_args* args, void* out, rf_mem out_mem, void* stream, bool f64_out);
// score_hint on a per-candidate scan of a long query (reference: levenshtein.rs:1069-1088, the band of max(hint, 31) doubled until the
// distance fits -- results never depend on the hint, :2153-2160).  rf_hint.hip has the scheme: pass 1 over everything under the cutoff k1 = max(hint, 31)
// (the band kernel, or the banded multi-word scans), then the candidates it left unresolved are gathered into dense tiles and scanned under the caller's own
// cutoff.  Applies to Levenshtein distance with a common weight factor (the path that reads the hint in the reference, levenshtein.rs:1307-1316), queries of
// more than 64 symbols, k1 below the caller's cutoff and below the longest string, corpora of >= RF_HINT_MIN_TILES tiles (default 1024: the pass costs a
// stream synchronization and four launches); everything else ignores the hint, as before.  Returns false when it does not apply.
static bool hint_pass_applies(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, bool f64_out, uint32_t* k1, uint32_t* factor)
{
    static const uint32_t min_tiles = [] { const char* e = getenv("RF_HINT_MIN_TILES"); return e ? (uint32_t)atoll(e) : 1024u; }();  // 0xFFFFFFFF: hints are ignored (A/B)
    if (f64_out || op != RF_OP_DISTANCE || c->metric != RF_LEVENSHTEIN || args->score_hint_usize == RF_NO_CUTOFF || c->words < 2) return false;
    const uint64_t f = args->insertion_cost;
    if (f < 1 || f > 0xFFFF || args->deletion_cost != f || args->substitution_cost != f) return false;
    if (corpus->n_tiles < min_tiles || corpus->borrowed || corpus->n >= 0xFFFFFFFFull / 2) return false;
    const uint64_t longest = std::max<uint64_t>(c->s1.size(), corpus->max_len);
    const uint64_t raw_hint = args->score_hint_usize / f + (args->score_hint_usize % f != 0);  // ceil_div, levenshtein.rs:1308-1311
    const uint64_t k = std::max<uint64_t>(raw_hint, 31);                                      // :1069
    const uint64_t raw_cut = args->cutoff_usize == RF_NO_CUTOFF ? UINT64_MAX : args->cutoff_usize / f;
    if (k >= raw_cut || k >= longest) return false;
    *k1 = (uint32_t)k;
    *factor = (uint32_t)f;
    return true;
}

static rf_status run_many_hinted(const rf_comparator* c_in, const rf_corpus* corpus_in, const rf_comparator* c, const rf_corpus* corpus, const rf_args* args,
                                 uint32_t* out, rf_mem out_mem, hipStream_t st, uint32_t k1, uint32_t factor)
{
    const rf_op op = RF_OP_DISTANCE;
    const size_t out_bytes = corpus->n * sizeof(uint32_t);
    // everything this call allocates, released in stream order on every way out
    struct Scratch {
        hipStream_t st;
        std::vector<void*> blocks;
        ~Scratch()
        {
            for (void* b : blocks) scratch_free(b, st);
        }
        hipError_t get(void** p, size_t bytes)
        {
            const hipError_t e = scratch_alloc(p, bytes, st);
            if (e == hipSuccess) blocks.push_back(*p);
            return e;
        }
    } sc{st, {}};
    uint32_t* d_out = out;
    if (out_mem == RF_MEM_HOST) RF_HIP(sc.get((void**)&d_out, out_bytes));
    rf_args a1 = *args;  // pass 1: everything under the cutoff k1 (in result units: times the common factor)
    a1.score_hint_usize = RF_NO_CUTOFF;
    a1.cutoff_usize = (uint64_t)k1 * factor;
    rf_args a2 = *args;  // the caller's own scan
    a2.score_hint_usize = RF_NO_CUTOFF;
    const uint64_t raw_cut64 = args->cutoff_usize == RF_NO_CUTOFF ? 0xFFFFFFFFull : std::min<uint64_t>(args->cutoff_usize / factor, 0xFFFFFFFFull);
    const uint32_t raw_cut = (uint32_t)raw_cut64;
    // ---- is the hint any good?  Pass 1 over a corpus with near-duplicates in every tile is a full-length band scan -- 1.6 of the 3.0 ms a full scan of 10 M x 256
    // takes -- so a hint that is wrong for most candidates costs up to + 55 % (profiles/bench_hint16_neardup50.json).  Corpora of >= RF_HINT_SAMPLE_MIN_TILES tiles
    // (default 16384) first run pass 1's scan over every (tiles / 512)-th tile of its length window (ONE launch of the compiled multi-word kernel, which takes a
    // tile step) and count: with fewer than 70 % of the candidates that matter resolved (break-even is ~73 %) the hint is dropped and the plain scan runs.  Costs
    // one more stream synchronization.
    static const uint32_t sample_min = [] { const char* e = getenv("RF_HINT_SAMPLE_MIN_TILES"); return e ? (uint32_t)atoll(e) : 16384u; }();
    // (round 6: the sample is skipped while the hint has been proving itself on this corpus -- the pass itself counts what it left unresolved, and that count comes to
    // the host anyway; a hinted call in the steady state then synchronizes once.  A hint that turns bad costs ONE call its first pass, RF_HINT_TRUST=0: sample always.)
    static const bool trust_on = [] { const char* e = getenv("RF_HINT_TRUST"); return !e || atoi(e) != 0; }();
    const uint32_t trust = corpus->hint_trust.load(std::memory_order_relaxed);
    const bool trusted = trust_on && trust != 0 && (trust & 15u) != 0;
    if (corpus->n_tiles >= sample_min && !trusted) {
        ScanParams p1;
        RawKind raw1 = RAW_LEV;
        if (const rf_status rs = plan(c, corpus, op, &a1, false, &p1, &raw1); rs != RF_OK) return rs;
        if (!p1.long_words_pad && p1.words <= (uint32_t)kMaxWords && p1.tile_end > p1.tile_begin) {  // (the sample runs the register-resident scan: queries of <= 512 symbols)
            if (const rf_status rs = comparator_device_pm(c, corpus->device, &p1.pm); rs != RF_OK) return rs;
            p1.out = d_out;
            p1.prefill_none = 1;  // (the count below reads out[]; a tile the cutoff kernel abandons without storing must read as None, not as stale bytes)
            p1.band = 0;  // (the compiled LevState<W> scan under the same cutoff: the same values, and it walks every tile_step-th tile)
            p1.mixed = nullptr, p1.mixed_begin = p1.mixed_end = 0;  // (the views of the mixed section are ordinary tiles to that kernel)
            p1.heads8 = nullptr, p1.heads6 = nullptr;
            p1.tile_step = std::max<uint32_t>(1, (p1.tile_end - p1.tile_begin) / 512);
            uint32_t* d_acc = nullptr;
            RF_HIP(sc.get((void**)&d_acc, 2 * sizeof(uint32_t)));
            RF_HIP(hipMemsetAsync(d_acc, 0, 2 * sizeof(uint32_t), st));
            RF_HIP(launch_scan(raw1, p1, st, nullptr));
            RF_HIP(launch_hint_sample(p1, d_out, p1.tile_begin, p1.tile_end, p1.tile_step, d_acc, st));
            uint32_t acc[2] = {0, 0};
            RF_HIP(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, st));
            RF_HIP(hipStreamSynchronize(st));
            // candidates outside pass 1's length window but inside the caller's: unresolved without looking (tiles are 64 slots; the estimate ignores padding lanes)
            uint64_t in_k1 = 0, in_cut = 0;
            for (size_t i = 0; i < corpus->lengths.size(); ++i) {
                const uint32_t first = corpus->length_first_tile[i], end = i + 1 < corpus->lengths.size() ? corpus->length_first_tile[i + 1] : corpus->n_tiles;
                const uint32_t L = corpus->lengths[i], gap = p1.len1 > L ? p1.len1 - L : L - p1.len1;
                if (L == 0) continue;
                if (gap <= k1) in_k1 += end - first;
                if (gap <= raw_cut) in_cut += end - first;
            }
            const double share = acc[0] ? (double)acc[1] / (double)acc[0] : 0.0;
            const double resolved = in_cut ? share * (double)in_k1 / (double)in_cut : 1.0;
            static const bool trace_sample = getenv("RF_TRACE_PLAN") != nullptr;
            if (trace_sample) std::fprintf(stderr, "[rf plan] hint sample: %u of %u sampled candidates within max(hint, 31) = %u, ~%.2f of the candidates that matter\n", acc[1], acc[0], k1, resolved);
            if (resolved < 0.70) {
                corpus->hint_trust.store(0, std::memory_order_relaxed);
                const rf_status rs = run_many(c_in, corpus_in, op, &a2, d_out, RF_MEM_DEVICE, st, false);
                if (rs != RF_OK) return rs;
                if (out_mem == RF_MEM_HOST) {
                    RF_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
                    RF_HIP(hipStreamSynchronize(st));
                }
                return RF_OK;
            }
        }
    }
    // ---- round 6, single-length corpora whose first pass is the band kernel (k1 <= 31) and whose hint is credible (sampled just now, or trusted): the pass LISTS the
    // lanes it answers None (tile + lane mask, rf_band.hip band_list_kernel), lane_list_pack_kernel numbers them, and the caller's own multi-word scan walks them 64 to a
    // wavefront (rf_sparse.hip sparse_words_kernel) -- no mark pass, no sums, no host in between, no copy of the payload (whose reads the second pass now does itself).
    // What the pass left comes back through pinned memory and is read by the NEXT such call (more than 30 % unresolved: the credit is gone and that call takes the road
    // below, with its sample and its count on the host).  RF_HINT_LISTS=0: the road below always.
    static const bool lists_on = [] { const char* e = getenv("RF_HINT_LISTS"); return !e || atoi(e) != 0; }();
    const bool sampled_ok = corpus->n_tiles >= sample_min && !trusted;  // (a sample that said no has returned above)
    if (lists_on && (trusted || sampled_ok) && corpus->uniform && !corpus->borrowed && k1 <= 31 && c->words >= 2 && c->words <= (size_t)kMaxWords &&
        corpus->uniform_len >= (uint32_t)kChunk) {
        const uint32_t len1 = (uint32_t)c->s1.size(), gap = len1 > corpus->uniform_len ? len1 - corpus->uniform_len : corpus->uniform_len - len1;
        ScanParams p1, p2;
        RawKind raw1 = RAW_LEV, raw2 = RAW_LEV;
        if (gap <= k1 && plan(c, corpus, op, &a1, false, &p1, &raw1) == RF_OK && plan(c, corpus, op, &a2, false, &p2, &raw2) == RF_OK && raw1 == RAW_LEV && raw2 == RAW_LEV &&
            p1.band && !p1.tiles && p1.tile_end > p1.tile_begin && !p2.band && !p2.long_words_pad && !p2.tiles && p2.words >= 2 && p2.words <= (uint32_t)kMaxWords &&
            p2.tile_begin == p1.tile_begin && p2.tile_end == p1.tile_end) {
            if (const rf_status rs = comparator_device_pm(c, corpus->device, &p1.pm); rs != RF_OK) return rs;
            p2.pm = p1.pm;
            std::unique_lock<std::mutex> list_lock(corpus->filter_enqueue_mu);
            uint32_t* buf = corpus_tile_list(corpus, st);
            uint32_t *packed_at = nullptr, *first_at = nullptr;
            p1.tile_list_buf = buf;
            bool go = buf != nullptr && band_list_geometry(p1, &packed_at, &first_at);
            volatile uint32_t* report = nullptr;
            if (go) {
                rf_corpus::TileList& tl = corpus->tile_lists.front();  // (corpus_tile_list moved this stream's list to the front)
                if (!tl.band_report) {
                    void* hp = nullptr;
                    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
                        std::memset(hp, 0, 64);
                        tl.band_report = static_cast<volatile uint32_t*>(hp);
                    } else {
                        (void)hipGetLastError();
                    }
                }
                report = tl.band_report;
                if (report && report[10] != 0) {  // the last such call on this stream: did its first pass leave more than 30 % of the corpus?
                    const uint64_t left = report[8], of = report[9];
                    report[10] = 0;
                    if (left * 10 > of * 3) {
                        corpus->hint_trust.store(0, std::memory_order_relaxed);
                        go = false;
                    }
                }
            }
            if (go) {
                p1.out = d_out;
                p1.lane_list = 1;
                p1.band_list = 1;
                RF_HIP(launch_scan(raw1, p1, st, nullptr));  // (its None pre-fill if any, band_list_kernel, lane_list_pack_kernel)
                p2.out = d_out;
                p2.prefill_none = 0;
                p2.tile_list = packed_at;
                p2.lane_first = first_at;
                p2.tile_list_count = buf;
                p2.band_report = const_cast<uint32_t*>(report);
                RF_HIP(launch_sparse_words(p2, st));
                corpus_tile_list_done(corpus, st);
                list_lock.unlock();
                corpus->hint_trust.store(std::min<uint32_t>(trust + 1, 0x7FFFFFFFu), std::memory_order_relaxed);
                static const bool trace_lists = getenv("RF_TRACE_PLAN") != nullptr;
                if (trace_lists) std::fprintf(stderr, "[rf plan] hint lists: k1=%u, the band pass lists what it leaves, %u-word scan over the list\n", k1, p2.words);
                if (out_mem == RF_MEM_HOST) {
                    RF_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
                    RF_HIP(hipStreamSynchronize(st));
                }
                return RF_OK;
            }
        }
    }
    // ---- pass 1
    if (const rf_status rs = run_many(c_in, corpus_in, op, &a1, d_out, RF_MEM_DEVICE, st, false); rs != RF_OK) return rs;
    // ---- the caller's own scan, planned for the corpus and re-aimed at the dense tiles below
    ScanParams p;
    RawKind raw = RAW_LEV;
    if (const rf_status rs = plan(c, corpus, op, &a2, false, &p, &raw); rs != RF_OK) return rs;
    if (const rf_status rs = comparator_device_pm(c, corpus->device, &p.pm); rs != RF_OK) return rs;
    const uint64_t zero64 = (uint64_t)p.len1 * factor;
    const uint32_t zero_value = (args->cutoff_usize == RF_NO_CUTOFF || zero64 <= args->cutoff_usize) ? (uint32_t)zero64 : RF_NONE_U32;
    // ---- mark: per tile the lanes pass 1 left unresolved, numbered in slot order; the sums at the length runs' boundaries come to the host
    const uint32_t R = (uint32_t)corpus->lengths.size();
    std::vector<uint32_t> run_first(corpus->length_first_tile.begin(), corpus->length_first_tile.end());
    run_first.push_back(corpus->n_tiles);
    uint32_t *d_run_first = nullptr, *d_count = nullptr, *d_prefix = nullptr, *d_run_prefix = nullptr;
    uint64_t* d_mask = nullptr;
    void* d_temp = nullptr;
    const size_t temp_bytes = hint_scan_temp_bytes(corpus->n_tiles);
    RF_HIP(sc.get((void**)&d_run_first, (R + 1) * sizeof(uint32_t)));
    RF_HIP(sc.get((void**)&d_run_prefix, (R + 1) * sizeof(uint32_t)));
    RF_HIP(sc.get((void**)&d_count, ((size_t)corpus->n_tiles + 1) * sizeof(uint32_t)));
    RF_HIP(sc.get((void**)&d_prefix, ((size_t)corpus->n_tiles + 1) * sizeof(uint32_t)));
    RF_HIP(sc.get((void**)&d_mask, (size_t)corpus->n_tiles * sizeof(uint64_t)));
    RF_HIP(sc.get(&d_temp, temp_bytes));
    RF_HIP(hipMemcpyAsync(d_run_first, run_first.data(), (R + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    RF_HIP(launch_hint_mark(p, d_out, raw_cut, zero_value, d_mask, d_count, d_prefix, d_temp, temp_bytes, d_run_first, R, d_run_prefix, st));
    std::vector<uint32_t> run_prefix(R + 1);
    RF_HIP(hipMemcpyAsync(run_prefix.data(), d_run_prefix, (R + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RF_HIP(hipStreamSynchronize(st));  // (the one synchronization of a hinted call: the dense tiles are sized by what pass 1 left)
    std::vector<uint32_t> cnt(R);
    for (uint32_t r = 0; r < R; ++r) cnt[r] = run_prefix[r + 1] - run_prefix[r];
    std::vector<uint32_t> tile_base(R + 1, 0u), run_len(R);
    std::vector<uint64_t> tables;  // (what goes up to the device, in one copy; lives as long as the other staging vectors of this call)
    std::vector<uint64_t> data_base(R, 0ull);
    uint64_t bytes2 = 0, tiles2_64 = 0;
    for (uint32_t r = 0; r < R; ++r) {
        tile_base[r] = (uint32_t)tiles2_64;
        data_base[r] = bytes2;
        run_len[r] = corpus->lengths[r];
        const uint64_t t = ((uint64_t)cnt[r] + kWave - 1) / kWave;
        tiles2_64 += t;
        bytes2 += t * tile_bytes(corpus->lengths[r]);
    }
    tile_base[R] = (uint32_t)tiles2_64;
    static const bool trace_plan = getenv("RF_TRACE_PLAN") != nullptr;
    if (trace_plan) std::fprintf(stderr, "[rf plan] hint pass: k1=%u, %llu dense tiles (%llu bytes) of %u left for the full scan\n", k1, (unsigned long long)tiles2_64, (unsigned long long)bytes2, corpus->n_tiles);
    // what the pass left, for the next hinted call on this corpus: at most 30 % unresolved (the sample's own line) and the hint keeps its credit
    if ((uint64_t)run_prefix[R] * 10 <= (uint64_t)corpus->n * 3)
        corpus->hint_trust.store(std::min<uint32_t>(trust + 1, 0x7FFFFFFFu), std::memory_order_relaxed);
    else
        corpus->hint_trust.store(0, std::memory_order_relaxed);
    if ((uint64_t)run_prefix[R] * 4 > (uint64_t)corpus->n * 3) {
        // the hint was wrong for more than three quarters of the corpus: gathering them costs more than scanning the few resolved ones again
        if (const rf_status rs = run_many(c_in, corpus_in, op, &a2, d_out, RF_MEM_DEVICE, st, false); rs != RF_OK) return rs;
    } else if (tiles2_64 != 0) {
        const uint32_t n_tiles2 = (uint32_t)tiles2_64;
        uint32_t *d_tile_base = nullptr, *d_run_len = nullptr, *d_orig2 = nullptr;
        uint64_t* d_data_base = nullptr;
        uint8_t* d_data2 = nullptr;
        TileDesc* d_tiles2 = nullptr;
        // (the three small tables in one block and one copy: every scratch block is a lock and an event, every copy from pageable memory a staged transfer)
        tables.assign(R + (2 * (size_t)R + 1 + 1) / 2, 0);  // R x u64 data_base | (R + 1) x u32 tile_base | R x u32 run_len
        std::memcpy(tables.data(), data_base.data(), R * sizeof(uint64_t));
        std::memcpy(reinterpret_cast<uint32_t*>(tables.data() + R), tile_base.data(), (R + 1) * sizeof(uint32_t));
        std::memcpy(reinterpret_cast<uint32_t*>(tables.data() + R) + (R + 1), run_len.data(), R * sizeof(uint32_t));
        RF_HIP(sc.get((void**)&d_data_base, tables.size() * sizeof(uint64_t)));
        d_tile_base = reinterpret_cast<uint32_t*>(d_data_base + R);
        d_run_len = d_tile_base + (R + 1);
        RF_HIP(sc.get((void**)&d_data2, bytes2 + kTailPad));
        RF_HIP(sc.get((void**)&d_tiles2, (size_t)n_tiles2 * sizeof(TileDesc)));
        RF_HIP(sc.get((void**)&d_orig2, (size_t)n_tiles2 * kWave * sizeof(uint32_t)));
        RF_HIP(hipMemcpyAsync(d_data_base, tables.data(), tables.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        RF_HIP(hipMemsetAsync(d_data2 + bytes2, 0, kTailPad, st));  // (the scans prefetch one chunk row past the last tile)
        RF_HIP(launch_hint_gather(p, d_run_first, R, d_run_prefix, d_prefix, d_mask, d_tile_base, d_data_base, d_run_len, n_tiles2, d_data2, d_tiles2, d_orig2, st));
        // ---- pass 2: a general corpus of exact tiles whose orig[] holds ORIGINAL candidate indices: results land in the caller's vector
        p.data = d_data2;
        p.tiles = d_tiles2;
        p.orig = d_orig2;
        p.n_tiles = p.n_exact = n_tiles2;
        p.tile_begin = 0, p.tile_end = n_tiles2;
        p.mixed = nullptr, p.mixed_len = nullptr, p.mixed_orig = nullptr;
        p.mixed_begin = p.mixed_end = 0;
        p.joint_begin = p.joint_end = 0;
        p.zero_begin[0] = p.zero_end[0] = p.zero_begin[1] = p.zero_end[1] = 0;
        p.uniform_len = 0, p.uniform_tile_bytes = 0;
        p.heads8 = nullptr, p.heads6 = nullptr, p.run_orig = nullptr;
        p.head_need = p.head_k = 0;
        p.prefill_none = 0;
        p.out = d_out;
        if (p.long_words_pad) {  // (queries beyond 512 symbols: the multi-sweep kernel's scratch strips, sized for the dense tiles)
            p.long_grid = (uint32_t)std::max(1, std::min<int>((int)p.long_grid, scan_grid(n_tiles2)));
            if (const size_t scratch = launch_scratch_bytes(p, raw)) RF_HIP(sc.get((void**)&p.long_scratch, scratch));
        }
        RF_HIP(launch_scan(raw, p, st, nullptr));
    }
    if (out_mem == RF_MEM_HOST) {
        RF_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        RF_HIP(hipStreamSynchronize(st));
    }
    return RF_OK;
}

rf_status run_many(const rf_comparator* c_in, const rf_corpus* corpus_in, rf_op op, const rf_args* args, void* out,
                          rf_mem out_mem, void* stream, bool f64_out)
{
    if (!c_in || !corpus_in || !args) {
        set_error("null handle or args");
        return RF_ERR_INVALID_ARG;
    }
    AccelTimer call_timer("rf_many call (host side)");
    Effective eff;
    if (const rf_status rs = make_effective(c_in, corpus_in, (hipStream_t)stream, &eff); rs != RF_OK) return rs;
    const rf_comparator* c = eff.c;
    const rf_corpus* corpus = eff.corpus;
    DeviceGuard guard(corpus->n ? corpus->device : -1);  // (before plan(): grids are sized from the current device's CU count)
    if (corpus->n && !guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    ScanParams p;
    RawKind raw = RAW_LEV;
    rf_status s = plan(c, corpus, op, args, f64_out, &p, &raw);
    if (s != RF_OK) return s;
    if (corpus->n == 0) return RF_OK;
    if (!out) {
        set_error("null output");
        return RF_ERR_INVALID_ARG;
    }
    s = comparator_device_pm(c, corpus->device, &p.pm);
    if (s != RF_OK) return s;

    hipStream_t st = (hipStream_t)stream;
    // RF_FLAG_SLOT_ORDER (round 6): out[slot] instead of out[original index] -- a length-bucketed corpus then needs neither scattered stores nor the gather pass
    // (rfgpu.h; rf_corpus_slot_index gives the map).  A single-length corpus' slots ARE its indices: nothing to do.
    const bool want_slots = (args->flags & kFlagSlotsInternal) != 0 && !corpus->uniform && corpus->d_orig != nullptr;  // (set by rf_many_* from RF_FLAG_SLOT_ORDER: rf_host.hpp)
    if (want_slots && (corpus != corpus_in || corpus->borrowed || !corpus->n_slots)) {
        set_error("RF_FLAG_SLOT_ORDER: not available for this corpus / query pair (a u32 query with overflow symbols, or a streamed segment)");
        return RF_ERR_UNSUPPORTED;
    }
    // normalized_distance / normalized_similarity of a multi-word Levenshtein scan: the u32 distance scan (the asm kernels with the
    // Ukkonen band: 3.3 instead of 2.5 Gpairs/s at 256 x 256) into a temporary, then ONE pass that runs the very arithmetic of emit_fin (rf_device.hpp) on every
    // distance -- dist / maximum, 1.0 - nd, the cutoff compare -- 12 bytes per candidate against a scan of >= 128 symbols each.  RF_NORM_TWO_STEP=0: the A/B switch.
    static const bool norm_two_step = [] { const char* e = getenv("RF_NORM_TWO_STEP"); return !e || atoi(e) != 0; }();
    // (round 6) ... and under an f64 cutoff that leaves at most 31 raw edits -- normalized_similarity >= 0.9 of a 256-symbol query: 25 -- the u32 scan runs under THAT
    // cutoff, i.e. the small-band kernel (rf_band.hip: 75 Gpairs/s where the compiled f64 early-out scan walks four words per column), and the normalizing pass decides
    // Some / None from the exact distance as before.  The raw cutoff is the largest distance the f64 test can pass for the longest candidate, plus one edit of slack for
    // the rounding of c x maximum: whatever the band answers None is beyond it, and None stays None (details/distance.rs:246-250, :273; common.rs:43-45).
    // RF_NORM_BAND=0: the compiled f64 scan.
    static const bool norm_band = [] { const char* e = getenv("RF_NORM_BAND"); return !e || atoi(e) != 0; }();
    uint64_t norm_raw_cut = RF_NO_CUTOFF;
    if (norm_band && f64_out && raw == RAW_LEV && p.has_cutoff && p.finish == FIN_LEV && p.factor >= 1 && c->words >= 2 &&
        (op == RF_OP_NORMALIZED_DISTANCE || op == RF_OP_NORMALIZED_SIMILARITY)) {
        const double c_nd = op == RF_OP_NORMALIZED_DISTANCE ? p.cutoff_f64 : 1.0 - p.cutoff_f64;
        const double maximum = (double)p.factor * (double)std::max<uint64_t>(p.len1, corpus->max_len);  // (FIN_LEV: maximum = factor x the longer string, ascending in len2)
        if (c_nd >= 0.0 && c_nd * maximum < 32.0 * (double)p.factor) {
            const uint64_t k_units = (uint64_t)std::floor(c_nd * maximum) + p.factor;
            if (k_units / p.factor <= 31) norm_raw_cut = k_units;
        }
    }
    if (norm_two_step && !want_slots && f64_out && raw == RAW_LEV && p.words >= 2 && p.words <= kMaxWords && (!p.early || norm_raw_cut != RF_NO_CUTOFF) && !p.long_words_pad && corpus == corpus_in && !corpus->borrowed &&
        corpus->n_tiles >= 16 && (uint64_t)p.len1 + corpus->max_len < 0x7FFFFFFFu && (corpus->uniform || (corpus->d_orig && corpus->d_tiles))) {
        // (length-bucketed corpora: the candidates' lengths in original order, 4 bytes each, built once per corpus from the tile descriptors)
        const uint32_t* len_of = nullptr;
        if (!corpus->uniform) {
            std::lock_guard<std::mutex> lock(corpus->scratch_mu);
            if (!corpus->d_len_of) {
                uint32_t* l = nullptr;
                hipError_t e1 = hipMalloc((void**)&l, corpus->n * sizeof(uint32_t));
                if (e1 == hipSuccess) e1 = launch_len_of(corpus->d_tiles, corpus->n_tiles, corpus->d_orig, l, st);
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);  // (other streams may use it as soon as the lock is released)
                if (e1 != hipSuccess) {
                    if (l) (void)hipFree(l);
                    (void)hipGetLastError();  // (no room: the compiled f64 scan below, same values)
                } else
                    corpus->d_len_of = l;
            }
            len_of = corpus->d_len_of;
        }
        if (corpus->uniform || len_of) {
            rf_args a = *args;
            a.cutoff_usize = norm_raw_cut;  // (RF_NO_CUTOFF unless the f64 cutoff leaves <= 31 raw edits)
            a.score_hint_usize = RF_NO_CUTOFF;
            // (the temporaries are optional -- no room for them is not an error, the compiled f64 scan below needs none: same values)
            uint32_t* d_dist = nullptr;
            double* d_out = static_cast<double*>(out);
            bool have_tmp = scratch_alloc((void**)&d_dist, corpus->n * sizeof(uint32_t), st) == hipSuccess;
            if (have_tmp && out_mem == RF_MEM_HOST && scratch_alloc((void**)&d_out, corpus->n * sizeof(double), st) != hipSuccess) {
                scratch_free(d_dist, st);
                have_tmp = false;
            }
            if (!have_tmp) (void)hipGetLastError();
          if (have_tmp) {
            const rf_status rs = run_many(c_in, corpus_in, RF_OP_DISTANCE, &a, d_dist, RF_MEM_DEVICE, st, false);
            hipError_t e = hipSuccess;
            if (rs == RF_OK) e = launch_normalize(d_dist, len_of, corpus->uniform_len, d_out, (uint32_t)corpus->n, p.len1, p.fin_mS, p.fin_mM, p.op, p.has_cutoff, p.cutoff_f64, st);
            scratch_free(d_dist, st);
            if (rs == RF_OK && e == hipSuccess && out_mem == RF_MEM_HOST) {
                e = hipMemcpyAsync(out, d_out, corpus->n * sizeof(double), hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
            }
            if (out_mem == RF_MEM_HOST && d_out != out) scratch_free(d_out, st);
            if (rs != RF_OK) return rs;
            RF_HIP(e);
            return RF_OK;
          }
        }
    }
    {
        uint32_t k1 = 0, factor = 1;
        if (corpus == corpus_in && !want_slots && hint_pass_applies(c, corpus, op, args, f64_out, &k1, &factor)) {
            const rf_status rs = run_many_hinted(c_in, corpus_in, c, corpus, args, static_cast<uint32_t*>(out), out_mem, st, k1, factor);
            if (rs != RF_ERR_OOM) return rs;
            //  no room for the hinted passes' temporaries (masks, sums, dense tiles): not an error -- the plain scan below needs none and returns
            // the same values (results never depend on the hint, levenshtein.rs:2153-2160)
            (void)hipGetLastError();
        }
    }
    p.heads8 = corpus_head8_plane(corpus, p, raw, st);
    p.heads6 = p.heads8 ? corpus_head6_plane(corpus, st) : nullptr;
    // the HBM-bound scans of a single-length corpus stream the 6-bit payload where there is one (Indel / LCS, one word, no early-out)
    // (where the asm scan over it applies -- rf_scan.hip launch_state; f64 results through a table of the <= 256 values there are: rf_stream_asm.hip stream_asm_f64_table)
    // (lengths that are not whole chunks: the 32-bit column runs the packer's fill columns too -- 57 symbols, query 30: 83.8 -> 100.7 Gpairs/s; the issue-bound 64-bit
    // column shifts the partial chunk into place and runs the real columns only)
    // (bucketed corpora: the tiles kernels over the same payload, u32 results -- the ragged scans read 16-byte chunk rows for every started 16 symbols, and 12 here)
    p.data6 = (raw == RAW_LCS && p.words == 1 && !p.early && (corpus->uniform || (!f64_out && corpus->d_orig != nullptr)))
                  ? corpus_data6(corpus, st)
                  : nullptr;
    if (p.data6) p.max_stored_sym = corpus_max_stored_symbol(corpus, st);  // (< 63: the scans may zero the table row of the fill code)
    if (corpus->uniform) plan_band_filter(c, corpus, op, f64_out, &p, corpus->uniform_len);  // (bucketed corpora: per length run, launch_scan_runs)
    static const bool jaro_priv = [] { const char* e = getenv("RF_JARO_PRIV"); return e && atoi(e) != 0; }();  // (off by default: rf_jaro.hip launch_jaro_word)
    if (jaro_priv && raw == RAW_JARO && corpus->uniform && !p.has_cutoff) p.max_stored_sym = corpus_max_stored_symbol(corpus, st);
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    const size_t out_bytes = (want_slots ? corpus->n_slots : corpus->n) * elem;
    void* d_out = out;
    if (out_mem == RF_MEM_HOST) RF_HIP(scratch_alloc(&d_out, out_bytes, st));  // (kept by the pool: no hipMalloc / hipFree pair -- and no device-wide sync -- per call)
    p.out = d_out;
    // Large ragged corpora: results in slot order into a temporary, then ONE gather into original order (rf_pack.hip
    // "gather_results_kernel" has the why: the scattered out[orig[slot]] stores of a length-bucketed corpus cost more than the scan).
    // The mixed section is walked through its one-length views so that every candidate has exactly one slot; a cutoff's length
    // window pre-fills the TEMPORARY with None (launch_scan: p.out, p.n slots).  RF_UNSCATTER_MIN=<candidates> moves the threshold
    // (0 = never).
    // Full no-cutoff scans of the VALU-bound single-word kernels (Levenshtein for queries > 32, OSA): walk the tiles BY ORIGIN
    // (tiles_by_origin()) with the workgroups dealt to them XCD by XCD (ScanParams::xcd_deal) and store straight through orig[].
    // The tiles in flight on one XCD then write one compact window of `out`, the partial lines meet in that XCD's L2, and what is
    // left of the scatter (64 L2 transactions per wavefront store instead of 4) hides under the kernel's arithmetic: bench.py
    // --ragged 62.2 -> 74.7 (Levenshtein), 49.5 -> 59.5 (OSA) against the gather below, which stays for the kernels that are
    // short of memory system instead (query <= 32: 72 -> 66, Indel: 80 -> 68 this way; profiles/ragged_result_order_r03.txt).
    // RF_TILE_ORDER: 0 = never, 1 = by origin without the deal, 2 = default, 3 = also the kernels that lose by it.
    const int tile_order = tile_order_knob();
    const bool valu_bound = p.words == 1 && ((raw == RAW_LEV && p.len1 > 32) || raw == RAW_OSA);
    // (Jaro: when every exact tile takes the single-word kernel -- launch_jaro splits the tiles BY POSITION where the lengths pass
    // 64 symbols, which needs the length order)
    const bool jaro_word_only = raw == RAW_JARO && !p.has_cutoff && p.jaro_split >= corpus->n_exact && !p.jaro_long;
    const bool by_origin = tile_order && !want_slots && corpus->d_tiles_by_origin && !corpus->borrowed && !p.early && !p.prefill_none && !p.band && !p.long_words_pad &&
                           ((valu_bound && !p.out_f64) || jaro_word_only || (tile_order >= 3 && (raw == RAW_LEV || raw == RAW_LCS || raw == RAW_OSA))) &&
                           p.tile_begin == 0 && p.tile_end == corpus->n_tiles;
    if (by_origin) {
        p.tiles = corpus->d_tiles_by_origin;
        p.xcd_deal = tile_order >= 2 ? 1u : 0u;
    }
    static const size_t unscatter_min = [] { const char* e = getenv("RF_UNSCATTER_MIN"); return e ? (size_t)atoll(e) : (size_t)1 << 20; }();
    void* d_tmp = nullptr;
    bool tmp_owned = false;                 // d_tmp is this call's own stream-ordered allocation
    rf_corpus::GatherTmp* kept_tmp = nullptr;  // ... or this kept buffer of the corpus (valid while tmp_lock is held)
    std::unique_lock<std::mutex> tmp_lock;  // held while a kept temporary's scan + gather are enqueued
    // (under a cutoff only the tiles of the passing length window write through orig[]; the gather is a fixed 10-12 bytes per
    // candidate of the WHOLE corpus, so it pays from a window of ~30 % of the tiles on: measured break-even, bench.py --ragged --cutoff)
    const bool wide_window = (uint64_t)(p.tile_end - p.tile_begin) * 10 >= (uint64_t)corpus->n_tiles * 3;
    const bool by_runs = !by_origin && scan_runs_applies(corpus, p, raw);  // small-cutoff scans of a bucketed corpus: one single-length view per length run
    if (want_slots || (unscatter_min && corpus->n >= unscatter_min && corpus->d_orig && !corpus->borrowed && corpus->n_slots && wide_window && !by_origin && !by_runs)) {
        {
            std::lock_guard<std::mutex> lock(corpus->scratch_mu);
            if (!corpus->d_slot_ident) {
                AccelTimer timer("slot maps + window table");
                // once per corpus: the slot -> slot map the scans store through, and what the gather needs -- the window table
                // (rf_pack.hip window_gather_kernel) when the slots are few enough ascending runs, else the candidate -> slot map
                static const bool use_windows = [] { const char* e = getenv("RF_GATHER_WINDOWS"); return !e || atoi(e) != 0; }();
                uint32_t *so = nullptr, *si = nullptr, *list = nullptr, *table = nullptr;
                uint32_t n_runs = 0, n_rows = 0;
                std::vector<uint32_t> runs(kMaxGatherRuns + 2, 0u);  // [0] = count, then the run starts
                hipError_t e1 = hipMalloc((void**)&si, corpus->n_slots * sizeof(uint32_t));
                if (e1 == hipSuccess && use_windows) {
                    e1 = hipMalloc((void**)&list, runs.size() * sizeof(uint32_t));
                    if (e1 == hipSuccess) e1 = hipMemsetAsync(list, 0, sizeof(uint32_t), st);
                    if (e1 == hipSuccess) e1 = launch_run_starts(corpus->d_orig, (uint32_t)corpus->n_slots, list + 1, kMaxGatherRuns, list, st);
                    if (e1 == hipSuccess) e1 = hipMemcpyAsync(runs.data(), list, (kMaxGatherRuns + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
                    if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);
                    if (e1 == hipSuccess && runs[0] >= 1 && runs[0] <= kMaxGatherRuns) {
                        n_runs = runs[0];
                        std::sort(runs.begin() + 1, runs.begin() + 1 + n_runs);
                        runs[1 + n_runs] = (uint32_t)corpus->n_slots;
                        n_rows = (uint32_t)((corpus->n + kGatherWindow - 1) / kGatherWindow) + 1;
                        e1 = hipMemcpyAsync(list, runs.data() + 1, (n_runs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st);
                        if (e1 == hipSuccess) e1 = hipMalloc((void**)&table, (size_t)n_rows * n_runs * sizeof(uint32_t));
                        if (e1 == hipSuccess) e1 = launch_window_table(corpus->d_orig, list, n_runs, n_rows, table, st);
                    }
                }
                if (e1 == hipSuccess && !table) {
                    e1 = hipMalloc((void**)&so, corpus->n * sizeof(uint32_t));
                    if (e1 == hipSuccess) e1 = hipMemsetAsync(so, 0xFF, corpus->n * sizeof(uint32_t), st);
                }
                if (e1 == hipSuccess) e1 = launch_slot_maps(corpus->d_orig, (uint32_t)corpus->n_slots, so, si, st);
                // (round 5) the window gather reads a slot's place inside its span from 2 bytes instead of the 4 of orig[] (RF_GATHER_OFF16=0: the A/B switch;
                // no room for them: the gather reads orig[] as before)
                static const bool use_off16 = [] { const char* e = getenv("RF_GATHER_OFF16"); return !e || atoi(e) != 0; }();
                uint16_t* o16 = nullptr;
                if (e1 == hipSuccess && table && use_off16) {
                    if (hipMalloc((void**)&o16, corpus->n_slots * sizeof(uint16_t)) != hipSuccess) (void)hipGetLastError(), o16 = nullptr;
                    if (o16) e1 = launch_slot_off16(corpus->d_orig, (uint32_t)corpus->n_slots, o16, st);
                }
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);  // (other streams may use the maps as soon as the lock is released)
                if (list) (void)hipFree(list);
                if (e1 != hipSuccess) {
                    // Not an error: an HBM-tight caller keeps what round 2 gave it -- the launch below stores straight
                    // through orig[] (scattered, slower, same values).  The next call tries again.
                    if (so) (void)hipFree(so);
                    if (si) (void)hipFree(si);
                    if (table) (void)hipFree(table);
                    if (o16) (void)hipFree(o16);
                    (void)hipGetLastError();
                } else {
                    corpus->d_slot_of = so;
                    corpus->d_window_table = table;
                    corpus->d_slot_off16 = o16;
                    corpus->gather_runs = n_runs;
                    corpus->gather_rows = n_rows;
                    corpus->d_slot_ident = si;
                }
            }
        }
        if (want_slots) {
            if (!corpus->d_slot_ident) {
                if (out_mem == RF_MEM_HOST) scratch_free(d_out, st);
                set_error("RF_FLAG_SLOT_ORDER: no device memory for the slot map");
                return RF_ERR_OOM;
            }
            // the caller's vector IS the slot-ordered one: n_slots entries, no temporary, no gather
            p.orig = corpus->d_slot_ident;
            p.slot_store = 1;
            p.mixed = nullptr;  // views, not scan_kernel_mixed: one slot per candidate
            p.mixed_end = 0;
            p.n = (uint32_t)corpus->n_slots;
            p.prefill_window = (args->flags & kFlagWindowInternal) ? 1u : 0u;
        } else if (corpus->d_slot_ident) {
        // the temporary: this stream's kept buffer (grown if this call needs f64 where u32 was kept); beyond 4 streams per corpus a
        // stream-ordered allocation for the call
        const size_t tmp_bytes = corpus->n_slots * elem;
        tmp_lock = std::unique_lock<std::mutex>(corpus->gather_enqueue_mu);
        hipError_t ea = hipSuccess;
        for (rf_corpus::GatherTmp& t : corpus->gather_tmp)
            if (t.stream == st) {
                if (t.bytes < tmp_bytes) {
                    void* bigger = nullptr;
                    ea = hipMalloc(&bigger, tmp_bytes);
                    if (ea == hipSuccess) {
                        (void)hipFree(t.ptr);  // (synchronizes with the work that used it)
                        t.ptr = bigger;
                        t.bytes = tmp_bytes;
                    }
                }
                d_tmp = t.ptr;
                kept_tmp = &t;
                if (t.done && hipStreamWaitEvent(st, t.done, 0) != hipSuccess) (void)hipGetLastError();
                break;
            }
        if (!d_tmp && ea == hipSuccess) {
            if (corpus->gather_tmp.size() < 4) {
                AccelTimer timer("gather temporary");
                ea = hipMalloc(&d_tmp, tmp_bytes);
                if (ea == hipSuccess) {
                    hipEvent_t ev = nullptr;
                    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) (void)hipGetLastError(), ev = nullptr;
                    corpus->gather_tmp.push_back({st, d_tmp, tmp_bytes, ev});
                    kept_tmp = &corpus->gather_tmp.back();
                }
            } else {
                ea = scratch_alloc(&d_tmp, tmp_bytes, st);
                tmp_owned = true;
            }
        }
        if (ea != hipSuccess) {  // no room for the temporary: scattered stores through orig[] as before (not an error)
            (void)hipGetLastError();
            d_tmp = nullptr;
            tmp_owned = false;
            tmp_lock.unlock();
        } else {
            p.out = d_tmp;
            p.orig = corpus->d_slot_ident;
            p.slot_store = 1;  // (d_slot_ident is kPad on padding lanes and the identity elsewhere; the temporary has n_slots entries)
            p.mixed = nullptr;  // views, not scan_kernel_mixed: one slot per candidate
            p.mixed_end = 0;
            p.n = (uint32_t)corpus->n_slots;
        }
        }
    }
    if (const size_t scratch = launch_scratch_bytes(p, raw)) {
        const hipError_t ea = scratch_alloc((void**)&p.long_scratch, scratch, st);
        if (ea != hipSuccess) {
            if (out_mem == RF_MEM_HOST) scratch_free(d_out, st);
            if (d_tmp && tmp_owned) (void)scratch_free(d_tmp, st);  // (this call's own temporary must not outlive the failure)
        }
        RF_HIP(ea);
    }
    std::unique_lock<std::mutex> filter_lock;  // held while a filter pass and the scan over its list are enqueued
    // (... and the small-band scans of a single-length corpus: tiles with a few lanes left are listed for a dense second pass, rf_band.hip launch_band)
    const bool band_lists = p.band && raw == RAW_LEV && corpus->uniform && !p.tiles && !want_slots && p.tile_step == 1 && !by_runs;
    // (... and of a bucketed corpus: its long length runs are walked as single-length corpora of their own, launch_band_runs.  RF_BAND_RUNS=0: one launch over the tiles)
    static const bool band_runs_on = [] { const char* e = getenv("RF_BAND_RUNS"); return !e || atoi(e) != 0; }();
    const bool band_runs = band_runs_on && p.band && raw == RAW_LEV && !corpus->uniform && p.tiles == corpus->d_tiles && corpus->d_orig != nullptr && !want_slots &&
                           p.tile_step == 1 && !by_runs && !by_origin && !p.topk_k && band_has_long_run(p, corpus);
    int band_runs_mode = 2;  // 2: every long run through launch_band, 1: the longest one only, 0: none (one launch over the tiles)
    if (p.heads8 || band_lists || band_runs) {  // (the head-plane scans: band prefilter or first look as a streaming pass, then the cutoff scan over its list)
        filter_lock = std::unique_lock<std::mutex>(corpus->filter_enqueue_mu);
        p.tile_list_buf = corpus_tile_list(corpus, st);
        corpus_lane_buffers(corpus, &p);
        if ((band_lists || band_runs) && p.tile_list_buf) {
            p.band_defer_seen = p.tile_list_buf + tile_list_words(corpus->n_tiles) - 4;
            // Which form this launch takes.  Handing tiles over pays when many tiles hold a few near candidates; on a corpus with none, or with most lanes of most tiles
            // near, it buys nothing and costs the list kernels and ~3 % of the first pass' columns.  The second pass of this stream's LAST hand-over launch left what it
            // listed in pinned memory (E tiles, S lanes, of N tiles at column a); read here without waiting -- whatever launch it is from, it only steers speed:
            //   columns saved ~ E (L - a) - S / 64 L   against   N a + (tiles alive at a) (L - a)  of the plain launch
            // (tiles alive at a: the listed ones, or -- when the listed ones sit right at the lane limit, so that most live tiles are above it -- all of them).
            // Below 25 % the plain kernel runs, and every 16th launch looks again.  RF_BAND_DEFER_ADAPT=0: always hand over.
            static const bool adapt = [] { const char* e = getenv("RF_BAND_DEFER_ADAPT"); return !e || atoi(e) != 0; }();
            rf_corpus::TileList& tl = corpus->tile_lists.front();  // (corpus_tile_list moved this stream's list to the front)
            if (!tl.band_report) {
                void* hp = nullptr;
                if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
                    std::memset(hp, 0, 64);
                    tl.band_report = static_cast<volatile uint32_t*>(hp);
                } else {
                    (void)hipGetLastError();
                }
            }
            p.band_report = const_cast<uint32_t*>(tl.band_report);
            if (adapt && tl.band_report) {
                bool pays = false;
                if (tl.band_report[3] != 0) {
                    const double E = tl.band_report[0], S = tl.band_report[1], N = tl.band_report[2], a = tl.band_report[4], lanes_max = tl.band_report[5];
                    const double L = tl.band_report[6], rest = L > a ? L - a : 0.0;  // (the length of the launch that reported: a bucketed corpus' last run)
                    const double saved = E * rest - S / 64.0 * L;
                    const bool at_the_limit = E > 0 && S / E > 0.75 * lanes_max;
                    const double plain = N * a + (at_the_limit ? N : E) * rest;
                    pays = saved > 0.25 * plain;  // (the estimate runs ~15 points above what is measured: gathered chunk loads, the list kernels)
                }
                // (no report yet: the stream's FIRST launch hands over, the ones enqueued behind it before its report is in do not -- a host that enqueues ahead
                // would otherwise send a dozen launches down a road nobody has measured)
                const bool look = (tl.band_plain_calls++ & 15u) == 0;
                if (!pays && !look) p.band_defer_seen = nullptr;  // (launch_band: the plain kernel)
                // (a bucketed corpus pays for the hand-over with a launch sequence PER length run -- 13 runs of 770 k candidates: 626 instead of 183 us on random
                // rows: all runs only while it pays; a look peels off the longest run alone, the rest stays one launch over the tiles)
                band_runs_mode = pays ? 2 : (look ? 1 : 0);
            }
        }
    }
    static const bool trace_plan = getenv("RF_TRACE_PLAN") != nullptr;  // one line per rf_many_* call on stderr: which path the plan took
    if (trace_plan)
        std::fprintf(stderr, "[rf plan] raw=%d words=%u early=%u first_check=%u band=%u heads8=%d head_need=%u head_k=%u tile_list=%d by_runs=%d by_origin=%d gather=%d "
                             "tiles=[%u,%u) of %u prefill=%u data6=%d\n",
                     (int)raw, p.words, p.early, p.first_check, p.band, p.heads8 != nullptr, p.head_need, p.head_k, p.tile_list_buf != nullptr, (int)by_runs, (int)by_origin,
                     d_tmp != nullptr, p.tile_begin, p.tile_end, corpus->n_tiles, p.prefill_none, p.data6 != nullptr);
    hipError_t e = by_runs ? launch_scan_runs(raw, p, c, corpus, op, f64_out, st) : (band_runs && band_runs_mode != 0 && p.band_defer_seen ? launch_band_runs(raw, p, corpus, st, band_runs_mode == 1) : launch_scan(raw, p, st, nullptr));
    if (filter_lock.owns_lock()) {
        if (p.tile_list_buf) corpus_tile_list_done(corpus, st);
        filter_lock.unlock();
    }
    if (p.long_scratch) (void)scratch_free(p.long_scratch, st);
    if (d_tmp) {
        if (e == hipSuccess)
            e = corpus->d_window_table ? launch_window_gather(d_tmp, corpus->d_orig, corpus->d_slot_off16, corpus->d_window_table, corpus->gather_runs, corpus->gather_rows, d_out,
                                                              (uint32_t)corpus->n, f64_out, st)
                                       : launch_gather_results(d_tmp, corpus->d_slot_of, d_out, (uint32_t)corpus->n, f64_out, st);
        if (tmp_owned) (void)scratch_free(d_tmp, st);
        else if (kept_tmp && kept_tmp->done && hipEventRecord(kept_tmp->done, st) != hipSuccess) (void)hipGetLastError();
        if (tmp_lock.owns_lock()) tmp_lock.unlock();
    }
    if (e == hipSuccess && out_mem == RF_MEM_HOST) {
        e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (out_mem == RF_MEM_HOST) scratch_free(d_out, st);
    if (e != hipSuccess) {
        set_error(std::string("scan launch: ") + hipGetErrorString(e));
        return e == hipErrorInvalidValue ? RF_ERR_UNSUPPORTED : RF_ERR_HIP;
    }
    return RF_OK;
}

rf_status rf_many_u32(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, uint32_t* out,
                      rf_mem out_mem, void* stream)
try {
    if (!args) return run_many(c, corpus, op, args, out, out_mem, stream, false);
    const rf_args a = sanitized_args(args, true);
    return run_many(c, corpus, op, &a, out, out_mem, stream, false);
}
RF_ABI_CATCH

rf_status rf_many_f64(const rf_comparator* c, const rf_corpus* corpus, rf_op op, const rf_args* args, double* out,
                      rf_mem out_mem, void* stream)
try {
    if (!args) return run_many(c, corpus, op, args, out, out_mem, stream, true);
    const rf_args a = sanitized_args(args, true);
    return run_many(c, corpus, op, &a, out, out_mem, stream, true);
}
RF_ABI_CATCH

static rf_status run_one(const rf_comparator* c, const uint8_t* s2, size_t len2, rf_op op, const rf_args* args_in, int device, void* out,
                         int* is_some, bool f64_out)
{
    if (!c || !args_in || !out || !is_some || (len2 && !s2)) {
        set_error("rf_one: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    const rf_args args_v = sanitized_args(args_in, false), *args = &args_v;
    const uint64_t offsets[2] = {0, len2};
    rf_corpus* corpus = nullptr;
    rf_status s = rf_corpus_pack(s2, offsets, 1, device, &corpus);
    if (s != RF_OK) return s;
    if (f64_out) {
        double v = 0.0;
        s = run_many(c, corpus, op, args, &v, RF_MEM_HOST, nullptr, true);
        *static_cast<double*>(out) = v;
        *is_some = !std::isnan(v);
    } else {
        uint32_t v = 0;
        s = run_many(c, corpus, op, args, &v, RF_MEM_HOST, nullptr, false);
        *static_cast<uint32_t*>(out) = v;
        *is_some = v != RF_NONE_U32;
    }
    rf_corpus_free(corpus);
    return s;
}
rf_status rf_one_u32(const rf_comparator* c, const uint8_t* s2, size_t len2, rf_op op, const rf_args* args, int device, uint32_t* out,
                     int* is_some)
try {
    return run_one(c, s2, len2, op, args, device, out, is_some, false);
}
RF_ABI_CATCH
rf_status rf_one_f64(const rf_comparator* c, const uint8_t* s2, size_t len2, rf_op op, const rf_args* args, int device, double* out,
                     int* is_some)
try {
    return run_one(c, s2, len2, op, args, device, out, is_some, true);
}
RF_ABI_CATCH

// ---------------------------------------------------------------------------------------------------
// many queries x one corpus
// ---------------------------------------------------------------------------------------------------
// Queries whose recurrences fit one machine word and agree on the kernel family are fused kMaxMulti (then 2) at a
// time into scan_multi_kernel launches, which read every candidate byte once per group; the rest go through the
// single-query launch.  Either way row q of `out` is exactly what rf_many_* gives for cs[q].
static rf_status run_many_multi(const rf_comparator* const* cs_in, uint32_t q, const rf_corpus* corpus, rf_op op, const rf_args* args_in,
                                void* out, rf_mem out_mem, void* stream, bool f64_out)
{
    if (!cs_in || !corpus || !args_in) {
        set_error("null handle or args");
        return RF_ERR_INVALID_ARG;
    }
    const rf_args args_v = sanitized_args(args_in, false), *args = &args_v;  // (rows of n entries: RF_FLAG_SLOT_ORDER is rf_many_*'s alone)
    if (q == 0 || corpus->n == 0) return RF_OK;
    if (!out) {
        set_error("null output");
        return RF_ERR_INVALID_ARG;
    }
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    const size_t row_bytes = corpus->n * elem;
    std::vector<ScanParams> ps(q);
    std::vector<RawKind> raws(q, RAW_LEV);
    std::vector<const rf_comparator*> eff(q, nullptr);
    std::vector<ComparatorRef> holds(q);
    for (uint32_t i = 0; i < q; ++i) {
        bool overflow_hit = false;
        if (resolve(cs_in[i], corpus, &eff[i], &holds[i], &overflow_hit) != RF_OK && overflow_hit && corpus->d_raw) {
            // a query with overflow-class symbols needs its own translated image of the corpus: one launch per query
            for (uint32_t j = 0; j < q; ++j) {
                const rf_status sj = run_many(cs_in[j], corpus, op, args, static_cast<char*>(out) + (size_t)j * row_bytes, out_mem, stream, f64_out);
                if (sj != RF_OK) return sj;
            }
            return RF_OK;
        }
    }
    for (uint32_t i = 0; i < q; ++i) {
        rf_status s = resolve(cs_in[i], corpus, &eff[i], &holds[i]);
        if (s == RF_OK) s = plan(eff[i], corpus, op, args, f64_out, &ps[i], &raws[i]);
        if (s != RF_OK) return s;
    }
    const rf_comparator* const* cs = eff.data();
    DeviceGuard guard(corpus->device);
    if (!guard.ok) {
        set_error("cannot select the corpus' device");
        return RF_ERR_NO_DEVICE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* d_out = static_cast<char*>(out);
    if (out_mem == RF_MEM_HOST) RF_HIP(scratch_alloc((void**)&d_out, row_bytes * q, st));

    // fusable: single-word Levenshtein / LCS-family recurrences; the group key is what the kernel cannot vary per query
    // (a tight cutoff is better served by one early-out launch per query than by the fused kernel, which runs every column)
    auto fusable = [&](uint32_t i) { return (raws[i] == RAW_LEV || raws[i] == RAW_LCS) && cs[i]->words == 1 && !ps[i].long_words_pad && !ps[i].early; };
    auto same_group = [&](uint32_t a, uint32_t b) {
        return raws[a] == raws[b] && ps[a].finish == ps[b].finish && ps[a].factor == ps[b].factor && ps[a].op == ps[b].op &&
               (ps[a].len1 <= 32) == (ps[b].len1 <= 32);
    };
    std::vector<char> done(q, 0);
    rf_status status = RF_OK;
    hipError_t e = hipSuccess;
    for (uint32_t i = 0; i < q && status == RF_OK && e == hipSuccess; ++i) {
        if (done[i]) continue;
        std::vector<uint32_t> group{i};
        if (fusable(i))
            for (uint32_t j = i + 1; j < q && group.size() < (size_t)kMaxMulti; ++j)
                if (!done[j] && fusable(j) && same_group(i, j) && j == group.back() + 1) group.push_back(j);  // contiguous rows of out
        if (group.size() == 3) group.pop_back();
        for (uint32_t g : group) done[g] = 1;
        ScanParams p = ps[i];
        p.out = d_out + (size_t)i * row_bytes;
        if (group.size() == 1) {
            // (through run_many: a general corpus' results take the cheapest way into original order there)
            status = run_many(cs_in[i], corpus, op, args, p.out, RF_MEM_DEVICE, stream, f64_out);
        } else {
            p.early = 0;  // the fused kernel always runs every column of every tile (values are the same either way)
            p.tile_begin = 0, p.tile_end = p.n_tiles, p.prefill_none = 0;
            p.multi_q = (uint32_t)group.size();
            for (size_t k = 0; k < group.size() && status == RF_OK; ++k) {
                p.multi_len1[k] = ps[group[k]].len1;
                status = comparator_device_pm(cs[group[k]], corpus->device, &p.multi_pm[k]);
            }
            if (status != RF_OK) break;
            // general corpora, LCS family: the tiles are walked by origin with the XCD deal (run_many has the why).  20 M ragged
            // candidates x 4 queries: Indel 1.06 -> 0.89 ms.  Not the fused Levenshtein kernels: 1.19 -> 1.68 ms -- their code (4
            // recurrences x 16 tail entries) is large, and by origin the wavefronts of a CU run different tail lengths at the same
            // time where the storage order keeps them on the same path (instruction cache); their scatter already hides under 4
            // queries' arithmetic.
            if (raws[i] == RAW_LCS && tile_order_knob() && corpus->d_tiles_by_origin && !corpus->borrowed) {
                p.tiles = corpus->d_tiles_by_origin;
                p.xcd_deal = tile_order_knob() >= 2 ? 1u : 0u;
            }
            e = launch_scan_multi(raws[i], p.len1 <= 32, p, st);
        }
    }
    if (status == RF_OK && e == hipSuccess && out_mem == RF_MEM_HOST) {
        e = hipMemcpyAsync(out, d_out, row_bytes * q, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (out_mem == RF_MEM_HOST) {
        if (status != RF_OK || e != hipSuccess) (void)hipStreamSynchronize(st);
        scratch_free(d_out, st);
    }
    if (status != RF_OK) return status;
    if (e != hipSuccess) {
        set_error(std::string("multi-query scan: ") + hipGetErrorString(e));
        return e == hipErrorInvalidValue ? RF_ERR_UNSUPPORTED : RF_ERR_HIP;
    }
    return RF_OK;
}

rf_status rf_many_multi_u32(const rf_comparator* const* cs, uint32_t q, const rf_corpus* corpus, rf_op op, const rf_args* args,
                            uint32_t* out, rf_mem out_mem, void* stream)
try {
    return run_many_multi(cs, q, corpus, op, args, out, out_mem, stream, false);
}
RF_ABI_CATCH

rf_status rf_many_multi_f64(const rf_comparator* const* cs, uint32_t q, const rf_corpus* corpus, rf_op op, const rf_args* args,
                            double* out, rf_mem out_mem, void* stream)
try {
    return run_many_multi(cs, q, corpus, op, args, out, out_mem, stream, true);
}
RF_ABI_CATCH


}  // extern "C"
