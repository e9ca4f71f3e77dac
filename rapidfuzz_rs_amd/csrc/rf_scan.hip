// rf_scan.hip -- the one-vs-many scan kernels over the packed corpus and their dispatch:
//   scan_body / scan_kernel*      multi-word states and every cutoff run (early-out)
//   stream_body / stream_kernel   the lean no-cutoff loop of the single-word states (the headline path)
//   scan_multi_kernel             Q queries per pass over the corpus
//   topk_final_kernel             selection of the k best candidate keys
// Device helpers live in rf_device.hpp; the long-pattern, generalized-weights and Jaro kernels in rf_long.hip / rf_jaro.hip.
#include <atomic>
#include <type_traits>

#include "rf_device.hpp"

namespace rf {

// kFirst >= 0: a cutoff scan (`early` is a compile-time fact) whose first look inside a tile's first chunk sits at column kFirst
// (16 = at the chunk's end); kFirst < 0: `early` and the look are read from ScanParams at run time.
template <class State, bool kUniform, int kFirst = -1>
__device__ __forceinline__ void scan_body(const ScanParams& p, typename State::Word* lds_pm, uint64_t (*lds_topk)[kWave])
{
    constexpr int W = State::kWords;
    // stage the PM table; the 32-bit states keep the low half of each (single-word) entry
    // (the corpus stores renamed symbols sigma(c), see rf_corpus: row c of the table goes to row sigma(c))
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock)
        lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (typename State::Word)p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    const bool early = kFirst >= 0 ? true : (State::kCanPrune && p.early != 0);
    WaveTopK best;
    best.init();
    // offers are filtered by `limit` = min(launch-wide pruning bound as last seen, own list's worst key), a scalar.
    // This body is at its VGPR budget (cutoff state + early-out), so the bound is re-read only every 8th tile and
    // consumed at once instead of riding in registers across a tile like stream_body's.
    uint64_t limit = ~0ull;
    uint32_t tiles_done = 0;

    // Each wavefront walks its tiles as one continuous stream of 16-column chunks.  The load of the NEXT
    // chunk (the next 16 columns of this tile, or the first 16 of the wavefront's next tile) is always issued
    // before the current chunk is processed, so exactly one 1 KiB request per wavefront is in flight and the
    // wait before each chunk is a counted vmcnt(1), never a drain.
    // With a cutoff (`early`) the bet is the opposite: after 16 columns nearly every wavefront of a random
    // corpus is past the cutoff, so the prefetch goes to the NEXT TILE and a surviving wavefront fetches its
    // own next chunk on demand -- a dead tile costs 16 of its 64+ bytes per candidate in HBM traffic.
    uint32_t t = p.tile_begin + (dealt_workgroup(p) * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        TileView cur_tile = load_tile<kUniform>(p, t);
        uint4 cur = load_chunk(cur_tile.src + lane);  // the packed buffer carries one chunk of tail padding: always readable

        while (true) {
            const uint32_t t_next = t + stride;
            const bool has_next = t_next < p.tile_end;
            const TileView next_tile = load_tile<kUniform>(p, has_next ? t_next : t);

            const uint32_t len2 = cur_tile.len;
            const uint32_t slot = cur_tile.slot0 + lane;
            uint32_t idx = slot;
            if (!kUniform) idx = p.orig[slot];  // issued early; consumed after the columns

            State st;
            st.init();
            const uint32_t nch = (len2 + kChunk - 1) / kChunk;
            const TileFin fin = tile_fin(p, p.len1, len2);
            bool dead = false;
            uint4 ahead = make_uint4(0, 0, 0, 0);
            if (early || nch == 0) ahead = load_chunk(next_tile.src + lane);
            for (uint32_t c = 0; c < nch; ++c) {
                uint4 nxt = ahead;
                if (!early) {
                    const uint4* nsrc = (c + 1 < nch) ? cur_tile.src + (size_t)(c + 1) * kWave : next_tile.src;
                    nxt = load_chunk(nsrc + lane);
                }
                const uint32_t cols = len2 - c * kChunk;
#ifndef RF_NO_COMPILED_BAND  // (measurement builds: tools/build_scan_variants.sh "-DRF_NO_COMPILED_BAND=" 1 -- round 4's multi-word scan for the A/B)
                if constexpr (has_band<State>::value) st.set_band(p.len1, len2, p.trim_k1 ? p.trim_k1 - 1u : 0xFFFFFFFFu, c);
#endif
                if (cols >= kChunk) {
                    if (early && c == 0) {
                        // first chance to stop: an unrelated candidate gains almost one edit per column, so the first look is
                        // taken a few columns past the cutoff (p.first_check = 4 ... 16, chosen by plan() from the raw
                        // distance the cutoff still allows -- a choice of speed only, every look is value-preserving)
#define RF_FIRST_LOOK(J)                                                                  \
    {                                                                                     \
        process_chunk_full<State, 0, J>(st, lds_pm, cur);                                 \
        if (__ballot(may_pass(p, fin, st.bound_first(p.len1, J, len2))) == 0) {           \
            dead = true;                                                                  \
            break;                                                                        \
        }                                                                                 \
        process_chunk_full<State, J, kChunk>(st, lds_pm, cur);                            \
    }
                        if constexpr (kFirst == 16) {
                            process_chunk_full<State>(st, lds_pm, cur);
                        } else if constexpr (kFirst >= 0) {
                            RF_FIRST_LOOK(kFirst)
                        } else if constexpr (W == 1) {  // (the multi-word kernels keep the one look at column 8: six copies of their chunk code would not pay)
                            if (p.first_check == 4) RF_FIRST_LOOK(4)
                            else if (p.first_check == 6) RF_FIRST_LOOK(6)
                            else if (p.first_check == 10) RF_FIRST_LOOK(10)
                            else if (p.first_check == 12) RF_FIRST_LOOK(12)
                            else if (p.first_check == 14) RF_FIRST_LOOK(14)
                            else if (p.first_check == 16) process_chunk_full<State>(st, lds_pm, cur);  // loose cutoff: the look at the chunk's end is the first
                            else RF_FIRST_LOOK(8)
                        } else RF_FIRST_LOOK(8)
#undef RF_FIRST_LOOK
                    } else {
                        process_chunk_full<State>(st, lds_pm, cur);
                    }
                } else {
                    process_chunk_tail<State>(st, lds_pm, cur, cols);
                }
                if (early) {
                    const uint32_t j = min(len2, (c + 1) * kChunk);
                    if (__ballot(may_pass(p, fin, st.bound(p.len1, j, len2))) == 0) {
                        dead = true;  // the whole wavefront is beyond the cutoff: stop reading this tile
                        break;
                    }
                    if (c + 1 < nch) nxt = load_chunk(cur_tile.src + (size_t)(c + 1) * kWave + lane);
                }
                cur = nxt;
            }
            if (dead || nch == 0) cur = ahead;

            const bool valid = kUniform ? slot < p.n : idx != kPad;
            const uint32_t raw = st.result(p.len1, len2);
            if (p.out && valid) {
                if (dead)
                    emit_none(p, idx);
                else
                    emit_fin(p, fin, raw, idx, p.out);
            }
            if (topk && !dead) {
                bool keep;
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);
                if ((tiles_done++ & 7u) == 0) topk_refresh_bound(p, limit);
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);
            }

            if (!has_next) break;
            t = t_next;
            cur_tile = next_tile;
        }
    }

    if (topk) topk_block_publish(p, best, lds_topk, wave, lane, limit);
}

// Two entry points over the same body: the single-word kernels are pinned to 8 wavefronts per SIMD (otherwise the
// scalar state of the tile loop pushes them to 96 SGPRs = 7 resident workgroups per CU); W >= 2 keeps the
// compiler's own register budget (forcing 8 would spill VGPRs).
template <class State, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void scan_kernel(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    scan_body<State, kUniform>(p, lds_pm, lds_topk);
}
// The cutoff scan of a SINGLE-LENGTH corpus, written around its common case: a tile whose every lane is past the cutoff at the
// first look.  That case is a straight line here -- state, kFirst columns, the diagonal bound, one ballot, the None results, the
// next tile -- with the length-dependent values (chunk count, finishing coefficients, row masks) computed once per wavefront;
// everything a surviving tile needs (the rest of its first chunk, later chunks fetched on demand, looks at every chunk end,
// finishing, top-k) sits behind one branch.  scan_body spends 50 scalar instructions and 21 branches on a dead tile even with
// the look compiled in; four SIMDs share one scalar unit, so that is time the 136 vector instructions cannot overlap.
// kHead8: the first look reads the candidates' first 8 symbols from the HEAD PLANE (rf_pack.hip head8_plane_kernel: 8 bytes per
// candidate, tile t's row at t * 512 B) instead of their first 16-byte chunk rows: a scan under a cutoff <= 5 decides nearly every
// tile from <= 8 columns, so the bytes it has to move halve (and the look itself runs on 32-bit words: the narrow look below).
// A surviving tile fetches its first chunk row from the ordinary payload and starts over.
// the band prefilter's table (below): per STORED symbol, bit i = "the symbol occurs in query[i - K .. i + K]"
__device__ __forceinline__ void build_band_table(const ScanParams& p, uint8_t* lds_band)
{
    const uint32_t K = p.head_k;
    for (int c = threadIdx.x; c < 256; c += kWave * kWavesPerBlock) {
        const uint32_t row = (uint32_t)p.pm[(size_t)c * p.words];  // query positions 0..31 hold every band of the first 8 columns (K <= 3)
        uint32_t bits = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            const uint32_t lo = i > K ? i - K : 0u, hi = i + K;  // rows lo..hi
            const uint32_t mask = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
            bits |= (row & mask) ? 1u << i : 0u;
        }
        lds_band[p.sigma[c]] = (uint8_t)bits;
    }
}
__device__ __forceinline__ uint32_t band_hits(const uint8_t* lds_band, uint32_t lo, uint32_t hi)
{
    uint32_t hit = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t sym = ((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xFFu;
        hit |= (uint32_t)lds_band[sym] & (1u << i);
    }
    return (uint32_t)__popc(hit);
}

// The band prefilter as a pass of its own (full scans, tile_step == 1): a streaming kernel over the head plane that does nothing
// else -- 16 bytes per lane per load (TWO candidates; a wavefront takes a PAIR of tiles per trip: lanes 0..31 hold tile t, lanes
// 32..63 tile t + 1), two loads in flight, ~45 instructions per pair -- writes None for the tiles it decides and leaves the others
// in ITS OWN segment of a list (no atomics: one shared counter took 12 ns per survivor, 0.4 ms of a 0.5 ms pass);
// tile_list_pack_kernel packs the segments, and the cutoff scan proper (early_lean_kernel) walks the packed list.
// Inside the cutoff kernel the same filter was held to 4.2-4.6 TB/s of head plane by that kernel's loop (25 scalar instructions
// and 12 branches per tile around it, one tile per trip); a plain streaming read of the plane in 16-byte-per-lane rows moves
// 6.7 TB/s (tools/membw.hip).
// The pass also takes THE FIRST LOOK itself: for the pairs the band test lets through (a few percent: their bytes are in registers,
// and a tile that passes the band test but not the look would otherwise cost the cutoff kernel a list entry, a row fetch and the
// same look), and for every pair when the band filter does not apply (cutoffs that allow 4..5 edits, or corpora whose alphabet
// leaves the filter nothing to decide): the 32-bit recurrence over the first kFirst columns of BOTH candidates of a lane (two
// independent chains) and the diagonal bound.  Inside early_head8_kernel that look sits in a loop of 25 scalar instructions and
// 12 branches per tile, one tile per trip.
// buf: [0] packed count | G per-wavefront counts | G words unused | G segments of `cap` tiles | the packed list   (G = wavefronts of this launch)
// kPlane6: the pass reads the 6-bit plane (ScanParams::heads6: three coalesced dword loads per lane and pair, 768 B per wavefront instead
// of 1024) and widens the two candidates to the byte form of the 8-byte plane in registers -- ~30 more vector instructions per pair in a
// pass that is bound by the stream (C5 shape, 100 M candidates: 138 -> ~105 us).
__device__ __forceinline__ uint32_t widen24(uint32_t x)  // four 6-bit symbols on bits 0..23 -> four bytes
{
    return (x & 0x3Fu) | ((x << 2) & 0x3F00u) | ((x << 4) & 0x3F0000u) | ((x << 6) & 0x3F000000u);
}
// kLanes (round 6): the pass lists its surviving tiles WITH the mask of their surviving lanes (16-byte entries), and writes the None of every candidate it decides,
// whatever the rest of its tile does: what is left for the second pass is the surviving CANDIDATES, gathered 64 to a wavefront (rf_sparse.hip).  A lane survives when
// its own symbols pass the band test AND its own first look passes -- both are necessary conditions per candidate.
template <class State, int kFirst, bool kPlane6, bool kLanes = false>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void head_filter_kernel(const ScanParams p, uint32_t* __restrict__ buf, uint32_t cap)
{
    using Look = typename std::conditional<std::is_same<State, LevState<1>>::value, Lev32State,
                                           typename std::conditional<std::is_same<State, OsaState<1>>::value, Osa32State, State>::type>::type;
    constexpr int kLookPitch = (std::is_same<State, LevState<1>>::value || std::is_same<State, OsaState<1>>::value) ? 2 : 1;
    static_assert(kFirst <= 8, "the head plane holds 8 symbols per candidate");
    __shared__ typename State::Word lds_pm[256];
    __shared__ uint8_t lds_band[256];
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds_pm[p.sigma[i]] = (typename State::Word)p.pm[i];
    const uint32_t need = p.head_need;  // 0 = no band test
    if (need) build_band_table(p, lds_band);
    __syncthreads();
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t pairs = (p.tile_end - p.tile_begin + 1) / 2, stride = gridDim.x * kWavesPerBlock;
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;
    // kLanes: [0] packed count | [1] survivors | 2 pad | G (tile count, lane count) pairs | G segments of `cap` 16-byte entries | the packed entries
    uint32_t* seg = kLanes ? buf + 4 + 2 * (size_t)stride + (size_t)gw * cap * 4 : buf + 1 + 2 * (size_t)stride + (size_t)gw * cap;
    uint32_t kept = 0, kept_lanes = 0;
    uint32_t pr = gw;
    if (pr < pairs) {
        const uint32_t len1 = p.len1, len2 = p.uniform_len;
        const TileFin fin = tile_fin(p, len1, len2);
        const typename Look::Word* pm = reinterpret_cast<const typename Look::Word*>(lds_pm);
        const uint8_t* base = p.heads8 + (size_t)lane * sizeof(v4u);
        const uint32_t* base6 = kPlane6 ? p.heads6 + (size_t)(p.tile_begin / 2) * (3 * kWave) + lane : nullptr;  // (the launcher sends even tile_begin here)
        auto load_pair = [&](uint32_t q) {
            if constexpr (kPlane6) {
                const uint32_t* row = base6 + (size_t)q * (3 * kWave);
                v4u v;
                v.x = __builtin_nontemporal_load(row);
                v.y = __builtin_nontemporal_load(row + kWave);
                v.z = __builtin_nontemporal_load(row + 2 * kWave);
                v.w = 0u;
                return v;
            } else {
                return __builtin_nontemporal_load(reinterpret_cast<const v4u*>(base + ((uint64_t)p.tile_begin + 2ull * q) * (kWave * 8)));
            }
        };
        auto widen = [&](v4u v) {  // A[31:0] | A[47:32] + B[15:0] << 16 | B[47:16] -> the two candidates' 8 bytes each
            if constexpr (kPlane6) {
                v4u r;
                r.x = widen24(v.x);
                r.y = widen24(__builtin_amdgcn_alignbit(v.y, v.x, 24) & 0xFFFFFFu);
                r.z = widen24(__builtin_amdgcn_alignbit(v.z, v.y, 16));
                r.w = widen24(v.z >> 8);
                return r;
            } else {
                return v;
            }
        };
        v4u packed = load_pair(pr);
        v4u ahead = load_pair(pr + stride < pairs ? pr + stride : pr);
        while (true) {
            const uint32_t pr_next = pr + stride;
            const v4u ahead2 = load_pair(pr_next + stride < pairs ? pr_next + stride : pr);
            const v4u cur = widen(packed);
            const uint32_t t0 = p.tile_begin + 2 * pr;
            uint64_t m = ~0ull;
            if constexpr (kLanes) {
                // Who survives the pass, per LANE.  With the band test on: whoever passes IT -- the look is not run here at all: on a random corpus one candidate in
                // a thousand passes the band test and the second pass disposes of it at its first chunk end, while on a corpus that shares prefixes with the query the
                // look (~150 instructions per pair, all lanes, whenever ONE lane passed the band test) made this streaming pass VALU-bound: 119 -> 236 us per 100 M at
                // 1 % prefix sharers (profiles/survivors_r06.txt).  Without a band test (cutoffs of 4..5 edits): whoever passes the look, as before.
                const bool tile_ok = lane < 32 || t0 + 1 < p.tile_end;  // (an odd tile count: the last pair's second half is the plane's pad row)
                const uint32_t ia = t0 * kWave + 2 * lane;              // (lanes 32..63 run on into tile t0 + 1)
                bool pa = tile_ok && ia < p.n, pb = tile_ok && ia + 1 < p.n;
                if (need) {
                    pa = pa && band_hits(lds_band, cur.x, cur.y) >= need;
                    pb = pb && band_hits(lds_band, cur.z, cur.w) >= need;
                } else {
                    Look a, b;
                    a.init();
                    b.init();
                    process_chunk_full<Look, 0, kFirst, kLookPitch>(a, pm, make_uint4(cur.x, cur.y, 0u, 0u));
                    process_chunk_full<Look, 0, kFirst, kLookPitch>(b, pm, make_uint4(cur.z, cur.w, 0u, 0u));
                    pa = pa && may_pass(p, fin, a.bound_first(len1, kFirst, len2));
                    pb = pb && may_pass(p, fin, b.bound_first(len1, kFirst, len2));
                }
                m = __ballot(pa || pb);
                uint64_t mask0 = 0, mask1 = 0;
                if (m != 0) {
                    // bit k of a tile's mask = its candidate k, which sits in lane k / 2 (+ 32 for the pair's second tile), slot k & 1
                    const int v = (pa ? 1 : 0) | (pb ? 2 : 0);
                    const int s0 = __shfl(v, (int)(lane >> 1), kWave), s1 = __shfl(v, (int)(32 + (lane >> 1)), kWave);
                    mask0 = __ballot((s0 >> (lane & 1)) & 1);
                    mask1 = __ballot((s1 >> (lane & 1)) & 1);
                }
                if (p.out && !p.run_orig && tile_ok) {  // (a length run of a bucketed corpus: `out` is pre-filled with None)
                    // None for the whole pair, survivors included: one coalesced 8-byte store per lane whatever the masks say -- the second pass runs behind this
                    // kernel and overwrites the survivors that pass (per-lane stores around the survivors cost the 1 % corpus 60 us per 100 M)
                    const uint32_t idx = t0 * kWave + 2 * lane;
                    if (!p.out_f64 && idx + 1 < p.n) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(p.out) + idx) = make_uint2(RF_NONE_U32, RF_NONE_U32);
                    } else {
                        if (idx < p.n) emit_none(p, idx);
                        if (idx + 1 < p.n) emit_none(p, idx + 1);
                    }
                }
                // the list: one 16-byte entry per surviving tile -- tile, its lane mask, (filled in by the pack kernel) the survivors in front of it
                if (mask0 != 0) {
                    if (lane == 0) reinterpret_cast<uint4*>(seg)[kept] = make_uint4(t0, (uint32_t)mask0, (uint32_t)(mask0 >> 32), 0u);
                    ++kept;
                    kept_lanes += (uint32_t)__popcll(mask0);
                }
                if (mask1 != 0) {
                    if (lane == 0) reinterpret_cast<uint4*>(seg)[kept] = make_uint4(t0 + 1, (uint32_t)mask1, (uint32_t)(mask1 >> 32), 0u);
                    ++kept;
                    kept_lanes += (uint32_t)__popcll(mask1);
                }
                if (pr_next >= pairs) break;
                pr = pr_next;
                packed = ahead;
                ahead = ahead2;
                continue;
            }
            if (need) m = __ballot(band_hits(lds_band, cur.x, cur.y) >= need || band_hits(lds_band, cur.z, cur.w) >= need);
            if (m != 0) {
                Look a, b;
                a.init();
                b.init();
                process_chunk_full<Look, 0, kFirst, kLookPitch>(a, pm, make_uint4(cur.x, cur.y, 0u, 0u));
                process_chunk_full<Look, 0, kFirst, kLookPitch>(b, pm, make_uint4(cur.z, cur.w, 0u, 0u));
                m = __ballot(may_pass(p, fin, a.bound_first(len1, kFirst, len2)) || may_pass(p, fin, b.bound_first(len1, kFirst, len2)));
            }
            const bool alive0 = (uint32_t)m != 0, alive1 = (uint32_t)(m >> 32) != 0 && t0 + 1 < p.tile_end;
            if (p.out && !p.run_orig) {  // (a length run of a bucketed corpus: `out` is pre-filled with None)
                const bool mine_dead = lane < 32 ? !alive0 : (!alive1 && t0 + 1 < p.tile_end);
                const uint32_t idx = t0 * kWave + 2 * lane;
                if (mine_dead) {
                    if (!p.out_f64 && idx + 1 < p.n) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(p.out) + idx) = make_uint2(RF_NONE_U32, RF_NONE_U32);
                    } else {
                        if (idx < p.n) emit_none(p, idx);
                        if (idx + 1 < p.n) emit_none(p, idx + 1);
                    }
                }
            }
            if (alive0) {
                if (lane == 0) seg[kept] = t0;
                ++kept;
            }
            if (alive1) {
                if (lane == 0) seg[kept] = t0 + 1;
                ++kept;
            }
            if (pr_next >= pairs) break;
            pr = pr_next;
            packed = ahead;
            ahead = ahead2;
        }
    }
    if (lane == 0) {
        if constexpr (kLanes) {
            reinterpret_cast<uint2*>(buf + 4)[gw] = make_uint2(kept, kept_lanes);  // (one 8-byte word per wavefront: the pack kernel adds them up in front of every block)
        } else {
            buf[1 + gw] = kept;
        }
    }
}

// packing the segments: ONE launch, a workgroup per 256 segments.  Every workgroup adds up the counts in front of its block itself
// (<= 16 K coalesced loads spread over 256 threads: cheaper than a launch that would do it once), scans its own 256 counts and
// copies its segments behind one another; the last workgroup leaves the total in buf[0].  (First version: one workgroup doing
// everything, every thread walking 16 segments one dependent load after another: 123 us.  Second: an offsets kernel whose threads
// read 16 CONSECUTIVE counts each -- 64-byte lane stride, 17 us through one CU's L1 -- and a copy kernel, 6 us.)
// buf: [0] packed count | G per-wavefront counts | G words unused | G segments of `cap` tiles | the packed list
__global__ __launch_bounds__(256) void tile_list_pack_kernel(uint32_t* __restrict__ buf, uint32_t G, uint32_t cap)
{
    constexpr uint32_t kWaves = 256 / kWave;
    __shared__ uint32_t own[kWaves], front[kWaves];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t first = blockIdx.x * 256, s = first + threadIdx.x;
    uint32_t before = 0;  // this thread's share of the counts in front of the block
#pragma unroll 8
    for (uint32_t k = threadIdx.x; k < first; k += 256) before += buf[1 + k];
    const uint32_t n = s < G ? buf[1 + s] : 0u;
    uint32_t incl = n;  // inclusive scan of n inside the wavefront
#pragma unroll
    for (uint32_t d = 1; d < (uint32_t)kWave; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
    }
#pragma unroll
    for (uint32_t d = kWave / 2; d >= 1; d >>= 1) before += __shfl_xor(before, d, kWave);  // wavefront sum
    if (lane == kWave - 1) own[wave] = incl;
    if (lane == 0) front[wave] = before;
    __syncthreads();
    uint32_t at = incl - n, block_total = 0, in_front = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) {
        in_front += front[w];
        if (w < wave) at += own[w];
        block_total += own[w];
    }
    at += in_front;
    if (s < G) {
        const uint32_t* seg = buf + 1 + 2 * (size_t)G + (size_t)s * cap;
        uint32_t* packed = buf + 1 + 2 * (size_t)G + (size_t)G * cap;
        for (uint32_t j = 0; j < n; ++j) packed[at + j] = seg[j];
    }
    if (first + 256 >= G && threadIdx.x == 0) buf[0] = in_front + block_total;
}

// The same for the LANE lists of head_filter_kernel<..., kLanes = true> (round 6): 16-byte entries (tile, lane mask lo / hi, -), two counts per segment -- tiles and
// surviving lanes -- and two running sums: an entry's last word becomes the number of survivors in front of it, which is what rf_sparse.hip searches.
// first[j] = the packed entry that holds survivor 64 j: where dense tile j of the second pass starts looking (every entry holds >= 1 survivor, so the 64 entries
// from there on hold all of the tile's 64).
// buf: [0] packed entries | [1] survivors | 2 pad | G (tile count, lane count) pairs | G segments of `cap` entries | the packed entries | first[]
__global__ __launch_bounds__(256) void lane_list_pack_kernel(uint32_t* __restrict__ buf, uint32_t G, uint32_t cap, uint32_t* __restrict__ first_of)
{
    constexpr uint32_t kWaves = 256 / kWave;
    __shared__ uint32_t own[kWaves], front[kWaves], own_l[kWaves], front_l[kWaves];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t first = blockIdx.x * 256, s = first + threadIdx.x;
    const uint2* __restrict__ counts = reinterpret_cast<const uint2*>(buf + 4);  // (tiles, surviving lanes) per segment
    uint32_t before = 0, before_l = 0;  // this thread's share of the counts in front of the block
#pragma unroll 8
    for (uint32_t k = threadIdx.x; k < first; k += 256) {  // (independent loads, eight in flight: the last block's chain is what the kernel takes)
        const uint2 c2 = counts[k];
        before += c2.x;
        before_l += c2.y;
    }
    const uint2 mine2 = s < G ? counts[s] : make_uint2(0u, 0u);
    const uint32_t n = mine2.x, nl = mine2.y;
    uint32_t incl = n, incl_l = nl;  // inclusive scans inside the wavefront
#pragma unroll
    for (uint32_t d = 1; d < (uint32_t)kWave; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, kWave), vl = __shfl_up(incl_l, d, kWave);
        if (lane >= d) incl += v, incl_l += vl;
    }
#pragma unroll
    for (uint32_t d = kWave / 2; d >= 1; d >>= 1) {
        before += __shfl_xor(before, d, kWave);
        before_l += __shfl_xor(before_l, d, kWave);
    }
    if (lane == kWave - 1) own[wave] = incl, own_l[wave] = incl_l;
    if (lane == 0) front[wave] = before, front_l[wave] = before_l;
    __syncthreads();
    uint32_t at = incl - n, lat = incl_l - nl, block_total = 0, block_total_l = 0, in_front = 0, in_front_l = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) {
        in_front += front[w];
        in_front_l += front_l[w];
        if (w < wave) at += own[w], lat += own_l[w];
        block_total += own[w];
        block_total_l += own_l[w];
    }
    at += in_front;
    lat += in_front_l;
    // the copy: a thread per segment, FOUR entries' loads in flight at a time (one dependent 16-byte load and store per entry was the longest kernel of the step at
    // 5 % prefix sharers, ~90 entries per segment; a wavefront per segment was tried and is no faster: it walks its 64 segments one memory round trip after another)
    if (s < G) {
        const uint4* seg = reinterpret_cast<const uint4*>(buf + 4 + 2 * (size_t)G) + (size_t)s * cap;
        uint4* packed = reinterpret_cast<uint4*>(buf + 4 + 2 * (size_t)G) + (size_t)G * cap;
        uint32_t run = lat;
        auto place = [&](uint4 e, uint32_t j) {
            e.w = run;
            packed[at + j] = e;
            const uint32_t c = (uint32_t)__popc(e.y) + (uint32_t)__popc(e.z), kk = (run + kWave - 1) / kWave;  // (c <= 64: at most one multiple of 64 in [run, run + c))
            if (kk * kWave < run + c) first_of[kk] = at + j;
            run += c;
        };
        uint32_t j = 0;
        for (; j + 4 <= n; j += 4) {
            const uint4 e0 = seg[j], e1 = seg[j + 1], e2 = seg[j + 2], e3 = seg[j + 3];
            place(e0, j), place(e1, j + 1), place(e2, j + 2), place(e3, j + 3);
        }
        for (; j < n; ++j) place(seg[j], j);
    }
    if (first + 256 >= G && threadIdx.x == 0) {
        buf[0] = in_front + block_total;
        buf[1] = in_front_l + block_total_l;
    }
}

hipError_t launch_lane_list_pack(uint32_t* buf, uint32_t G, uint32_t cap, uint32_t* first_of, hipStream_t stream)
{
    hipLaunchKernelGGL(lane_list_pack_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, buf, G, cap, first_of);
    return hipGetLastError();
}

// THE BAND PREFILTER (kHead8, p.head_need != 0: cutoffs that allow at most K = p.head_k <= 3 edits).  Any alignment of cost <= K
// stays within K of the main diagonal, and every candidate symbol it does not MATCH costs at least one edit (a substitution or an
// insertion; an OSA transposition costs one for two symbols that both equal a query symbol one off their path position, which is
// still within K of their own).  So a candidate within the cutoff has at least 8 - K symbols among its first 8 that equal a query
// symbol at a position within K of their own -- a necessary condition that costs 3 instructions per symbol: a 256-byte LDS table
// holds, per stored symbol, bit i = "occurs in query[i - K .. i + K]" (from the table rows the kernel loads anyway), the lane ORs
// (table[c_i] & (1 << i)) over its 8 head symbols and counts the bits.  A tile in which no lane reaches 8 - K is None as a whole
// without running the recurrence (which is 14 instructions per column for 6 columns at cutoff 3); any other tile takes the first
// look as before, all lanes, so no value depends on the filter.  The host switches it on per launch from the corpus' symbol
// frequencies (plan_band_filter, rf_api_scan.hip): on a 62-symbol alphabet 6 % of the tiles pass at K = 3.
template <class State, int kFirst, bool kHead8 = false>
__device__ __forceinline__ void early_lean_body(const ScanParams& p, typename State::Word* lds_pm, uint64_t (*lds_topk)[kWave], uint8_t* lds_band = nullptr)
{
    constexpr int W = State::kWords;
    static_assert(W == 1 && kFirst >= 4 && kFirst <= 16, "single-word states, first look inside the first chunk");
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock)
        lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (typename State::Word)p.pm[i];
    const bool band_filter = kHead8 && p.head_need != 0;
    if constexpr (kHead8) {
        if (band_filter) build_band_table(p, lds_band);
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const bool topk = p.topk_k != 0;
    WaveTopK best;
    best.init();
    uint64_t limit = ~0ull;
    uint32_t tiles_done = 0;

    const uint32_t len2 = p.uniform_len, len1 = p.len1;
    const uint32_t nch = (len2 + kChunk - 1) / kChunk;  // >= 1: the launcher sends lengths >= 16 here
    const TileFin fin = tile_fin(p, len1, len2);
    // the narrow first look (below): 64-bit Levenshtein only, and only when the diagonal through (len1, len2) crosses column kFirst
    // inside the first 32 rows.  RF_NARROW_LOOK=0 is the A/B switch (read by the launcher into p.flags_narrow).
    constexpr bool kWide = std::is_same<State, LevState<1>>::value || std::is_same<State, OsaState<1>>::value;  // 64-bit states with a 32-bit twin
    constexpr bool kNarrowLook = kWide && kFirst < 16;
    const int32_t look_row = (int32_t)kFirst + (int32_t)len1 - (int32_t)len2;
    const bool narrow = kNarrowLook && p.narrow_look && look_row >= 1 && look_row <= 32;
    // the tiles of this launch: the range [tile_begin, tile_end) in steps of tile_step, or (after head_filter_kernel) a list
    const uint32_t n_trips = p.tile_list ? *p.tile_list_count : (p.tile_end > p.tile_begin ? (p.tile_end - p.tile_begin + p.tile_step - 1) / p.tile_step : 0u);
    const uint32_t trip_stride = gridDim.x * kWavesPerBlock;
    auto tile_of = [&](uint32_t k) { return p.tile_list ? p.tile_list[k] : p.tile_begin + k * p.tile_step; };
    uint32_t trip = blockIdx.x * kWavesPerBlock + wave;
    if (trip < n_trips) {
        uint32_t t = tile_of(trip);
        static_assert(!kHead8 || kFirst <= 8, "the head plane holds 8 symbols per candidate");
        // the state the first look runs on, and its table row pitch in its words
        using Look = typename std::conditional<std::is_same<State, LevState<1>>::value, Lev32State,
                                               typename std::conditional<std::is_same<State, OsaState<1>>::value, Osa32State, State>::type>::type;
        constexpr int kLookPitch = kWide ? 2 : 1;
        const uint8_t* base = kHead8 ? p.heads8 + (size_t)lane * sizeof(uint2) : p.data + (size_t)lane * sizeof(uint4);
        const uint32_t row_pitch = kHead8 ? (p.exp_flags & 1u ? 0u : (uint32_t)(kWave * sizeof(uint2))) : p.uniform_tile_bytes;  // (exp_flags bit 0: RF_EXP_NOHBM)
        auto load_row = [&](uint32_t tile) -> uint4 {
            if constexpr (kHead8) {
                typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                const v2u v = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(base + (uint64_t)tile * row_pitch));
                return make_uint4(v.x, v.y, 0u, 0u);
            } else {
                return load_chunk(reinterpret_cast<const uint4*>(base + (uint64_t)tile * row_pitch));
            }
        };
        uint4 cur = load_row(t);
        // The head plane's rows are 512 B per wavefront: with ONE row in flight per wavefront the chip has 8 x 4 x 256 x 512 B = 4 MB
        // on its way, which at ~1 us of loaded HBM latency is 4 TB/s -- exactly where the head-plane scans stood once the band
        // prefilter had made a dead tile cheap.  They keep TWO rows in flight (the next tile's and the one after).
        uint4 ahead_kept = kHead8 ? load_row(trip + trip_stride < n_trips ? tile_of(trip + trip_stride) : t) : cur;
        while (true) {
            const uint32_t trip_next = trip + trip_stride;
            const bool has_next = trip_next < n_trips;
            const uint32_t t_next = has_next ? tile_of(trip_next) : t;
            // the first chunk row of the next tile (past the last tile: a cached re-read of this one)
            uint4 ahead;
            if constexpr (kHead8) {
                ahead = ahead_kept;
                const uint32_t trip_next2 = trip_next + trip_stride;
                ahead_kept = load_row(has_next && trip_next2 < n_trips ? tile_of(trip_next2) : t);
            } else {
                ahead = load_row(t_next);
            }
            const uint32_t idx = t * kWave + lane;
            const bool valid = idx < p.n;
            State st;
            st.init();
            bool dead;
            if constexpr (kHead8) {
                dead = false;
                if (band_filter) dead = __ballot(band_hits(lds_band, cur.x, cur.y) >= p.head_need) == 0;
                if (!dead) {
                    Look lo;
                    lo.init();
                    process_chunk_full<Look, 0, kFirst, kLookPitch>(lo, reinterpret_cast<const typename Look::Word*>(lds_pm), cur);
                    dead = __ballot(may_pass(p, fin, lo.bound_first(len1, kFirst, len2))) == 0;
                }
                if (!dead) {  // rare: the whole first chunk row from the tile itself, and the full-width state from column 0
                    cur = load_chunk(reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes) + lane);
                    process_chunk_full<State, 0, kFirst>(st, lds_pm, cur);
                }
            } else if constexpr (kNarrowLook) {
                // The first look only asks for D[kFirst + len1 - len2][kFirst] (the diagonal bound), i.e. for the first <= 32 pattern
                // rows: the first columns of a 64-bit Levenshtein scan run on the LOW words of the table with the 32-bit recurrence
                // (10 instead of 16 VALU instructions per column, profiles/head_plane_r03.txt: a dead tile's cost is its instruction
                // count).  A surviving tile -- rare -- starts over on the full words.
                if (narrow) {
                    Look lo;
                    lo.init();
                    process_chunk_full<Look, 0, kFirst, 2>(lo, reinterpret_cast<const uint32_t*>(lds_pm), cur);
                    dead = __ballot(may_pass(p, fin, lo.bound_first(len1, kFirst, len2))) == 0;
                    if (!dead) process_chunk_full<State, 0, kFirst>(st, lds_pm, cur);
                } else {
                    process_chunk_full<State, 0, kFirst>(st, lds_pm, cur);
                    dead = __ballot(may_pass(p, fin, st.bound_first(len1, kFirst, len2))) == 0;
                }
            } else if constexpr (kFirst < 16) {
                process_chunk_full<State, 0, kFirst>(st, lds_pm, cur);
                dead = __ballot(may_pass(p, fin, st.bound_first(len1, kFirst, len2))) == 0;
            } else {
                process_chunk_full<State>(st, lds_pm, cur);
                dead = __ballot(may_pass(p, fin, st.bound(len1, kChunk, len2))) == 0;
            }
            if (dead) {
                if (p.out && valid && !p.run_orig) emit_none(p, idx);
            } else {
                // a tile with a lane still in the race: the general walk (looks at every chunk end, chunks fetched on demand)
                if constexpr (kFirst < 16) {
                    process_chunk_full<State, kFirst, kChunk>(st, lds_pm, cur);
                    dead = __ballot(may_pass(p, fin, st.bound(len1, kChunk, len2))) == 0;
                }
                const uint4* src = reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes);
                for (uint32_t c = 1; c < nch && !dead; ++c) {
                    const uint4 more = load_chunk(src + (size_t)c * kWave + lane);
                    const uint32_t cols = len2 - c * kChunk;
                    if (cols >= (uint32_t)kChunk)
                        process_chunk_full<State>(st, lds_pm, more);
                    else
                        process_chunk_tail<State>(st, lds_pm, more, cols);
                    const uint32_t j = min(len2, (c + 1) * kChunk);
                    dead = __ballot(may_pass(p, fin, st.bound(len1, j, len2))) == 0;
                }
                const uint32_t raw = st.result(len1, len2);
                // where this lane's result goes: its index -- or, for a length run of a bucketed corpus, the candidate's original
                // index (read here, in the rare surviving tile, not per tile)
                uint32_t oi = idx;
                bool real = valid;
                if (p.run_orig) {
                    oi = p.run_orig[idx];
                    real = oi != kPad;
                }
                if (p.out && real) {
                    if (dead) {
                        if (!p.run_orig) emit_none(p, oi);
                    } else {
                        emit_fin(p, fin, raw, oi, p.out);
                    }
                }
                if (topk && !dead) {
                    bool keep;
                    const uint32_t v = usize_value(p, raw, len2, &keep, len1);
                    const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + oi);
                    if ((tiles_done++ & 7u) == 0) topk_refresh_bound(p, limit);
                    if (best.offer(mine, real && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);
                }
            }
            if (!has_next) break;
            trip = trip_next;
            t = t_next;
            cur = ahead;
        }
    }
    if (topk) topk_block_publish(p, best, lds_topk, wave, lane, limit);
}
template <class State, int kFirst>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void early_lean_kernel(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    early_lean_body<State, kFirst>(p, lds_pm, lds_topk);
}
template <class State, int kFirst>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void early_head8_kernel(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    __shared__ uint8_t lds_band[256];
    early_lean_body<State, kFirst, true>(p, lds_pm, lds_topk, lds_band);
}

template <class State, bool kUniform, int kFirst>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void early_kernel(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    scan_body<State, kUniform, kFirst>(p, lds_pm, lds_topk);
}
template <class State, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void scan_kernel_occ8(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    scan_body<State, kUniform>(p, lds_pm, lds_topk);
}

// ---------------------------------------------------------------------------------------------------
// The no-cutoff scan of the single-word states as a stream: every wavefront sees its tiles as ONE sequence of
// 16-column chunks and keeps kDepth chunk loads (1 KiB each) in flight ahead of the chunk it is working on, across
// tile boundaries.  A FETCH cursor runs kDepth chunks ahead of the PROCESS cursor; both walk (tile, chunk) pairs, and
// a zero-length tile counts as one (unused) chunk so the two stay in lock-step.  Past the wavefront's last tile the
// fetch cursor parks on its last valid chunk (a cached re-read).  Compared with scan_body (which also carries the
// cutoff early-out) this loop has about half the scalar/branch instructions per chunk.
// ---------------------------------------------------------------------------------------------------
template <class State, bool kUniform, int kDepth>
__device__ __forceinline__ void stream_body(const ScanParams& p, typename State::Word* lds_pm, uint64_t (*lds_topk)[kWave])
{
    constexpr int W = State::kWords;
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock)
        lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (typename State::Word)p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    WaveTopK best;
    best.init();
    // offers are filtered by `limit` = min(launch-wide pruning bound as last seen, own list's worst key), a scalar;
    // `bound` is the bound fetch in flight (topk_refresh_bound)
    // (the first value is waited for once: it is the sampled bound, and without it the first tile of EVERY wavefront
    // would insert 64 keys and then hit the one bound word with an atomic -- 32768 serialized device-scope atomics)
    uint64_t limit = ~0ull;
    if (topk) topk_refresh_bound(p, limit);
    uint32_t tiles_done = 0;

    uint32_t t = p.tile_begin + (dealt_workgroup(p) * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        // fetch cursor
        uint32_t ft = t, fc = 0;
        TileView fv = load_tile<kUniform>(p, ft);
        uint32_t fn = max(1u, (fv.len + kChunk - 1) / kChunk);
        auto fetch = [&]() {
            const uint4 v = load_chunk(fv.src + (size_t)fc * kWave + lane);
            if (++fc == fn) {
                const uint32_t nt = ft + stride;
                if (nt < p.tile_end) {
                    ft = nt;
                    fv = load_tile<kUniform>(p, ft);
                    fn = max(1u, (fv.len + kChunk - 1) / kChunk);
                    fc = 0;
                } else {
                    fc = fn - 1;
                }
            }
            return v;
        };
        // Ring of kDepth + 1 chunk buffers with STATIC names: the loop below is unrolled over the ring phase, so
        // no buffer is ever copied (a copy of a register with a load in flight would force a wait for that load)
        // and the compiler's vmcnt bookkeeping stays exact.
        uint4 buf[kDepth + 1];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) buf[d] = fetch();

        // process cursor
        TileView cur_tile = load_tile<kUniform>(p, t);
        uint32_t c = 0;
        // (gather path: `orig` is the identity, 4 bytes per slot the HBM-bound scans can leave unread -- ScanParams::slot_store)
        const bool by_slot = !kUniform && p.slot_store && !topk;
        uint32_t idx = cur_tile.slot0 + lane;
        if (!kUniform && !by_slot) idx = p.orig[idx];
        State st;
        st.init();
        bool done = false;
        auto step = [&](const uint4& use, uint4& refill) {
            refill = fetch();
            const uint32_t len2 = cur_tile.len;
            const uint32_t nch = (len2 + kChunk - 1) / kChunk;
            const uint32_t cols = len2 - c * kChunk;
            if (cols >= kChunk)
                process_chunk_full<State>(st, lds_pm, use);
            else if (nch)
                process_chunk_tail<State>(st, lds_pm, use, cols);
            if (++c < max(1u, nch)) return;

            // tile finished
            const uint32_t slot = cur_tile.slot0 + lane;
            const bool valid = kUniform ? slot < p.n : idx != kPad;
            const uint32_t raw = st.result(p.len1, len2);
            if (p.out && valid) emit_usize(p, raw, len2, idx);
            if (topk) {
                bool keep;
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);
                // every 8th tile: the bound word is ONE line that every wavefront of the chip reads at agent scope, i.e.
                // from the memory side (the XCDs' L2s are not coherent) -- per tile that was 1.5 M requests per launch
                // on one channel and cost the Indel scan +27 % (rocprofv3: 1.32 -> 1.68 ms); the sampled bound is tight
                // from the start, so a staler copy loses nothing measurable
                if ((++tiles_done & 7u) == 0) topk_refresh_bound(p, limit);
            }
            t += stride;
            if (t >= p.tile_end) {
                done = true;
                return;
            }
            cur_tile = load_tile<kUniform>(p, t);
            c = 0;
            idx = cur_tile.slot0 + lane;
            if (!kUniform && !by_slot) idx = p.orig[idx];
            st.init();
        };
        while (!done) {
#pragma unroll
            for (int ph = 0; ph <= kDepth; ++ph) {
                step(buf[ph], buf[(ph + kDepth) % (kDepth + 1)]);
                if (done) break;
            }
        }
    }

    if (topk) topk_block_publish(p, best, lds_topk, wave, lane, limit);
}
template <class State, bool kUniform, int kDepth>
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void stream_kernel_occ8(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256 * State::kWords];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    stream_body<State, kUniform, kDepth>(p, lds_pm, lds_topk);
}

// ---------------------------------------------------------------------------------------------------
// Many queries x one corpus: Q pattern-match tables sit side by side in LDS and every 16-column
// chunk a wavefront loads from HBM is run through Q recurrences before the next chunk is touched, so the candidate
// bytes are read ONCE for Q queries -- the arithmetic intensity per HBM byte rises Q-fold, which is what the
// HBM-bound kernels (LCS / Indel / the 32-bit forms) need.  out is [Q][n], original candidate order per query.
// ---------------------------------------------------------------------------------------------------
template <class State, int Q, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void scan_multi_kernel(const ScanParams p)
{
    using Word = typename State::Word;
    static_assert(State::kWords == 1, "multi-query kernels are single-word");
    __shared__ Word lds_pm[Q][256];
    for (int i = threadIdx.x; i < Q * 256; i += kWave * kWavesPerBlock) {
        const int q = i / 256, c = i % 256;
        lds_pm[q][p.sigma[c]] = (Word)p.multi_pm[q][c];  // single-word tables: row stride 1
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    uint32_t t = dealt_workgroup(p) * kWavesPerBlock + wave;
    if (t >= p.n_tiles) return;
    TileView cur_tile = load_tile<kUniform>(p, t);
    uint4 cur = load_chunk(cur_tile.src + lane);

    while (true) {
        const uint32_t t_next = t + stride;
        const bool has_next = t_next < p.n_tiles;
        const TileView next_tile = load_tile<kUniform>(p, has_next ? t_next : t);
        const uint32_t len2 = cur_tile.len;
        const uint32_t slot = cur_tile.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];

        State st[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) st[q].init();
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        for (uint32_t c = 0; c < nch; ++c) {
            const uint4* nsrc = (c + 1 < nch) ? cur_tile.src + (size_t)(c + 1) * kWave : next_tile.src;
            const uint4 nxt = load_chunk(nsrc + lane);
            const uint32_t cols = len2 - c * kChunk;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (cols >= kChunk)
                    process_chunk_full<State>(st[q], lds_pm[q], cur);
                else
                    process_chunk_tail<State>(st[q], lds_pm[q], cur, cols);
            }
            cur = nxt;
        }
        if (nch == 0) cur = load_chunk(next_tile.src + lane);

        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t raw = st[q].result(p.multi_len1[q], len2);
                char* out = reinterpret_cast<char*>(p.out) + (size_t)q * p.n * (p.out_f64 ? sizeof(double) : sizeof(uint32_t));
                emit_usize(p, raw, len2, idx, out, p.multi_len1[q]);
            }
        }
        if (!has_next) break;
        t = t_next;
        cur_tile = next_tile;
    }
}

template <class State, int Q>
static hipError_t launch_multi_q(const ScanParams& p, hipStream_t stream, int grid)
{
    const dim3 g(grid), b(kWave * kWavesPerBlock);
    if (p.tiles)
        hipLaunchKernelGGL((scan_multi_kernel<State, Q, false>), g, b, 0, stream, p);
    else
        hipLaunchKernelGGL((scan_multi_kernel<State, Q, true>), g, b, 0, stream, p);
    return hipGetLastError();
}
template <class State>
static hipError_t launch_multi_state(const ScanParams& p, hipStream_t stream, int grid)
{
    switch (p.multi_q) {
    case 2: return launch_multi_q<State, 2>(p, stream, grid);
    case 4: return launch_multi_q<State, 4>(p, stream, grid);
    default: return hipErrorInvalidValue;
    }
}
// raw: RAW_LEV or RAW_LCS; all queries single-word; `narrow` = every query <= 32 symbols
hipError_t launch_scan_multi(RawKind raw, bool narrow, const ScanParams& p, hipStream_t stream)
{
    if (p.n_tiles == 0) return hipSuccess;
    const int grid = scan_grid_full(p.n_tiles);
    if (raw == RAW_LEV) return narrow ? launch_multi_state<Lev32State>(p, stream, grid) : launch_multi_state<LevState<1>>(p, stream, grid);
    if (raw == RAW_LCS) return narrow ? launch_multi_state<Lcs32State>(p, stream, grid) : launch_multi_state<LcsState<1>>(p, stream, grid);
    return hipErrorInvalidValue;
}

// Stand-alone selection (the post-all-gather merge, rf_topk_merge_keys_device): `count` keys -> the k smallest, ascending,
// ~0 = empty.  The scans select inside their last workgroup (topk_block_publish) with the same topk_select().
constexpr int kFinalThreads = kWave * kWavesPerBlock;
__global__ __launch_bounds__(kFinalThreads) void topk_final_kernel(const uint64_t* __restrict__ keys, uint32_t count, uint32_t k, uint64_t* __restrict__ out)
{
    __shared__ uint64_t lists[kWavesPerBlock][kWave];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = uniform(threadIdx.x / kWave);
    WaveTopK best;
    topk_select(keys, count, k, lists, wave, lane, best);
    if (wave == 0 && lane < k) out[lane] = best.key;
}

hipError_t launch_topk_final(const uint64_t* keys, uint32_t count, uint32_t k, uint64_t* out, hipStream_t stream)
{
    hipLaunchKernelGGL(topk_final_kernel, dim3(1), dim3(kFinalThreads), 0, stream, keys, count, k, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------------
// Workgroups per launch: 8 (32 wavefronts) are resident per CU; launching a multiple of that lets early finishers be
// replaced.  Two factors, measured on the C2 / C3 corpora (same-session sweeps):
//   * full scans (no cutoff): the more the better up to ~256-512 per CU -- Levenshtein +1.2 %, Indel +4 % over 32 -- the
//     static grid-stride deal leaves a tail when wavefronts own dozens of tiles each;
//   * short-running launches (cutoff early-out, the band kernel, the completeness paths): 32 per CU -- every workgroup
//     stages the pattern table before its first tile, and with 6 tiles per wavefront that prologue shows (the band kernel
//     lost 37 % at 256 per CU).
// RF_SCAN_BLOCKS_PER_CU / RF_SCAN_BLOCKS_PER_CU_FULL override.  The CU count is the current device's (256 on a whole
// MI355X, 32 on a CPX partition), cached per device.
static int device_cus()
{
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
static int env_or(const char* name, int dflt)
{
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : dflt;
}
int scan_max_grid()
{
    static const int per_cu = env_or("RF_SCAN_BLOCKS_PER_CU", 32);
    return device_cus() * per_cu;
}
int scan_max_grid_full()
{
    static const int per_cu = env_or("RF_SCAN_BLOCKS_PER_CU_FULL", 256);
    return device_cus() * per_cu;
}

// the A/B switches of the cutoff scans (read once per process)
static bool env_on(const char* name)
{
    const char* e = getenv(name);
    return !e || atoi(e) != 0;
}
static bool sw_early_static() { static const bool v = env_on("RF_EARLY_STATIC"); return v; }
static bool sw_early_lean() { static const bool v = env_on("RF_EARLY_LEAN"); return v; }          // early_lean_kernel
static bool sw_narrow_look() { static const bool v = env_on("RF_NARROW_LOOK"); return v; }        // the 32-bit first look of early_lean_kernel
static bool sw_head_two_pass() { static const bool v = env_on("RF_HEAD_TWO_PASS"); return v; }    // head_filter_kernel + a second pass
static bool sw_head_look_pass() { static const bool v = env_on("RF_HEAD_LOOK_PASS"); return v; }  // ... also where the band filter does not apply
static bool sw_lane_compact() { static const bool v = env_on("RF_LANE_COMPACT"); return v; }      // the second pass over surviving LANES (rf_sparse.hip) instead of surviving tiles

// Will launch_scan run this launch as head_filter_kernel + a second pass?  (The conditions of RF_EARLY_CASE below, for callers that must know before they
// launch: rf_filter_* hands the second pass a compact result instead of a dense vector.)
bool head_two_pass_applies(RawKind raw, const ScanParams& p)
{
    if ((raw != RAW_LEV && raw != RAW_OSA) || p.words != 1 || !p.early || p.band || p.long_words_pad || p.tile_step != 1) return false;
    if (p.mixed && p.mixed_end > p.mixed_begin) return false;
    if (!sw_early_static() || !sw_early_lean() || p.tiles || p.uniform_len < (uint32_t)kChunk || !p.heads8) return false;
    if (p.first_check < 4 || p.first_check > 8 || (p.first_check & 1u)) return false;
    const bool lev32 = raw == RAW_LEV && p.len1 <= 32;
    const int32_t look_row = (int32_t)p.first_check + (int32_t)p.len1 - (int32_t)p.uniform_len;
    if (!lev32 && (look_row < 1 || look_row > 32)) return false;
    if (!(p.head_need ? sw_head_two_pass() : sw_head_look_pass())) return false;
    return p.tile_list_buf != nullptr && p.tile_end > p.tile_begin;
}

template <class State>
static hipError_t launch_state(const ScanParams& p, hipStream_t stream, int grid)
{
    const dim3 g(grid), b(kWave * kWavesPerBlock);
    if constexpr (State::kWords == 1) {
        // no cutoff early-out to serve: the leaner stream loop.  Ring depth: 1 for the issue-bound recurrences (2 and 3 measured
        // 1-3 % slower), 2 for the LCS states, whose 5-instruction column makes a chunk short enough for a second load in flight to
        // pay (Indel 80.8 -> 81.6 Gpairs/s, depth 3: 79.6); RF_STREAM=0 selects scan_body for A/B.
        constexpr int kDepth = (std::is_same<State, LcsState<1>>::value || std::is_same<State, Lcs32State>::value) ? 2 : 1;
        static const bool use_stream = [] { const char* e = getenv("RF_STREAM"); return !e || atoi(e) != 0; }();
        if (!p.early && use_stream) {
            // the headline case has its chunk in hand-scheduled asm (rf_lev_asm.hip); RF_ASM_CHUNK=0 selects the compiled loop
            static const bool use_asm = [] { const char* e = getenv("RF_ASM_CHUNK"); return !e || atoi(e) != 0; }();
            // per-candidate u32 results of the Levenshtein / OSA states: the whole-kernel asm scans (rf_stream_asm.hip), whatever the
            // corpus looks like.  Zero-length tiles have no column to run: the compiled loop fills those runs in.  RF_ASM_STREAM=0
            // keeps round 2's fixed-shape asm kernels + the compiled loop for A/B.
            static const bool use_stream_asm = [] { const char* e = getenv("RF_ASM_STREAM"); return !e || atoi(e) != 0; }();
            constexpr int kAsmKind = std::is_same<State, LevState<1>>::value ? 0 : (std::is_same<State, Lev32State>::value ? 1 : (std::is_same<State, OsaState<1>>::value ? 2 : -1));
            // the single-word LCS scans of a single-length corpus that keeps its payload at 6 bits per symbol too (ScanParams::data6): the asm scan over that
            // (whole chunks only: lengths that are multiples of 16; u32 results).  RF_PACK6=0 (no such payload) is the A/B switch.
            if ((std::is_same<State, LcsState<1>>::value || std::is_same<State, Lcs32State>::value) && use_asm && use_stream_asm && p.data6 && !p.tiles &&
                stream_asm_serves(p))
                return launch_stream_asm(std::is_same<State, Lcs32State>::value ? 6 : 5, p, stream, std::max(1, scan_grid_full(p.tile_end - p.tile_begin)));
            // ... and the same scans over the 6-bit payload of a length-bucketed corpus (tiles; u32 results)
            constexpr bool kLcs1 = std::is_same<State, LcsState<1>>::value || std::is_same<State, Lcs32State>::value;
            const int asm_kind = kAsmKind >= 0 ? kAsmKind : ((kLcs1 && p.data6 && p.tiles) ? (std::is_same<State, Lcs32State>::value ? 6 : 5) : -1);
            if (asm_kind >= 0 && use_asm && use_stream_asm && stream_asm_serves(p)) {
                uint32_t at = p.tile_begin;
                for (int r = 0; r <= 2 && at < p.tile_end; ++r) {
                    const uint32_t zb = r < 2 ? std::min(std::max(p.zero_begin[r], at), p.tile_end) : p.tile_end;
                    const uint32_t ze = r < 2 ? std::min(std::max(p.zero_end[r], zb), p.tile_end) : p.tile_end;
                    if (r < 2 && p.zero_end[r] <= p.zero_begin[r]) continue;  // no such run
                    if (zb > at) {
                        ScanParams q = p;
                        q.tile_begin = at, q.tile_end = zb;
                        const hipError_t e = launch_stream_asm(asm_kind, q, stream, std::max(1, scan_grid_full(zb - at)));
                        if (e != hipSuccess) return e;
                    }
                    if (ze > zb) {
                        ScanParams q = p;
                        q.tile_begin = zb, q.tile_end = ze;
                        hipLaunchKernelGGL((stream_kernel_occ8<State, false, kDepth>), dim3(std::max(1, scan_grid(ze - zb))), b, 0, stream, q);
                    }
                    at = ze;
                }
                return hipGetLastError();
            }
            if (std::is_same<State, LevState<1>>::value && !p.tiles && use_asm && p.uniform_len >= (uint32_t)kChunk && p.uniform_len % kChunk == 0)
                return launch_lev1_asm(p, stream, grid);
            if (std::is_same<State, Lev32State>::value && !p.tiles && use_asm && p.uniform_len >= (uint32_t)kChunk && p.uniform_len % kChunk == 0)
                return launch_lev32_asm(p, stream, grid);
            if (std::is_same<State, OsaState<1>>::value && !p.tiles && use_asm && p.uniform_len >= (uint32_t)kChunk && p.uniform_len % kChunk == 0)
                return launch_osa1_asm(p, stream, grid);
            if (p.tiles)
                hipLaunchKernelGGL((stream_kernel_occ8<State, false, kDepth>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((stream_kernel_occ8<State, true, kDepth>), g, b, 0, stream, p);
            return hipGetLastError();
        }
        if (p.early) {
            // cutoff runs: the compiler's own register budget.  Pinned to 8 wavefronts per SIMD this body spills (32 B of
            // scratch traffic per tile in a loop that only runs 8 columns); 7 resident wavefronts without spills
            // measured +8 % at cutoff 3 and +3 % at cutoff 10 on the C2 corpus.
            // The single-word states get the look's column (and `early` itself) as compile-time facts: the run-time dispatch
            // and flags cost the short per-tile loop of a cutoff scan 83 scalar instructions and 34 branches per tile -- with
            // four SIMDs sharing one scalar unit about as much time as its 139 vector instructions (cutoff 3: 217 -> 237
            // Gpairs/s, top-16 237 -> 263).  RF_EARLY_STATIC=0 selects the run-time form for A/B.
            const bool early_static = sw_early_static(), lean = sw_early_lean(), narrow_look = sw_narrow_look();
            ScanParams pn = p;
            pn.narrow_look = narrow_look ? 1u : 0u;
            const bool two_pass = sw_head_two_pass(), look_pass = sw_head_look_pass();
#ifdef RF_EXPERIMENTS  // measurement builds only (tools/build_stream_variant.sh): the shipping library has no switch that changes a result
            static const bool exp_nohbm = getenv("RF_EXP_NOHBM") != nullptr;  // every tile reads tile 0's head row (results are wrong on purpose)
            pn.exp_flags = exp_nohbm ? 1u : 0u;
#else
            pn.exp_flags = 0u;
#endif
            if constexpr (std::is_same<State, LevState<1>>::value || std::is_same<State, Lev32State>::value || std::is_same<State, OsaState<1>>::value ||
                          std::is_same<State, LcsState<1>>::value || std::is_same<State, Lcs32State>::value) {
                if (early_static) {
#define RF_EARLY_CASE(J)                                                                   \
    case J:                                                                                \
        if (p.tiles)                                                                       \
            hipLaunchKernelGGL((early_kernel<State, false, J>), g, b, 0, stream, p);       \
        else if (lean && p.uniform_len >= (uint32_t)kChunk) {                              \
            if constexpr (J <= 8 && (std::is_same<State, LevState<1>>::value || std::is_same<State, Lev32State>::value || std::is_same<State, OsaState<1>>::value)) { \
                const int32_t look_row = J + (int32_t)p.len1 - (int32_t)p.uniform_len;      \
                if (p.heads8 && (std::is_same<State, Lev32State>::value || (look_row >= 1 && look_row <= 32))) { \
                    if ((p.head_need ? two_pass : look_pass) && p.tile_step == 1 && p.tile_list_buf && p.tile_end > p.tile_begin) { \
                        /* band prefilter and first look as a streaming pass of its own, then the cutoff scan over the tiles it left */ \
                        const uint32_t pairs = (p.tile_end - p.tile_begin + 1) / 2;        \
                        const uint32_t fgrid = std::min<uint32_t>((pairs + kWavesPerBlock - 1) / kWavesPerBlock, std::min<uint32_t>((uint32_t)device_cus() * 16u, 4096u)); /* G <= 16 K: rf_api_scan.hip sizes the list buffer for that */ \
                        const uint32_t G = fgrid * kWavesPerBlock, cap = 2 * ((pairs + G - 1) / G); \
                        if (pn.lane_list && sw_lane_compact()) {                          \
                            /* round 6: the pass leaves LANE masks with its surviving tiles, the second pass walks the surviving lanes 64 to a wavefront (rf_sparse.hip) */ \
                            const uint32_t cap4 = cap;                                     \
                            if (pn.heads6 && (p.tile_begin & 1u) == 0)                     \
                                hipLaunchKernelGGL((head_filter_kernel<State, J, true, true>), dim3(fgrid), b, 0, stream, pn, p.tile_list_buf, cap4); \
                            else                                                           \
                                hipLaunchKernelGGL((head_filter_kernel<State, J, false, true>), dim3(fgrid), b, 0, stream, pn, p.tile_list_buf, cap4); \
                            uint32_t* packed_at = p.tile_list_buf + 4 + 2 * (size_t)G + 4 * (size_t)G * cap4; \
                            uint32_t* first_at = packed_at + 4 * (size_t)(2 * pairs + 2);   \
                            hipLaunchKernelGGL(lane_list_pack_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, p.tile_list_buf, G, cap4, first_at); \
                            ScanParams p2 = pn;                                            \
                            p2.heads8 = nullptr;                                           \
                            p2.tile_list = packed_at;                                      \
                            p2.lane_first = first_at;                                      \
                            p2.tile_list_count = p.tile_list_buf;                          \
                            hipError_t el = hipGetLastError();                             \
                            if (el == hipSuccess) el = launch_sparse_lean(std::is_same<State, Lev32State>::value ? 1 : (std::is_same<State, OsaState<1>>::value ? 2 : 0), p2, stream); \
                            return el;                                                     \
                        }                                                                  \
                        if (pn.heads6 && (p.tile_begin & 1u) == 0)                         \
                            hipLaunchKernelGGL((head_filter_kernel<State, J, true>), dim3(fgrid), b, 0, stream, pn, p.tile_list_buf, cap); \
                        else                                                               \
                            hipLaunchKernelGGL((head_filter_kernel<State, J, false>), dim3(fgrid), b, 0, stream, pn, p.tile_list_buf, cap); \
                        hipLaunchKernelGGL(tile_list_pack_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, p.tile_list_buf, G, cap); \
                        ScanParams p2 = pn;                                                \
                        p2.heads8 = nullptr;                                               \
                        p2.tile_list = p.tile_list_buf + 1 + 2 * (size_t)G + (size_t)G * cap; \
                        p2.tile_list_count = p.tile_list_buf;                              \
                        hipLaunchKernelGGL((early_lean_kernel<State, J>), dim3((uint32_t)device_cus() * 8u), b, 0, stream, p2); \
                        return hipGetLastError();                                          \
                    }                                                                      \
                    hipLaunchKernelGGL((early_head8_kernel<State, J>), g, b, 0, stream, pn); \
                    return hipGetLastError();                                              \
                }                                                                          \
            }                                                                              \
            hipLaunchKernelGGL((early_lean_kernel<State, J>), g, b, 0, stream, pn);        \
        }                                                                                  \
        else                                                                               \
            hipLaunchKernelGGL((early_kernel<State, true, J>), g, b, 0, stream, p);        \
        return hipGetLastError();
                    switch (p.first_check) {
                        RF_EARLY_CASE(4)
                        RF_EARLY_CASE(6)
                        RF_EARLY_CASE(8)
                        RF_EARLY_CASE(10)
                        RF_EARLY_CASE(12)
                        RF_EARLY_CASE(14)
                        RF_EARLY_CASE(16)
                    default: break;
                    }
#undef RF_EARLY_CASE
                }
            }
            if (p.tiles)
                hipLaunchKernelGGL((scan_kernel<State, false>), g, b, 0, stream, p);
            else
                hipLaunchKernelGGL((scan_kernel<State, true>), g, b, 0, stream, p);
            return hipGetLastError();
        }
        if (p.tiles)
            hipLaunchKernelGGL((scan_kernel_occ8<State, false>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((scan_kernel_occ8<State, true>), g, b, 0, stream, p);
    } else {
        // queries of 65 .. 512 symbols, one u32 per candidate, no cutoff: the multi-word asm scans (rf_stream_asm.hip, word planes in
        // LDS); zero-length tiles go to the compiled kernel like above.  RF_ASM_BLOCK=0 keeps the compiled kernel for A/B.
        static const bool use_block_asm = [] { const char* e = getenv("RF_ASM_BLOCK"); return !e || atoi(e) != 0; }();
        constexpr bool kLevW = std::is_same<State, LevState<State::kWords>>::value && State::kWords >= 2 && State::kWords <= 8;  // (round 5: 5 .. 8 words too)
        if (kLevW && use_block_asm && stream_asm_serves(p)) {
            uint32_t at = p.tile_begin;
            for (int r = 0; r <= 2 && at < p.tile_end; ++r) {
                const uint32_t zb = r < 2 ? std::min(std::max(p.zero_begin[r], at), p.tile_end) : p.tile_end;
                const uint32_t ze = r < 2 ? std::min(std::max(p.zero_end[r], zb), p.tile_end) : p.tile_end;
                if (r < 2 && p.zero_end[r] <= p.zero_begin[r]) continue;  // no such run
                if (zb > at) {
                    ScanParams q = p;
                    q.tile_begin = at, q.tile_end = zb;
                    const hipError_t e = launch_stream_asm(3, q, stream, std::max(1, scan_grid_full(zb - at)));
                    if (e != hipSuccess) return e;
                }
                if (ze > zb) {
                    ScanParams q = p;
                    q.tile_begin = zb, q.tile_end = ze;
                    hipLaunchKernelGGL((scan_kernel<State, false>), dim3(std::max(1, scan_grid(ze - zb))), b, 0, stream, q);
                }
                at = ze;
            }
            return hipGetLastError();
        }
        if (p.tiles)
            hipLaunchKernelGGL((scan_kernel<State, false>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((scan_kernel<State, true>), g, b, 0, stream, p);
    }
    return hipGetLastError();
}

template <template <int> class StateT>
static hipError_t launch_words(const ScanParams& p, hipStream_t stream, int grid)
{
    switch (p.words) {
    case 1: return launch_state<StateT<1>>(p, stream, grid);
    case 2: return launch_state<StateT<2>>(p, stream, grid);
    case 3: return launch_state<StateT<3>>(p, stream, grid);
    case 4: return launch_state<StateT<4>>(p, stream, grid);
    case 5: return launch_state<StateT<5>>(p, stream, grid);
    case 6: return launch_state<StateT<6>>(p, stream, grid);
    case 7: return launch_state<StateT<7>>(p, stream, grid);
    case 8: return launch_state<StateT<8>>(p, stream, grid);
    default: return hipErrorInvalidValue;
    }
}

int scan_grid(uint32_t n_tiles)
{
    return (int)std::min<uint32_t>((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (uint32_t)scan_max_grid());
}
// Full (no-cutoff) scans of the register-resident kernels; also the capacity of the top-k scratch.  A wavefront walks tiles t, t + stride,
// ...: what matters is HOW MANY.  One or two tiles per wavefront and the launch is made of workgroup prologues (pattern table into the
// LDS behind two dependent global loads and a barrier, a cold first chunk): 20 M ragged candidates ran at 46.5 (Levenshtein) / 27.7
// (Jaro-Winkler) Gpairs/s on 256 workgroups per CU = 1.2 tiles per wavefront and run at 64.0 / 43.1 on 64 per CU.  Dozens of tiles per
// wavefront and the statically dealt wavefronts drift apart (tile lengths differ), the stores of one output window no longer meet in the
// L2 and the last partial round of resident workgroups weighs more: 100 M ragged Jaro-Winkler 47.6 (6 tiles per wavefront) -> 43.4 (24)
// -> 40.9 (95).  So: RF_SCAN_TILES_PER_WAVE (5) tiles per wavefront, at least 32 workgroups per CU where the corpus has them (3 M candidates
// on 8 per CU lost 6-9 % to the old one-tile-per-wavefront grid), at most
// RF_SCAN_BLOCKS_PER_CU_FULL (256) per CU, a multiple of 8 (the XCD deal).  profiles/grid_sweep_r04.txt
int scan_grid_full(uint32_t n_tiles)
{
    static const uint32_t per_wave = (uint32_t)env_or("RF_SCAN_TILES_PER_WAVE", 5);
    const uint32_t by_tiles = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;  // one tile per wavefront
    const uint32_t want = (n_tiles + kWavesPerBlock * per_wave - 1) / (kWavesPerBlock * per_wave);
    const uint32_t fill = std::min<uint32_t>(by_tiles, (uint32_t)device_cus() * 32u);  // (below ~10 M candidates: as many workgroups as there are)
    uint32_t g = std::min<uint32_t>(std::max(want, fill), (uint32_t)scan_max_grid_full());
    if (g >= 8u) g = std::min<uint32_t>((g + 7u) & ~7u, std::max<uint32_t>(8u, (uint32_t)scan_max_grid_full() & ~7u));
    return (int)std::max<uint32_t>(g, 1u);
}
hipError_t launch_scan(RawKind raw, const ScanParams& p_in, hipStream_t stream, int* grid_used)
{
    if (p_in.n_tiles == 0) return hipSuccess;
#ifdef RF_EXPERIMENTS  // measurement builds only: RF_EXP_TILE_BYTES=<bytes> walks a single-length corpus at another tile pitch (results are wrong on purpose:
                       // what would a scan gain if its payload were that much smaller, with its arithmetic unchanged?)
    ScanParams p_exp = p_in;
    static const uint32_t exp_tile_bytes = [] { const char* e = getenv("RF_EXP_TILE_BYTES"); return e ? (uint32_t)atoi(e) : 0u; }();
    if (exp_tile_bytes && !p_exp.tiles) p_exp.uniform_tile_bytes = exp_tile_bytes;
    const ScanParams& p = p_exp;
#else
    const ScanParams& p = p_in;
#endif
    // A corpus with a mixed section (leftovers of every length pooled into tiles with per-lane lengths) is scanned in two
    // launches by the register-resident Levenshtein / LCS / OSA kernels: the exact tiles as always, the mixed tiles by
    // scan_kernel_mixed.  Top-k, multi-query and the other kernel families walk the one-length views instead (p stays as is).
    // (band launches walk the one-length views instead: band_kernel has no per-lane-length form)
    if (p.mixed && p.mixed_end > p.mixed_begin && !p.topk_k && !p.long_words_pad && !p.band && p.words <= (uint32_t)kMaxWords &&
        (raw == RAW_LEV || raw == RAW_LCS || raw == RAW_OSA) && p.tile_step == 1) {
        ScanParams q = p;
        q.mixed = nullptr;
        q.tile_end = std::min(p.tile_end, p.n_exact);
        q.tile_begin = std::min(p.tile_begin, q.tile_end);
        // A small corpus costs two launches and the gap between them: up to RF_JOINT_MAX_TILES exact tiles the mixed kernel walks them
        // too, as tiles whose 64 lanes share one length.  configs[0] (query 32 x 10 k candidates): 12.5 -> 6.9 us per call; 500 k
        // candidates 18.0 -> 11.2; 1 M 30 -> 22; 2 M 38.6 -> 38.1 (beyond that the asm kernels' rate wins): profiles/joint_launch_r04.txt
        static const uint32_t joint_max = [] { const char* e = getenv("RF_JOINT_MAX_TILES"); return e ? (uint32_t)atoi(e) : 16384u; }();
        if (!p.prefill_none && p.tiles && p.orig && !p.run_orig && q.tile_end - q.tile_begin <= joint_max) {
            ScanParams m = p;
            m.joint_begin = q.tile_begin;
            m.joint_end = q.tile_end;
            m.tile_begin = p.mixed_begin;
            m.tile_end = p.mixed_end;
            if (grid_used) *grid_used = scan_grid(m.tile_end - m.tile_begin + m.joint_end - m.joint_begin);
            return launch_scan_mixed(raw, m, stream);
        }
        hipError_t e = hipSuccess;
        if (q.tile_end > q.tile_begin || p.prefill_none) e = launch_scan(raw, q, stream, grid_used);  // (also does the None pre-fill)
        if (e != hipSuccess) return e;
        ScanParams m = p;
        m.tile_begin = p.mixed_begin;
        m.tile_end = p.mixed_end;
        m.prefill_none = 0;
        return launch_scan_mixed(raw, m, stream);
    }
    const uint32_t launch_tiles = p.tile_end > p.tile_begin ? (p.tile_end - p.tile_begin + p.tile_step - 1) / p.tile_step : 0;
    const bool full_scan = !p.early && !p.band && !p.long_words_pad && p.tile_step == 1 && (raw == RAW_LEV || raw == RAW_LCS || raw == RAW_OSA);
    const int grid = p.long_words_pad ? (int)p.long_grid : std::max(1, full_scan ? scan_grid_full(launch_tiles) : scan_grid(launch_tiles));
    if (grid_used) *grid_used = grid;
    if (p.prefill_none && p.out) {  // candidates outside the cutoff's length window (plan()): None without being read
        // (for f64 outputs two all-ones words per entry: a NaN, which is all "None" promises)
        // (prefill_window: the reader looks at the window's own slots only -- 64 per tile, exact tiles and views alike)
        const size_t w = p.out_f64 ? 2 : 1, from = p.prefill_window ? (size_t)p.tile_begin * kWave : 0, count = p.prefill_window ? (size_t)(p.tile_end - p.tile_begin) * kWave : (size_t)p.n;
        const hipError_t e = count ? hipMemsetD32Async((hipDeviceptr_t)(reinterpret_cast<uint32_t*>(p.out) + from * w), (int)RF_NONE_U32, count * w, stream) : hipSuccess;
        if (e != hipSuccess) return e;
    }
    // long query, small distance cutoff: one word down the diagonal (rf_band.hip); exact tiles and one-length views alike
    if (p.band && raw == RAW_LEV && !p.topk_k) return launch_band(p, stream);
    if (p.long_words_pad && (raw == RAW_LEV || raw == RAW_LCS || raw == RAW_OSA)) return launch_long(raw, p, stream, grid);
    switch (raw) {
    case RAW_LEV: return p.len1 <= 32 ? launch_state<Lev32State>(p, stream, grid) : launch_words<LevState>(p, stream, grid);
    case RAW_LCS: return p.len1 <= 32 ? launch_state<Lcs32State>(p, stream, grid) : launch_words<LcsState>(p, stream, grid);
    case RAW_OSA: return launch_words<OsaState>(p, stream, grid);
    case RAW_WF: return launch_wf(p, stream);
    case RAW_JARO: return launch_jaro(p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rf
