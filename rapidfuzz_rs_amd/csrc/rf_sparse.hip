// rf_sparse.hip -- LANE COMPACTION of the head-plane cutoff scans (round 6; VERDICT r5 "make the cutoff path robust to survivors").
//
// reference: the cutoff only decides Some / None after the loops (levenshtein.rs:492-496, common.rs:43-45), so anything that proves a candidate beyond the cutoff
// without changing a value that is within it is allowed.  The head-plane scans (rf_scan.hip head_filter_kernel) prove that for nearly every candidate of a random
// corpus from its first 8 symbols; until round 5 what they handed on was a list of TILES with at least one lane left, and the second pass (early_lean_kernel) ran
// such a tile with all 64 lanes.  On corpora whose candidates share prefixes with the query -- URLs, names, SKUs: 2 % of the candidates carrying the query's first
// 8..12 symbols leave 1 - 0.98^64 = 73 % of the tiles alive -- that second pass read a 1 KiB chunk row and ran >= 16 columns on 64 lanes for one or two live ones.
//
// Now the first pass attaches a 64-bit LANE mask to every tile it lists (every other candidate gets its None there and then; a tile without survivors costs the
// pass exactly what it cost before), lane_list_pack_kernel (rf_scan.hip) packs the 16-byte entries and numbers the surviving candidates -- in index order, because
// the list is -- and sparse_lean_kernel walks DENSE tiles: wavefront lane l of dense tile j takes survivor 64 j + l -- a binary search over the entries' running
// sums for its tile, the n-th set bit of that tile's mask for its lane -- and reads its own candidate's chunk rows (one 16-byte load per lane and chunk;
// neighbouring survivors of one tile share cache lines).  (A first version kept a mask for EVERY tile and a hipcub sum over all of them: one more store per tile
// pair in the first pass's loop cost the random-corpus case 14 %, profiles/survivors_r06.txt.)  Columns run from 0 on the full-width state with a look at every chunk
// end (wavefront ballot over the dense tile), the next chunk fetched only by a tile that is still alive.  Results go where the caller wants them: out[candidate] (dense vector: the first pass has written
// every other entry), a length run's run_orig[], the in-scan top-k lists, or -- rf_filter_*, no dense vector at all -- lane_val / lane_idx at the survivor's number.
// Everything is exact for every input: a survivor is merely a candidate the first pass could not rule out.
#include <algorithm>
#include <type_traits>

#include "rf_device.hpp"

namespace rf {

namespace {

struct Source {
    const uint4* src;  // this lane's first chunk
    uint32_t idx;      // candidate index inside the launch's corpus (view): tile * 64 + lane
    bool have;         // a survivor sits in this dense lane
};

template <class State>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void sparse_lean_kernel(const ScanParams p)
{
    __shared__ typename State::Word lds_pm[256];
    __shared__ uint64_t lds_topk[kWavesPerBlock][kWave];
    const uint4* __restrict__ list = reinterpret_cast<const uint4*>(p.tile_list);  // (tile, lane mask lo, hi, survivors in front): lane_list_pack_kernel
    const uint32_t entries = uniform(p.tile_list_count[0]), total = uniform(p.tile_list_count[1]);
    const uint32_t n_dense = (total + kWave - 1) / kWave;
    const bool topk = p.topk_k != 0;
    // (the survivors' number is only known here: the grid is sized for many, and a workgroup without a dense tile leaves before it stages the table -- unless the
    // launch keeps top-k lists, whose selection counts the workgroups in)
    if (!topk && blockIdx.x * kWavesPerBlock >= n_dense) return;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds_pm[p.sigma[i]] = (typename State::Word)p.pm[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    WaveTopK best;
    best.init();
    uint64_t limit = ~0ull;
    uint32_t tiles_done = 0;

    const uint32_t len1 = p.len1, len2 = p.uniform_len;
    const uint32_t nch = (len2 + kChunk - 1) / kChunk;
    const TileFin fin = tile_fin(p, len1, len2);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    auto locate = [&](uint32_t j) {
        Source s;
        const uint32_t g = j * kWave + lane;
        s.have = g < total;
        const uint32_t gg = s.have ? g : total - 1;  // (idle lanes of the last dense tile shadow its last survivor: defined bytes, no result)
        // the 64 entries from first[j] on hold all 64 survivors of this dense tile (an entry holds at least one): one coalesced load, then the lane's own entry --
        // the last one whose running sum is <= its survivor number -- by a binary search across the LANES (the sums ascend with the lane)
        const uint32_t e0 = uniform(p.lane_first[j]);
        const uint4 ent = list[min(e0 + lane, entries - 1)];
        uint32_t lo = 0, hi = kWave;
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t wm = (uint32_t)__shfl((int)ent.w, (int)mid, kWave);
            if (wm <= gg)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t t = (uint32_t)__shfl((int)ent.x, (int)lo, kWave), mlo = (uint32_t)__shfl((int)ent.y, (int)lo, kWave), mhi = (uint32_t)__shfl((int)ent.z, (int)lo, kWave),
                       w = (uint32_t)__shfl((int)ent.w, (int)lo, kWave);
        const uint32_t ls = nth_set_bit(((uint64_t)mhi << 32) | mlo, gg - w);
        s.idx = t * kWave + ls;
        s.src = reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes) + ls;
        return s;
    };
    uint32_t j = blockIdx.x * kWavesPerBlock + wave;
    if (j < n_dense) {
        Source cur = locate(j);
        uint4 chunk = load_chunk(cur.src);
        while (true) {
            const uint32_t j_next = j + stride;
            const bool has_next = j_next < n_dense;
            // the next dense tile's sources and first chunks: requested before this tile's columns run
            Source nxt = cur;
            uint4 chunk_next = chunk;
            if (has_next) {
                nxt = locate(j_next);
                chunk_next = load_chunk(nxt.src);
            }
            State st;
            st.init();
            bool dead = false;
            for (uint32_t c = 0; c < nch; ++c) {
                // (the next chunk is fetched only once this one has not killed the tile: a lane's 16 bytes pull a whole 128-byte line out of a row it shares with no
                // other survivor, and nearly every survivor of a prefix-sharing corpus dies at its first chunk end -- fetching one chunk ahead doubled the traffic of a
                // pass that is bound by exactly that, 5 % prefix sharers: 332 us per 100 M, profiles/survivors5_cutoff3_sparse_r06.txt)
                const uint32_t cols = len2 - c * kChunk;
                if (cols >= (uint32_t)kChunk)
                    process_chunk_full<State>(st, lds_pm, chunk);
                else
                    process_chunk_tail<State>(st, lds_pm, chunk, cols);
                const uint32_t jj = min(len2, (c + 1) * kChunk);
                if (__ballot(cur.have && may_pass(p, fin, st.bound(len1, jj, len2))) == 0) {
                    dead = true;  // no lane of this dense tile can pass the cutoff any more
                    break;
                }
                if (c + 1 < nch) chunk = load_chunk(cur.src + (size_t)(c + 1) * kWave);
            }
            const uint32_t raw = st.result(len1, len2);
            // where this lane's result goes
            uint32_t oi = cur.idx;
            bool real = cur.have && cur.idx < p.n;
            if (p.run_orig && cur.have) {  // a length run of a bucketed corpus: the candidate's original index
                oi = p.run_orig[cur.idx];
                real = oi != kPad;
            }
            if (p.lane_val) {  // rf_filter_*: value (or None) and index at the survivor's own number
                const uint32_t g = j * kWave + lane;
                if (cur.have && g < p.lane_cap) {
                    p.lane_idx[g] = real ? oi : kPad;
                    if (dead || !real) {
                        if (!p.out_f64)
                            reinterpret_cast<uint32_t*>(p.lane_val)[g] = RF_NONE_U32;
                        else
                            reinterpret_cast<double*>(p.lane_val)[g] = __longlong_as_double(0x7FF8000000000000ll);
                    } else {
                        emit_fin(p, fin, raw, g, p.lane_val);
                    }
                }
            } else if (p.out && real) {
                if (dead) {
                    if (!p.run_orig) emit_none(p, oi);  // (a run's vector is pre-filled)
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
            if (!has_next) break;
            j = j_next;
            cur = nxt;
            chunk = chunk_next;
        }
    }
    if (topk) topk_block_publish(p, best, lds_topk, wave, lane, limit);
}

// The second pass of a score_hint scan of a single-length corpus (round 6; rf_api_scan.hip run_many_hinted): the candidates the band pass left unresolved -- listed by
// that pass itself, tile and lane mask (rf_band.hip band_list_kernel) -- run the caller's own multi-word scan 64 to a wavefront, each lane reading its own candidate's chunk
// rows.  No host in between: the number of dense tiles is read here.  (Round 5 marked the lanes in a pass of its own, summed them, brought the sums to the host to size
// dense tiles, copied the payload into them and scanned the copy: the copy moved what this kernel's loads move, once more.)
template <class State>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void sparse_words_kernel(const ScanParams p)
{
    constexpr int W = State::kWords;
    __shared__ typename State::Word lds_pm[256 * W];
    const uint4* __restrict__ list = reinterpret_cast<const uint4*>(p.tile_list);
    const uint32_t entries = uniform(p.tile_list_count[0]), total = uniform(p.tile_list_count[1]);
    const uint32_t n_dense = (total + kWave - 1) / kWave;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.band_report) {  // what the first pass left, for the host's bookkeeping of the hint (read there without waiting)
        __hip_atomic_store(p.band_report + 8, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.band_report + 9, p.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.band_report + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (blockIdx.x * kWavesPerBlock >= n_dense) return;
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock) lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (typename State::Word)p.pm[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t len1 = p.len1, len2 = p.uniform_len;
    const uint32_t nch = (len2 + kChunk - 1) / kChunk;
    const TileFin fin = tile_fin(p, len1, len2);
    const bool early = State::kCanPrune && p.early != 0;
    for (uint32_t j = blockIdx.x * kWavesPerBlock + wave; j < n_dense; j += gridDim.x * kWavesPerBlock) {
        const DenseLane dl = dense_lane_source(list, entries, total, p.lane_first, j, lane);
        const uint4* src = reinterpret_cast<const uint4*>(p.data + (uint64_t)dl.tile * p.uniform_tile_bytes) + dl.lane_in_tile;
        const uint32_t idx = dl.tile * kWave + dl.lane_in_tile;
        const bool valid = dl.have && idx < p.n;
        State st;
        st.init();
        bool dead = false;
        uint4 cur = nch ? load_chunk(src) : make_uint4(0, 0, 0, 0);
        for (uint32_t c = 0; c < nch; ++c) {
            uint4 nxt = cur;
            if (c + 1 < nch) nxt = load_chunk(src + (size_t)(c + 1) * kWave);
            const uint32_t cols = len2 - c * kChunk;
            if constexpr (has_band<State>::value) st.set_band(len1, len2, p.trim_k1 ? p.trim_k1 - 1u : 0xFFFFFFFFu, c);
            if (cols >= (uint32_t)kChunk)
                process_chunk_full<State>(st, lds_pm, cur);
            else
                process_chunk_tail<State>(st, lds_pm, cur, cols);
            if (early) {
                const uint32_t jj = min(len2, (c + 1) * kChunk);
                if (__ballot(valid && may_pass(p, fin, st.bound(len1, jj, len2))) == 0) {
                    dead = true;  // no lane of this dense tile can pass the caller's cutoff any more
                    break;
                }
            }
            cur = nxt;
        }
        const uint32_t raw = st.result(len1, len2);
        if (p.out && valid) {
            if (dead)
                emit_none(p, idx);
            else
                emit_fin(p, fin, raw, idx, p.out);
        }
    }
}

}  // namespace

hipError_t launch_sparse_lean(int state_kind, const ScanParams& p, hipStream_t stream)
{
    // (the survivors' number is only known on the device: a fixed grid of 8 workgroups per CU, like early_lean_kernel over its list)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const uint32_t tiles = p.tile_end > p.tile_begin ? p.tile_end - p.tile_begin : 1u;
    // (a wavefront per dense tile while they are few: a tile is a chain of dependent loads -- list entry, chunk rows -- and the workgroups beyond the survivors leave at once)
    // (a length run of a bucketed corpus is one of several launches of its call and a fraction of the corpus: the smaller grid, or the empty workgroups of eight runs add up)
    const uint32_t want = (uint32_t)cus * ((p.topk_k || p.run_orig) ? 8u : 32u), most = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 g(std::max(1u, std::min(want, most))), b(kWave * kWavesPerBlock);
    switch (state_kind) {
    case 0: hipLaunchKernelGGL((sparse_lean_kernel<LevState<1>>), g, b, 0, stream, p); break;
    case 1: hipLaunchKernelGGL((sparse_lean_kernel<Lev32State>), g, b, 0, stream, p); break;
    case 2: hipLaunchKernelGGL((sparse_lean_kernel<OsaState<1>>), g, b, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sparse_words(const ScanParams& p, hipStream_t stream)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const uint32_t tiles = p.tile_end > p.tile_begin ? p.tile_end - p.tile_begin : 1u;
    const dim3 g(std::max(1u, std::min((uint32_t)cus * 16u, (tiles + kWavesPerBlock - 1) / kWavesPerBlock))), b(kWave * kWavesPerBlock);
    switch (p.words) {
    case 2: hipLaunchKernelGGL((sparse_words_kernel<LevState<2>>), g, b, 0, stream, p); break;
    case 3: hipLaunchKernelGGL((sparse_words_kernel<LevState<3>>), g, b, 0, stream, p); break;
    case 4: hipLaunchKernelGGL((sparse_words_kernel<LevState<4>>), g, b, 0, stream, p); break;
    case 5: hipLaunchKernelGGL((sparse_words_kernel<LevState<5>>), g, b, 0, stream, p); break;
    case 6: hipLaunchKernelGGL((sparse_words_kernel<LevState<6>>), g, b, 0, stream, p); break;
    case 7: hipLaunchKernelGGL((sparse_words_kernel<LevState<7>>), g, b, 0, stream, p); break;
    case 8: hipLaunchKernelGGL((sparse_words_kernel<LevState<8>>), g, b, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rf
