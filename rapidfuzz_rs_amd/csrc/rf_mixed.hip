// rf_mixed.hip -- the Levenshtein / LCS / OSA scan over MIXED tiles: 64 leftover candidates of neighbouring lengths per
// tile, one per lane, each with its own length (rf_api.hip HostLayout).  The column loop runs to the longest lane of the
// tile; chunks that lie inside every lane's length run the ordinary unmasked chunk code, later chunks step a lane only
// while it still has symbols (a divergent `if`: the compiler turns it into an EXEC mask per column).  A finished lane's
// state stays frozen at its own last column, so its result is read after the loop like everybody else's.  Finishing
// arithmetic, cutoff early-out and the None encoding are those of scan_body, evaluated per lane.
//
// Why: round 1 padded every candidate length to whole 64-lane tiles, so a corpus of few, long, all-different-length
// candidates paid 64x in HBM bytes and in time.  With mixed tiles such a corpus scans at (nearly) the rate of a dense one.
#include "rf_device.hpp"

namespace rf {

template <class State>
__device__ __forceinline__ void process_chunk_masked(State& st, const typename State::Word* lds_pm, uint4 c, uint32_t j0, uint32_t my_len)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
#pragma unroll 4
    for (uint32_t b = 0; b < (uint32_t)kChunk; ++b) {
        Word x[W];
        load_pm<Word, W>(x, lds_pm, c.x & 0xFFu);  // (a lane past its end reads the row of its padding byte: unused)
        if (j0 + b < my_len) st.step(x);
        c.x = __builtin_amdgcn_alignbit(c.y, c.x, 8);
        c.y = __builtin_amdgcn_alignbit(c.z, c.y, 8);
        c.z = __builtin_amdgcn_alignbit(c.w, c.z, 8);
        c.w >>= 8;
    }
}

template <class State>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void scan_kernel_mixed(const ScanParams p)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    __shared__ Word lds_pm[256 * W];
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock) lds_pm[(uint32_t)p.sigma[i / W] * W + i % W] = (Word)p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const bool early = State::kCanPrune && p.early != 0;
    // JOINT launches (small corpora, launch_scan): the exact tiles [joint_begin, joint_end) come first, as tiles whose lanes all have
    // the same length, then the mixed tiles -- one launch instead of two where the launch IS the cost.
    const uint32_t n_joint = p.joint_end - p.joint_begin, n_virtual = n_joint + (p.tile_end - p.tile_begin);
    for (uint32_t v = blockIdx.x * kWavesPerBlock + wave; v < n_virtual; v += gridDim.x * kWavesPerBlock) {
        // the descriptor is wavefront-uniform and read-only: scalar loads through the constant address space
        typedef const __attribute__((address_space(4))) uint32_t* cptr;
        uint64_t off;
        uint32_t max_len, min_len, my_len, idx;
        if (v < n_joint) {
            cptr td = (cptr)(uintptr_t)(p.tiles + p.joint_begin + v);
            off = ((uint64_t)td[1] << 32) | td[0];
            max_len = min_len = my_len = td[2];
            idx = p.orig[td[3] + lane];
        } else {
            cptr md = (cptr)(uintptr_t)(p.mixed + p.tile_begin + (v - n_joint));
            off = ((uint64_t)md[1] << 32) | md[0];
            max_len = md[2], min_len = md[3];
            const uint32_t slot0 = md[4];
            my_len = p.mixed_len[slot0 + lane];
            idx = p.mixed_orig[slot0 + lane];
        }
        const uint4* src = reinterpret_cast<const uint4*>(p.data + off);
        const bool valid = idx != kPad;

        State st;
        st.init();
        const uint32_t nch = (max_len + kChunk - 1) / kChunk, full = min_len / kChunk;
        const TileFin fin = tile_fin(p, p.len1, my_len);  // per lane here
        bool dead = false;
        uint4 cur = load_chunk(src + lane);  // (an empty tile reads the next block or the tail padding: always readable)
        for (uint32_t c = 0; c < nch; ++c) {
            uint4 nxt = cur;
            if (c + 1 < nch) nxt = load_chunk(src + (size_t)(c + 1) * kWave + lane);
            if (c < full)
                process_chunk_full<State>(st, lds_pm, cur);
            else
                process_chunk_masked<State>(st, lds_pm, cur, c * kChunk, my_len);
            if (early) {
                const uint32_t j = min(my_len, (c + 1) * kChunk);
                if (__ballot(valid && may_pass(p, fin, st.bound(p.len1, j, my_len))) == 0) {
                    dead = true;  // no lane of the tile can pass the cutoff any more
                    break;
                }
            }
            cur = nxt;
        }
        if (valid) {
            if (dead)
                emit_none(p, idx);
            else
                emit_fin(p, fin, st.result(p.len1, my_len), idx, p.out);
        }
    }
}

template <class State>
static hipError_t launch_mixed_state(const ScanParams& p, hipStream_t stream)
{
    const uint32_t tiles = p.tile_end - p.tile_begin + (p.joint_end - p.joint_begin);
    const dim3 g(std::max(1, scan_grid(tiles))), b(kWave * kWavesPerBlock);
    hipLaunchKernelGGL((scan_kernel_mixed<State>), g, b, 0, stream, p);
    return hipGetLastError();
}
template <template <int> class StateT>
static hipError_t launch_mixed_words(const ScanParams& p, hipStream_t stream)
{
    switch (p.words) {
    case 1: return launch_mixed_state<StateT<1>>(p, stream);
    case 2: return launch_mixed_state<StateT<2>>(p, stream);
    case 3: return launch_mixed_state<StateT<3>>(p, stream);
    case 4: return launch_mixed_state<StateT<4>>(p, stream);
    case 5: return launch_mixed_state<StateT<5>>(p, stream);
    case 6: return launch_mixed_state<StateT<6>>(p, stream);
    case 7: return launch_mixed_state<StateT<7>>(p, stream);
    case 8: return launch_mixed_state<StateT<8>>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

// p.mixed / p.mixed_len / p.mixed_orig and [p.tile_begin, p.tile_end) describe the mixed section to scan
hipError_t launch_scan_mixed(RawKind raw, const ScanParams& p, hipStream_t stream)
{
    if (p.tile_end <= p.tile_begin && p.joint_end <= p.joint_begin) return hipSuccess;
    switch (raw) {
    case RAW_LEV: return p.len1 <= 32 ? launch_mixed_state<Lev32State>(p, stream) : launch_mixed_words<LevState>(p, stream);
    case RAW_LCS: return p.len1 <= 32 ? launch_mixed_state<Lcs32State>(p, stream) : launch_mixed_words<LcsState>(p, stream);
    case RAW_OSA: return launch_mixed_words<OsaState>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rf
