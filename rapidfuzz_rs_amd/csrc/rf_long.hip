// rf_long.hip -- the two completeness paths of the Levenshtein family that do not fit the register-resident scan:
//   wf_kernel     generalized weight tables: Wagner-Fischer rows in LDS (levenshtein.rs:212-259)
//   long_kernel   patterns beyond 512 symbols: 8 words per sweep, carries parked in HBM (levenshtein.rs:769-1019,
//                 lcs_seq.rs:267-341)
#include "rf_device.hpp"

namespace rf {

// ---------------------------------------------------------------------------------------------------
// Generalized weights (levenshtein.rs:212-259 generalized_wagner_fischer, reached from _distance_with_pm :1328-1330
// for every weight table that is neither (f,f,f) nor (f,f,>=2f)): the O(len1 * len2) row DP, one candidate per lane.
//   new[i+1] = s1[i] == ch2 ? old[i] : min(new[i] + del, old[i] + sub, old[i+1] + ins)
// The row (len1 + 1 u32 per lane) lives in LDS as [i][lane] -- conflict-free, 256 B per row entry and wavefront -- so
// the workgroup has as many wavefronts as fit (plan(): wf_waves).  The query, renamed like the corpus, is rebuilt
// from the PM table into LDS and read back 4 symbols at a time with a wavefront-uniform (broadcast) address.
// A completeness path (~10 VALU + 2.25 LDS operations per cell); the reference's common-affix stripping and minimum-
// edits test (:286-309) change nothing in the value and are not replayed.
// ---------------------------------------------------------------------------------------------------
template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void wf_kernel(const ScanParams p)
{
    extern __shared__ uint32_t lds_wf[];
    const uint32_t len1 = p.len1;
    const uint32_t qwords = (len1 + 3) / 4 + 1;  // query bytes, 4 per word, one word of slack
    uint8_t* lds_q = reinterpret_cast<uint8_t*>(lds_wf);
    for (uint32_t i = threadIdx.x; i < qwords; i += blockDim.x) lds_wf[i] = 0;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) {  // PM row c, bit i  <=>  s1[i] == c
        const uint8_t stored = p.sigma[c];
        for (uint32_t w = 0; w * 64 < len1; ++w) {
            uint64_t bits = p.pm[(size_t)c * p.words + w];
            while (bits) {
                lds_q[64 * w + (__ffsll((unsigned long long)bits) - 1)] = stored;
                bits &= bits - 1;
            }
        }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t waves = blockDim.x / kWave;
    // row[i * kWave] = cache[i] of this lane: in LDS, or -- for queries whose row does not fit (beyond ~590 symbols) -- in a
    // global scratch strip per wavefront (same [i][lane] layout, coalesced, L2-resident): slower, but no length limit
    uint32_t* row = p.wf_global ? p.long_scratch + ((size_t)blockIdx.x * waves + wave) * (len1 + 1) * kWave + lane
                                : lds_wf + qwords + (size_t)wave * (len1 + 1) * kWave + lane;
    const uint32_t ins = p.w_ins, del = p.w_del, sub = p.w_sub;

    for (uint32_t t = blockIdx.x * waves + wave; t < p.n_tiles; t += gridDim.x * waves) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        for (uint32_t i = 0; i <= len1; ++i) row[i * kWave] = i * del;  // :219-221
        uint32_t top = 0;  // cache[0] = j * ins, the same in every lane
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        for (uint32_t c = 0; c < nch; ++c) {
            const uint4 data = load_chunk(tv.src + (size_t)c * kWave + lane);
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            for (uint32_t j = 0; j < cols; ++j) {
                const uint32_t word = j < 4 ? data.x : (j < 8 ? data.y : (j < 12 ? data.z : data.w));
                const uint32_t ch2 = (word >> (8 * (j & 3))) & 0xFFu;
                uint32_t diag = top;  // old[i]
                top += ins;           // :226
                uint32_t left = top;  // new[i]
                for (uint32_t i = 0; i < len1; i += 4) {
                    const uint32_t q4 = lds_wf[i / 4];  // wavefront-uniform address: one broadcast read for 4 symbols
                    const uint32_t lim = min(4u, len1 - i);
                    for (uint32_t k = 0; k < lim; ++k) {
                        const uint32_t up = row[(i + k + 1) * kWave];  // old[i+1]
                        const uint32_t x = ((q4 >> (8 * k)) & 0xFFu) == ch2 ? diag : min(min(left + del, diag + sub), up + ins);
                        row[(i + k + 1) * kWave] = x;
                        diag = up;
                        left = x;
                    }
                }
            }
        }
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
            const uint32_t dist = len1 ? row[len1 * kWave] : top;
            // _maximum, levenshtein.rs:263-277 (not affine in the lengths for a general table)
            const uint32_t max_dist = len1 * del + len2 * ins;
            const uint32_t alt = len1 >= len2 ? len2 * sub + (len1 - len2) * del : len1 * sub + (len2 - len1) * ins;
            TileFin f;
            f.max = min(max_dist, alt);
            f.d0 = 0;
            f.v0 = p.fin_flip ? f.max : 0;  // similarity = maximum - distance (fin_vR = -1), distance = raw (fin_vR = +1)
            emit_fin(p, f, dist, idx, p.out);
        }
    }
}

// The same recurrence with the row in REGISTERS for queries of at most kMax <= 64 symbols: kMax + 1 VGPRs of row, the
// (renamed) query bytes arrive in the kernel arguments and so live in SGPRs, both loops over the row are fully
// unrolled: 6 VALU per cell (compare, three adds, min3, select) and no LDS at all.  Row entries beyond len1 are computed
// and never read (every entry only depends on lower ones).
template <bool kUniform, int kMax>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void wf_reg_kernel(const ScanParams p)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t len1 = p.len1;
    const uint32_t ins = p.w_ins, del = p.w_del, sub = p.w_sub;
    for (uint32_t t = blockIdx.x * kWavesPerBlock + wave; t < p.n_tiles; t += gridDim.x * kWavesPerBlock) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        uint32_t row[kMax + 1];
#pragma unroll
        for (int i = 0; i <= kMax; ++i) row[i] = (uint32_t)i * del;
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        for (uint32_t c = 0; c < nch; ++c) {
            uint4 data = load_chunk(tv.src + (size_t)c * kWave + lane);
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            for (uint32_t j = 0; j < cols; ++j) {
                const uint32_t ch2 = data.x & 0xFFu;
                uint32_t diag = row[0];
                row[0] += ins;
                uint32_t left = row[0];
#pragma unroll
                for (int i = 0; i < kMax; ++i) {
                    const uint32_t qi = (p.wf_query[i / 4] >> (8 * (i % 4))) & 0xFFu;  // scalar
                    const uint32_t up = row[i + 1];
                    const uint32_t x = qi == ch2 ? diag : min(min(left + del, diag + sub), up + ins);
                    row[i + 1] = x;
                    diag = up;
                    left = x;
                }
                data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                data.w >>= 8;
            }
        }
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) {
            uint32_t dist = row[0];
#pragma unroll
            for (int i = 1; i <= kMax; ++i) dist = (uint32_t)i == len1 ? row[i] : dist;  // row[len1] without dynamic indexing
            const uint32_t max_dist = len1 * del + len2 * ins;
            const uint32_t alt = len1 >= len2 ? len2 * sub + (len1 - len2) * del : len1 * sub + (len2 - len1) * ins;
            TileFin f;
            f.max = min(max_dist, alt);
            f.d0 = 0;
            f.v0 = p.fin_flip ? f.max : 0;
            emit_fin(p, f, dist, idx, p.out);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Patterns longer than 512 symbols (the reference's hyrroe2003_block / lcs_blockwise territory,
// levenshtein.rs:769-1019, lcs_seq.rs:267-341): the pattern is cut into groups of 8 words (512 rows).  A
// wavefront sweeps the candidate once per group with that group's 16 bit-vectors in registers; the horizontal
// deltas crossing the group boundary (2 bits per column and lane for Levenshtein, the adder carry for LCS) wait in
// a chunk-interleaved HBM scratch strip between sweeps.  PM words come straight from global memory (the table of a
// long pattern does not fit LDS; it is L2-resident).  Throughput path for completeness, not for the roofline.
// ---------------------------------------------------------------------------------------------------
constexpr int kLongGroup = 8;
enum LongKind : int { LONG_LEV = 0, LONG_LCS = 1, LONG_OSA = 2 };

template <int kKind, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void long_kernel(const ScanParams p)
{
    constexpr bool kLcs = kKind == LONG_LCS, kOsa = kKind == LONG_OSA;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + wave;  // global wavefront id: owns one scratch strip
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    const uint32_t words_pad = p.long_words_pad;
    const uint32_t groups = words_pad / kLongGroup;
    // one strip of carries per wavefront (OSA: two, the second for the transposition bit: osa.rs:180 across word groups)
    const size_t strip_words = (size_t)p.long_chunks_max * kWave;
    uint32_t* strip = p.long_scratch + (size_t)gw * strip_words * (kOsa ? 2 : 1);
    uint32_t* strip_tr = strip + strip_words;
    __shared__ uint8_t lds_unrename[256];  // stored symbol -> original symbol (the PM table stays in global memory)
    lds_unrename[p.sigma[threadIdx.x & 255]] = (uint8_t)(threadIdx.x & 255);
    __syncthreads();

    for (uint32_t t = gw; t < p.n_tiles; t += stride) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        int32_t acc = 0;

        for (uint32_t g = 0; g < groups; ++g) {
            LevState<kLongGroup> lev;
            LcsState<kLongGroup> lcs;
            OsaState<kLongGroup> osa;
            if (kLcs)
                lcs.init();
            else if (kOsa)
                osa.init();
            else
                lev.init();
            for (uint32_t c = 0; c < nch; ++c) {
                uint4 data = tv.src[(size_t)c * kWave + lane];
                // carries entering word 0 of this group for the 16 columns of the chunk:
                // bits 0..15 = hp (or the LCS adder carry), bits 16..31 = hn; OSA: the transposition bits in a strip of their own
                uint32_t cin = g == 0 ? (kLcs ? 0u : 0x0000FFFFu) : strip[(size_t)c * kWave + lane];
                uint32_t tin = (kOsa && g != 0) ? strip_tr[(size_t)c * kWave + lane] : 0u;
                uint32_t cout = 0, tout = 0;
                const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
                for (uint32_t j = 0; j < cols; ++j) {
                    const uint32_t ch = lds_unrename[data.x & 0xFFu];
                    const uint64_t* row = p.pm + (size_t)ch * words_pad + (size_t)g * kLongGroup;
                    uint64_t x[kLongGroup];
#pragma unroll
                    for (int w = 0; w < kLongGroup; ++w) x[w] = row[w];
                    if (kLcs) {
                        uint32_t carry = (cin >> j) & 1u;
                        lcs.step_carry(x, carry);
                        cout |= carry << j;
                    } else if (kOsa) {
                        uint32_t hp_c = (cin >> j) & 1u, hn_c = (cin >> (16 + j)) & 1u, tr_c = (tin >> j) & 1u;
                        osa.step_carry(x, hp_c, hn_c, tr_c);
                        cout |= (hp_c << j) | (hn_c << (16 + j));
                        tout |= tr_c << j;
                    } else {
                        uint32_t hp_c = (cin >> j) & 1u, hn_c = (cin >> (16 + j)) & 1u;
                        lev.step_carry(x, hp_c, hn_c);
                        cout |= (hp_c << j) | (hn_c << (16 + j));
                    }
                    data.x = __builtin_amdgcn_alignbit(data.y, data.x, 8);
                    data.y = __builtin_amdgcn_alignbit(data.z, data.y, 8);
                    data.z = __builtin_amdgcn_alignbit(data.w, data.z, 8);
                    data.w >>= 8;
                }
                if (g + 1 < groups) {
                    strip[(size_t)c * kWave + lane] = cout;
                    if (kOsa) strip_tr[(size_t)c * kWave + lane] = tout;
                }
            }
            acc += kLcs ? (int32_t)lcs.result(0, 0) : (kOsa ? osa.delta_sum(p.len1, g * kLongGroup) : lev.delta_sum(p.len1, g * kLongGroup));
        }
        const uint32_t raw = kLcs ? (uint32_t)acc : (uint32_t)((int32_t)len2 + acc);
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) emit_usize(p, raw, len2, idx);
    }
}

template <int kKind>
static hipError_t launch_long_kind(const ScanParams& p, hipStream_t stream, int grid)
{
    const dim3 g(grid), b(kWave * kWavesPerBlock);
    if (p.tiles)
        hipLaunchKernelGGL((long_kernel<kKind, false>), g, b, 0, stream, p);
    else
        hipLaunchKernelGGL((long_kernel<kKind, true>), g, b, 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_long(RawKind raw, const ScanParams& p, hipStream_t stream, int grid)
{
    if (raw == RAW_LCS) return launch_long_kind<LONG_LCS>(p, stream, grid);
    if (raw == RAW_OSA) return launch_long_kind<LONG_OSA>(p, stream, grid);
    return launch_long_kind<LONG_LEV>(p, stream, grid);
}

template <int kMax>
static hipError_t launch_wf_reg(const ScanParams& p, hipStream_t stream)
{
    const dim3 g(std::max(1, scan_grid(p.n_tiles))), b(kWave * kWavesPerBlock);
    if (p.tiles)
        hipLaunchKernelGGL((wf_reg_kernel<false, kMax>), g, b, 0, stream, p);
    else
        hipLaunchKernelGGL((wf_reg_kernel<true, kMax>), g, b, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_wf(const ScanParams& p, hipStream_t stream)
{
    static const bool use_reg = [] { const char* e = getenv("RF_WF_REG"); return !e || atoi(e) != 0; }();  // A/B switch
    if (use_reg && p.len1 <= 16) return launch_wf_reg<16>(p, stream);
    if (use_reg && p.len1 <= 32) return launch_wf_reg<32>(p, stream);
    if (use_reg && p.len1 <= 64) return launch_wf_reg<64>(p, stream);
    const size_t lds = ((size_t)(p.len1 + 3) / 4 + 1) * 4 + (p.wf_global ? 0 : (size_t)p.wf_waves * (p.len1 + 1) * kWave * 4);
    const dim3 g(p.wf_global ? std::max(1u, p.long_grid) : (uint32_t)std::max(1, scan_grid(p.n_tiles))), b(kWave * p.wf_waves);
    auto k = p.tiles ? wf_kernel<false> : wf_kernel<true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, g, b, lds, stream, p);
    return hipGetLastError();
}

}  // namespace rf
