// rf_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the one-vs-many scan.
//
// Execution shape: one candidate per wavefront lane, one 64-candidate tile per wavefront
// at a time, 4 wavefronts (one per SIMD) per workgroup, grid-stride over tiles.  The query's
// pattern-match table (256 x W u64, src/details/pattern_match_vector.rs:194-321) is staged once per
// workgroup into LDS; candidate bytes arrive as one coalesced 1 KiB global_load_dwordx4 per wavefront per
// 16 columns; the bit-vectors of the recurrence (VP/VN, S, P/T flags) never leave VGPRs.
// Integer/bitwise work only: no MFMA.  3-input boolean terms use v_bitop3_b32 (new on gfx950).
//
// Reference algorithms restated here for the device (cited per function):
//   hyrroe2003 / hyrroe2003_block      src/distance/levenshtein.rs:435-507, :769-1019 (advance_block :838-875)
//   lcs_unroll                         src/distance/lcs_seq.rs:199-261
//   flag_similar_characters_word,
//   count_transpositions_word          src/distance/jaro.rs:147-190, :339-368
//   MetricUsize / Metricf64 defaults   src/details/distance.rs:154-385
#include <algorithm>

#include "rf_internal.hpp"

namespace rf {

// ---------------------------------------------------------------------------------------------------
// v_bitop3_b32: arbitrary 3-input boolean function; the truth table is f(0xF0, 0xCC, 0xAA)
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <uint32_t TT>
__device__ __forceinline__ uint64_t lut3(uint64_t a, uint64_t b, uint64_t c)
{
    uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT & 0xFF);
    uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), TT & 0xFF);
    return ((uint64_t)hi << 32) | lo;
}
constexpr uint32_t T_XOR_OR = (TA ^ TB) | TC;       // (a ^ b) | c
constexpr uint32_t T_OR_NOR = TA | (~(TB | TC));    // a | ~(b | c)

// (x << 1) | carry_in.  Measured on gfx950 (tools/microbench.hip, profiles/microbench_r01.txt): the 64-bit VALU
// forms v_lshl_add_u64 / v_lshlrev_b64 issue at the same (half) rate as ONE v_alignbit_b32 / v_lshl_or_b32, so a
// single 64-bit instruction beats the two-instruction 32-bit pair hipcc otherwise builds from split halves.
// Plain VALU on VGPR pairs: no memory counters, no hazard padding needed (guide 5.7).
// (hipcc canonicalises x + x + 1 back into shift-or on split halves, hence the asm; it is plain VALU on VGPR
// pairs: nothing to count, no hazard padding needed -- guide 5.7.)
template <int CIN>
__device__ __forceinline__ uint64_t shl1_const(uint64_t x)
{
    uint64_t r;
    if (CIN)
        asm("v_lshl_add_u64 %0, %1, 1, 1" : "=v"(r) : "v"(x));
    else
        asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint64_t shl1_var(uint64_t x, uint32_t cin)
{
    uint64_t r;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return r | cin;
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------------------------------------------
// Levenshtein: one column of Hyyro's recurrence over W 64-bit words (levenshtein.rs:466-490 for W == 1,
// advance_block :838-875 for the carries between words).  The running score of :476-477 is NOT tracked:
// after the last column VP/VN hold the vertical deltas of column len2, so
//   D[len1][len2] = len2 + popcount(VP & valid) - popcount(VN & valid)        (D[0][len2] = len2)
// which removes the per-column mask tests from the hot loop.
// ---------------------------------------------------------------------------------------------------
template <int W>
struct LevState {
    uint64_t vp[W], vn[W];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            vp[w] = ~0ull;  // levenshtein.rs:454-455
            vn[w] = 0;
        }
    }
    __device__ __forceinline__ void step(const uint64_t* __restrict__ pm_row)
    {
        uint32_t hp_c = 1, hn_c = 0;  // levenshtein.rs:824-825
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint64_t x = pm_row[w];
            if (w > 0) x |= hn_c;                            // :847
            const uint64_t p = vp[w], n = vn[w];
            const uint64_t sum = (x & p) + p;
            const uint64_t e = lut3<T_XOR_OR>(sum, p, x);    // (sum ^ vp) | x
            const uint64_t d0 = e | n;                       // :848
            const uint64_t hn = e & p;                       // == d0 & vp because vp & vn == 0   (:852)
            const uint64_t hp = lut3<T_OR_NOR>(n, d0, p);    // vn | ~(d0 | vp)                  (:851)
            const uint64_t hps = w == 0 ? shl1_const<1>(hp) : shl1_var(hp, hp_c);  // :865-866
            const uint64_t hns = w == 0 ? shl1_const<0>(hn) : shl1_var(hn, hn_c);
            if (w + 1 < W) {                                 // :857-858
                hp_c = (uint32_t)(hp >> 63);
                hn_c = (uint32_t)(hn >> 63);
            }
            vn[w] = hps & d0;                                // :869
            vp[w] = lut3<T_OR_NOR>(hns, hps, d0);            // hn | ~(d0 | hp)                  (:868)
        }
    }
    // D[len1][len2] from the final column's vertical deltas
    __device__ __forceinline__ uint32_t result(uint32_t len1, uint32_t len2) const
    {
        int32_t d = (int32_t)len2;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int32_t bits = (int32_t)len1 - 64 * w;  // valid pattern rows in this word
            uint64_t valid = bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1));
            d += __popcll(vp[w] & valid) - __popcll(vn[w] & valid);
        }
        return (uint32_t)d;
    }
};

// ---------------------------------------------------------------------------------------------------
// LCS: Hyyro's bit-parallel LCS length (lcs_seq.rs:222-252): S' = (S + (S & M)) | (S - (S & M)) with the
// add's carry chained across words; similarity = sum popcount(~S).
// ---------------------------------------------------------------------------------------------------
template <int W>
struct LcsState {
    uint64_t s[W];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int w = 0; w < W; ++w) s[w] = ~0ull;  // lcs_seq.rs:215
    }
    __device__ __forceinline__ void step(const uint64_t* __restrict__ pm_row)
    {
        uint64_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t sw = s[w];
            const uint64_t u = sw & pm_row[w];
            uint64_t x = sw + u;  // carrying_add, src/details/intrinsics.rs:22-26
            uint64_t c = x < sw;
            if (w > 0) {
                const uint64_t x2 = x + carry;
                c |= (uint64_t)(x2 < x);
                x = x2;
            }
            carry = c;
            s[w] = x | (sw - u);  // lcs_seq.rs:230
        }
    }
    __device__ __forceinline__ uint32_t result(uint32_t, uint32_t) const
    {
        uint32_t sim = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) sim += __popcll(~s[w]);  // lcs_seq.rs:254-257
        return sim;
    }
};

// ---------------------------------------------------------------------------------------------------
// finishing arithmetic: raw primitive -> the value `<op>_with_args` returns (or None)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lev_maximum(const ScanParams& p, uint32_t len1, uint32_t len2)
{
    // levenshtein.rs:263-277 _maximum
    const uint32_t max_dist = len1 * p.w_del + len2 * p.w_ins;
    const uint32_t alt = len1 >= len2 ? len2 * p.w_sub + (len1 - len2) * p.w_del : len1 * p.w_sub + (len2 - len1) * p.w_ins;
    return min(max_dist, alt);
}

// norm_sim_to_norm_dist, src/details/common.rs:4-7
__device__ __forceinline__ double norm_sim_to_norm_dist(double c) { return fmin(1.0 - c + 0.00001, 1.0); }

struct UsizeResult {
    uint32_t dist, maximum;
};

__device__ __forceinline__ UsizeResult usize_result(const ScanParams& p, uint32_t raw, uint32_t len2)
{
    UsizeResult r;
    const uint32_t len1 = p.len1;
    if (p.finish == FIN_LEV) {  // uniform weights: distance * factor (levenshtein.rs:1308-1316)
        r.dist = raw * p.factor;
        r.maximum = lev_maximum(p, len1, len2);
    } else if (p.finish == FIN_LCS) {  // details/distance.rs:157-179 over lcs_seq.rs:772-793
        r.maximum = max(len1, len2);
        r.dist = r.maximum - raw;
    } else {  // indel.rs:365-367; FIN_LEV_INDEL = levenshtein weights (f, f, >= 2f), levenshtein.rs:1321-1327
        r.dist = (len1 + len2 - 2 * raw) * p.factor;
        r.maximum = p.finish == FIN_LEV_INDEL ? lev_maximum(p, len1, len2) : (len1 + len2);
    }
    return r;
}

// Which value the op yields and whether `score()` (src/common.rs:43-45 / :83-85) keeps it.  All kernels on
// this path are exact, so the CPU-side cutoff plumbing (details/distance.rs:157-274) reduces to
// "compute the value, then compare with the user's cutoff" -- see DESIGN.md "cutoff equivalence".
__device__ __forceinline__ void emit_usize(const ScanParams& p, uint32_t raw, uint32_t len2, uint32_t idx)
{
    const UsizeResult r = usize_result(p, raw, len2);
    if (!p.out_f64) {
        uint32_t v;
        bool keep;
        if (p.op == RF_OP_DISTANCE) {
            v = r.dist;
            keep = !p.has_cutoff || v <= p.cutoff_u32;
        } else {  // similarity = maximum - distance (details/distance.rs:209-210)
            v = r.maximum - r.dist;
            keep = !p.has_cutoff || v >= p.cutoff_u32;
        }
        reinterpret_cast<uint32_t*>(p.out)[idx] = keep ? v : RF_NONE_U32;
    } else {
        // details/distance.rs:246-250: dist / maximum (0.0 when maximum == 0)
        const double nd = r.maximum == 0 ? 0.0 : (double)r.dist / (double)r.maximum;
        double v;
        bool keep;
        if (p.op == RF_OP_NORMALIZED_DISTANCE) {
            v = nd;
            keep = !p.has_cutoff || v <= p.cutoff_f64;
        } else {  // details/distance.rs:273: 1.0 - norm_dist
            v = 1.0 - nd;
            keep = !p.has_cutoff || v >= p.cutoff_f64;
        }
        reinterpret_cast<double*>(p.out)[idx] = keep ? v : __longlong_as_double(0x7FF8000000000000ll);
    }
}

// ---------------------------------------------------------------------------------------------------
// the scan kernel
// ---------------------------------------------------------------------------------------------------
template <class State, int W>
__device__ __forceinline__ void process_chunk_full(State& st, const uint64_t* lds_pm, const uint4& c)
{
    const uint32_t dw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ch = (dw[d] >> (8 * k)) & 0xFFu;
            st.step(lds_pm + ch * W);
        }
    }
}

template <class State, int W>
__device__ __forceinline__ void process_chunk_tail(State& st, const uint64_t* lds_pm, uint4 c, uint32_t rem)
{
    for (uint32_t j = 0; j < rem; ++j) {  // rem is wavefront-uniform (tile length)
        const uint32_t ch = c.x & 0xFFu;
        st.step(lds_pm + ch * W);
        c.x = __builtin_amdgcn_alignbit(c.y, c.x, 8);
        c.y = __builtin_amdgcn_alignbit(c.z, c.y, 8);
        c.z = __builtin_amdgcn_alignbit(c.w, c.z, 8);
        c.w >>= 8;
    }
}

struct TileView {
    const uint4* src;  // wavefront-uniform base of the tile payload
    uint32_t len, slot0;
};
template <bool kUniform>
__device__ __forceinline__ TileView load_tile(const ScanParams& p, uint32_t t)
{
    TileView v;
    if (!kUniform) {
        // t is wavefront-uniform and the descriptors are read-only for the whole launch: read them through
        // the constant address space so they become scalar s_load_dwordx4 (no VGPRs, no vmcnt traffic)
        typedef const __attribute__((address_space(4))) uint32_t* cptr;
        cptr td = (cptr)(uintptr_t)(p.tiles + t);
        const uint32_t off_lo = td[0], off_hi = td[1];
        v.len = td[2];
        v.slot0 = td[3];
        v.src = reinterpret_cast<const uint4*>(p.data + (((uint64_t)off_hi << 32) | off_lo));
    } else {  // single-length corpus: tile t is at t * tile_bytes, no descriptor traffic at all
        v.len = p.uniform_len;
        v.slot0 = t * kWave;
        v.src = reinterpret_cast<const uint4*>(p.data + (uint64_t)t * p.uniform_tile_bytes);
    }
    return v;
}

template <class State, int W, bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void scan_kernel(const ScanParams p)
{
    __shared__ uint64_t lds_pm[256 * W];
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock) lds_pm[i] = p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock;

    // Each wavefront walks its tiles as one continuous stream of 16-column chunks.  The load of the NEXT
    // chunk (the next 16 columns of this tile, or the first 16 of the wavefront's next tile) is always issued
    // before the current chunk is processed, so exactly one 1 KiB request per wavefront is in flight and the
    // wait before each chunk is a counted vmcnt(1), never a drain.
    uint32_t t = blockIdx.x * kWavesPerBlock + wave;
    if (t >= p.n_tiles) return;
    TileView cur_tile = load_tile<kUniform>(p, t);
    uint4 cur = cur_tile.src[lane];  // the packed buffer carries one chunk of tail padding: always readable

    while (true) {
        const uint32_t t_next = t + stride;
        const bool has_next = t_next < p.n_tiles;
        const TileView next_tile = load_tile<kUniform>(p, has_next ? t_next : t);

        const uint32_t len2 = cur_tile.len;
        const uint32_t slot = cur_tile.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];  // issued early; consumed after the columns

        State st;
        st.init();
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        for (uint32_t c = 0; c < nch; ++c) {
            const uint4* nsrc = (c + 1 < nch) ? cur_tile.src + (size_t)(c + 1) * kWave : next_tile.src;
            const uint4 nxt = nsrc[lane];
            const uint32_t cols = len2 - c * kChunk;
            if (cols >= kChunk)
                process_chunk_full<State, W>(st, lds_pm, cur);
            else
                process_chunk_tail<State, W>(st, lds_pm, cur, cols);
            cur = nxt;
        }
        if (nch == 0) cur = next_tile.src[lane];

        const uint32_t raw = st.result(p.len1, len2);
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        if (valid) emit_usize(p, raw, len2, idx);

        if (!has_next) break;
        t = t_next;
        cur_tile = next_tile;
    }
}

// ---------------------------------------------------------------------------------------------------
// corpus packing on the device: row-major fixed-length rows -> chunk-interleaved tiles
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t* __restrict__ rows, size_t n, uint32_t len,
                                                        size_t stride, uint8_t* __restrict__ packed, uint32_t n_tiles)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t chunks = (len + kChunk - 1) / kChunk;
    const size_t tile_bytes = (size_t)chunks * kWave * kChunk;
    for (size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += (size_t)gridDim.x * 4) {
        const size_t row = t * kWave + lane;
        uint8_t* dst = packed + t * tile_bytes + (size_t)lane * kChunk;
        const uint8_t* src = rows + row * stride;
        for (uint32_t c = 0; c < chunks; ++c) {
            uint32_t w[4] = {0, 0, 0, 0};
            if (row < n) {
                const uint32_t base = c * kChunk;
                if (base + kChunk <= len && ((reinterpret_cast<uintptr_t>(src + base) & 3) == 0)) {
                    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + base);
                    w[0] = s4[0];
                    w[1] = s4[1];
                    w[2] = s4[2];
                    w[3] = s4[3];
                } else {
                    for (uint32_t b = 0; b < kChunk && base + b < len; ++b) w[b / 4] |= (uint32_t)src[base + b] << (8 * (b % 4));
                }
            }
            *reinterpret_cast<uint4*>(dst + (size_t)c * kWave * kChunk) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

hipError_t launch_pack_rows(const uint8_t* rows, size_t n, uint32_t len, size_t stride, uint8_t* packed, uint32_t n_tiles,
                            hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n_tiles + 3) / 4, 256 * 16);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, len, stride, packed, n_tiles);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------------
int scan_max_grid() { return 256 * 8; }  // 256 CUs x 8 workgroups (32 waves/CU): the whole chip resident

template <template <int> class StateT, int W>
static hipError_t launch_one(const ScanParams& p, hipStream_t stream, int grid)
{
    if (p.tiles)
        hipLaunchKernelGGL((scan_kernel<StateT<W>, W, false>), dim3(grid), dim3(kWave * kWavesPerBlock), 0, stream, p);
    else
        hipLaunchKernelGGL((scan_kernel<StateT<W>, W, true>), dim3(grid), dim3(kWave * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}

template <template <int> class StateT>
static hipError_t launch_words(const ScanParams& p, hipStream_t stream, int grid)
{
    switch (p.words) {
    case 1: return launch_one<StateT, 1>(p, stream, grid);
    case 2: return launch_one<StateT, 2>(p, stream, grid);
    case 3: return launch_one<StateT, 3>(p, stream, grid);
    case 4: return launch_one<StateT, 4>(p, stream, grid);
    case 5: return launch_one<StateT, 5>(p, stream, grid);
    case 6: return launch_one<StateT, 6>(p, stream, grid);
    case 7: return launch_one<StateT, 7>(p, stream, grid);
    case 8: return launch_one<StateT, 8>(p, stream, grid);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan(RawKind raw, const ScanParams& p, hipStream_t stream, int* grid_used)
{
    if (p.n_tiles == 0) return hipSuccess;
    const int grid = (int)std::min<uint32_t>((p.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, (uint32_t)scan_max_grid());
    if (grid_used) *grid_used = grid;
    switch (raw) {
    case RAW_LEV: return launch_words<LevState>(p, stream, grid);
    case RAW_LCS: return launch_words<LcsState>(p, stream, grid);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rf
