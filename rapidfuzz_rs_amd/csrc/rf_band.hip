// rf_band.hip -- Levenshtein with a long query and a small distance cutoff: one 64-bit word sliding down the diagonal.
//
// The reference's hyrroe2003_small_band_with_pm (src/distance/levenshtein.rs:509-617; Hyyro's band variant, after Ukkonen):
// if the distance is to be at most k, only the cells within k of the main diagonal matter, and 2k + 1 <= 64 of them per
// column fit ONE machine word whatever the query length.  The word covers pattern rows [j + k - 63, j + k] at column j and
// moves down one row per column: where the full-matrix recurrence shifts the horizontal deltas up by one (HP << 1,
// HN << 1), the band recurrence shifts the diagonal term down instead (D0 >> 1).  The multi-word scan kernel spends
// ceil(len1 / 64) words of VALU per column on such a query (4 for BASELINE.json configs[2], 8 at 512 symbols); this one
// spends one, plus fetching the pattern bits at a bit offset:
//   * the PM table is staged in LDS as 32-bit words with 64 zero bits below row 0 and above the last row of every symbol's
//     bit-vector, row stride odd (bank spread), so the 64 pattern bits at ANY window position v = start_pos + 64 are three
//     consecutive dwords d0 d1 d2 at index v / 32 funnel-shifted by v % 32: two v_alignbit_b32 with a scalar shift amount
//     (the window position depends on the column only: wavefront-uniform);
//   * the running score follows the reference: down the diagonal while the window's top row is above the last pattern row
//     (score += the diagonal delta, bit 63 of D0 clear), then along the last row (+HP, -HN at a bit that moves down).
// Exact whenever the distance is <= k; any value > k otherwise, which the finishing compare turns into None -- the same
// contract the reference relies on (levenshtein.rs:1059-1066).  A wavefront abandons a tile as soon as every lane is past
// break_score (levenshtein.rs:519-523, :568-570), and candidates whose length differs from the query's by more than k are
// None without being read.
#include "rf_device.hpp"
#include "rf_band_asm.inc"

namespace rf {

constexpr uint32_t band_row_dwords(uint32_t words) { return 2 * words + 5; }  // 2 zero dwords + 2W + 2 zero dwords + 1 (odd stride)

template <bool kUniform>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void band_kernel(const ScanParams p)
{
    extern __shared__ uint32_t lds_band[];  // 256 rows x band_row_dwords(words)
    const uint32_t W = p.words, stride = band_row_dwords(W);
    for (uint32_t i = threadIdx.x; i < 256 * stride; i += kWave * kWavesPerBlock) {
        const uint32_t c = i / stride, d = i % stride;
        uint32_t v = 0;
        if (d >= 2 && d < 2 + 2 * W) {
            const uint64_t word = p.pm[(size_t)c * W + (d - 2) / 2];
            v = (d & 1) ? (uint32_t)(word >> 32) : (uint32_t)word;  // d - 2 even -> low half
        }
        lds_band[(uint32_t)p.sigma[c] * stride + d] = v;  // the corpus stores renamed symbols
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t len1 = p.len1, k = p.band_k;
    const uint32_t pitch_bytes = stride * 4u;
    const bool asm_run = p.band_asm != 0;  // (RF_ASM_BAND=0: the compiled column everywhere, the A/B switch)
    for (uint32_t t = p.tile_begin + blockIdx.x * kWavesPerBlock + wave; t < p.tile_end; t += gridDim.x * kWavesPerBlock) {
        const TileView tv = load_tile<kUniform>(p, t);
        const uint32_t len2 = tv.len;
        const uint32_t slot = tv.slot0 + lane;
        uint32_t idx = slot;
        if (!kUniform) idx = p.orig[slot];
        const bool valid = kUniform ? slot < p.n : idx != kPad;
        const uint32_t diff = len1 > len2 ? len1 - len2 : len2 - len1;
        if (diff > k) {  // levenshtein.rs:1389-1391: the distance is at least the length difference
            if (valid) emit_none(p, idx);
            continue;
        }
        // levenshtein.rs:515-531
        uint64_t vp = ~0ull << (63 - k), vn = 0;
        const uint32_t break_score = 2 * k + len2 - len1;  // (diff <= k: never negative)
        const uint32_t first = min(len1 - k, len2);        // columns walked down the diagonal (len1 > 64 > k)
        uint32_t diag_hits = 0;                            // set diagonal deltas seen in phase 1: score = k + columns - hits
        uint32_t score = k;                                // phase 2 continues from the phase-1 total
        uint32_t v = k + 1;                                // window position + 64: bit v - 64 + b of the pattern is window bit b
        bool dead = false;
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        uint4 cur = nch ? load_chunk(tv.src + lane) : make_uint4(0, 0, 0, 0);
        for (uint32_t c = 0; c < nch && !dead; ++c) {
            uint4 nxt = cur;
            if (c + 1 < nch) nxt = load_chunk(tv.src + (size_t)(c + 1) * kWave + lane);
            const uint32_t cols = min((uint32_t)kChunk, len2 - c * kChunk);
            // 16 columns = 4 dwords x 4 bytes with COMPILE-TIME byte positions: `((dw >> 8k) & 0xFF) * stride` is one
            // v_mul_u32_u24_sdwa; walking the chunk with a running byte shift cost three v_alignbit_b32 + a shift + an and + a
            // v_mul_lo_u32 per column -- six instructions, five of them half-rate on gfx950 (profiles/issue_rates_r02.txt).
            // Columns go in runs of 8 (the reference's break test, :568-570 / :607-609, is evaluated per wavefront after each run);
            // a run that lies entirely on the diagonal walk and inside the chunk -- the common case -- is a straight line with no
            // per-column scalar tests: the first version spent 340 scalar instructions per tile next to 550 vector ones, and four
            // SIMDs share one scalar unit.
            const uint32_t dws[4] = {cur.x, cur.y, cur.z, cur.w};
            auto column = [&](uint32_t sym, uint32_t j, bool diagonal) {
                const uint32_t* row = lds_band + sym * stride + (v >> 5);
                const uint32_t d0w = row[0], d1w = row[1], d2w = row[2];
                const uint32_t sh = v & 31;
                const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(d2w, d1w, sh) << 32) | __builtin_amdgcn_alignbit(d1w, d0w, sh);
                const uint64_t sum = (x & vp) + vp;
                const uint64_t e = lut3<T_XOR_OR>(sum, vp, x);
                const uint64_t d0 = e | vn;                     // levenshtein.rs:556 / :593
                const uint64_t hp = lut3<T_OR_NOR>(vn, e, vp);  // vn | ~(d0 | vp)
                const uint64_t hn = e & vp;                     // d0 & vp (vp & vn == 0)
                if (diagonal) {                                 // :560-562
                    diag_hits += (uint32_t)(d0 >> 63);
                } else {                                        // :597-600: the last row, at a bit that moves down
                    const uint64_t hmask = 1ull << (62 - (j - first));
                    score += (hp & hmask) != 0;
                    score -= (hn & hmask) != 0;
                }
                const uint64_t d0s = d0 >> 1;
                vp = lut3<T_OR_NOR>(hn, d0s, hp);               // :571 / :611
                vn = d0s & hp;
                ++v;
            };
#pragma unroll
            for (int run = 0; run < 2; ++run) {
                const uint32_t b0 = (uint32_t)run * 8, j0 = c * kChunk + b0;
                if (b0 >= cols || dead) break;  // wavefront-uniform
                if (b0 + 8 <= cols && j0 + 8 <= first) {
                    if (asm_run) {
                        // the eight columns as one asm block (tools/gen_band_asm.py: 23 VALU per column against the ~30 hipcc writes); the diagonal bits come back
                        // as a shift register
                        uint32_t vpl = (uint32_t)vp, vph = (uint32_t)(vp >> 32), vnl = (uint32_t)vn, vnh = (uint32_t)(vn >> 32), acc = 0;
                        asm volatile(RF_BAND_RUN8_ASM
                                     : [vpl] "+v"(vpl), [vph] "+v"(vph), [vnl] "+v"(vnl), [vnh] "+v"(vnh), [acc] "+v"(acc)
                                     : [dw0] "v"(dws[run * 2]), [dw1] "v"(dws[run * 2 + 1]), [pitch] "v"(pitch_bytes), [v0] "s"(uniform(v))
                                     : RF_BAND_RUN8_CLOBBERS);
                        vp = ((uint64_t)vph << 32) | vpl;
                        vn = ((uint64_t)vnh << 32) | vnl;
                        diag_hits += (uint32_t)__popc(acc & 0xFFu);
                        v += 8;
                    } else {
#pragma unroll
                        for (int kb = 0; kb < 8; ++kb) column((dws[run * 2 + kb / 4] >> (8 * (kb % 4))) & 0xFFu, j0 + kb, true);
                    }
                    score = k + (j0 + 8) - diag_hits;           // :561: the running total of the diagonal walk
                } else {
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) {
                        if (b0 + kb < cols) {
                            const uint32_t j = j0 + kb;
                            const bool diagonal = j < first;
                            column((dws[run * 2 + kb / 4] >> (8 * (kb % 4))) & 0xFFu, j, diagonal);
                            if (diagonal) score = k + (j + 1) - diag_hits;
                        }
                    }
                }
                if (__ballot(valid && score <= break_score) == 0) dead = true;
            }
            cur = nxt;
        }
        if (valid) {
            if (dead || score > k)
                emit_none(p, idx);
            else
                emit_usize(p, score, len2, idx);
        }
    }
}

hipError_t launch_band(const ScanParams& p, hipStream_t stream)
{
    if (p.tile_end <= p.tile_begin) return hipSuccess;
    const size_t lds = (size_t)256 * band_row_dwords(p.words) * sizeof(uint32_t);
    // every workgroup stages the 32-bit band table (256 x (2W + 5) dwords: 13 KiB at 256 symbols) before its first tile, and a
    // tile dies after 8-16 columns: half the general launches' workgroups per CU measured best (configs[2] corpus, cutoff 8:
    // 8 or 16 per CU 76.7 Gpairs/s, 24: 72.7, 32: 67.8, 64: 55.5)
    const int band_grid = std::max(1, std::min(scan_grid(p.tile_end - p.tile_begin), (scan_max_grid() + 1) / 2));
    const dim3 g(band_grid), b(kWave * kWavesPerBlock);
    static const bool use_asm = [] { const char* e = getenv("RF_ASM_BAND"); return !e || atoi(e) != 0; }();
    ScanParams pa = p;
    pa.band_asm = use_asm ? 1u : 0u;
    auto kern = p.tiles ? band_kernel<false> : band_kernel<true>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, g, b, lds, stream, pa);
    return hipGetLastError();
}

}  // namespace rf
