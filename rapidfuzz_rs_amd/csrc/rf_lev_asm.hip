// rf_lev_asm.hip -- the headline scan (one query of <= 64 symbols against a single-length corpus whose length is a multiple of
// 16, no cutoff early-out) with its 16-column chunk as ONE hand-scheduled asm block.
//
// Why: the recurrence column is a dependent chain that mixes full-rate 32-bit ops with half-rate 64-bit ones, and on gfx950 a
// half-rate VALU instruction issued right behind the instructions that feed it costs SIMD issue time that eight resident
// wavefronts do not hide -- an s_nop 0 in the right places buys it back (profiles/issue_rates_r02.txt,
// profiles/lev_schedule_experiments_r02.txt).  hipcc cannot be told where those places are: 32 compiler-visible variants of nop
// placement all landed within 2 % of each other, because the scheduler moves everything else around them.  So this kernel pins
// the chunk -- byte extraction, the 8-column look-ahead of LDS table reads carried ACROSS chunks with counted lgkmcnt waits, the
// 16 columns and their s_nops -- on physical VGPRs v30..v63 (tools/gen_lev_chunk_asm.py writes rf_lev_chunk_asm.inc).
// Measured against stream_kernel_occ8<LevState<1>, true, 1> on the same box: 44.5 vs 42.4 Gpairs/s (100 M x len-64).
//
// Everything around the chunk -- the fetch ring, the tile loop, results, in-scan top-k -- is stream_body (rf_scan.hip) again,
// written with macros instead of lambdas because the recurrence state and the look-ahead rows live in register-asm variables
// (their address cannot be taken).  Those variables are touched by asm statements ONLY: one C++ use and hipcc keeps them in
// other registers and copies them in and out around every block (measured: 8 v_mov per chunk).
//
// The pattern table must sit at LDS address 0 (the asm's ds_read addresses are symbol * 8 with no base): the kernel has exactly
// one LDS object with the table first.  lgkmcnt: the block waits with lgkmcnt(7) for the oldest of its 8 outstanding reads; LDS
// returns in order, so any other outstanding LGKM operation can only make that wait longer, never shorter than needed.
#include "rf_device.hpp"
#include "rf_lev_chunk_asm.inc"

#ifndef RF_LEV_CHUNK_VARIANT
#define RF_LEV_CHUNK_VARIANT RF_LEV_CHUNK_ASM_WG
#endif

namespace rf {

struct LevAsmLds {
    uint64_t pm[256];  // LDS address 0
    uint64_t topk[kWavesPerBlock][kWave];
};

__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void lev1_asm_kernel(const ScanParams p)
{
    __shared__ LevAsmLds lds;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds.pm[p.sigma[i]] = p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    WaveTopK best;
    best.init();
    uint64_t limit = ~0ull;
    if (topk) topk_refresh_bound(p, limit);
    uint32_t tiles_done = 0;

    uint32_t t = p.tile_begin + (blockIdx.x * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        const uint32_t len2 = p.uniform_len;
        const uint32_t nch = len2 / kChunk;  // the launcher sends only whole-chunk lengths >= 16 here
        const uint32_t three = 3u;
        // D[len1][len2] = len2 + popcount(VP & valid) - popcount(VN & valid)  (LevState::result)
        const uint64_t valid_rows = p.len1 >= 64 ? ~0ull : ((1ull << p.len1) - 1);
        const uint32_t valid_lo = (uint32_t)valid_rows, valid_hi = (uint32_t)(valid_rows >> 32);

        // fetch cursor: chunk loads run two chunks ahead of the chunk being processed, across tile boundaries
        uint32_t ft = t, fc = 0;
        const uint4* fsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)ft * p.uniform_tile_bytes);
#define RF_FETCH(dst)                                                                                \
    {                                                                                                \
        dst = load_chunk(fsrc + (size_t)fc * kWave + lane);                                          \
        if (++fc == nch) {                                                                           \
            const uint32_t nt = ft + stride;                                                         \
            if (nt < p.tile_end) {                                                                   \
                ft = nt;                                                                             \
                fsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)ft * p.uniform_tile_bytes); \
                fc = 0;                                                                              \
            } else {                                                                                 \
                fc = nch - 1; /* parked on the last valid chunk: a cached re-read */                  \
            }                                                                                        \
        }                                                                                            \
    }
        // ring of three chunk buffers with static names: the chunk being processed, the next one (the table rows of its first 8
        // columns are gathered while this one is computed) and the load in flight
        uint4 b0, b1, b2;
        RF_FETCH(b0);
        RF_FETCH(b1);

        // the recurrence state of the tile in flight (levenshtein.rs:454-455) and the look-ahead table rows
        register uint32_t vpl asm("v60"), vph asm("v61"), vnl asm("v62"), vnh asm("v63");
        register uint32_t r34 asm("v34"), r35 asm("v35"), r36 asm("v36"), r37 asm("v37"), r38 asm("v38"), r39 asm("v39"), r40 asm("v40"), r41 asm("v41");
        register uint32_t r42 asm("v42"), r43 asm("v43"), r44 asm("v44"), r45 asm("v45"), r46 asm("v46"), r47 asm("v47"), r48 asm("v48"), r49 asm("v49");
#define RF_ROWS(c) c(r34), c(r35), c(r36), c(r37), c(r38), c(r39), c(r40), c(r41), c(r42), c(r43), c(r44), c(r45), c(r46), c(r47), c(r48), c(r49)
#define RF_OUT(r) "=v"(r)
#define RF_INOUT(r) "+v"(r)
#define RF_STATE_INIT asm volatile("v_mov_b32 %0, -1\n\tv_mov_b32 %1, -1\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(vpl), "=v"(vph), "=v"(vnl), "=v"(vnh))
        RF_STATE_INIT;
        asm volatile("s_waitcnt lgkmcnt(0)\n\t" RF_LEV_PROLOGUE_ASM : RF_ROWS(RF_OUT) : [c0] "v"(b0.x), [c1] "v"(b0.y), [k3] "v"(three) : "v30", "v31", "v32", "v33");
        uint32_t c = 0;
        bool done = false;

#define RF_STEP(use, refill, nextuse)                                                                                 \
    {                                                                                                                 \
        RF_FETCH(refill);                                                                                             \
        asm volatile(RF_LEV_CHUNK_VARIANT                                                                             \
                     : "+v"(vpl), "+v"(vph), "+v"(vnl), "+v"(vnh), RF_ROWS(RF_INOUT)                                  \
                     : [c2] "v"(use.z), [c3] "v"(use.w), [n0] "v"(nextuse.x), [n1] "v"(nextuse.y), [k3] "v"(three)    \
                     : RF_LEV_CHUNK_CLOBBERS);                                                                        \
        if (++c >= nch) { /* tile finished */                                                                         \
            uint32_t pp, pn, tmp;                                                                                     \
            asm volatile("v_and_b32 %0, %7, %3\n\tv_and_b32 %2, %8, %4\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_bcnt_u32_b32 %0, %2, %0\n\t" \
                         "v_and_b32 %1, %7, %5\n\tv_and_b32 %2, %8, %6\n\tv_bcnt_u32_b32 %1, %1, 0\n\tv_bcnt_u32_b32 %1, %2, %1"     \
                         : "=&v"(pp), "=&v"(pn), "=&v"(tmp)                                                           \
                         : "v"(vpl), "v"(vph), "v"(vnl), "v"(vnh), "s"(valid_lo), "s"(valid_hi));                     \
            const uint32_t idx = t * kWave + lane;                                                                    \
            const bool valid = idx < p.n;                                                                             \
            const uint32_t raw = len2 + pp - pn;                                                                      \
            if (p.out && valid) emit_usize(p, raw, len2, idx);                                                        \
            if (topk) {                                                                                               \
                bool keep;                                                                                            \
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);                                          \
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);            \
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);  \
                if ((++tiles_done & 7u) == 0) topk_refresh_bound(p, limit); /* see stream_body */                     \
            }                                                                                                         \
            t += stride;                                                                                              \
            if (t >= p.tile_end) {                                                                                    \
                done = true;                                                                                          \
            } else {                                                                                                  \
                c = 0;                                                                                                \
                RF_STATE_INIT;                                                                                        \
            }                                                                                                         \
        }                                                                                                             \
    }
        while (true) {
            RF_STEP(b0, b2, b1);
            if (done) break;
            RF_STEP(b1, b0, b2);
            if (done) break;
            RF_STEP(b2, b1, b0);
            if (done) break;
        }
#undef RF_STEP
#undef RF_STATE_INIT
#undef RF_INOUT
#undef RF_OUT
#undef RF_ROWS
#undef RF_FETCH
    }
    if (topk) topk_block_publish(p, best, lds.topk, wave, lane, limit);
}

// ---- OSA (OsaState<1>): the same kernel around the OSA column; the previous column's table row stays in the look-ahead ring ----
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void osa1_asm_kernel(const ScanParams p)
{
    __shared__ LevAsmLds lds;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds.pm[p.sigma[i]] = p.pm[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    WaveTopK best;
    best.init();
    uint64_t limit = ~0ull;
    if (topk) topk_refresh_bound(p, limit);
    uint32_t tiles_done = 0;

    uint32_t t = p.tile_begin + (blockIdx.x * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        const uint32_t len2 = p.uniform_len;
        const uint32_t nch = len2 / kChunk;  // the launcher sends only whole-chunk lengths >= 16 here
        const uint32_t three = 3u;
        // D[len1][len2] = len2 + popcount(VP & valid) - popcount(VN & valid)  (LevState::result)
        const uint64_t valid_rows = p.len1 >= 64 ? ~0ull : ((1ull << p.len1) - 1);
        const uint32_t valid_lo = (uint32_t)valid_rows, valid_hi = (uint32_t)(valid_rows >> 32);

        // fetch cursor: chunk loads run two chunks ahead of the chunk being processed, across tile boundaries
        uint32_t ft = t, fc = 0;
        const uint4* fsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)ft * p.uniform_tile_bytes);
#define RF_FETCH(dst)                                                                                \
    {                                                                                                \
        dst = load_chunk(fsrc + (size_t)fc * kWave + lane);                                          \
        if (++fc == nch) {                                                                           \
            const uint32_t nt = ft + stride;                                                         \
            if (nt < p.tile_end) {                                                                   \
                ft = nt;                                                                             \
                fsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)ft * p.uniform_tile_bytes); \
                fc = 0;                                                                              \
            } else {                                                                                 \
                fc = nch - 1; /* parked on the last valid chunk: a cached re-read */                  \
            }                                                                                        \
        }                                                                                            \
    }
        // ring of three chunk buffers with static names: the chunk being processed, the next one (the table rows of its first 8
        // columns are gathered while this one is computed) and the load in flight
        uint4 b0, b1, b2;
        RF_FETCH(b0);
        RF_FETCH(b1);

        // the recurrence state of the tile in flight (levenshtein.rs:454-455) and the look-ahead table rows
        register uint32_t vpl asm("v60"), vph asm("v61"), vnl asm("v62"), vnh asm("v63"), d0l asm("v58"), d0h asm("v59");
        register uint32_t r34 asm("v34"), r35 asm("v35"), r36 asm("v36"), r37 asm("v37"), r38 asm("v38"), r39 asm("v39"), r40 asm("v40"), r41 asm("v41");
        register uint32_t r42 asm("v42"), r43 asm("v43"), r44 asm("v44"), r45 asm("v45"), r46 asm("v46"), r47 asm("v47"), r48 asm("v48"), r49 asm("v49");
#define RF_ROWS(c) c(r34), c(r35), c(r36), c(r37), c(r38), c(r39), c(r40), c(r41), c(r42), c(r43), c(r44), c(r45), c(r46), c(r47), c(r48), c(r49)
#define RF_OUT(r) "=v"(r)
#define RF_INOUT(r) "+v"(r)
/* osa.rs:74-77, :125-135: VP all ones, VN = D0 = 0, and no previous column: its table row (v[48:49] at a block's start) is zero */ \
#define RF_STATE_INIT                                                                                                        \
    asm volatile("v_mov_b32 %0, -1\n\tv_mov_b32 %1, -1\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %5, 0\n\tv_mov_b32 %6, 0\n\tv_mov_b32 %7, 0" \
                 : "=v"(vpl), "=v"(vph), "=v"(vnl), "=v"(vnh), "=v"(d0l), "=v"(d0h), "=v"(r48), "=v"(r49))
        RF_STATE_INIT;
        asm volatile("s_waitcnt lgkmcnt(0)\n\t" RF_OSA_PROLOGUE_ASM : RF_ROWS(RF_OUT) : [c0] "v"(b0.x), [c1] "v"(b0.y), [k3] "v"(three) : "v30", "v31", "v32", "v33");
        uint32_t c = 0;
        bool done = false;

#define RF_STEP(use, refill, nextuse)                                                                                 \
    {                                                                                                                 \
        RF_FETCH(refill);                                                                                             \
        asm volatile(RF_OSA_CHUNK_ASM                                                                                 \
                     : "+v"(vpl), "+v"(vph), "+v"(vnl), "+v"(vnh), "+v"(d0l), "+v"(d0h), RF_ROWS(RF_INOUT)            \
                     : [c1] "v"(use.y), [c2] "v"(use.z), [c3] "v"(use.w), [n0] "v"(nextuse.x), [n1] "v"(nextuse.y), [k3] "v"(three) \
                     : RF_OSA_CHUNK_CLOBBERS);                                                                        \
        if (++c >= nch) { /* tile finished */                                                                         \
            uint32_t pp, pn, tmp;                                                                                     \
            asm volatile("v_and_b32 %0, %7, %3\n\tv_and_b32 %2, %8, %4\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_bcnt_u32_b32 %0, %2, %0\n\t" \
                         "v_and_b32 %1, %7, %5\n\tv_and_b32 %2, %8, %6\n\tv_bcnt_u32_b32 %1, %1, 0\n\tv_bcnt_u32_b32 %1, %2, %1"     \
                         : "=&v"(pp), "=&v"(pn), "=&v"(tmp)                                                           \
                         : "v"(vpl), "v"(vph), "v"(vnl), "v"(vnh), "s"(valid_lo), "s"(valid_hi));                     \
            const uint32_t idx = t * kWave + lane;                                                                    \
            const bool valid = idx < p.n;                                                                             \
            const uint32_t raw = len2 + pp - pn;                                                                      \
            if (p.out && valid) emit_usize(p, raw, len2, idx);                                                        \
            if (topk) {                                                                                               \
                bool keep;                                                                                            \
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);                                          \
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);            \
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);  \
                if ((++tiles_done & 7u) == 0) topk_refresh_bound(p, limit); /* see stream_body */                     \
            }                                                                                                         \
            t += stride;                                                                                              \
            if (t >= p.tile_end) {                                                                                    \
                done = true;                                                                                          \
            } else {                                                                                                  \
                c = 0;                                                                                                \
                RF_STATE_INIT;                                                                                        \
            }                                                                                                         \
        }                                                                                                             \
    }
        while (true) {
            RF_STEP(b0, b2, b1);
            if (done) break;
            RF_STEP(b1, b0, b2);
            if (done) break;
            RF_STEP(b2, b1, b0);
            if (done) break;
        }
#undef RF_STEP
#undef RF_STATE_INIT
#undef RF_INOUT
#undef RF_OUT
#undef RF_ROWS
#undef RF_FETCH
    }
    if (topk) topk_block_publish(p, best, lds.topk, wave, lane, limit);
}

// ---- the same kernel for queries of <= 32 symbols (Lev32State: 32-bit words, 4-byte table rows) ----
struct Lev32AsmLds {
    uint32_t pm[256];  // LDS address 0
    uint64_t topk[kWavesPerBlock][kWave];
};
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void lev32_asm_kernel(const ScanParams p)
{
    __shared__ Lev32AsmLds lds;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds.pm[p.sigma[i]] = (uint32_t)p.pm[i];  // the low half of each single-word entry
    __syncthreads();

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform(threadIdx.x / kWave);
    const uint32_t stride = gridDim.x * kWavesPerBlock * p.tile_step;
    const bool topk = p.topk_k != 0;
    WaveTopK best;
    best.init();
    uint64_t limit = ~0ull;
    if (topk) topk_refresh_bound(p, limit);
    uint32_t tiles_done = 0;

    uint32_t t = p.tile_begin + (blockIdx.x * kWavesPerBlock + wave) * p.tile_step;
    if (t < p.tile_end) {
        const uint32_t len2 = p.uniform_len;
        const uint32_t nch = len2 / kChunk;  // the launcher sends only whole-chunk lengths >= 16 here
        const uint32_t two = 2u;
        // D[len1][len2] = len2 + popcount(VP & valid) - popcount(VN & valid)  (LevState::result)
        const uint32_t valid_rows = p.len1 >= 32 ? ~0u : ((1u << p.len1) - 1);

        // fetch cursor: chunk loads run two chunks ahead of the chunk being processed, across tile boundaries
        uint32_t ft = t, fc = 0;
        const uint4* fsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)ft * p.uniform_tile_bytes);
#define RF_FETCH(dst)                                                                                \
    {                                                                                                \
        dst = load_chunk(fsrc + (size_t)fc * kWave + lane);                                          \
        if (++fc == nch) {                                                                           \
            const uint32_t nt = ft + stride;                                                         \
            if (nt < p.tile_end) {                                                                   \
                ft = nt;                                                                             \
                fsrc = reinterpret_cast<const uint4*>(p.data + (uint64_t)ft * p.uniform_tile_bytes); \
                fc = 0;                                                                              \
            } else {                                                                                 \
                fc = nch - 1; /* parked on the last valid chunk: a cached re-read */                  \
            }                                                                                        \
        }                                                                                            \
    }
        // ring of three chunk buffers with static names: the chunk being processed, the next one (the table rows of its first 8
        // columns are gathered while this one is computed) and the load in flight
        uint4 b0, b1, b2;
        RF_FETCH(b0);
        RF_FETCH(b1);

        // the recurrence state of the tile in flight (levenshtein.rs:454-455) and the look-ahead table rows
        register uint32_t vp asm("v60"), vn asm("v61");
        register uint32_t r34 asm("v34"), r35 asm("v35"), r36 asm("v36"), r37 asm("v37"), r38 asm("v38"), r39 asm("v39"), r40 asm("v40"), r41 asm("v41");
#define RF_ROWS(c) c(r34), c(r35), c(r36), c(r37), c(r38), c(r39), c(r40), c(r41)
#define RF_OUT(r) "=v"(r)
#define RF_INOUT(r) "+v"(r)
#define RF_STATE_INIT asm volatile("v_mov_b32 %0, -1\n\tv_mov_b32 %1, 0" : "=v"(vp), "=v"(vn))
        RF_STATE_INIT;
        asm volatile("s_waitcnt lgkmcnt(0)\n\t" RF_LEV32_PROLOGUE_ASM : RF_ROWS(RF_OUT) : [c0] "v"(b0.x), [c1] "v"(b0.y), [k2] "v"(two) : "v30", "v31", "v32", "v33");
        uint32_t c = 0;
        bool done = false;

#define RF_STEP(use, refill, nextuse)                                                                                 \
    {                                                                                                                 \
        RF_FETCH(refill);                                                                                             \
        asm volatile(RF_LEV32_CHUNK_ASM                                                                               \
                     : "+v"(vp), "+v"(vn), RF_ROWS(RF_INOUT)                                                          \
                     : [c2] "v"(use.z), [c3] "v"(use.w), [n0] "v"(nextuse.x), [n1] "v"(nextuse.y), [k2] "v"(two)      \
                     : RF_LEV32_CHUNK_CLOBBERS);                                                                      \
        if (++c >= nch) { /* tile finished */                                                                         \
            uint32_t pp, pn;                                                                                          \
            asm volatile("v_and_b32 %0, %4, %2\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_and_b32 %1, %4, %3\n\tv_bcnt_u32_b32 %1, %1, 0"   \
                         : "=&v"(pp), "=&v"(pn) : "v"(vp), "v"(vn), "s"(valid_rows));                                 \
            const uint32_t idx = t * kWave + lane;                                                                    \
            const bool valid = idx < p.n;                                                                             \
            const uint32_t raw = len2 + pp - pn;                                                                      \
            if (p.out && valid) emit_usize(p, raw, len2, idx);                                                        \
            if (topk) {                                                                                               \
                bool keep;                                                                                            \
                const uint32_t v = usize_value(p, raw, len2, &keep, p.len1);                                          \
                const uint64_t mine = ((uint64_t)(p.topk_desc ? ~v : v) << 32) | (p.key_index_base + idx);            \
                if (best.offer(mine, valid && keep, p.topk_k, lane, limit)) topk_list_changed(p, best, lane, limit);  \
                if ((++tiles_done & 7u) == 0) topk_refresh_bound(p, limit); /* see stream_body */                     \
            }                                                                                                         \
            t += stride;                                                                                              \
            if (t >= p.tile_end) {                                                                                    \
                done = true;                                                                                          \
            } else {                                                                                                  \
                c = 0;                                                                                                \
                RF_STATE_INIT;                                                                                        \
            }                                                                                                         \
        }                                                                                                             \
    }
        while (true) {
            RF_STEP(b0, b2, b1);
            if (done) break;
            RF_STEP(b1, b0, b2);
            if (done) break;
            RF_STEP(b2, b1, b0);
            if (done) break;
        }
#undef RF_STEP
#undef RF_STATE_INIT
#undef RF_INOUT
#undef RF_OUT
#undef RF_ROWS
#undef RF_FETCH
    }
    if (topk) topk_block_publish(p, best, lds.topk, wave, lane, limit);
}

hipError_t launch_lev1_asm(const ScanParams& p, hipStream_t stream, int grid)
{
    hipLaunchKernelGGL(lev1_asm_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}

// Measurement probe (rf_probe_issue_rate mode 2): the very asm block of the kernel above in a loop -- table rows gathered from
// LDS, look-ahead carried from iteration to iteration -- with the chunk's dwords constant: no HBM traffic, no tile loop.
__global__ __launch_bounds__(kWave* kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void lev1_asm_probe_kernel(uint32_t* out, int iters, uint32_t seed)
{
    __shared__ LevAsmLds lds;
    for (int i = threadIdx.x; i < 256; i += kWave * kWavesPerBlock) lds.pm[i] = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    __syncthreads();
    uint32_t h = (threadIdx.x + 1) * 2654435761u + seed, dw[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            h = h * 1664525u + 1013904223u;
            v |= ((h >> 16) % 62u) << (8 * b);
        }
        dw[d] = v;
    }
    const uint32_t three = 3u;
    register uint32_t vpl asm("v60"), vph asm("v61"), vnl asm("v62"), vnh asm("v63");
    register uint32_t r34 asm("v34"), r35 asm("v35"), r36 asm("v36"), r37 asm("v37"), r38 asm("v38"), r39 asm("v39"), r40 asm("v40"), r41 asm("v41");
    register uint32_t r42 asm("v42"), r43 asm("v43"), r44 asm("v44"), r45 asm("v45"), r46 asm("v46"), r47 asm("v47"), r48 asm("v48"), r49 asm("v49");
#define RF_ROWS(c) c(r34), c(r35), c(r36), c(r37), c(r38), c(r39), c(r40), c(r41), c(r42), c(r43), c(r44), c(r45), c(r46), c(r47), c(r48), c(r49)
#define RF_OUT(r) "=v"(r)
#define RF_INOUT(r) "+v"(r)
    asm volatile("v_mov_b32 %0, -1\n\tv_mov_b32 %1, -1\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(vpl), "=v"(vph), "=v"(vnl), "=v"(vnh));
    asm volatile("s_waitcnt lgkmcnt(0)\n\t" RF_LEV_PROLOGUE_ASM : RF_ROWS(RF_OUT) : [c0] "v"(dw[0]), [c1] "v"(dw[1]), [k3] "v"(three) : "v30", "v31", "v32", "v33");
    for (int i = 0; i < iters; ++i)
        asm volatile(RF_LEV_CHUNK_VARIANT
                     : "+v"(vpl), "+v"(vph), "+v"(vnl), "+v"(vnh), RF_ROWS(RF_INOUT)
                     : [c2] "v"(dw[2]), [c3] "v"(dw[3]), [n0] "v"(dw[0]), [n1] "v"(dw[1]), [k3] "v"(three)
                     : RF_LEV_CHUNK_CLOBBERS);
    uint32_t sink;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(sink) : "v"(vpl), "v"(vnh));
    if ((int)threadIdx.x == iters) out[0] = sink;  // never true, keeps the state live
#undef RF_INOUT
#undef RF_OUT
#undef RF_ROWS
}

void launch_lev1_asm_probe(dim3 g, dim3 b, uint32_t* out, int iters, uint32_t seed)
{
    hipLaunchKernelGGL(lev1_asm_probe_kernel, g, b, 0, 0, out, iters, seed);
}

hipError_t launch_osa1_asm(const ScanParams& p, hipStream_t stream, int grid)
{
    hipLaunchKernelGGL(osa1_asm_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_lev32_asm(const ScanParams& p, hipStream_t stream, int grid)
{
    hipLaunchKernelGGL(lev32_asm_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}

}  // namespace rf
