// rf_probe.hip -- measurement probes exported through the C ABI (rf_probe_issue_rate).
//
// The single-word scans are bound by VALU issue, not by HBM, so the honest yardstick next to the HBM
// roofline is "how fast does this chip run the very same column code with nothing else in the way".  The probe kernels
// below instantiate the PRODUCT recurrence states (rf_device.hpp State::step -- not a copy of them) on PM words that
// come from registers: no HBM traffic, no LDS gather, no byte extraction, no tile bookkeeping.  bench.py calls the probe
// in the same process and reports `roofline.issue_bound` from it, so the ceiling can never go stale against the kernel.
#include "rf_device.hpp"
#include "rf_internal.hpp"

namespace rf {

template <class State>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void probe_regs_kernel(uint32_t* out, int iters, uint32_t seed)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    State st;
    st.init();
    constexpr int kX = W == 1 ? 4 : 2;  // distinct pattern rows in rotation (multi-word states: fewer, for the register budget)
    Word x[kX][W];
#pragma unroll
    for (int j = 0; j < kX; ++j)
#pragma unroll
        for (int w = 0; w < W; ++w) x[j][w] = (Word)((threadIdx.x + 1) * 0x9E3779B97F4A7C15ull * (2 * j + 3) + seed + w);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / kX; ++r)  // 16 columns per iteration, like one chunk
#pragma unroll
            for (int j = 0; j < kX; ++j) st.step(x[j]);
    }
    if ((int)threadIdx.x == iters) out[0] = st.result(64 * W, 0);  // never true (iters >> 256), but the compiler cannot know: keeps the state live
}

// The same columns fed the way the scans feed them: 16 symbol bytes per lane in four dwords, byte extraction, the PM
// row gathered from LDS (process_chunk_full -- the scans' own chunk code, software pipelining included) -- but still no
// HBM traffic and no tile loop.  The symbols are uniformly random over `symbols` table rows (fixed per lane).
template <class State>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void probe_lds_kernel(uint32_t* out, int iters, uint32_t seed, uint32_t symbols)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    __shared__ Word lds_pm[256 * W];
    for (int i = threadIdx.x; i < 256 * W; i += kWave * kWavesPerBlock) lds_pm[i] = (Word)((i + 1) * 0x9E3779B97F4A7C15ull + seed);
    __syncthreads();
    uint32_t h = (threadIdx.x + 1) * 2654435761u + seed;
    uint32_t dw[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            h = h * 1664525u + 1013904223u;
            v |= ((h >> 16) % symbols) << (8 * b);
        }
        dw[d] = v;
    }
    const uint4 chunk = make_uint4(dw[0], dw[1], dw[2], dw[3]);
    State st;
    st.init();
    for (int i = 0; i < iters; ++i) {
        process_chunk_full<State>(st, lds_pm, chunk);
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((int)threadIdx.x == iters) out[0] = st.result(64 * W, 0);
}

template <class State>
static hipError_t probe_run(uint32_t mode, int blocks_per_cu, int iters, double* wave_columns_per_ns)
{
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    uint32_t* d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    e = hipMalloc((void**)&d_out, 64);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    // Exactly as many workgroups as stay resident at once: a grid of 8 per CU for a kernel of which 7 fit (the multi-word states)
    // runs its last workgroup per CU alone, and the "ceiling" then reads BELOW the real kernel (round 2: configs[2] 1.15 of it).
    int resident = blocks_per_cu;
    if (mode != 2) {
        int occ = 0;
        const void* fn = mode == 0 ? reinterpret_cast<const void*>(probe_regs_kernel<State>) : reinterpret_cast<const void*>(probe_lds_kernel<State>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kWave * kWavesPerBlock, 0) == hipSuccess && occ > 0) resident = std::min(resident, occ);
    }
    const dim3 g(cus * resident), b(kWave * kWavesPerBlock);
    auto launch = [&](int n, uint32_t seed) {
        if (mode == 0)
            hipLaunchKernelGGL((probe_regs_kernel<State>), g, b, 0, 0, d_out, n, seed);
        else if (mode == 2)
            launch_lev1_asm_probe(g, b, d_out, n, seed);
        else
            hipLaunchKernelGGL((probe_lds_kernel<State>), g, b, 0, 0, d_out, n, seed, 62u);
    };
    float ms = 0;
    if (e == hipSuccess) {
        launch(iters / 8, 1u);  // warm-up (clocks, code upload)
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipEventRecord(e0, 0);
    if (e == hipSuccess) {
        launch(iters, 2u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, 0);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && ms > 0) {
        const double cols = (double)g.x * kWavesPerBlock * (double)iters * 16.0;
        *wave_columns_per_ns = cols / ((double)ms * 1e6);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_out) (void)hipFree(d_out);
    return e;
}

hipError_t launch_probe(RawKind raw, uint32_t len1, uint32_t mode, int blocks_per_cu, int iters, double* wave_columns_per_ns)
{
    if (mode > 2) return hipErrorInvalidValue;
    if (mode == 2 && !(raw == RAW_LEV && len1 > 32 && len1 <= 64)) return hipErrorInvalidValue;  // the asm chunk exists for LevState<1> only
    switch (raw) {
    case RAW_LEV:
        if (len1 <= 32) return probe_run<Lev32State>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 64) return probe_run<LevState<1>>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 128) return probe_run<LevState<2>>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 256) return probe_run<LevState<4>>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        return hipErrorInvalidValue;
    case RAW_LCS:
        if (len1 <= 32) return probe_run<Lcs32State>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 64) return probe_run<LcsState<1>>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        return hipErrorInvalidValue;
    case RAW_OSA:
        if (len1 <= 64) return probe_run<OsaState<1>>(mode, blocks_per_cu, iters, wave_columns_per_ns);
        return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
}

// ---- the core clock the chip actually runs at, sampled WHILE something else runs (rf_probe_core_clock) -------------------------
// The issue ceiling above is measured with idle HBM; the scans stream HBM, and the chip then clocks lower (power management):
// GRBM_GUI_ACTIVE / duration reads 2.39 GHz for the probes and 2.05-2.16 GHz for the streaming scans, for the SAME cycle count
// (tools/clock_of.sh).  One wavefront on a stream of its own sleeps `sleeps` x s_sleep 127 (64 core cycles per unit) between two
// readings of the constant-rate 100 MHz counter (s_memrealtime) and of s_memtime: bench.py launches it beside back-to-back scans.
__global__ void core_clock_kernel(uint64_t* out, uint32_t sleeps)
{
    const uint64_t r0 = wall_clock64(), c0 = clock64();
    for (uint32_t i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    const uint64_t r1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = r1 - r0;
        out[1] = c1 - c0;
    }
}
hipError_t launch_core_clock(uint64_t* d_out, uint32_t sleeps, hipStream_t stream)
{
    hipLaunchKernelGGL(core_clock_kernel, dim3(1), dim3(64), 0, stream, d_out, sleeps);
    return hipGetLastError();
}

}  // namespace rf
