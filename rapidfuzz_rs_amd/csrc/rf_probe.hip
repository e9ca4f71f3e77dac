// rf_probe.hip -- measurement probes exported through the C ABI (rf_probe_issue_rate).
//
// The single-word scans are bound by VALU issue, not by HBM, so the honest yardstick next to the HBM
// roofline is "how fast does this chip run the very same column code with nothing else in the way".  The probe kernels
// below instantiate the PRODUCT recurrence states (rf_device.hpp State::step -- not a copy of them) on PM words that
// come from registers: no HBM traffic, no LDS gather, no byte extraction, no tile bookkeeping.  bench.py calls the probe
// in the same process and reports `roofline.issue_bound` from it, so the ceiling can never go stale against the kernel.
#include "rf_device.hpp"

namespace rf {

template <class State>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void probe_regs_kernel(uint32_t* out, int iters, uint32_t seed)
{
    using Word = typename State::Word;
    constexpr int W = State::kWords;
    State st;
    st.init();
    Word x[4][W];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < W; ++w) x[j][w] = (Word)((threadIdx.x + 1) * 0x9E3779B97F4A7C15ull * (2 * j + 3) + seed + w);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // 16 columns per iteration, like one chunk
            st.step(x[0]);
            st.step(x[1]);
            st.step(x[2]);
            st.step(x[3]);
        }
    }
    if (st.result(64 * W, 0) == 0x12345678u) out[0] = 1;  // keeps the state live
}

template <class State>
static hipError_t probe_run(int blocks_per_cu, int iters, double* wave_columns_per_ns)
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
    const dim3 g(cus * blocks_per_cu), b(kWave * kWavesPerBlock);
    float ms = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL((probe_regs_kernel<State>), g, b, 0, 0, d_out, iters / 8, 1u);  // warm-up (clocks, code upload)
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipEventRecord(e0, 0);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((probe_regs_kernel<State>), g, b, 0, 0, d_out, iters, 2u);
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

hipError_t launch_probe(RawKind raw, uint32_t len1, int blocks_per_cu, int iters, double* wave_columns_per_ns)
{
    switch (raw) {
    case RAW_LEV:
        if (len1 <= 32) return probe_run<Lev32State>(blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 64) return probe_run<LevState<1>>(blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 128) return probe_run<LevState<2>>(blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 256) return probe_run<LevState<4>>(blocks_per_cu, iters, wave_columns_per_ns);
        return hipErrorInvalidValue;
    case RAW_LCS:
        if (len1 <= 32) return probe_run<Lcs32State>(blocks_per_cu, iters, wave_columns_per_ns);
        if (len1 <= 64) return probe_run<LcsState<1>>(blocks_per_cu, iters, wave_columns_per_ns);
        return hipErrorInvalidValue;
    case RAW_OSA:
        if (len1 <= 64) return probe_run<OsaState<1>>(blocks_per_cu, iters, wave_columns_per_ns);
        return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rf
