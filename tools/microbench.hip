// tools/microbench.hip -- instruction-issue microbenchmarks for the ops of the Myers/Hyyro inner loop on
// gfx950.  Not part of the product; run on the GPU box to price design choices:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench && gpurun_out/microbench
// Each kernel runs `iters` x 32 copies of one instruction on 8 independent register sets per lane, with the
// whole chip resident (256 CUs x 32 waves); reported: wave-instructions per nanosecond and the implied
// cycles per wave-instruction per SIMD at the measured shader clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x x))

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed)            \
    {                                                                                               \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t b0 = a0 ^ 0x55, b1 = a1 ^ 0x33, b2 = a2 ^ 0x0f, b3 = a3 ^ 0xff, b4 = a4 ^ 1, b5 = a5 ^ 2, b6 = a6 ^ 4, b7 = a7 ^ 8; \
        uint64_t t0 = clock64();                                                                    \
        for (int i = 0; i < iters; ++i) { REP4(body) }                                              \
        uint64_t t1 = clock64();                                                                    \
        uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
        if (r == 0x12345678) out[0] = r;                                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[1] = (uint32_t)(t1 - t0); }                  \
    }

// 8 independent instructions per body (dependent only on their own register set)
#define I8(op) \
    asm volatile(op " %0, %0, %1" : "+v"(a0) : "v"(b0)); asm volatile(op " %0, %0, %1" : "+v"(a1) : "v"(b1)); \
    asm volatile(op " %0, %0, %1" : "+v"(a2) : "v"(b2)); asm volatile(op " %0, %0, %1" : "+v"(a3) : "v"(b3)); \
    asm volatile(op " %0, %0, %1" : "+v"(a4) : "v"(b4)); asm volatile(op " %0, %0, %1" : "+v"(a5) : "v"(b5)); \
    asm volatile(op " %0, %0, %1" : "+v"(a6) : "v"(b6)); asm volatile(op " %0, %0, %1" : "+v"(a7) : "v"(b7));
#define I8_3(op, tail) \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a0) : "v"(b0), "v"(a1)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a1) : "v"(b1), "v"(a2)); \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a2) : "v"(b2), "v"(a3)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a3) : "v"(b3), "v"(a4)); \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a4) : "v"(b4), "v"(a5)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a5) : "v"(b5), "v"(a6)); \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a6) : "v"(b6), "v"(a7)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a7) : "v"(b7), "v"(a0));

KERNEL(k_and, I8("v_and_b32"))
KERNEL(k_or3, I8_3("v_or3_b32", ""))
KERNEL(k_bitop3, I8_3("v_bitop3_b32", " bitop3:0xbe"))
KERNEL(k_bfi, I8_3("v_bfi_b32", ""))
KERNEL(k_alignbit, I8_3("v_alignbit_b32", ""))
KERNEL(k_lshl_or, I8_3("v_lshl_or_b32", ""))
KERNEL(k_add_u32, I8("v_add_u32"))

// 64-bit forms on register pairs
#define KERNEL64(name, body)                                                                        \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed)            \
    {                                                                                               \
        uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint64_t b0 = a0 ^ 0x55, b1 = a1 ^ 0x33, b2 = a2 ^ 0x0f, b3 = a3 ^ 0xff, b4 = a4 ^ 1, b5 = a5 ^ 2, b6 = a6 ^ 4, b7 = a7 ^ 8; \
        uint64_t t0 = clock64();                                                                    \
        for (int i = 0; i < iters; ++i) { REP4(body) }                                              \
        uint64_t t1 = clock64();                                                                    \
        uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
        if (r == 0x12345678) out[0] = (uint32_t)r;                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[1] = (uint32_t)(t1 - t0); }                  \
    }
#define J8(fmt) \
    asm volatile(fmt : "+v"(a0) : "v"(b0)); asm volatile(fmt : "+v"(a1) : "v"(b1)); asm volatile(fmt : "+v"(a2) : "v"(b2)); asm volatile(fmt : "+v"(a3) : "v"(b3)); \
    asm volatile(fmt : "+v"(a4) : "v"(b4)); asm volatile(fmt : "+v"(a5) : "v"(b5)); asm volatile(fmt : "+v"(a6) : "v"(b6)); asm volatile(fmt : "+v"(a7) : "v"(b7));
KERNEL64(k_lshl_add_u64, J8("v_lshl_add_u64 %0, %0, 1, %1"))
KERNEL64(k_lshlrev_b64, J8("v_lshlrev_b64 %0, 1, %0"))
// the two-instruction 64-bit add on explicit lo/hi registers (4 pairs = 8 wave-instructions per body)
#define P4 \
    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(b1) : "vcc"); \
    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a2), "+v"(a3) : "v"(b2), "v"(b3) : "vcc"); \
    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a4), "+v"(a5) : "v"(b4), "v"(b5) : "vcc"); \
    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a6), "+v"(a7) : "v"(b6), "v"(b7) : "vcc");
KERNEL(k_add_co_pair, P4)

// LDS: ds_read_b64 at pseudo-random 8-byte slots of a 2 KiB table (62 distinct symbols like the benchmark alphabet)
__global__ __launch_bounds__(256) void k_lds_b64(uint32_t* out, int iters, uint32_t seed)
{
    __shared__ uint64_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint32_t x = (threadIdx.x * 2654435761u) ^ seed;
    uint64_t acc = 0;
    uint64_t t0 = clock64();
    for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            acc += tab[48 + ((x >> 24) % 62)];  // address math is part of the cost here; see the ratio vs k_lds_same
        }
    }
    uint64_t t1 = clock64();
    if (acc == 0x12345678) out[0] = (uint32_t)acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (uint32_t)(t1 - t0);
}
__global__ __launch_bounds__(256) void k_lds_same(uint32_t* out, int iters, uint32_t seed)
{
    __shared__ uint64_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint32_t x = (threadIdx.x * 2654435761u) ^ seed;
    uint64_t acc = 0;
    uint64_t t0 = clock64();
    for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            acc += tab[(threadIdx.x & 31) + ((x >> 31) & 1)];  // conflict-free: lane l -> slot l (+0/1)
        }
    }
    uint64_t t1 = clock64();
    if (acc == 0x12345678) out[0] = (uint32_t)acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (uint32_t)(t1 - t0);
}

typedef void (*kern_t)(uint32_t*, int, uint32_t);

static int run(const char* name, kern_t k, int iters, double instr_per_iter_per_wave, uint32_t* d_out)
{
    const int blocks = 256 * 8, threads = 256;  // 32 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters / 8, 1u);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t h[2];
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    const double waves = (double)blocks * threads / 64;
    const double winstr = waves * iters * instr_per_iter_per_wave;
    const double per_ns = winstr / (ms * 1e6);
    // clock64() ticks at 100 MHz (s_memrealtime) on gfx9; derive the shader clock from nothing here -- report both
    const double cyc_at_2p4 = (1024.0 * 2.4) / per_ns;  // SIMD-cycles per wave-instruction if the clock were 2.4 GHz
    printf("%-16s %8.3f ms  %8.2f wave-instr/ns  => %5.2f cycles/wave-instr/SIMD @2.4GHz   (clock64 ticks %u)\n", name, ms, per_ns,
           cyc_at_2p4, h[1]);
    return 0;
}

int main_lev();
int main()
{
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemset(d_out, 0, 64));
    const int iters = 20000;
    run("v_and_b32", k_and, iters, 32, d_out);
    run("v_add_u32", k_add_u32, iters, 32, d_out);
    run("v_or3_b32", k_or3, iters, 32, d_out);
    run("v_bitop3_b32", k_bitop3, iters, 32, d_out);
    run("v_bfi_b32", k_bfi, iters, 32, d_out);
    run("v_alignbit_b32", k_alignbit, iters, 32, d_out);
    run("v_lshl_or_b32", k_lshl_or, iters, 32, d_out);
    run("v_lshl_add_u64", k_lshl_add_u64, iters, 32, d_out);
    run("v_lshlrev_b64", k_lshlrev_b64, iters, 32, d_out);
    run("add_co+addc", k_add_co_pair, iters, 32, d_out);
    run("ds_read_b64 rnd", k_lds_b64, iters / 10, 32, d_out);
    run("ds_read_b64 lin", k_lds_same, iters / 10, 32, d_out);
    return main_lev();
}

// ---- the Levenshtein column itself, without HBM: (a) PM word from registers, (b) PM word from LDS at a random
//      symbol per lane.  Reports wave-columns per ns (= Gpairs/s-equivalent x 64 columns / 64 lanes).
constexpr uint32_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <uint32_t TT>
__device__ __forceinline__ uint64_t lut3(uint64_t a, uint64_t b, uint64_t c)
{
    uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT & 0xFF);
    uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), TT & 0xFF);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ void lev_col(uint64_t& vp, uint64_t& vn, uint64_t x)
{
    const uint64_t sum = (x & vp) + vp;
    const uint64_t e = lut3<(TA ^ TB) | TC>(sum, vp, x);
    const uint64_t d0 = e | vn;
    const uint64_t hn = e & vp;
    const uint64_t hp = lut3<TA | (~(TB | TC))>(vn, d0, vp);
    uint64_t hps, hns;
    asm("v_lshl_add_u64 %0, %1, 1, 1" : "=v"(hps) : "v"(hp));
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(hns) : "v"(hn));
    vn = hps & d0;
    vp = lut3<TA | (~(TB | TC))>(hns, hps, d0);
}
__global__ __launch_bounds__(256) void k_lev_regs(uint32_t* out, int iters, uint32_t seed)
{
    uint64_t vp = ~0ull, vn = 0;
    uint64_t x0 = threadIdx.x * 0x9E3779B97F4A7C15ull + seed, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lev_col(vp, vn, x0); lev_col(vp, vn, x1); lev_col(vp, vn, x2); lev_col(vp, vn, x3);
        }
    }
    if ((vp ^ vn) == 0x12345678) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_lev_lds(uint32_t* out, int iters, uint32_t seed)
{
    __shared__ uint64_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint64_t vp = ~0ull, vn = 0;
    uint32_t w = (threadIdx.x * 2654435761u) ^ seed;
    for (int i = 0; i < iters; ++i) {
        // 16 columns from 4 dwords of pseudo-random alphanumeric-like symbols (48 + 0..61)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            w = w * 1664525u + 1013904223u;
            const uint32_t c0 = 48 + ((w >> 8) & 63) % 62, c1 = 48 + ((w >> 14) & 63) % 62, c2 = 48 + ((w >> 20) & 63) % 62, c3 = 48 + ((w >> 26) & 63) % 62;
            const uint64_t p0 = tab[c0], p1 = tab[c1], p2 = tab[c2], p3 = tab[c3];
            lev_col(vp, vn, p0); lev_col(vp, vn, p1); lev_col(vp, vn, p2); lev_col(vp, vn, p3);
        }
    }
    if ((vp ^ vn) == 0x12345678) out[0] = 1;
}
static int run_lev(const char* name, kern_t k, int iters, uint32_t* d_out, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters / 8, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cols = (double)blocks * threads / 64 * iters * 16;
    printf("%-14s %d blocks/CU %8.3f ms  %7.2f wave-columns/ns (= Gpairs/s for 64-column candidates)\n", name, blocks_per_cu, ms, cols / (ms * 1e6));
    return 0;
}
int main_lev()
{
    uint32_t* d_out;
    hipMalloc(&d_out, 64);
    for (int b : {2, 4, 8}) run_lev("lev col regs", k_lev_regs, 40000, d_out, b);
    for (int b : {2, 4, 8}) run_lev("lev col lds", k_lev_lds, 40000, d_out, b);
    return 0;
}
