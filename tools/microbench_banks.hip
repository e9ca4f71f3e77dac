// VGPR bank-conflict probe: does v_bitop3_b32 (3 VGPR sources) slow down when the sources share a register bank
// (bank = index mod 4)?  Explicit registers inside one asm block (v0..v31 clobbered).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define R8(a, b, c)                                   \
    "v_bitop3_b32 v16, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v17, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v18, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v19, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v20, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v21, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v22, " a ", " b ", " c " bitop3:0x96\n" \
    "v_bitop3_b32 v23, " a ", " b ", " c " bitop3:0x96\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23"

#define KERN(name, BODY)                                                              \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters)            \
    {                                                                                 \
        for (int i = 0; i < iters; ++i) asm volatile(BODY BODY BODY BODY ::: CLOB);  \
        if (iters < 0) out[0] = 1;                                                    \
    }
KERN(k_same, R8("v0", "v4", "v8"))     // banks 0,0,0
KERN(k_two, R8("v0", "v4", "v9"))      // banks 0,0,1
KERN(k_diff, R8("v1", "v6", "v11"))    // banks 1,2,3
// 2-source VOP2 forms
#define A8(a, b)                       \
    "v_and_b32 v16, " a ", " b "\n" "v_and_b32 v17, " a ", " b "\n" "v_and_b32 v18, " a ", " b "\n" "v_and_b32 v19, " a ", " b "\n" \
    "v_and_b32 v20, " a ", " b "\n" "v_and_b32 v21, " a ", " b "\n" "v_and_b32 v22, " a ", " b "\n" "v_and_b32 v23, " a ", " b "\n"
KERN(k_and_same, A8("v0", "v4"))
KERN(k_and_diff, A8("v1", "v6"))
// dependent chain of length 8 on one register vs 8 independent
#define D8 "v_and_b32 v16, v16, v1\n" "v_and_b32 v16, v16, v2\n" "v_and_b32 v16, v16, v3\n" "v_and_b32 v16, v16, v5\n" \
           "v_and_b32 v16, v16, v6\n" "v_and_b32 v16, v16, v7\n" "v_and_b32 v16, v16, v9\n" "v_and_b32 v16, v16, v10\n"
KERN(k_and_dep, D8)

typedef void (*kern_t)(uint32_t*, int);
static void run(const char* name, kern_t k, uint32_t* d, int per_iter, int blocks_per_cu)
{
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %d blk/CU %8.3f ms  %8.2f wave-instr/ns\n", name, blocks_per_cu, ms, (double)blocks * 4 * iters * per_iter / (ms * 1e6));
}
int main()
{
    uint32_t* d; hipMalloc(&d, 64);
    for (int b : {8, 2}) {
        run("bitop3 same", k_same, d, 32, b);
        run("bitop3 two", k_two, d, 32, b);
        run("bitop3 diff", k_diff, d, 32, b);
        run("and same", k_and_same, d, 32, b);
        run("and diff", k_and_diff, d, 32, b);
        run("and dep", k_and_dep, d, 32, b);
    }
    return 0;
}
