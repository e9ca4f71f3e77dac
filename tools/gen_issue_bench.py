"""Generates tools/microbench_issue.hip: issue-rate probes for the instructions of the bit-parallel columns (gfx950)."""
K = {}
def rep(lines, n=32):
    out = []
    while len(out) < n:
        out += lines
    return out[:n]
K["and_indep"] = rep([f"v_and_b32 v{16 + i}, v1, v6" for i in range(8)])
K["bitop3_indep_diffbank"] = rep([f"v_bitop3_b32 v{16 + i}, v1, v6, v11 bitop3:0x96" for i in range(8)])
K["bitop3_dep_chain"] = rep([f"v_bitop3_b32 v16, v16, v{5 + i}, v{10 + i} bitop3:0x96" for i in range(4)])
K["lshladd64_indep_cls01"] = rep([f"v_lshl_add_u64 v[{16 + 2 * i}:{17 + 2 * i}], v[0:1], 1, v[6:7]" for i in range(8)])
K["lshladd64_indep_cls00"] = rep([f"v_lshl_add_u64 v[{16 + 2 * i}:{17 + 2 * i}], v[0:1], 1, v[4:5]" for i in range(8)])
K["lshladd64_indep_const"] = rep([f"v_lshl_add_u64 v[{16 + 2 * i}:{17 + 2 * i}], v[0:1], 1, 1" for i in range(8)])
K["lshladd64_inplace_dep"] = rep(["v_lshl_add_u64 v[16:17], v[16:17], 1, v[6:7]"])
K["lshladd64_then_bitop3_dep"] = rep(["v_lshl_add_u64 v[16:17], v[16:17], 1, v[6:7]", "v_bitop3_b32 v16, v16, v5, v10 bitop3:0x96", "v_bitop3_b32 v17, v17, v6, v11 bitop3:0x96"], 33)
K["bitop3_then_lshladd64_dep"] = rep(["v_bitop3_b32 v16, v16, v5, v10 bitop3:0x96", "v_bitop3_b32 v17, v17, v6, v11 bitop3:0x96", "v_lshl_add_u64 v[16:17], v[16:17], 1, v[6:7]"], 33)
K["addco_addc_indep"] = rep([x for i in range(4) for x in (f"v_add_co_u32 v{16 + 2 * i}, vcc, v0, v6", f"v_addc_co_u32 v{17 + 2 * i}, vcc, v1, v7, vcc")])
K["add_u32_indep"] = rep([f"v_add_u32 v{16 + i}, v1, v6" for i in range(8)])
K["lshl_add_u32_indep"] = rep([f"v_lshl_add_u32 v{16 + i}, v1, 1, v6" for i in range(8)])
K["alignbit_indep"] = rep([f"v_alignbit_b32 v{16 + i}, v1, v6, 31" for i in range(8)])
K["sdwa_lshl_indep"] = rep([f"v_lshlrev_b32_sdwa v{16 + i}, v1, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}" for i in range(8)])
K["bfe_indep"] = rep([f"v_bfe_u32 v{16 + i}, v1, 8, 8" for i in range(8)])
K["bcnt_indep"] = rep([f"v_bcnt_u32_b32 v{16 + i}, v1, v6" for i in range(8)])

T3 = ["v_lshl_add_u64 v[16:17], v[16:17], 1, v[6:7]", "v_bitop3_b32 v16, v16, v5, v10 bitop3:0x96", "v_bitop3_b32 v17, v17, v6, v11 bitop3:0x96"]
K["trip_nop0_after64"] = rep([T3[0], "s_nop 0", T3[1], T3[2]], 32)
K["trip_nop1_after64"] = rep([T3[0], "s_nop 1", T3[1], T3[2]], 32)
K["trip_nop0_before64"] = rep([T3[0], T3[1], T3[2], "s_nop 0"], 32)
K["trip_nop0_both"] = rep([T3[0], "s_nop 0", T3[1], T3[2], "s_nop 0"], 30)
K["trip_indep_filler"] = rep([T3[0], "v_and_b32 v30, v1, v2", T3[1], T3[2]], 32)
K["pair_and_then_64_dep"] = rep(["v_and_b32 v16, v16, v5", "v_and_b32 v17, v17, v6", "v_lshl_add_u64 v[16:17], v[16:17], 1, v[6:7]"], 33)
K["addco_dep_pair"] = rep(["v_bitop3_b32 v16, v16, v5, v10 bitop3:0x96", "v_bitop3_b32 v17, v17, v6, v11 bitop3:0x96", "v_add_co_u32 v16, vcc, v16, v6", "v_addc_co_u32 v17, vcc, v17, v7, vcc"], 32)

def ind(fmt, n=8): return rep([fmt.format(d=16 + i, d2=f"[{16 + 2 * i}:{17 + 2 * i}]") for i in range(n)])
MORE = {
    "v_lshlrev_b32 (VOP2)": "v_lshlrev_b32 v{d}, 3, v6", "v_lshrrev_b32 (VOP2)": "v_lshrrev_b32 v{d}, 3, v6", "v_or_b32": "v_or_b32 v{d}, v1, v6",
    "v_xor_b32": "v_xor_b32 v{d}, v1, v6", "v_not_b32": "v_not_b32 v{d}, v6", "v_mov_b32": "v_mov_b32 v{d}, v6", "v_sub_u32": "v_sub_u32 v{d}, v1, v6",
    "v_add3_u32": "v_add3_u32 v{d}, v1, v6, v11", "v_and_or_b32": "v_and_or_b32 v{d}, v1, v6, v11", "v_or3_b32": "v_or3_b32 v{d}, v1, v6, v11",
    "v_lshl_or_b32": "v_lshl_or_b32 v{d}, v1, 3, v6", "v_xad_u32": "v_xad_u32 v{d}, v1, v6, v11", "v_bfi_b32": "v_bfi_b32 v{d}, v1, v6, v11",
    "v_perm_b32": "v_perm_b32 v{d}, v1, v6, v11", "v_ffbl_b32": "v_ffbl_b32 v{d}, v6", "v_ffbh_u32": "v_ffbh_u32 v{d}, v6",
    "v_min_u32": "v_min_u32 v{d}, v1, v6", "v_max_u32": "v_max_u32 v{d}, v1, v6", "v_cndmask_b32 (vcc)": "v_cndmask_b32 v{d}, v1, v6, vcc",
    "v_cmp_eq_u32 (vcc)": "v_cmp_eq_u32 vcc, v1, v6", "v_cmp_lt_u32 e64 (sgpr pair)": "v_cmp_lt_u32 s[20:21], v1, v6",
    "v_mul_lo_u32": "v_mul_lo_u32 v{d}, v1, v6", "v_mul_u32_u24": "v_mul_u32_u24 v{d}, v1, v6", "v_mad_u32_u24": "v_mad_u32_u24 v{d}, v1, v6, v11",
    "v_lshlrev_b64": "v_lshlrev_b64 v{d2}, 1, v[6:7]", "v_lshrrev_b64": "v_lshrrev_b64 v{d2}, 1, v[6:7]", "v_mov_b64": "v_mov_b64 v{d2}, v[6:7]",
    "v_mbcnt_lo_u32_b32": "v_mbcnt_lo_u32_b32 v{d}, v1, v6", "v_mov_b32 dpp row_shr:1": "v_mov_b32_dpp v{d}, v6 row_shr:1 row_mask:0xf bank_mask:0xf",
    "v_mov_b32 dpp wave_shr:1": "v_mov_b32_dpp v{d}, v6 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_and_b32 sdwa byte": "v_and_b32_sdwa v{d}, v1, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1",
    "v_cvt_f64_u32": "v_cvt_f64_u32 v{d2}, v6", "v_add_f64": "v_add_f64 v{d2}, v[0:1], v[6:7]", "v_mul_f64": "v_mul_f64 v{d2}, v[0:1], v[6:7]", "v_fma_f64": "v_fma_f64 v{d2}, v[0:1], v[6:7], v[8:9]",
    "v_add_f32": "v_add_f32 v{d}, v1, v6", "v_readlane_b32": "v_readlane_b32 s20, v6, 3", "v_readfirstlane_b32": "v_readfirstlane_b32 s20, v6",
    "v_pk_add_u16": "v_pk_add_u16 v{d}, v1, v6", "v_sad_u32": "v_sad_u32 v{d}, v1, v6, v11", "v_alignbyte_b32": "v_alignbyte_b32 v{d}, v1, v6, 1",
    "v_subrev_u32": "v_subrev_u32 v{d}, v1, v6", "v_ashrrev_i32": "v_ashrrev_i32 v{d}, 3, v6", "v_bfm_b32": "v_bfm_b32 v{d}, v1, v6",
}
K2 = {}
K["cndmask_e64_sgprpair"] = rep([f"v_cndmask_b32 v{16 + i}, v1, v6, s[20:21]" for i in range(8)])
K["cndmask_vcc_after_cmp"] = rep([x for i in range(4) for x in ("v_cmp_eq_u32 vcc, v1, v6", f"v_cndmask_b32 v{16 + i}, v1, v6, vcc")])
K["cndmask_sgpr_after_cmp"] = rep([x for i in range(4) for x in ("v_cmp_eq_u32 s[20:21], v1, v6", f"v_cndmask_b32 v{16 + i}, v1, v6, s[20:21]")])
K["cmp_then_4_cndmask"] = rep(["v_cmp_eq_u32 vcc, v1, v6"] + [f"v_cndmask_b32 v{16 + i}, v1, v6, vcc" for i in range(3)])
K["cmp_nop_cndmask"] = rep([x for i in range(4) for x in ("v_cmp_eq_u32 vcc, v1, v6", "s_nop 0", f"v_cndmask_b32 v{16 + i}, v1, v6, vcc")], 36)
K["cmp64_then_cndmask"] = rep([x for i in range(4) for x in ("v_cmp_eq_u64 vcc, v[0:1], v[6:7]", f"v_cndmask_b32 v{16 + i}, v1, v6, vcc")])
K["cmp_then_2_cndmask_vcc"] = rep(["v_cmp_eq_u32 vcc, v1, v6", "v_cndmask_b32 v16, v1, v6, vcc", "v_cndmask_b32 v17, v2, v7, vcc"], 33)
K["cmp_then_2_cndmask_e64"] = rep(["v_cmp_eq_u32 s[20:21], v1, v6", "v_cndmask_b32 v16, v1, v6, s[20:21]", "v_cndmask_b32 v17, v2, v7, s[20:21]"], 33)
K["cmp_2cnd_vcc_nop_between"] = rep(["v_cmp_eq_u32 vcc, v1, v6", "v_cndmask_b32 v16, v1, v6, vcc", "s_nop 0", "v_cndmask_b32 v17, v2, v7, vcc"], 32)
K["cmp_2cnd_vcc_valu_between"] = rep(["v_cmp_eq_u32 vcc, v1, v6", "v_cndmask_b32 v16, v1, v6, vcc", "v_and_b32 v20, v3, v4", "v_cndmask_b32 v17, v2, v7, vcc"], 32)
K["cnd_vcc_x8_valu_between"] = rep(["v_cndmask_b32 v16, v1, v6, vcc", "v_and_b32 v20, v3, v4"], 32)
K["cnd_vcc_x8_nop_between"] = rep(["v_cndmask_b32 v16, v1, v6, vcc", "s_nop 0"], 32)
K["cnd_vcc_dep_chain"] = rep(["v_cndmask_b32 v16, v16, v6, vcc"], 32)
K["cnd_vcc_distinct_dst"] = rep([f"v_cndmask_b32 v{16+i}, v{1+i}, v{8+i}, vcc" for i in range(8)], 32)
K["addc_chain_vcc"] = rep(["v_addc_co_u32 v16, vcc, v1, v6, vcc"], 32)
K["bfe_i32"] = rep([f"v_bfe_i32 v{16 + i}, v1, 3, 1" for i in range(8)])
K["sub_and_mask_trick"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, 1, v6", f"v_sub_u32 v{16 + i}, 0, v{16 + i}")])
for nm, fmt in MORE.items(): K2[nm] = ind(fmt)

def mix(nv, ns, sop):
    v = [f"v_and_b32 v{16 + i % 8}, v1, v6" for i in range(nv)]
    out = []
    per = nv // max(ns, 1) if ns else nv + 1
    k = 0
    for i, x in enumerate(v):
        out.append(x)
        if ns and (i + 1) % per == 0 and k < ns:
            out.append(sop.format(k=k % 4)); k += 1
    return out
for ns in (0, 2, 4, 8, 16):
    K[f"valu16_salu{ns}_s_add"] = mix(16, ns, "s_add_u32 s{k}, s{k}, 1") * 2
    K[f"valu16_salu{ns}_s_nop"] = mix(16, ns, "s_nop 0") * 2
K["salu_only_s_add"] = ["s_add_u32 s0, s0, 1", "s_add_u32 s1, s1, 1", "s_add_u32 s2, s2, 1", "s_add_u32 s3, s3, 1"] * 8
K["salu_only_s_nop"] = ["s_nop 0"] * 32
K["salu_only_waitcnt"] = ["s_waitcnt lgkmcnt(0)"] * 32
K["salu_dep_chain"] = ["s_add_u32 s0, s0, 1"] * 32
K["s_cmp_cbranch_like"] = ["s_cmp_lt_u32 s0, s1", "s_cselect_b32 s2, s0, s1"] * 16
clob = ",".join(f'"v{i}"' for i in range(0, 40)) + ',"vcc","scc","s0","s1","s2","s3","s20","s21"'
src = ["// GENERATED by tools/gen_issue_bench.py", "#include <hip/hip_runtime.h>", "#include <stdint.h>", "#include <stdio.h>", f"#define CLOB {clob}"]
import re as _re
ALL = dict(K)
for nm, lines in K2.items(): ALL["op_" + _re.sub(r"[^a-z0-9]+", "_", nm.lower()).strip("_")] = lines
LABEL = {("op_" + _re.sub(r"[^a-z0-9]+", "_", nm.lower()).strip("_")): nm for nm in K2}
for name, lines in ALL.items():
    body = "".join(f'        "{l}\\n"\n' for l in lines)
    src.append(f"__global__ __launch_bounds__(256) void k_{name}(uint32_t* out, int iters)\n{{\n    for (int i = 0; i < iters; ++i) asm volatile(\n{body}        ::: CLOB);\n    if (iters < 0) out[0] = 1;\n}}")
src.append('''typedef void (*kern_t)(uint32_t*, int);
static void run(const char* name, kern_t k, uint32_t* d, int per_iter, int blocks_per_cu)
{
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters / 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %d blk/CU %8.3f ms  %8.3f ns per wave-instruction per SIMD\\n", name, blocks_per_cu, ms, ms * 1e6 / ((double)blocks * 4 * iters * per_iter / 1024));
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 64);
    for (int b : {8}) {''')
for name, lines in ALL.items():
    src.append(f'        run("{LABEL.get(name, name)}", k_{name}, d, {len(lines)}, b);')
src.append("    }\n    return 0;\n}")
open(__file__.replace("gen_issue_bench.py", "microbench_issue.hip"), "w").write("\n".join(src) + "\n")
