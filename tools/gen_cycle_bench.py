"""Generates tools/microbench_cycles.hip: VALU issue cost in SHADER CYCLES (s_memtime), not nanoseconds.

profiles/issue_rates_r02.txt timed instruction streams with HIP events: ~1.0 ns per wavefront-instruction and SIMD for the
"full rate" forms -- 2.4-2.5 cycles if the chip ran at 2.38 GHz, 2.0 if the all-SIMDs-busy probe was itself power-limited to
~2.0 GHz.  MI355X_MICROARCH.md says a wave64 VALU instruction issues over 2 cycles.  This bench settles it: every wavefront
brackets its instruction stream with s_memtime (one tick = one shader cycle, MI355X_MICROARCH.md "Per-instruction cycle constants"),
so the result is cycles per wavefront-instruction per SIMD whatever the clock does, at 1 / 2 / 4 / 8 wavefronts per SIMD (dynamic LDS
sized so that exactly that many 256-thread workgroups fit a CU), next to the wall-clock figure of the same launch (-> the clock).

  python tools/gen_cycle_bench.py && hipcc --offload-arch=gfx950 -O2 -o tools/bin/microbench_cycles tools/microbench_cycles.hip
"""
import os

N = 256  # instructions per unrolled block


def rep(lines, n=N):
    out = []
    while len(out) < n:
        out += lines
    return out[:n]


K = {}
# one VGPR source / none / two in different banks / two in one bank / SGPR + VGPR / three sources
K["mov_v"] = rep([f"v_mov_b32 v{16 + i}, v{1 + i % 4}" for i in range(8)])
K["mov_const"] = rep([f"v_mov_b32 v{16 + i}, 0" for i in range(8)])
K["and_v1_v6"] = rep([f"v_and_b32 v{16 + i}, v1, v6" for i in range(8)])
K["and_v4_v8_samebank"] = rep([f"v_and_b32 v{16 + i}, v4, v8" for i in range(8)])
K["and_s_v"] = rep([f"v_and_b32 v{16 + i}, s20, v6" for i in range(8)])
K["and_e64"] = rep([f"v_and_b32_e64 v{16 + i}, v1, v6" for i in range(8)])
K["xor_dep_chain"] = rep(["v_xor_b32 v16, v16, v6"])
K["xor_2chains"] = rep(["v_xor_b32 v16, v16, v6", "v_xor_b32 v17, v17, v7"])
K["xor_4chains"] = rep([f"v_xor_b32 v{16 + i}, v{16 + i}, v{5 + i}" for i in range(4)])
K["bitop3_diffbank"] = rep([f"v_bitop3_b32 v{16 + i}, v1, v6, v11 bitop3:0x96" for i in range(8)])
K["bitop3_dep_chain"] = rep([f"v_bitop3_b32 v16, v16, v{5 + i}, v{10 + i} bitop3:0x96" for i in range(4)])
K["add_u32"] = rep([f"v_add_u32 v{16 + i}, v1, v6" for i in range(8)])
K["add_f32"] = rep([f"v_add_f32 v{16 + i}, v1, v6" for i in range(8)])
K["fma_f32"] = rep([f"v_fma_f32 v{16 + i}, v1, v6, v11" for i in range(8)])
K["pk_fma_f32"] = rep([f"v_pk_fma_f32 v[{16 + 2 * i}:{17 + 2 * i}], v[0:1], v[6:7], v[10:11]" for i in range(8)])
K["lshl_add_u64"] = rep([f"v_lshl_add_u64 v[{16 + 2 * i}:{17 + 2 * i}], v[0:1], 1, v[6:7]" for i in range(8)])
K["lshlrev_b32"] = rep([f"v_lshlrev_b32 v{16 + i}, 3, v6" for i in range(8)])
K["sdwa_lshl"] = rep([f"v_lshlrev_b32_sdwa v{16 + i}, v1, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}" for i in range(8)])
K["perm_b32"] = rep([f"v_perm_b32 v{16 + i}, v1, v6, v11" for i in range(8)])
K["and_nop_alt"] = rep([x for i in range(8) for x in (f"v_and_b32 v{16 + i}, v1, v6", "s_nop 0")])

# (round 5: what unpacking a 6-bit symbol would cost -- v_bfe_u32 + v_and with a literal against the one SDWA shift of the 8-bit payload)
K["bfe_u32"] = rep([f"v_bfe_u32 v{16 + i}, v6, {3 + 6 * (i % 4)}, 9" for i in range(8)])
K["bfe_and_lit"] = rep([x for i in range(8) for x in (f"v_bfe_u32 v{16 + i}, v6, {3 + 6 * (i % 4)}, 9", f"v_and_b32 v{16 + i}, 0x1f8, v{16 + i}")])
K["bfe_and_vgpr"] = rep([x for i in range(8) for x in (f"v_bfe_u32 v{16 + i}, v6, {3 + 6 * (i % 4)}, 9", f"v_and_b32 v{16 + i}, v11, v{16 + i}")])
K["alignbit_and"] = rep([x for i in range(8) for x in (f"v_alignbit_b32 v{16 + i}, v7, v6, 27", f"v_and_b32 v{16 + i}, v11, v{16 + i}")])
K["lshr_and"] = rep([x for i in range(8) for x in (f"v_lshrrev_b32 v{16 + i}, {3 + 6 * (i % 4)}, v6", f"v_and_b32 v{16 + i}, v11, v{16 + i}")])
K["and_lit8"] = rep([f"v_and_b32 v{16 + i}, 0x12345678, v6" for i in range(8)])                       # VOP2 + 32-bit literal: 8 bytes
K["bitop3_nop_alt"] = rep([x for i in range(8) for x in (f"v_bitop3_b32 v{16 + i}, v1, v6, v11 bitop3:0x96", "s_nop 0")])
K["bitop3_2nop"] = rep([x for i in range(8) for x in (f"v_bitop3_b32 v{16 + i}, v1, v6, v11 bitop3:0x96", "s_nop 0", "s_nop 0")])
K["and_bitop3_alt"] = rep([x for i in range(8) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_bitop3_b32 v{24 + i}, v1, v6, v11 bitop3:0x96")])
K["and_3_bitop3_1"] = rep([x for i in range(8) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_and_b32 v{24 + i}, v2, v7", f"v_and_b32 v{16 + i}, v3, v5", f"v_bitop3_b32 v{24 + i}, v1, v6, v11 bitop3:0x96")])
K["and_lshladd_alt"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[6:7]")])
K["and2_lshladd"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_and_b32 v{20 + i}, v2, v7", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[6:7]")], 255)

K["and4_lshladd"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_and_b32 v{20 + i}, v2, v7", f"v_and_b32 v{16 + i}, v3, v5", f"v_and_b32 v{20 + i}, v1, v7", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[6:7]")], 255)
K["and2_nop_lshladd_nop"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_and_b32 v{20 + i}, v2, v7", "s_nop 0", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[6:7]", "s_nop 0")], 255)
K["and2_lshladd_otherregs"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v2", f"v_and_b32 v{20 + i}, v3, v5", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[8:9], 1, v[12:13]")], 255)
K["bitop3x2_lshladd"] = rep([x for i in range(4) for x in (f"v_bitop3_b32 v{16 + i}, v1, v6, v11 bitop3:0x96", f"v_bitop3_b32 v{20 + i}, v2, v7, v10 bitop3:0x96", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[4:5]")], 255)

K["e64and2_lshladd"] = rep([x for i in range(4) for x in (f"v_and_b32_e64 v{16 + i}, v1, v6", f"v_and_b32_e64 v{20 + i}, v2, v7", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[6:7]")], 255)
K["and2_sdwa"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_and_b32 v{20 + i}, v2, v7", f"v_lshlrev_b32_sdwa v{24 + i}, v1, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")], 255)
K["and2_lshlrev"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_and_b32 v{20 + i}, v2, v7", f"v_lshlrev_b32 v{24 + i}, 3, v6")], 255)
K["bitop3x4_lshladd"] = rep([x for i in range(4) for x in (f"v_bitop3_b32 v{16 + i}, v1, v6, v11 bitop3:0x96", f"v_bitop3_b32 v{20 + i}, v2, v7, v10 bitop3:0x96", f"v_bitop3_b32 v{16 + i}, v3, v5, v11 bitop3:0x96", f"v_bitop3_b32 v{20 + i}, v1, v7, v9 bitop3:0x96", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[4:5]")], 255)
K["and_bitop3_lshladd"] = rep([x for i in range(4) for x in (f"v_and_b32 v{16 + i}, v1, v6", f"v_bitop3_b32 v{20 + i}, v2, v7, v10 bitop3:0x96", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[4:5]")], 255)
K["bitop3_and_lshladd"] = rep([x for i in range(4) for x in (f"v_bitop3_b32 v{20 + i}, v2, v7, v10 bitop3:0x96", f"v_and_b32 v{16 + i}, v1, v6", f"v_lshl_add_u64 v[{24 + 2 * i}:{25 + 2 * i}], v[0:1], 1, v[4:5]")], 255)

# the Levenshtein column of rf_stream_asm (tools/gen_stream_asm.py lev64), pattern words register-resident (v[34:35] ...), no LDS
VP, VN, A, E, HN, HP, T = (60, 61), (62, 63), (58, 59), (56, 57), (54, 55), (52, 53), (50, 51)


def pr(r):
    return f"v[{r[0]}:{r[1]}]"


def column(i, nop_mask, sdwa=True, vop3=False, order=None):
    PM = (34 + 2 * (i % 8), 35 + 2 * (i % 8))
    AND = (lambda d, a, b: f"v_bitop3_b32 v{d}, v{a}, v{b}, v{b} bitop3:0xc0") if vop3 else (lambda d, a, b: f"v_and_b32 v{d}, v{a}, v{b}")
    toks = [
        [AND(A[h], PM[h], VP[h]) for h in (0, 1)],
        [f"v_lshl_add_u64 {pr(A)}, {pr(A)}, 0, {pr(VP)}"],
        [f"v_bitop3_b32 v{E[h]}, v{A[h]}, v{VP[h]}, v{PM[h]} bitop3:0xbe" for h in (0, 1)],
        [f"v_bitop3_b32 v{HP[h]}, v{VN[h]}, v{E[h]}, v{VP[h]} bitop3:0xf1" for h in (0, 1)],
        [AND(HN[h], E[h], VP[h]) for h in (0, 1)],
        [f"v_lshl_add_u64 {pr(HP)}, {pr(HP)}, 1, 1"],
        [f"v_bitop3_b32 v{T[h]}, v{E[h]}, v{VN[h]}, v{HP[h]} bitop3:0x01" for h in (0, 1)],
        [f"v_bitop3_b32 v{VN[h]}, v{HP[h]}, v{E[h]}, v{VN[h]} bitop3:0xe0" for h in (0, 1)],
        [f"v_lshl_add_u64 {pr(VP)}, {pr(HN)}, 1, {pr(T)}"],
    ]
    L = []
    for j, t in enumerate(toks):
        L += t
        if nop_mask >> j & 1:
            L.append("s_nop 0")
    if sdwa:
        L.append(f"v_lshlrev_b32_sdwa v30, v10, v{18 + i % 4} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")
    return L


def column2(i, nop_mask):
    """columns of two independent states (A: v50..63, B: v64..77) interleaved token by token"""
    def toks(o, PM):
        VP_, VN_, A_, E_, HN_, HP_, T_ = [(r[0] + o, r[1] + o) for r in (VP, VN, A, E, HN, HP, T)]
        return [
            [f"v_and_b32 v{A_[h]}, v{PM[h]}, v{VP_[h]}" for h in (0, 1)],
            [f"v_lshl_add_u64 {pr(A_)}, {pr(A_)}, 0, {pr(VP_)}"],
            [f"v_bitop3_b32 v{E_[h]}, v{A_[h]}, v{VP_[h]}, v{PM[h]} bitop3:0xbe" for h in (0, 1)],
            [f"v_bitop3_b32 v{HP_[h]}, v{VN_[h]}, v{E_[h]}, v{VP_[h]} bitop3:0xf1" for h in (0, 1)],
            [f"v_and_b32 v{HN_[h]}, v{E_[h]}, v{VP_[h]}" for h in (0, 1)],
            [f"v_lshl_add_u64 {pr(HP_)}, {pr(HP_)}, 1, 1"],
            [f"v_bitop3_b32 v{T_[h]}, v{E_[h]}, v{VN_[h]}, v{HP_[h]} bitop3:0x01" for h in (0, 1)],
            [f"v_bitop3_b32 v{VN_[h]}, v{HP_[h]}, v{E_[h]}, v{VN_[h]} bitop3:0xe0" for h in (0, 1)],
            [f"v_lshl_add_u64 {pr(VP_)}, {pr(HN_)}, 1, {pr(T_)}"],
        ]
    ta = toks(0, (34 + 2 * (i % 8), 35 + 2 * (i % 8)))
    tb = toks(14 + 14, (34 + 2 * ((i + 3) % 8), 35 + 2 * ((i + 3) % 8)))  # B: v78..v91
    L = []
    for j in range(9):
        L += ta[j] + tb[j]
        if nop_mask >> j & 1:
            L.append("s_nop 0")
    L.append(f"v_lshlrev_b32_sdwa v30, v10, v{18 + i % 4} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")
    L.append(f"v_lshlrev_b32_sdwa v31, v10, v{22 + i % 4} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")
    return L


def cols2(mask, n=8):
    L = []
    for i in range(n):
        L += column2(i, mask)
    return L


def cols(mask, sdwa=True, n=16, vop3=False):
    L = []
    for i in range(n):
        L += column(i, mask, sdwa, vop3)
    return L


# the two passes of the single-word Jaro kernel (tools/gen_jaro_chunk_asm.py: jaro.rs:147-190 flags, :339-368 transpositions) over one 16-column chunk,
# table rows and window rows register-resident (no LDS, no waits), the product's s_nop placement by default (round 5)
def jaro_pass1(m):
    P, PMJ, BELOW, Y, T16 = (60, 61), (26, 27), (24, 25), 23, 22
    L = [f"v_mov_b32 v{T16}, 0"]
    for i in range(16):
        pm, wn = (40 + 2 * (i % 8), 41 + 2 * (i % 8)), (32 + 2 * (i % 4), 33 + 2 * (i % 4))
        L += [f"v_bitop3_b32 v{PMJ[h]}, v{pm[h]}, v{wn[h]}, v{P[h]} bitop3:0x40" for h in (0, 1)]
        L += ["s_nop 0"] * (m & 1)
        L.append(f"v_lshl_add_u64 {pr(BELOW)}, {pr(PMJ)}, 0, -1")
        L += ["s_nop 0"] * (m >> 1 & 1)
        L += [f"v_bitop3_b32 v{P[h]}, v{P[h]}, v{PMJ[h]}, v{BELOW[h]} bitop3:0xf4" for h in (0, 1)]
        L.append(f"v_bitop3_b32 v{Y}, v{PMJ[1]}, v{BELOW[1]}, v{PMJ[1]} bitop3:0xf3")
        L += ["s_nop 0"] * (m >> 2 & 1)
        L.append(f"v_alignbit_b32 v{T16}, v{T16}, v{Y}, 31")
        L += ["s_nop 0"] * (m >> 3 & 1)
        L.append(f"v_lshlrev_b32_sdwa v{28 + i % 4}, v10, v{18 + i % 4} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")
    L += [f"v_lshlrev_b32 v{T16}, s20, v{T16}", f"v_bitop3_b32 v58, v58, v{T16}, s21 bitop3:0xf8", f"v_bitop3_b32 v59, v59, v{T16}, s21 bitop3:0xf4"]
    return L


def jaro_pass2(m):
    P, HITS, BELOW, M, Y, T16 = (60, 61), (62, 63), (24, 25), (56, 57), 23, 22
    L = [f"v_bitop3_b32 v{T16}, v58, v59, s21 bitop3:0xe4", f"v_lshrrev_b32 v{T16}, s20, v{T16}"]
    for i in range(16):
        pm = (40 + 2 * (i % 8), 41 + 2 * (i % 8))
        L += ["s_nop 0"] * (m & 1)
        L.append(f"v_bfe_i32 v{Y}, v{T16}, {15 - i}, 1")
        L += ["s_nop 0"] * (m >> 1 & 1)
        L.append(f"v_lshl_add_u64 {pr(BELOW)}, {pr(P)}, 0, -1")
        L += ["s_nop 0"] * (m >> 2 & 1)
        L += [f"v_bitop3_b32 v{M[h]}, v{P[h]}, v{BELOW[h]}, v{Y} bitop3:0x20" for h in (0, 1)]
        L += [f"v_bitop3_b32 v{HITS[h]}, v{HITS[h]}, v{pm[h]}, v{M[h]} bitop3:0xf8" for h in (0, 1)]
        L += [f"v_bitop3_b32 v{P[h]}, v{P[h]}, v{BELOW[h]}, v{Y} bitop3:0xd0" for h in (0, 1)]
        L += ["s_nop 0"] * (m >> 3 & 1)
        L.append(f"v_lshlrev_b32_sdwa v{28 + i % 4}, v10, v{18 + i % 4} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")
    return L


# the LCS column (lcs_seq.rs:222-231; rf_device.hpp LcsState<1>::step): u = S & M; x = S + u; S = x | (S & ~u), + the gather address of a later column.
# Variants (round 5): how the 64-bit add and the address are made -- half-rate single instructions (what hipcc emits) or pairs of full-rate 4-byte ones
def lcs_column(i, add, addr, and3=False, nop=0):
    S, U, X = (60, 61), (58, 59), (56, 57)
    M = (34 + 2 * (i % 8), 35 + 2 * (i % 8))
    L = [(f"v_bitop3_b32 v{U[h]}, v{S[h]}, v{M[h]}, v{M[h]} bitop3:0xc0" if and3 else f"v_and_b32 v{U[h]}, v{S[h]}, v{M[h]}") for h in (0, 1)]
    L += ["s_nop 0"] * (nop & 1)
    if add == "u64":
        L.append(f"v_lshl_add_u64 {pr(X)}, {pr(S)}, 0, {pr(U)}")
    else:
        L += [f"v_add_co_u32 v{X[0]}, vcc, v{S[0]}, v{U[0]}", f"v_addc_co_u32 v{X[1]}, vcc, v{S[1]}, v{U[1]}, vcc"]
    L += ["s_nop 0"] * (nop >> 1 & 1)
    L += [f"v_bitop3_b32 v{S[h]}, v{X[h]}, v{S[h]}, v{U[h]} bitop3:0xf4" for h in (0, 1)]   # x | (s & ~u)
    if addr == "sdwa":
        L.append(f"v_lshlrev_b32_sdwa v{28 + i % 4}, v10, v{18 + i % 4} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i % 4}")
    elif i % 4 == 0:  # byte 0: mask, then x 8
        L += [f"v_and_b32 v{28 + i % 4}, v11, v{18 + i % 4}", f"v_mul_u32_u24 v{28 + i % 4}, 8, v{28 + i % 4}"]
    else:  # (dw >> (8 k - 3)) & 0x7f8
        L += [f"v_lshrrev_b32 v{28 + i % 4}, {8 * (i % 4) - 3}, v{18 + i % 4}", f"v_and_b32 v{28 + i % 4}, v12, v{28 + i % 4}"]
    return L


def lcs_cols(add, addr, and3=False, nop=0):
    L = []
    for i in range(16):
        L += lcs_column(i, add, addr, and3, nop)
    return L


LCS = {"lcs_u64_sdwa": lcs_cols("u64", "sdwa"), "lcs_u64_sdwa_nop3": lcs_cols("u64", "sdwa", nop=3), "lcs_u64_sdwa_and3": lcs_cols("u64", "sdwa", and3=True),
       "lcs_addc_sdwa": lcs_cols("addc", "sdwa"), "lcs_u64_shift": lcs_cols("u64", "shift"), "lcs_addc_shift": lcs_cols("addc", "shift"),
       "lcs_addc_shift_and3": lcs_cols("addc", "shift", and3=True)}
# ... and over the 6-bit payload (rf_device.hpp row_offset6): the address of column j's table row is a shift (v_alignbit_b32 for the two fields that
# straddle a dword) and a mask instead of one SDWA shift.  Variants: the shift as VOP2 / VOP3, the mask as a literal v_and / v_and with a VGPR / v_bitop3
def addr6(i, shift, mask):
    o = 6 * i
    d, sh = o // 32, o % 32
    src, dst = 18 + d, 28 + i % 4
    if sh + 6 > 32:
        first = f"v_alignbit_b32 v{dst}, v{src + 1}, v{src}, {sh - 3}"
    elif sh >= 3:
        first = (f"v_lshrrev_b32_e64 v{dst}, {sh - 3}, v{src}" if shift == "e64" else f"v_lshrrev_b32 v{dst}, {sh - 3}, v{src}") if sh > 3 else None
    else:
        first = f"v_lshlrev_b32_e64 v{dst}, {3 - sh}, v{src}" if shift == "e64" else f"v_lshlrev_b32 v{dst}, {3 - sh}, v{src}"
    a = dst if first else src
    second = {"lit": f"v_and_b32 v{dst}, 0x1f8, v{a}", "vgpr": f"v_and_b32 v{dst}, v12, v{a}", "bitop3": f"v_bitop3_b32 v{dst}, v{a}, v12, v12 bitop3:0xc0"}[mask]
    return ([first] if first else []) + [second]


def lcs6_cols(shift, mask, nop=0, and3=True):
    L = []
    for i in range(16):
        c = lcs_column(i, "u64", "sdwa", and3, nop)[:-1]  # the column without its SDWA address ...
        L += c + addr6(i, shift, mask)                     # ... and the 6-bit one
    return L


LCS.update({"lcs6_e32_lit": lcs6_cols("e32", "lit"), "lcs6_e32_vgpr": lcs6_cols("e32", "vgpr"), "lcs6_e64_lit": lcs6_cols("e64", "lit"), "lcs6_e64_bitop3": lcs6_cols("e64", "bitop3"),
            "lcs6_e32_bitop3": lcs6_cols("e32", "bitop3"), "lcs6_e64_lit_nop1": lcs6_cols("e64", "lit", 1), "lcs6_e64_lit_nop2": lcs6_cols("e64", "lit", 2),
            "lcs6_e64_lit_nop3": lcs6_cols("e64", "lit", 3), "lcs6_e32_lit_plainand": lcs6_cols("e32", "lit", 0, False), "lcs6_e64_bitop3_nop3": lcs6_cols("e64", "bitop3", 3)})
# what decides?  isolated probes of the suspects
K["mul_u24"] = rep([f"v_mul_u32_u24 v{16 + i}, 8, v6" for i in range(8)])
K["lshrrev_b32"] = rep([f"v_lshrrev_b32 v{16 + i}, {5 + i}, v6" for i in range(8)])
K["add_co_addc"] = rep([x for i in range(4) for x in (f"v_add_co_u32 v{16 + 2 * i}, vcc, v1, v6", f"v_addc_co_u32 v{17 + 2 * i}, vcc, v2, v7, vcc")])

JARO = {"jaro_p1_prod": jaro_pass1(0x5), "jaro_p1_nonop": jaro_pass1(0), "jaro_p1_all": jaro_pass1(0xF), "jaro_p2_prod": jaro_pass2(0x3), "jaro_p2_nonop": jaro_pass2(0),
        "jaro_p2_all": jaro_pass2(0xF), "jaro_both_prod": jaro_pass1(0x5) + jaro_pass2(0x3)}

# (name -> (lines, number of VALU instructions in them)); the column kernels count VALU only
COLS = {"levcol_nonop": cols(0), "levcol_mask1B3": cols(0x1B3), "levcol_mask122": cols(0x122), "levcol_nosdwa_1B3": cols(0x1B3, False),
        "levcol3_nonop": cols(0, vop3=True), "levcol3_1B3": cols(0x1B3, vop3=True), "levcol3_122": cols(0x122, vop3=True), "levcol3_022": cols(0x022, vop3=True),
        "levcol3_1A2": cols(0x1A2, vop3=True), "levcol3_0A2": cols(0x0A2, vop3=True), "levcol3_102": cols(0x102, vop3=True), "levcol3_002": cols(0x002, vop3=True),
        "levcol3_020": cols(0x020, vop3=True), "levcol3_100": cols(0x100, vop3=True), "levcol3_1B2": cols(0x1B2, vop3=True), "levcol3_0B3": cols(0x0B3, vop3=True),
        "lev2col_nonop": cols2(0), "lev2col_mask022": cols2(0x022), "lev2col_mask122": cols2(0x122), "lev2col_mask1B3": cols2(0x1B3)}

clob = ",".join(f'"v{i}"' for i in range(0, 96)) + ',"vcc","scc","s20","s21"'
src = ["// GENERATED by tools/gen_cycle_bench.py", "#include <hip/hip_runtime.h>", "#include <stdint.h>", "#include <stdio.h>", "#include <vector>",
       "#include <algorithm>", "#include <map>", "#include <array>", f"#define CLOB {clob}"]
ALL = {}
for name, lines in K.items():
    ALL[name] = (lines, sum(1 for l in lines if l.startswith("v_")))
for name, lines in list(COLS.items()) + list(JARO.items()) + list(LCS.items()):
    ALL[name] = (lines, sum(1 for l in lines if l.startswith("v_")))
if os.environ.get("RF_CYCLE_ONLY"):  # e.g. RF_CYCLE_ONLY=jaro,levcol_mask1B3: only the kernels whose name starts with one of these
    ALL = {k: v for k, v in ALL.items() if any(k.startswith(x) for x in os.environ["RF_CYCLE_ONLY"].split(","))}
for name, (lines, nv) in ALL.items():
    body = "".join(f'        "{l}\\n"\n' for l in lines)
    src.append(f"""__global__ __launch_bounds__(256) void k_{name}(uint64_t* out, int iters)
{{
    extern __shared__ char lds[];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) asm volatile(
{body}        ::: CLOB);
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {{
        uint64_t* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
        o[0] = t0; o[1] = t1; o[2] = ((uint64_t)(xcc & 0xf) << 32) | (hw & 0xff30u);   // SIMD identity: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13] (not wave slot, pipe, workgroup, queue)
    }}
    if (iters < 0) lds[threadIdx.x] = 1;
}}""")
src.append('''typedef void (*kern_t)(uint64_t*, int);
static int g_cus = 256;
static void run(const char* name, kern_t k, uint64_t* d, int valu_per_iter, int waves_per_simd)
{
    const int iters = 2000, blocks = g_cus * waves_per_simd;
    // dynamic LDS so that exactly `waves_per_simd` workgroups fit one CU (160 KiB of LDS per CU; the 64 KiB cap per workgroup
    // is lifted with hipFuncSetAttribute); 8 per SIMD is the wave-slot limit itself
    int lds = 1024;
    if (waves_per_simd == 4) lds = 36 * 1024;   // 4 x 36 fit, 5 do not
    if (waves_per_simd == 2) lds = 64 * 1024;   // 2 x 64 KiB fit, 3 do not
    if (waves_per_simd == 1) lds = 96 * 1024;   // only one fits
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d, iters / 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * 4 * 3);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // per SIMD: (last end - first start) over the wavefronts that ran on it, / (their instruction count)
    std::map<uint64_t, std::array<uint64_t, 3>> simd;  // id -> {min t0, max t1, waves}
    for (size_t w = 0; w < h.size() / 3; ++w) {
        auto& e = simd.try_emplace(h[3 * w + 2], std::array<uint64_t, 3>{~0ull, 0, 0}).first->second;
        e[0] = std::min(e[0], h[3 * w]); e[1] = std::max(e[1], h[3 * w + 1]); e[2]++;
    }
    double instr = (double)iters * valu_per_iter;            // VALU instructions per wavefront
    std::vector<double> cpi;
    size_t odd = 0;
    for (auto& kv : simd) {
        if (kv.second[2] != (uint64_t)waves_per_simd) { odd++; continue; }   // a SIMD that got more or fewer wavefronts than planned
        cpi.push_back((double)(kv.second[1] - kv.second[0]) / (instr * kv.second[2]));
    }
    std::sort(cpi.begin(), cpi.end());
    double med = cpi.empty() ? 0 : cpi[cpi.size() / 2], mn = cpi.empty() ? 0 : cpi.front(), mx = cpi.empty() ? 0 : cpi.back();
    double ns = ms * 1e6 / (instr * waves_per_simd);        // wall clock, same unit (blocks == CUs x waves: one residency of the chip)
    printf("%-30s w/SIMD %d  cycles/VALU/SIMD %6.3f (min %6.3f max %6.3f; %zu SIMDs, %zu irregular)  ns %6.3f  -> clock %5.3f GHz\\n", name, waves_per_simd,
           med, mn, mx, cpi.size(), odd, ns, med / ns);
}
int main(int argc, char** argv)
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0); g_cus = p.multiProcessorCount;
    printf("device %s, %d CUs; s_memtime ticks per wavefront around %d x 2000 instructions\\n", p.gcnArchName, g_cus, ''' + str(N) + ''');
    uint64_t* d; (void)hipMalloc(&d, 8 * 3 * 4 * 256 * 16);
    for (int w : {4, 8}) {''')
for name, (lines, nv) in ALL.items():
    src.append(f'        run("{name} ({nv} VALU)", k_{name}, d, {nv}, w);')
src.append("    }\n    return 0;\n}")
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "microbench_cycles.hip"), "w").write("\n".join(src) + "\n")
