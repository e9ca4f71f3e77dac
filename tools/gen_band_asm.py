"""Generates rapidfuzz_rs_amd/csrc/rf_band_asm.inc: EIGHT diagonal-phase columns of the small-band Levenshtein kernel (rf_band.hip; the reference's
hyrroe2003_small_band_with_pm, src/distance/levenshtein.rs:509-617) as one asm block (round 6, VERDICT r5 item 5 / missing #5).

The compiled column costs ~30 VALU (hipcc: 64-bit shifts as v_lshrrev_b64, the window's funnel shift and the diagonal-bit count as separate shift / and / add chains).  Here, per column:
  address   v_mul_u32_u24_sdwa (symbol byte x row pitch) + v_add_u32 (the window's dword, a scalar)                          2
  window    ds_read2_b32 + ds_read_b32 (three consecutive dwords), two v_alignbit_b32 by the scalar bit offset              2
  e         x & VP (2), + VP through the carry flag (v_add_co / v_addc_co: 2), (sum ^ VP) | x as v_bitop3_b32 (2)            6
  D0 = e | VN (2), HP = VN | ~(e | VP) (2, bitop3), HN = e & VP (2)                                                          6
  diagonal bit: v_alignbit_b32 acc, acc, D0hi, 31 -- a shift register of the eight top bits, counted once per run            1
  D0 >> 1 (v_alignbit_b32 + v_lshrrev_b32), VP' = HN | ~(D0s | HP) (2, bitop3), VN' = D0s & HP (2)                            6
= 23 VALU + 5 scalar instructions.  The rows of columns 0..3 are requested up front and a row slot is refilled for column + 4 as soon as two columns have run (counted
lgkmcnt waits: LDS operations return in order), so two to four columns' reads are in flight ahead of the column being computed.

Operands: [dw0] [dw1] the eight symbols (two dwords of the chunk), [pitch] a VGPR holding the table's row pitch in BYTES, [v0] the window position + 64 at the run's
first column (SGPR), state [vpl] [vph] [vnl] [vnh] and the shift register [acc] in / out.  The band table is the kernel's only LDS object: LDS address 0.
Scratch: v36..v63, s76..s78, vcc (low enough that the kernel stays within the 96 SGPRs of eight wavefronts per SIMD).

  python tools/gen_band_asm.py [output path]        tests/test_docs.py checks the committed .inc is this script's output"""
import os
import sys

ADDR = [36, 37, 38, 39]
ROWS = [(40, 41, 48), (42, 43, 49), (44, 45, 50), (46, 47, 51)]  # (the ds_read2_b32 pair must start at an even register)
X, A, E, D, P, N = (52, 53), (54, 55), (56, 57), (58, 59), (60, 61), (62, 63)
S_V, S_I, S_SH = "s76", "s77", "s78"


def read(c):
    """the three window dwords of column c into row slot c % 4"""
    jj, g = c % 4, c // 4
    r0, r1, r2 = ROWS[jj]
    return [f"s_add_u32 {S_V}, %[v0], {c}", f"s_lshr_b32 {S_I}, {S_V}, 5", f"s_lshl_b32 {S_I}, {S_I}, 2",
            f"v_mul_u32_u24_sdwa v{ADDR[jj]}, %[dw{g}], %[pitch] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_{jj} src1_sel:DWORD",
            f"v_add_u32 v{ADDR[jj]}, {S_I}, v{ADDR[jj]}",
            f"ds_read2_b32 v[{r0}:{r1}], v{ADDR[jj]} offset1:1", f"ds_read_b32 v{r2}, v{ADDR[jj]} offset:8"]


def column(c, outstanding):
    """column c; `outstanding` = LDS operations that may still be in flight once its own rows have arrived (they return in order)"""
    jj = c % 4
    r0, r1, r2 = ROWS[jj]
    vp, vn = ("%[vpl]", "%[vph]"), ("%[vnl]", "%[vnh]")
    L = [f"s_waitcnt lgkmcnt({outstanding})", f"s_add_u32 {S_V}, %[v0], {c}", f"s_and_b32 {S_SH}, {S_V}, 31",
         f"v_alignbit_b32 v{X[0]}, v{r1}, v{r0}, {S_SH}", f"v_alignbit_b32 v{X[1]}, v{r2}, v{r1}, {S_SH}"]
    L += [f"v_and_b32 v{A[h]}, v{X[h]}, {vp[h]}" for h in (0, 1)]
    L += [f"v_add_co_u32 v{A[0]}, vcc, v{A[0]}, {vp[0]}", f"v_addc_co_u32 v{A[1]}, vcc, v{A[1]}, {vp[1]}, vcc"]
    L += [f"v_bitop3_b32 v{E[h]}, v{A[h]}, {vp[h]}, v{X[h]} bitop3:0xbe" for h in (0, 1)]          # e = (sum ^ VP) | x
    L += [f"v_or_b32 v{D[h]}, v{E[h]}, {vn[h]}" for h in (0, 1)]                                    # D0 = e | VN
    L += [f"v_bitop3_b32 v{P[h]}, {vn[h]}, v{E[h]}, {vp[h]} bitop3:0xf1" for h in (0, 1)]          # HP = VN | ~(e | VP)
    L += [f"v_and_b32 v{N[h]}, v{E[h]}, {vp[h]}" for h in (0, 1)]                                   # HN = e & VP
    L += [f"v_alignbit_b32 %[acc], %[acc], v{D[1]}, 31"]                                            # acc = acc << 1 | D0 >> 63
    L += [f"v_alignbit_b32 v{D[0]}, v{D[1]}, v{D[0]}, 1", f"v_lshrrev_b32 v{D[1]}, 1, v{D[1]}"]     # D0 >> 1
    L += [f"v_bitop3_b32 {vp[h]}, v{N[h]}, v{D[h]}, v{P[h]} bitop3:0xf1" for h in (0, 1)]          # VP' = HN | ~(D0s | HP)
    L += [f"v_and_b32 {vn[h]}, v{D[h]}, v{P[h]}" for h in (0, 1)]                                   # VN' = D0s & HP
    return L


def body():
    """rows of columns 0..3 requested up front; a column's row slot is refilled (column + 4) as soon as two columns have run, so that two to four columns' reads are always
    in flight ahead of the one being computed"""
    L = ["s_waitcnt lgkmcnt(0)"]  # (whatever the compiled code still has in flight: the counted waits below are about this block's own reads)
    for c in range(4):
        L += read(c)
    L += column(0, 6) + column(1, 4)
    L += read(4) + read(5)           # in flight now: columns 2, 3, 4, 5
    L += column(2, 6) + column(3, 4)
    L += read(6) + read(7)           # in flight: columns 4, 5, 6, 7
    L += column(4, 6) + column(5, 4) + column(6, 2) + column(7, 0)
    return L


def macro(name, lines):
    return [f"#define {name} \\"] + [f'    "{l}\\n\\t" \\' for l in lines[:-1]] + [f'    "{lines[-1]}\\n"']


def main():
    out = ["// GENERATED by tools/gen_band_asm.py -- do not edit.  Eight diagonal-phase columns of rf_band.hip's band_kernel as one asm block; see the generator for the",
           "// instruction budget (23 VALU per column) and the operands.",
           "#define RF_BAND_RUN8_CLOBBERS " + ", ".join(f'"v{r}"' for r in range(36, 64)) + ', "s76", "s77", "s78", "vcc", "scc"']
    out += macro("RF_BAND_RUN8_ASM", body())
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rapidfuzz_rs_amd", "csrc", "rf_band_asm.inc")
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
