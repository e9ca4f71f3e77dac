"""Generates rapidfuzz_rs_amd/csrc/rf_jaro_chunk_asm.inc: the two passes of the single-word Jaro kernel over one 16-column chunk
as hand-scheduled asm blocks (same reasons and same technique as tools/gen_lev_chunk_asm.py / rf_lev_asm.hip).

  pass 1 (jaro.rs:147-190, jaro_flag_chunk):      pm = PM[sym] & window[j] & ~P;  below = pm - 1;  P |= pm & ~below;
                                                  T bit = sign(pm | ~below), shifted into T16 (column j on bit 15 - j)
  pass 2 (jaro.rs:339-368, jaro_transpose_chunk): f = T bit j (all ones / zero);  below = P - 1;  m = P & ~below & f;
                                                  hits |= PM[sym] & m;  P &= below | ~f

Physical registers: P = v[60:61], hits = v[62:63], T flags = v58 (columns 0..31) / v59 (32..63) -- pinned, touched by asm only;
scratch v22..v57.  Named operands: c0..c3 the chunk's dwords, k3 a VGPR holding 3, wa a VGPR holding the LDS byte address of
this wavefront's window-mask row for the chunk's first column, sh / lo the scalar shift (0 or 16) and mask (all ones for
columns 0..31) that steer the chunk's sixteen T bits.  LDS operations return in order, so every wait is a counted lgkmcnt
computed here from the block's own issue order.

  python tools/gen_jaro_chunk_asm.py [output path]      RF_GEN_JMASK1 / RF_GEN_JMASK2 = nop masks for experiments"""
import os, sys
P, HITS, TLO, THI = (60, 61), (62, 63), 58, 59
PMB = [(40 + 2 * k, 41 + 2 * k) for k in range(8)]   # look-ahead table rows
WIN = [(32 + 2 * k, 33 + 2 * k) for k in range(4)]   # window-mask rows, four in flight
ADDR = [28, 29, 30, 31]
PMJ, BELOW, M = (26, 27), (24, 25), (56, 57)
Y, T16 = 23, 22
p = lambda r: f"v[{r[0]}:{r[1]}]"
class Block:
    def __init__(self): self.lines, self.issued, self.done_id = [], 0, {}
    def emit(self, s): self.lines.append(s)
    def lds(self, s, tag): self.lines.append(s); self.issued += 1; self.done_id[tag] = self.issued
    def wait(self, *tags):  # wait until every tagged LDS operation has returned
        need = max(self.done_id[t] for t in tags)
        self.lines.append(f"s_waitcnt lgkmcnt({self.issued - need})")
# PRIV (round 4): the conflict-free table gather for corpora of <= 64 stored symbols.  Row of symbol s for lane l at
# PRIV_OFF + s * 256 + (l & 31) * 8: the 32 lanes of each half-wavefront read 32 different bank pairs whatever their symbols are (a
# ds_read_b64 then costs 2 LDS cycles instead of the 4 that 62 symbols on 32 bank pairs average).  The address is ONE v_perm_b32
# (half rate, like the SDWA it replaces): byte 0 = the lane's bank offset %[lb], byte 1 = the chunk's symbol byte, selected by %[selK].
PRIV = False
PRIV_OFF = 0
def x(b, j):
    if PRIV: b.emit(f"v_perm_b32 v{ADDR[j % 4]}, %[c{j // 4}], %[lb], %[sel{j % 4}]")
    else: b.emit(f"v_lshlrev_b32_sdwa v{ADDR[j % 4]}, %[k3], %[c{j // 4}] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{j % 4}")
def d(b, j): b.lds(f"ds_read_b64 {p(PMB[j % 8])}, v{ADDR[j % 4]}" + (f" offset:{PRIV_OFF}" if PRIV else ""), ("pm", j))
# window-mask rows j, j + 1: every lane reads the SAME address, which a ds_read_b64 serves as a broadcast in 2 LDS cycles; the
# ds_read2_b64 of round 2 took 8 for the pair (MI355X_MICROARCH.md LDS table) -- and this kernel keeps the LDS ~80 % busy
def w2(b, j):
    if os.environ.get("RF_GEN_JWIN2", "0") == "1":
        b.lds(f"ds_read2_b64 v[{WIN[j % 4][0]}:{WIN[(j + 1) % 4][1]}], %[wa] offset0:{j} offset1:{j + 1}", ("w", j)); b.done_id[("w", j + 1)] = b.issued
    else:
        b.lds(f"ds_read_b64 {p(WIN[j % 4])}, %[wa] offset:{8 * j}", ("w", j))
        b.lds(f"ds_read_b64 {p(WIN[(j + 1) % 4])}, %[wa] offset:{8 * (j + 1)}", ("w", j + 1))
def nop(b, m, bit):
    if m >> bit & 1: b.emit("s_nop 0")
def pass1(m):
    b = Block()
    b.emit("s_waitcnt lgkmcnt(0)")
    b.emit(f"v_mov_b32 v{T16}, 0")
    for j in range(4): x(b, j)
    for j in range(4): d(b, j)
    w2(b, 0); w2(b, 2)
    for j in range(4, 8): x(b, j)
    for j in range(4, 8): d(b, j)
    for i in range(16):
        pm, wn = PMB[i % 8], WIN[i % 4]
        b.wait(("pm", i), ("w", i))
        for h in (0, 1): b.emit(f"v_bitop3_b32 v{PMJ[h]}, v{pm[h]}, v{wn[h]}, v{P[h]} bitop3:0x40")   # PM & window & ~P
        nop(b, m, 0)
        b.emit(f"v_lshl_add_u64 {p(BELOW)}, {p(PMJ)}, 0, -1")
        nop(b, m, 1)
        for h in (0, 1): b.emit(f"v_bitop3_b32 v{P[h]}, v{P[h]}, v{PMJ[h]}, v{BELOW[h]} bitop3:0xf4")  # P | (pm & ~below)
        b.emit(f"v_bitop3_b32 v{Y}, v{PMJ[1]}, v{BELOW[1]}, v{PMJ[1]} bitop3:0xf3")                    # pm | ~below: sign = (pm != 0)
        nop(b, m, 2)
        b.emit(f"v_alignbit_b32 v{T16}, v{T16}, v{Y}, 31")
        nop(b, m, 3)
        if i + 8 < 16: x(b, i + 8); d(b, i + 8)
        if i % 2 == 1 and i + 3 < 16: w2(b, i + 3)   # rows i+3, i+4 go where rows i-1, i (both consumed) were
    b.emit(f"v_lshlrev_b32 v{T16}, %[sh], v{T16}")
    b.emit(f"v_bitop3_b32 v{TLO}, v{TLO}, v{T16}, %[lo] bitop3:0xf8")   # t_lo |= bits & lo
    b.emit(f"v_bitop3_b32 v{THI}, v{THI}, v{T16}, %[lo] bitop3:0xf4")   # t_hi |= bits & ~lo
    return b.lines
def pass2(m):
    b = Block()
    b.emit("s_waitcnt lgkmcnt(0)")
    b.emit(f"v_bitop3_b32 v{T16}, v{TLO}, v{THI}, %[lo] bitop3:0xe4")   # this chunk's half: (t_lo & lo) | (t_hi & ~lo)
    b.emit(f"v_lshrrev_b32 v{T16}, %[sh], v{T16}")
    for j in range(4): x(b, j)
    for j in range(4): d(b, j)
    for j in range(4, 8): x(b, j)
    for j in range(4, 8): d(b, j)
    for i in range(16):
        pm = PMB[i % 8]
        nop(b, m, 0)
        b.emit(f"v_bfe_i32 v{Y}, v{T16}, {15 - i}, 1")
        nop(b, m, 1)
        b.emit(f"v_lshl_add_u64 {p(BELOW)}, {p(P)}, 0, -1")
        nop(b, m, 2)
        for h in (0, 1): b.emit(f"v_bitop3_b32 v{M[h]}, v{P[h]}, v{BELOW[h]}, v{Y} bitop3:0x20")        # lowest remaining flag, if flagged
        b.wait(("pm", i))
        for h in (0, 1): b.emit(f"v_bitop3_b32 v{HITS[h]}, v{HITS[h]}, v{pm[h]}, v{M[h]} bitop3:0xf8")   # hits | (PM & m)
        for h in (0, 1): b.emit(f"v_bitop3_b32 v{P[h]}, v{P[h]}, v{BELOW[h]}, v{Y} bitop3:0xd0")         # P & (below | ~f)
        nop(b, m, 3)
        if i + 8 < 16: x(b, i + 8); d(b, i + 8)
    return b.lines
def macro(name, lines): return [f"#define {name} \\"] + [f'    "{l}\\n\\t" \\' for l in lines[:-1]] + [f'    "{lines[-1]}\\n"']
m1 = int(os.environ.get("RF_GEN_JMASK1", "0x5"), 0)
m2 = int(os.environ.get("RF_GEN_JMASK2", "0x3"), 0)
out = ["// GENERATED by tools/gen_jaro_chunk_asm.py -- do not edit.  See that file.",
       '#define RF_JARO_CHUNK_CLOBBERS ' + ", ".join(f'"v{i}"' for i in range(22, 58))]
out += macro("RF_JARO_PASS1_ASM", pass1(m1)) + macro("RF_JARO_PASS2_ASM", pass2(m2))
PRIV, PRIV_OFF = True, int(os.environ.get("RF_GEN_JPRIV_OFF", "6736"))  # = sizeof(JaroWordLds) (rf_jaro.hip static_asserts it)
out += [f"#define RF_JARO_PRIV_OFF {PRIV_OFF}"] + macro("RF_JARO_PASS1_PRIV_ASM", pass1(m1)) + macro("RF_JARO_PASS2_PRIV_ASM", pass2(m2))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rapidfuzz_rs_amd", "csrc", "rf_jaro_chunk_asm.inc")
open(path, "w").write("\n".join(out) + "\n")
