"""Generates rapidfuzz_rs_amd/csrc/rf_stream_asm.inc: the no-cutoff scans of the single-word bit-parallel states (Levenshtein
64-bit, Levenshtein 32-bit, OSA) as WHOLE KERNELS in gfx950 assembly -- one asm body per (kind, corpus addressing) pair, wrapped
by rf_stream_asm.hip in a __global__ function whose only job is to hand over the kernarg pointer and the workgroup / thread ids.

Why whole kernels: round 2 pinned the recurrence state and the look-ahead table rows in physical registers and left the loop
around the 16-column blocks to hipcc.  Giving that kernel tile descriptors, partial last chunks and a rolled loop needs the chunk
ring pinned too, and with 46 + 14 of 64 VGPRs spoken for the register allocator starts splitting the pinned live ranges (copies
and scratch traffic around every block).  Here nothing is left to it: registers, waits and the schedule are all explicit.

One wavefront = one stream of 16-column chunks over its tiles (tile t, t + stride, ...):
  * STEP (x RING phases, the ring rotates by renaming): fetch the chunk RING-1 steps ahead into the ring (global_load_dwordx4, 1 KiB
    per wavefront), then the 16 recurrence columns of the current chunk (rf_device.hpp LevState<1>::step / Lev32State::step /
    OsaState<1>::step; levenshtein.rs:466-490, osa.rs:156-226) with the LDS gather of table rows running LA columns ahead, across
    chunk and tile boundaries, behind counted lgkmcnt waits.  The ring is waited for with counted vmcnt: this chunk at the top of
    the block, the next chunk's first dwords only at column 8.  s_nop 0 placement: the best of the placements measured in round 2
    (profiles/lev_schedule_experiments_r02.txt).
  * a tile's last, partial chunk (rem < 16 symbols) is shifted up by k = 16 - rem byte positions in place (wavefront-uniform) and
    the block is ENTERED AT COLUMN k: exactly rem columns run and the block still ends at column 15, where the look-ahead reads
    of the next tile's first columns have been issued the usual way.  Stub k waits for what is in flight (the rows the previous
    block gathered for this chunk's first columns came from the unshifted bytes), gathers the rows of columns k .. k+LA-1 in
    issue order -- positions >= 16 are the next chunk's first columns -- and jumps to column k.
  * tile epilogue (shared by the phases): D[len1][len2] = len2 + popcount(VP & valid) - popcount(VN & valid), the affine finishing
    map of rf_device.hpp ("Finishing": value = v0(tile) + vR * raw, None unless (value ^ flip) <= cflip), one u32 per candidate to
    out[orig[slot]] (general corpora) or out[slot] (single-length corpora), then the next tile's descriptor and state.
Zero-length tiles never reach these kernels (the launcher splits them off), nor do top-k / f64 / early-out launches.

Register map (all kinds):  SGPRs s8.. are loaded from the kernarg block (StreamAsmArgs, offsets below), s33.. are cursors;
VGPRs: v1 lane, v2 lane*16, v3 lane*4, v4 candidate index of the current tile, v5 zero, v6..v9 epilogue temporaries, v10 log2(row
size), ring buffers / ADDR v30..v33 / table rows v34.. / scratch v50..v59 / state v58..v63 per KINDS.

  python tools/gen_stream_asm.py [output path]        tests/test_docs.py checks the committed .inc is this script's output
"""
import os
import sys

ADDR = [30, 31, 32, 33]
ADDS32 = os.environ.get("RF_GEN_ADDS32", "0") == "1"
# RF_GEN_ADDC=64: the single-word kernel shifts HP through the carry flag too (v_addc_co_u32 pairs instead of v_lshl_add_u64) -- measured -1.4 %, an
# experiment knob.  The multi-word kernels always do (+4.6 % on configs[2], profiles/levw_addc_r04.txt; their hn_c through a second carry chain over an
# SGPR pair was built and measured 17 % slower in round 4 and is gone from the generator).
ADDC = set(filter(None, os.environ.get("RF_GEN_ADDC", "").replace(" ", "").split(",")))
VOP3_32 = os.environ.get("RF_GEN_VOP3_32", "0") == "1"
VOP3_OSA = os.environ.get("RF_GEN_VOP3_OSA", "0") == "1"
EARLY_FETCH = os.environ.get("RF_GEN_EARLY_FETCH", "1") == "1"  # 0: the first chunks are requested after the pattern table is staged (rounds 3-4: the A/B)
DESC_PREFETCH = os.environ.get("RF_GEN_DESC_PREFETCH", "0") == "1"  # (experiment, round 5) tiles kernels: the NEXT tile's descriptor is requested at a tile's start
S_PDESC = "s[56:59]"
BAND = os.environ.get("RF_GEN_BAND", "1") == "1"  # 0: the multi-word kernels run every word in every column (round 4's kernels: the A/B)
# kernarg block (struct StreamAsmArgs in rf_stream_asm.hip; static_asserts there hold the two together)
ARGS = [("data", 8), ("tiles", 8), ("orig", 8), ("pm", 8), ("sigma", 8), ("out", 8), ("tile_begin", 4), ("tile_end", 4), ("n", 4),
        ("uniform_len", 4), ("uniform_tile_bytes", 4), ("len1", 4), ("fin_vS", 4), ("fin_vM", 4), ("fin_vR", 4), ("fin_flip", 4),
        ("fin_cflip", 4), ("valid_lo", 4), ("valid_hi", 4), ("flags", 4), ("band_k", 4), ("valid_w", 64), ("pad1", 4), ("vtab", 2048)]  # vtab: the f64 value of every u32 value 0..255 (the 6-bit LCS kernels, flags bit 4); valid_w: (lo, hi) row masks of words 0..7; band_k: distances above it need not be exact (multi-word kernels)
# SGPR map
S_DATA, S_TILES, S_ORIG, S_PM, S_SIGMA, S_OUT = "s[8:9]", "s[10:11]", "s[12:13]", "s[14:15]", "s[16:17]", "s[18:19]"
(S_TBEGIN, S_TEND, S_N, S_ULEN, S_UBYTES, S_LEN1, S_VS, S_VM, S_VR, S_FLIP, S_CFLIP, S_VLO) = [f"s{i}" for i in range(20, 32)]
S_FLAGS = "s69"  # StreamAsmArgs::flags (loaded with valid_hi)
S_VHI = "s68"  # (s32 is the ABI's stack pointer: the compiler refuses it on a clobber list)
S_STRIDE, S_T, S_C, S_NCH, S_LEN2, S_SLOT0, S_V0 = "s33", "s34", "s35", "s36", "s37", "s38", "s39"
S_FT, S_FC, S_FN = "s40", "s41", "s42"
S_FBASE, S_FBASE_LO, S_FBASE_HI = "s[44:45]", "s44", "s45"
S_SRC, S_SRC_LO, S_SRC_HI = "s[46:47]", "s46", "s47"
S_K, S_Q, S_R8, S_SH = "s48", "s49", "s50", "s51"
T0, T1, T2, T3 = "s52", "s53", "s54", "s55"
S_DESC = "s[60:63]"  # TileDesc {u64 data_off, u32 len, u32 slot0}
S_NEXT, S_AFTER, S_EXEC = "s64", "s65", "s[66:67]"  # S_AFTER: this step follows a tile epilogue whose store (and index load) are still younger than the ring
S_KBAND, S_DHI, S_DLO, S_LIVE = "s88", "s89", "s90", "s91"  # multi-word kernels: the Ukkonen band (BlockKind)
V_LANE, V_OFF16, V_OFF4, V_IDX, V_ZERO, V_KS = "v1", "v2", "v3", "v4", "v5", "v10"


def pr(r):
    return f"v[{r[0]}:{r[1]}]"


VP, VN, A, E, HN, HP, T = (60, 61), (62, 63), (58, 59), (56, 57), (54, 55), (52, 53), (50, 51)
LEV_BASE = "a S e hp hn hq t vn vp".split()
VP32, VN32, A32, E32, HN32, HP32, T32 = 60, 61, 58, 56, 54, 52, 50
# OSA (osa.rs:156-226): the Levenshtein column with the transposition term; D0 is state, the previous column's table row is
# still in the ring (7 columns of look-ahead instead of 8, so slot (i - 1) % 8 has not been overwritten yet)
D0_, R1_, R2_ = (58, 59), (56, 57), (54, 55)
OSA_BASE = "t ts tr a S e d hn hp hq hs vn vp".split()
# LCS (Indel, LCS, fuzz::ratio; round 5): state S = v[60:61], u = v[58:59], x = v[56:57]
LS, LU, LX = (60, 61), (58, 59), (56, 57)
LCS_BASE = "u p s".split()


class Kind:
    def __init__(self, name, bits, la, bufs, state, nop_mask):
        self.name, self.bits, self.la, self.bufs, self.state, self.nop_mask = name, bits, la, bufs, list(state), nop_mask
        self.ring = len(bufs)
        o = os.environ.get({"lev64": "RF_GEN_ORDER64", "lev32": "RF_GEN_ORDER32"}.get(name, "RF_GEN_ORDER_NONE"), "")
        self.order = o.replace(",", " ").split() if o else None
        self.rows = [(34 + 2 * k, 35 + 2 * k) for k in range(8)] if bits == 64 else [(34 + k,) for k in range(8)]
        self.ks = 3 if bits == 64 else 2

    # LDS gather of the table row of window position j (0..15: this chunk, 16..23: the next chunk's first columns)
    def gather(self, j, use, nxt):
        src = use + (j // 4) if j < 16 else nxt + ((j - 16) // 4)
        a = ADDR[j % 4]
        x = f"v_lshlrev_b32_sdwa v{a}, {V_KS}, v{src} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{j % 4}"
        slot = self.rows[j % 8]
        return [x, f"ds_read_b64 {pr(slot)}, v{a}" if self.bits == 64 else f"ds_read_b32 v{slot[0]}, v{a}"]

    def op(self, tok, i):
        if self.name == "lev64":
            PM = self.rows[i % 8]
            return {"a": [f"v_and_b32 v{A[h]}, v{PM[h]}, v{VP[h]}" for h in (0, 1)],                               # x & VP
                    "S": [f"v_lshl_add_u64 {pr(A)}, {pr(A)}, 0, {pr(VP)}"],                                        # + VP
                    "e": [f"v_bitop3_b32 v{E[h]}, v{A[h]}, v{VP[h]}, v{PM[h]} bitop3:0xbe" for h in (0, 1)],       # e = (sum ^ VP) | x
                    "hp": [f"v_bitop3_b32 v{HP[h]}, v{VN[h]}, v{E[h]}, v{VP[h]} bitop3:0xf1" for h in (0, 1)],     # HP = VN | ~(e | VP)
                    "hn": [f"v_and_b32 v{HN[h]}, v{E[h]}, v{VP[h]}" for h in (0, 1)],                              # HN = e & VP
                    "hq": ([f"s_mov_b64 vcc, -1", f"v_addc_co_u32 v{HP[0]}, vcc, v{HP[0]}, v{HP[0]}, vcc", f"v_addc_co_u32 v{HP[1]}, vcc, v{HP[1]}, v{HP[1]}, vcc"]
                           if "64" in ADDC else [f"v_lshl_add_u64 {pr(HP)}, {pr(HP)}, 1, 1"]),                     # HP' = (HP << 1) + 1
                    "t": [f"v_bitop3_b32 v{T[h]}, v{E[h]}, v{VN[h]}, v{HP[h]} bitop3:0x01" for h in (0, 1)],       # T = ~(e | VN | HP')
                    "vn": [f"v_bitop3_b32 v{VN[h]}, v{HP[h]}, v{E[h]}, v{VN[h]} bitop3:0xe0" for h in (0, 1)],     # VN' = HP' & (e | VN)
                    "vp": [f"v_lshl_add_u64 {pr(VP)}, {pr(HN)}, 1, {pr(T)}"]}[tok]                                 # VP' = (HN << 1) + T
        if self.name == "lev32":
            PM = self.rows[i % 8][0]
            if VOP3_32:  # (experiment, round 5: the three 4-byte instructions in their long encodings -- the rule of profiles/lcs_cycles_r05.txt)
                return {"a": [f"v_and_b32_e64 v{A32}, v{PM}, v{VP32}"], "S": [f"v_add_u32_e64 v{A32}, v{A32}, v{VP32}"],
                        "e": [f"v_bitop3_b32 v{E32}, v{A32}, v{VP32}, v{PM} bitop3:0xbe"], "hp": [f"v_bitop3_b32 v{HP32}, v{VN32}, v{E32}, v{VP32} bitop3:0xf1"],
                        "hn": [f"v_and_b32_e64 v{HN32}, v{E32}, v{VP32}"], "hq": [f"v_lshl_or_b32 v{HP32}, v{HP32}, 1, 1"],
                        "t": [f"v_bitop3_b32 v{T32}, v{E32}, v{VN32}, v{HP32} bitop3:0x01"], "vn": [f"v_bitop3_b32 v{VN32}, v{HP32}, v{E32}, v{VN32} bitop3:0xe0"],
                        "vp": [f"v_lshl_add_u32 v{VP32}, v{HN32}, 1, v{T32}"]}[tok]
            return {"a": [f"v_and_b32 v{A32}, v{PM}, v{VP32}"], "S": [f"v_add_u32 v{A32}, v{A32}, v{VP32}"],
                    "e": [f"v_bitop3_b32 v{E32}, v{A32}, v{VP32}, v{PM} bitop3:0xbe"], "hp": [f"v_bitop3_b32 v{HP32}, v{VN32}, v{E32}, v{VP32} bitop3:0xf1"],
                    "hn": [f"v_and_b32 v{HN32}, v{E32}, v{VP32}"],
                    # RF_GEN_ADDS32=1 (experiment): the two half-rate shift-and-add forms as pairs of full-rate VOP2 adds
                    "hq": ([f"v_add_u32 v{HP32}, v{HP32}, v{HP32}", f"v_add_u32 v{HP32}, 1, v{HP32}"] if ADDS32 else [f"v_lshl_or_b32 v{HP32}, v{HP32}, 1, 1"]),
                    "t": [f"v_bitop3_b32 v{T32}, v{E32}, v{VN32}, v{HP32} bitop3:0x01"], "vn": [f"v_bitop3_b32 v{VN32}, v{HP32}, v{E32}, v{VN32} bitop3:0xe0"],
                    "vp": ([f"v_add_u32 v{HN32}, v{HN32}, v{HN32}", f"v_add_u32 v{VP32}, v{HN32}, v{T32}"] if ADDS32 else [f"v_lshl_add_u32 v{VP32}, v{HN32}, 1, v{T32}"])}[tok]
        if self.name == "lcs32":  # the same on 32-bit words (queries of <= 32 symbols; rf_device.hpp Lcs32State): every instruction full rate, in its long encoding
            PM = self.rows[i % 8][0]
            return {"u": [f"v_bitop3_b32 v{LU[0]}, v{LS[0]}, v{PM}, v{PM} bitop3:0xc0"],
                    "p": [f"v_add_u32_e64 v{LX[0]}, v{LS[0]}, v{LU[0]}"],
                    "s": [f"v_bitop3_b32 v{LS[0]}, v{LX[0]}, v{LS[0]}, v{LU[0]} bitop3:0xf4"]}[tok]
        if self.name == "lcs64":
            # lcs_seq.rs:222-231 (rf_device.hpp LcsState<1>::step): u = S & M; x = S + u; S = x | (S & ~u).  The two ands are written as v_bitop3_b32 ON
            # PURPOSE: 4-byte full-rate instructions next to half-rate ones (v_lshl_add_u64, SDWA) issue at 4 cycles each, 8-byte ones add up --
            # 24.0 -> 18.6 cycles per column in the register-only microbenchmark (profiles/lcs_cycles_r05.txt)
            PM = self.rows[i % 8]
            return {"u": [f"v_bitop3_b32 v{LU[h]}, v{LS[h]}, v{PM[h]}, v{PM[h]} bitop3:0xc0" for h in (0, 1)],
                    "p": [f"v_lshl_add_u64 {pr(LX)}, {pr(LS)}, 0, {pr(LU)}"],
                    "s": [f"v_bitop3_b32 v{LS[h]}, v{LX[h]}, v{LS[h]}, v{LU[h]} bitop3:0xf4" for h in (0, 1)]}[tok]
        PM, PMO = self.rows[i % 8], self.rows[(i - 1) % 8]
        AND = (lambda d, a, b: f"v_bitop3_b32 v{d}, v{a}, v{b}, v{b} bitop3:0xc0") if VOP3_OSA else (lambda d, a, b: f"v_and_b32 v{d}, v{a}, v{b}")
        if VOP3_OSA:  # (experiment, round 5: the eight ands in the long encoding)
            return {"t": [f"v_bitop3_b32 v{R1_[h]}, v{D0_[h]}, v{PM[h]}, v{PM[h]} bitop3:0x0c" for h in (0, 1)],
                    "ts": [f"v_lshlrev_b64 {pr(R1_)}, 1, {pr(R1_)}"],
                    "tr": [AND(R1_[h], R1_[h], PMO[h]) for h in (0, 1)],
                    "a": [AND(R2_[h], PM[h], VP[h]) for h in (0, 1)],
                    "S": [f"v_lshl_add_u64 {pr(R2_)}, {pr(R2_)}, 0, {pr(VP)}"],
                    "e": [f"v_bitop3_b32 v{R2_[h]}, v{R2_[h]}, v{VP[h]}, v{PM[h]} bitop3:0xbe" for h in (0, 1)],
                    "d": [f"v_bitop3_b32 v{D0_[h]}, v{R2_[h]}, v{VN[h]}, v{R1_[h]} bitop3:0xfe" for h in (0, 1)],
                    "hn": [AND(R1_[h], D0_[h], VP[h]) for h in (0, 1)],
                    "hp": [f"v_bitop3_b32 v{R2_[h]}, v{VN[h]}, v{D0_[h]}, v{VP[h]} bitop3:0xf1" for h in (0, 1)],
                    "hq": [f"v_lshl_add_u64 {pr(R2_)}, {pr(R2_)}, 1, 1"],
                    "hs": [f"v_lshlrev_b64 {pr(R1_)}, 1, {pr(R1_)}"],
                    "vn": [AND(VN[h], R2_[h], D0_[h]) for h in (0, 1)],
                    "vp": [f"v_bitop3_b32 v{VP[h]}, v{R1_[h]}, v{R2_[h]}, v{D0_[h]} bitop3:0xf1" for h in (0, 1)]}[tok]
        return {"t": [f"v_bitop3_b32 v{R1_[h]}, v{D0_[h]}, v{PM[h]}, v{PM[h]} bitop3:0x0c" for h in (0, 1)],       # ~D0 & PM
                "ts": [f"v_lshlrev_b64 {pr(R1_)}, 1, {pr(R1_)}"],
                "tr": [f"v_and_b32 v{R1_[h]}, v{R1_[h]}, v{PMO[h]}" for h in (0, 1)],                              # & PM_old
                "a": [f"v_and_b32 v{R2_[h]}, v{PM[h]}, v{VP[h]}" for h in (0, 1)],
                "S": [f"v_lshl_add_u64 {pr(R2_)}, {pr(R2_)}, 0, {pr(VP)}"],
                "e": [f"v_bitop3_b32 v{R2_[h]}, v{R2_[h]}, v{VP[h]}, v{PM[h]} bitop3:0xbe" for h in (0, 1)],
                "d": [f"v_bitop3_b32 v{D0_[h]}, v{R2_[h]}, v{VN[h]}, v{R1_[h]} bitop3:0xfe" for h in (0, 1)],      # D0 = e | VN | tr
                "hn": [f"v_and_b32 v{R1_[h]}, v{D0_[h]}, v{VP[h]}" for h in (0, 1)],
                "hp": [f"v_bitop3_b32 v{R2_[h]}, v{VN[h]}, v{D0_[h]}, v{VP[h]} bitop3:0xf1" for h in (0, 1)],
                "hq": [f"v_lshl_add_u64 {pr(R2_)}, {pr(R2_)}, 1, 1"],
                "hs": [f"v_lshlrev_b64 {pr(R1_)}, 1, {pr(R1_)}"],
                "vn": [f"v_and_b32 v{VN[h]}, v{R2_[h]}, v{D0_[h]}" for h in (0, 1)],
                "vp": [f"v_bitop3_b32 v{VP[h]}, v{R1_[h]}, v{R2_[h]}, v{D0_[h]} bitop3:0xf1" for h in (0, 1)]}[tok]  # hns | ~(hps | D0)

    def column(self, i, gather=()):
        """one recurrence column on row slot i % 8, and the look-ahead gather (`gather`: the lines of K.gather(i + la, ...)) where the
        kind's token order puts it -- token "x"; behind the column by default (RF_GEN_ORDER32 / RF_GEN_ORDER64: experiment knobs)"""
        L = [f"s_waitcnt lgkmcnt({self.la - 1})"]  # `la` reads in flight, in order: column i's row has arrived
        base = {"osa": OSA_BASE, "lcs64": LCS_BASE, "lcs32": LCS_BASE}.get(self.name, LEV_BASE)
        order = self.order or (base + ["x"])
        for tok in order:
            if tok == "x":
                L += list(gather)
                continue
            L += self.op(tok, i)
            if self.nop_mask >> base.index(tok) & 1:  # (mask bits are indexed by the canonical token list, whatever the order)
                L.append("s_nop 0")
        return L

    def state_init(self):
        if self.name == "lev64":  # levenshtein.rs:454-455
            return ["v_mov_b32 v60, -1", "v_mov_b32 v61, -1", "v_mov_b32 v62, 0", "v_mov_b32 v63, 0"]
        if self.name == "lev32":
            return ["v_mov_b32 v60, -1", "v_mov_b32 v61, 0"]
        if self.name == "lcs64":  # lcs_seq.rs:215
            return ["v_mov_b32 v60, -1", "v_mov_b32 v61, -1"]
        if self.name == "lcs32":
            return ["v_mov_b32 v60, -1"]
        # osa.rs:74-77, :125-135: D0 = 0 and no previous column: its table row (slot 7) is zero
        return ["v_mov_b32 v60, -1", "v_mov_b32 v61, -1", "v_mov_b32 v62, 0", "v_mov_b32 v63, 0", "v_mov_b32 v58, 0", "v_mov_b32 v59, 0",
                "v_mov_b32 v48, 0", "v_mov_b32 v49, 0"]


class BlockKind(Kind):
    """Levenshtein for queries of 65 .. 256 symbols: W = 2 .. 4 words per column with the horizontal carries of advance_block
    (levenshtein.rs:838-875) between them; rf_device.hpp LevState<W>::step is the compiled form.  Round 4.
      * pattern table in LDS as W word PLANES of 2 KiB (plane w at w * 2048, row sigma(c) * 8): a column's row is W ds_read_b64 of
        one address + immediate plane offsets, each with the 2-way conflicts of the single-word gather -- the compiled kernels'
        32-byte rows (2 ds_read_b128) put 62 symbols on 8 row positions of the 64 banks: 72 % of their LDS cycles were conflicts;
      * two row slots (2 x 2W VGPRs), look-ahead of 2 columns; chunk ring of 2 (a chunk is 16 x ~230 cycles: one load ahead is plenty);
      * carries: hp_c never leaves the carry flag -- HP' = HP + HP + carry as two v_addc_co_u32 per word, VCC set to all ones at the top of
        the column (word 0's + 1) and handed from each word's high half to the next word's low half (nothing else in a column touches
        VCC); this replaced v_lshl_add_u64 + one half-rate v_lshrrev_b32 ..., 31 per inter-word carry: configs[2] 2.83 -> 2.96 Gpairs/s
        (round 4, profiles/levw_addc_r04.txt; the single-word kernel LOSES 1.4 % by the same change and keeps v_lshl_add_u64).  hn_c as a
        value, one register per word boundary, OR-ed into the next word's table row and into T's low half (VP' = (HN << 1) + (T | hn_c):
        bit 0 of T is clear there);
      * THE BAND (round 5; the reference's Ukkonen trimming, levenshtein.rs:810-825, :906-985).  D <= k := min(band_k, max(len1, len2)), so
        a cell (i, j) with |i - j| + |(len1 - len2) - (i - j)| > k lies on no path that matters: with s = (k - |len1 - len2|) / 2 only rows
        j + dlo <= i <= j + dhi, dlo = min(0, len1 - len2) - s, dhi = max(0, len1 - len2) + s, have to be right in column j.  Word w is
        NOT RUN in a 16-column chunk in which all of its rows are outside that range (S_LIVE, one bit per word, made from the chunk index by
        a dozen scalar instructions; every word block of a column sits behind s_bitcmp1 + s_cbranch).  No other instruction changes:
        a word that has not started yet still holds its initial state VP = ~0, VN = 0 (D[i][j] = D[64w][j] + (i - 64w): an upper bound, which
        is all the cells outside the band have to be -- the min-recurrence is monotone); a word that is done keeps its last deltas, and the
        first live word above it runs with hp_c = 1 (VCC still holds the column's initial all-ones) and hn_c = 0 (the boundary register is
        cleared at the top of the chunk): its top boundary rises by one per column, again an upper bound, and exactly the one that keeps the
        epilogue's len2 + sum of popcounts right (frozen deltas + one per skipped column).  Results <= k are exact, results beyond are > k.
        At 256 x 256 that is word 3 in columns 1..64 and word 0 in columns 193..256: 12.5 % of the word-columns.
      * 64 VGPRs = 8 wavefronts per SIMD: v14..21 ring, v22..23 gather addresses, v24..39 row slots, v40..47 VP, v48..55 VN,
        v56..63 A E HN HP, v6..7 T, v11 / v3 / v12 hn_c of words 0 / 1 / 2 (v8, v9, v13 idle)."""
    TOKENS = "x a S e hp hn hnc hq t tor vn vp".split()

    def __init__(self, W, nop_mask):
        Kind.__init__(self, f"levw{W}", 64, 2, [14, 18], range(40, 56), nop_mask)
        self.W = W
        self.addr = [22, 23]
        if W <= 4:  # 64 VGPRs = 8 wavefronts per SIMD
            self.slots = [[(24 + 8 * sl + 2 * w, 25 + 8 * sl + 2 * w) for w in range(W)] for sl in range(2)]
            self.VP = [(40 + 2 * w, 41 + 2 * w) for w in range(W)]
            self.VN = [(48 + 2 * w, 49 + 2 * w) for w in range(W)]
            self.TMP = [(56, 57), (58, 59), (60, 61), (62, 63)]  # A E HN HP
            self.HNC = [11, 3, 12]
        else:  # queries of 257 .. 512 symbols (round 5): 104 VGPRs = 4 wavefronts per SIMD -- the compiled LevState<5..8> holds 133..184 = 3..2
            self.slots = [[(24 + 16 * sl + 2 * w, 25 + 16 * sl + 2 * w) for w in range(W)] for sl in range(2)]  # v24..55
            self.VP = [(56 + 2 * w, 57 + 2 * w) for w in range(W)]                                               # v56..71
            self.VN = [(72 + 2 * w, 73 + 2 * w) for w in range(W)]                                               # v72..87
            self.TMP = [(88, 89), (90, 91), (92, 93), (94, 95)]
            self.HNC = [11, 3, 12, 13, 9, 96, 97]
        self.uid = 0

    def gather(self, j, use, nxt):
        src = use + (j // 4) if j < 16 else nxt + ((j - 16) // 4)
        a = self.addr[j % 2]
        x = f"v_lshlrev_b32_sdwa v{a}, {V_KS}, v{src} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{j % 4}"
        return [x] + [f"ds_read_b64 {pr(self.slots[j % 2][w])}, v{a}" + (f" offset:{2048 * w}" if w else "") for w in range(self.W)]

    def tile_band(self):
        """per tile: S_DHI = dhi + 15, S_DLO = dlo of the band (class comment) from len1, len2 and band_k"""
        return [f"s_sub_i32 {T0}, {S_LEN1}, {S_LEN2}", f"s_max_u32 {T1}, {S_LEN1}, {S_LEN2}", f"s_min_u32 {T1}, {T1}, {S_KBAND}",  # k
                f"s_abs_i32 {T2}, {T0}", f"s_sub_i32 {T1}, {T1}, {T2}", f"s_max_i32 {T1}, {T1}, 0", f"s_lshr_b32 {T1}, {T1}, 1",    # s (0 when nothing can pass)
                f"s_max_i32 {T2}, {T0}, 0", f"s_add_i32 {S_DHI}, {T2}, {T1}", f"s_add_i32 {S_DHI}, {S_DHI}, 15",
                f"s_min_i32 {T2}, {T0}, 0", f"s_sub_i32 {S_DLO}, {T2}, {T1}"]

    def chunk_band(self):
        """per chunk c = S_C (columns 16c + 1 .. 16c + 16): S_LIVE = the words [lo, hi) with a row inside the band in one of these columns,
        hi = min(W, (16c + 16 + dhi - 1) / 64 + 1), lo = max(0, 16c + 1 + dlo - 1) / 64; the hn_c of every word that does not run is zero"""
        L = [f"s_lshl_b32 {T0}, {S_C}, 4", f"s_add_u32 {T1}, {T0}, {S_DHI}", f"s_lshr_b32 {T1}, {T1}, 6", f"s_add_u32 {T1}, {T1}, 1", f"s_min_u32 {T1}, {T1}, {self.W}",
             f"s_add_i32 {T2}, {T0}, {S_DLO}", f"s_max_i32 {T2}, {T2}, 0", f"s_lshr_b32 {T2}, {T2}, 6",
             f"s_bfm_b32 {S_LIVE}, {T1}, 0", f"s_bfm_b32 {T2}, {T2}, 0", f"s_andn2_b32 {S_LIVE}, {S_LIVE}, {T2}"]
        for w in range(self.W - 1):
            L += [f"s_bitcmp1_b32 {S_LIVE}, {w}", f"s_cselect_b32 {T0}, -1, 0", f"v_and_b32 v{self.HNC[w]}, {T0}, v{self.HNC[w]}"]
        return L

    def column(self, i, gather=()):
        W, HNC = self.W, self.HNC
        (A_, E_, HN_, HP_), T_ = self.TMP, (6, 7)
        R = self.slots[i % 2]
        # this column's W reads have arrived; the next column's W may still be in flight.  VCC = all ones: word 0's (or the first live word's) + 1
        L = [f"s_waitcnt lgkmcnt({W})", "s_mov_b64 vcc, -1"]
        self.uid += 1
        for w in range(W):
            VP_, VN_, PM = self.VP[w], self.VN[w], R[w]
            ops = {"x": [f"v_or_b32 v{PM[0]}, v{HNC[w - 1]}, v{PM[0]}"] if w else [],                                      # x |= hn_c (levenshtein.rs:847)
                   "a": [f"v_and_b32 v{A_[h]}, v{PM[h]}, v{VP_[h]}" for h in (0, 1)],
                   "S": [f"v_lshl_add_u64 {pr(A_)}, {pr(A_)}, 0, {pr(VP_)}"],
                   "e": [f"v_bitop3_b32 v{E_[h]}, v{A_[h]}, v{VP_[h]}, v{PM[h]} bitop3:0xbe" for h in (0, 1)],
                   "hp": [f"v_bitop3_b32 v{HP_[h]}, v{VN_[h]}, v{E_[h]}, v{VP_[h]} bitop3:0xf1" for h in (0, 1)],
                   "hn": [f"v_and_b32 v{HN_[h]}, v{E_[h]}, v{VP_[h]}" for h in (0, 1)],
                   "hnc": [f"v_lshrrev_b32 v{HNC[w]}, 31, v{HN_[1]}"] if w + 1 < W else [],                                  # :857-858
                   # HP' = HP + HP + carry, two v_addc_co_u32; VCC leaves word w's high half as word w + 1's carry (:865-866)
                   "hq": [f"v_addc_co_u32 v{HP_[h]}, vcc, v{HP_[h]}, v{HP_[h]}, vcc" for h in (0, 1)],
                   "t": [f"v_bitop3_b32 v{T_[h]}, v{E_[h]}, v{VN_[h]}, v{HP_[h]} bitop3:0x01" for h in (0, 1)],
                   "tor": [f"v_or_b32 v{T_[0]}, v{HNC[w - 1]}, v{T_[0]}"] if w else [],
                   "vn": [f"v_bitop3_b32 v{VN_[h]}, v{HP_[h]}, v{E_[h]}, v{VN_[h]} bitop3:0xe0" for h in (0, 1)],
                   "vp": [f"v_lshl_add_u64 {pr(VP_)}, {pr(HN_)}, 1, {pr(T_)}"]}
            skip = f"Lb{self.uid}w{w}_%="
            if BAND:
                L += [f"s_bitcmp1_b32 {S_LIVE}, {w}", f"s_cbranch_scc0 {skip}"]
            for j, tok in enumerate(self.TOKENS):
                L += ops[tok]
                if ops[tok] and self.nop_mask >> j & 1:
                    L.append("s_nop 0")
            if BAND:
                L.append(f"{skip}:")
        return L + list(gather)

    def state_init(self):  # levenshtein.rs:454-455 per word
        L = []
        for w in range(self.W):
            L += [f"v_mov_b32 v{self.VP[w][0]}, -1", f"v_mov_b32 v{self.VP[w][1]}, -1", f"v_mov_b32 v{self.VN[w][0]}, 0", f"v_mov_b32 v{self.VN[w][1]}, 0"]
        return L


class Lcs6Kind(Kind):
    """The LCS recurrence (Kind "lcs64") over the 6-BIT PAYLOAD of a single-length corpus (rf_pack.hip pack6_kernel; round 5): a chunk is three
    dwords per lane -- symbol j on bits 6 j .. 6 j + 5 of the 96-bit value -- fetched with global_load_dwordx3 (768 contiguous bytes per wavefront instead of 1 KiB).
    The Indel / LCS scans wait for HBM on the 8-bit payload (25 % fewer bytes: nothing, as long as the column costs the 25 cycles hipcc's form of it does;
    a 21-cycle column: nothing on 8-bit bytes; both together: this kernel).  The table row address of column j is a shift and a mask instead of one SDWA shift:
    v_lshrrev_b32 in its LONG encoding (a 4-byte one next to the half-rate v_lshl_add_u64 issues at 4 cycles) + v_and_b32 with the literal 0x1f8, v_alignbit_b32
    for the two fields that straddle a dword (profiles/lcs_cycles_r05.txt: 21.0 cycles per column register-only, 25.1 as hipcc writes it).
    Single-length corpora; whole chunks only: a length that is not a multiple of 16 is filled up with the code 63 by the packer (corpora that store at most 63
    symbols), whose table row the prologue zeroes (flags bit 3) -- an LCS column over it is a no-op."""
    chunk_dwords, chunk_pitch = 3, 768

    def __init__(self, bits, bufs, nop_mask):
        Kind.__init__(self, f"lcs{bits}", bits, 8, bufs, (60, 61)[: bits // 32], nop_mask)
        self.name6 = "lcs6" if bits == 64 else "lcs6n"  # (n = narrow: queries of <= 32 symbols, 32-bit words)
        # single-length corpora whose length is not a whole number of chunks: the 32-bit column waits for HBM, so it runs the packer's fill columns for free (whole
        # chunks only, no shifting); the issue-bound 64-bit column shifts the partial chunk into place and runs the real columns only (measured with fill columns:
        # 75.7 -> 74.6 Gpairs/s at 57 symbols, profiles/lcs_cycles_r05.txt)
        self.no_partial = bits == 32

    def gather(self, j, use, nxt):
        base, jj = (use, j) if j < 16 else (nxt, j - 16)
        o = 6 * jj
        d, sh = o // 32, o % 32
        a, src = ADDR[j % 4], base + d
        ks, mask = self.ks, hex(63 << self.ks)  # row offset = symbol x 8 (64-bit rows) or x 4
        if sh + 6 > 32:
            x = [f"v_alignbit_b32 v{a}, v{src + 1}, v{src}, {sh - ks}", f"v_and_b32 v{a}, {mask}, v{a}"]
        elif sh > ks:
            x = [f"v_lshrrev_b32_e64 v{a}, {sh - ks}, v{src}", f"v_and_b32 v{a}, {mask}, v{a}"]
        elif sh == ks:
            x = [f"v_and_b32 v{a}, {mask}, v{src}"]
        else:
            x = [f"v_lshlrev_b32_e64 v{a}, {ks - sh}, v{src}", f"v_and_b32 v{a}, {mask}, v{a}"]
        return x + [f"ds_read_b64 {pr(self.rows[j % 8])}, v{a}" if self.bits == 64 else f"ds_read_b32 v{self.rows[j % 8][0]}, v{a}"]


def dispatch(lo, hi, L, sfx):  # binary tree of scalar compares over k in [lo, hi]
    if lo == hi:
        L.append(f"s_branch Ls{lo}_{sfx}")
        return
    mid = (lo + hi + 1) // 2
    L += [f"s_cmp_lt_u32 {S_K}, {mid}", f"s_cbranch_scc0 Ld{mid}_{hi}_{sfx}"]
    dispatch(lo, mid - 1, L, sfx)
    L.append(f"Ld{mid}_{hi}_{sfx}:")
    dispatch(mid, hi, L, sfx)


def wait_vm(n, extra, tag, sfx):
    """s_waitcnt vmcnt(n) -- or vmcnt(n + extra) in the first step after a tile epilogue: the epilogue's result store and (general
    corpora) the next tile's index load were issued BEHIND the ring loads this wait is about, and memory operations return in
    issue order, so they may stay outstanding.  (Counting them in is exact only because the store is issued with every lane
    enabled, see the epilogue; waiting for them instead would put a store round trip at the top of every tile.)"""
    if not extra:
        return [f"s_waitcnt vmcnt({n})"]
    return [f"s_cmp_eq_u32 {S_AFTER}, 0", f"s_cbranch_scc1 Lw{tag}a_{sfx}", f"s_waitcnt vmcnt({n + extra})", f"s_branch Lw{tag}b_{sfx}",
            f"Lw{tag}a_{sfx}:", f"s_waitcnt vmcnt({n})", f"Lw{tag}b_{sfx}:"]


def chunk_load(K, buf):
    """this lane's next chunk into ring buffer `buf`: 16 bytes (the 8-bit payload), or 12 (the 6-bit one); v2 = lane x that"""
    n = getattr(K, "chunk_dwords", 4)
    return f"global_load_dwordx{n} v[{buf}:{buf + n - 1}], {V_OFF16}, {S_SRC} nt"


def step(K, P, extra, uniform=True):
    """fetch + the 16 columns of ring phase P; labels carry the phase and the statement's unique id"""
    R, sfx = K.ring, f"p{P}_%="
    use, nxt, refill = K.bufs[P], K.bufs[(P + 1) % R], K.bufs[(P + R - 1) % R]
    L = [chunk_load(K, refill)]  # the chunk RING-1 steps ahead
    if getattr(K, "no_partial", False) and uniform:  # (every chunk is whole: the launcher's condition)
        late = extra if R > 2 else 0
        L += wait_vm(R - 1, extra, "f", sfx)
        for i in range(16):
            if i == 8:
                L += wait_vm(R - 2, late, "h", sfx)
                L.append(f"s_mov_b32 {S_AFTER}, 0")
            L += K.column(i, K.gather(i + K.la, use, nxt))
        return L
    L += [f"s_cmp_eq_u32 {S_K}, 0", f"s_cbranch_scc1 Lfull_{sfx}"]
    # ---- a tile's last, partial chunk: shift its bytes up by k positions in place, gather its rows, enter at column k
    # (a ring of 2 fetches the NEXT chunk at the top of this very step: that load is younger than the epilogue's store and index load,
    # so nothing may be counted in -- the wait is a drain)
    late = extra if R > 2 else 0
    L += wait_vm(R - 2, late, "t", sfx)  # this chunk and the next one have arrived (only younger operations may still be out)
    L.append(f"s_mov_b32 {S_AFTER}, 0")
    if getattr(K, "chunk_dwords", 4) == 3:
        # the 6-bit payload: the chunk is a 96-bit value, k positions = 6 k bits (S_Q whole dwords, S_R8 bits: the phase's glue) -- at most 90
        u = [use + i for i in range(3)]
        L += [f"s_cmp_eq_u32 {S_R8}, 0", f"s_cbranch_scc1 Lq_{sfx}",
              f"v_alignbit_b32 v{u[2]}, v{u[2]}, v{u[1]}, {S_SH}", f"v_alignbit_b32 v{u[1]}, v{u[1]}, v{u[0]}, {S_SH}", f"v_lshlrev_b32 v{u[0]}, {S_R8}, v{u[0]}", f"Lq_{sfx}:",
              f"s_cmp_eq_u32 {S_Q}, 0", f"s_cbranch_scc1 Lsh_{sfx}", f"s_cmp_eq_u32 {S_Q}, 1", f"s_cbranch_scc0 Lq2_{sfx}",
              f"v_mov_b32 v{u[2]}, v{u[1]}", f"v_mov_b32 v{u[1]}, v{u[0]}", f"v_mov_b32 v{u[0]}, 0", f"s_branch Lsh_{sfx}",
              f"Lq2_{sfx}:", f"v_mov_b32 v{u[2]}, v{u[0]}", f"v_mov_b32 v{u[1]}, 0", f"v_mov_b32 v{u[0]}, 0", f"Lsh_{sfx}:"]
    else:
        u = [use + i for i in range(4)]
        L += [f"s_cmp_eq_u32 {S_R8}, 0", f"s_cbranch_scc1 Lq_{sfx}",
              f"v_alignbit_b32 v{u[3]}, v{u[3]}, v{u[2]}, {S_SH}", f"v_alignbit_b32 v{u[2]}, v{u[2]}, v{u[1]}, {S_SH}",
              f"v_alignbit_b32 v{u[1]}, v{u[1]}, v{u[0]}, {S_SH}", f"v_lshlrev_b32 v{u[0]}, {S_R8}, v{u[0]}", f"Lq_{sfx}:",
              f"s_cmp_eq_u32 {S_Q}, 0", f"s_cbranch_scc1 Lsh_{sfx}", f"s_cmp_eq_u32 {S_Q}, 1", f"s_cbranch_scc0 Lq2_{sfx}",
              f"v_mov_b32 v{u[3]}, v{u[2]}", f"v_mov_b32 v{u[2]}, v{u[1]}", f"v_mov_b32 v{u[1]}, v{u[0]}", f"v_mov_b32 v{u[0]}, 0", f"s_branch Lsh_{sfx}",
              f"Lq2_{sfx}:", f"s_cmp_eq_u32 {S_Q}, 2", f"s_cbranch_scc0 Lq3_{sfx}",
              f"v_mov_b32 v{u[3]}, v{u[1]}", f"v_mov_b32 v{u[2]}, v{u[0]}", f"v_mov_b32 v{u[1]}, 0", f"v_mov_b32 v{u[0]}, 0", f"s_branch Lsh_{sfx}",
              f"Lq3_{sfx}:", f"v_mov_b32 v{u[3]}, v{u[0]}", f"v_mov_b32 v{u[2]}, 0", f"v_mov_b32 v{u[1]}, 0", f"v_mov_b32 v{u[0]}, 0", f"Lsh_{sfx}:"]
    dispatch(1, 15, L, sfx)
    for k in range(1, 16):
        L += [f"Ls{k}_{sfx}:", "s_waitcnt lgkmcnt(0)"]
        if K.name == "osa" and (k - 1) % 8 != 7:  # the previous column's row (zero at a tile's start) sits in slot 7: move it to where column k looks
            L += [f"v_mov_b32 v{K.rows[(k - 1) % 8][h]}, v{K.rows[7][h]}" for h in (0, 1)]
        for j in range(k, k + K.la):
            L += K.gather(j, use, nxt)
        L.append(f"s_branch Lc{k}_{sfx}")
    # ---- a whole chunk
    L.append(f"Lfull_{sfx}:")
    L += wait_vm(R - 1, extra, "f", sfx)  # this chunk's dwords 2, 3 are about to be read
    for i in range(16):
        if i == 8:
            L += wait_vm(R - 2, late, "h", sfx)  # from here on the look-ahead reads the NEXT chunk's dwords 0, 1
            L.append(f"s_mov_b32 {S_AFTER}, 0")
        L.append(f"Lc{i}_{sfx}:")
        L += K.column(i, K.gather(i + K.la, use, nxt))
    return L


def tile_desc(tile_reg, uniform, fetch, six=False):
    """descriptor of tile `tile_reg` into the fetch cursor (fbase, fn) or the process cursor (len2, nch, slot0)"""
    L = []
    if not uniform:
        L += [f"s_lshl_b32 {T1}, {tile_reg}, 4", "s_nop 0", f"s_load_dwordx4 {S_DESC}, {S_TILES}, {T1}", "s_waitcnt lgkmcnt(0)"]
        if fetch:
            if six:  # the 6-bit payload mirrors the 8-bit one at 3/4 of every offset (tile payloads are whole KiB)
                L += ["s_lshr_b64 s[60:61], s[60:61], 2", f"s_mul_hi_u32 {T2}, s60, 3", "s_mul_i32 s61, s61, 3", f"s_add_u32 s61, s61, {T2}", "s_mul_i32 s60, s60, 3"]
            L += [f"s_add_u32 {S_FBASE_LO}, s8, s60", f"s_addc_u32 {S_FBASE_HI}, s9, s61", f"s_add_u32 {S_FN}, s62, 15", f"s_lshr_b32 {S_FN}, {S_FN}, 4"]
        else:
            L += [f"s_mov_b32 {S_LEN2}, s62", f"s_mov_b32 {S_SLOT0}, s63", f"s_add_u32 {S_NCH}, s62, 15", f"s_lshr_b32 {S_NCH}, {S_NCH}, 4"]
        if DESC_PREFETCH and not fetch:  # the descriptor came in a tile ago (tile_start): no scalar round trip at the end of a tile
            L = ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 {S_LEN2}, s58", f"s_mov_b32 {S_SLOT0}, s59", f"s_add_u32 {S_NCH}, s58, 15", f"s_lshr_b32 {S_NCH}, {S_NCH}, 4"]
    elif fetch:  # tile t at t * tile_bytes (64-bit product); chunk count fixed
        L += [f"s_mul_i32 {T1}, {tile_reg}, {S_UBYTES}", f"s_mul_hi_u32 {T2}, {tile_reg}, {S_UBYTES}",
              f"s_add_u32 {S_FBASE_LO}, s8, {T1}", f"s_addc_u32 {S_FBASE_HI}, s9, {T2}"]
    return L


def tile_start(K, uniform, sfx):
    """per tile: the finishing map's constant term, the candidate index of every lane, the recurrence state"""
    L = [f"s_add_u32 {T0}, {S_LEN1}, {S_LEN2}", f"s_max_u32 {T1}, {S_LEN1}, {S_LEN2}", f"s_mul_i32 {T0}, {T0}, {S_VS}", f"s_mul_i32 {T1}, {T1}, {S_VM}",
         f"s_add_u32 {S_V0}, {T0}, {T1}"]
    if getattr(K, "W", 1) > 1 and BAND:
        L += K.tile_band()
    if uniform:
        L.append(f"v_lshl_add_u32 {V_IDX}, {S_T}, 6, {V_LANE}")  # slot = index
    else:  # idx = orig[slot0 + lane]: lands long before the tile's epilogue (the next step's counted vmcnt wait is younger)
        # (flags bit 1, "slot store": `orig` is the slot -> slot identity of the gather path, rf_api_scan.hip run_many -- the load then reads the map's first
        # line over and over (same instruction stream, same vmcnt accounting, no HBM traffic) and the epilogue makes the index itself)
        L += [f"s_lshl_b32 {T0}, {S_SLOT0}, 2", f"s_lshr_b32 {T1}, {S_SLOT0}, 30", f"s_bitcmp1_b32 {S_FLAGS}, 1", f"s_cselect_b32 {T0}, 0, {T0}",
              f"s_cselect_b32 {T1}, 0, {T1}", f"s_add_u32 {T2}, s12, {T0}", f"s_addc_u32 {T3}, s13, {T1}"]
        if getattr(K, "W", 1) > 1:  # (v3 carries hn_c there: lane * 4 is made on the spot)
            L += [f"v_lshlrev_b32 v6, 2, {V_LANE}", f"global_load_dword {V_IDX}, v6, s[54:55]"]
        else:
            L.append(f"global_load_dword {V_IDX}, {V_OFF4}, s[54:55]")
    if DESC_PREFETCH and not uniform:  # (the load shares lgkmcnt with the LDS gathers: the columns' counted waits over-wait by one operation until it lands)
        L += [f"s_add_u32 {T0}, {S_T}, {S_STRIDE}", f"s_sub_u32 {T1}, {S_TEND}, 1", f"s_min_u32 {T0}, {T0}, {T1}", f"s_lshl_b32 {T0}, {T0}, 4",
              f"s_load_dwordx4 {S_PDESC}, {S_TILES}, {T0}"]
    return L + K.state_init()


def kernel(K, uniform):
    R = K.ring
    extra = 1 if uniform else 2  # memory operations a tile epilogue leaves younger than the ring: the result store (+ the index load)
    off, o = {}, 0
    for name, size in ARGS:
        off[name] = o
        o += size
    L = []
    W = getattr(K, "W", 1)
    six = getattr(K, "chunk_dwords", 4) == 3  # the 6-bit payload

    def fetch_glue(tag):  # src = address of the chunk under the fetch cursor; advance the cursor (parks on the last valid chunk)
        G = [f"s_mul_i32 {T0}, {S_FC}, {K.chunk_pitch}" if hasattr(K, "chunk_pitch") else f"s_lshl_b32 {T0}, {S_FC}, 10",
             f"s_add_u32 {S_SRC_LO}, {S_FBASE_LO}, {T0}", f"s_addc_u32 {S_SRC_HI}, {S_FBASE_HI}, 0",
             f"s_add_u32 {S_FC}, {S_FC}, 1", f"s_cmp_lt_u32 {S_FC}, {S_FN}", f"s_cbranch_scc1 Lfok_{tag}_%=",
             f"s_add_u32 {T0}, {S_FT}, {S_STRIDE}", f"s_cmp_lt_u32 {T0}, {S_TEND}", f"s_cbranch_scc0 Lfpark_{tag}_%=",
             f"s_mov_b32 {S_FT}, {T0}"]
        G += tile_desc(S_FT, uniform, fetch=True, six=six)
        G += [f"s_mov_b32 {S_FC}, 0", f"s_branch Lfok_{tag}_%=", f"Lfpark_{tag}_%=:", f"s_sub_u32 {S_FC}, {S_FN}, 1", f"Lfok_{tag}_%=:"]
        return G

    def off_of_lane(dst, lane):  # the lane's byte offset inside a chunk row: x 16, or x 12 on the 6-bit payload
        return f"v_mul_u32_u24 {dst}, 12, {lane}" if getattr(K, "chunk_dwords", 4) == 3 else f"v_lshlrev_b32 {dst}, 4, {lane}"

    # ---- kernarg block -> s8..s32
    L += [f"s_load_dwordx8 s[8:15], %[kp], {off['data']}", f"s_load_dwordx4 s[16:19], %[kp], {off['sigma']}",
          f"s_load_dwordx8 s[20:27], %[kp], {off['tile_begin']}", f"s_load_dwordx4 s[28:31], %[kp], {off['fin_vR']}",
          f"s_load_dwordx2 s[68:69], %[kp], {off['valid_hi']}", f"s_mov_b32 {S_STRIDE}, %[stride]"]
    if W > 1:
        L += [f"s_load_dwordx{16 if W > 4 else 8} s[72:{87 if W > 4 else 79}], %[kp], {off['valid_w']}", f"s_load_dword {S_KBAND}, %[kp], {off['band_k']}"]
    L += ["v_and_b32 v1, 0x3ff, %[tid]", "s_waitcnt lgkmcnt(0)"]
    # ---- this wavefront's first tile: the workgroup's place in the deal of tiles is its id, or (flags bit 0, "xcd deal") (id % 8) * (grid / 8) + id / 8,
    # so that consecutive tiles are walked by workgroups of ONE XCD (workgroups are dispatched to the 8 XCDs round-robin)
    first_tile = [f"v_readfirstlane_b32 {T0}, v1", f"s_lshr_b32 {T0}, {T0}, 6",  # wavefront within the workgroup
                  f"s_mov_b32 {T1}, %[wg]", f"s_bitcmp1_b32 {S_FLAGS}, 0", "s_cbranch_scc0 Lnodeal_%=",
                  f"s_and_b32 {T2}, %[wg], 7", f"s_lshr_b32 {T3}, {S_STRIDE}, 5", f"s_mul_i32 {T2}, {T2}, {T3}", f"s_lshr_b32 {T1}, %[wg], 3",
                  f"s_add_u32 {T1}, {T1}, {T2}", "Lnodeal_%=:",
                  f"s_lshl_b32 {T1}, {T1}, 2", f"s_add_u32 {T1}, {T1}, {T0}", f"s_add_u32 {S_T}, {S_TBEGIN}, {T1}"]
    cursors = [f"s_mov_b32 {S_FT}, {S_T}", f"s_mov_b32 {S_FC}, 0"]
    if uniform:
        cursors += [f"s_mov_b32 {S_LEN2}, {S_ULEN}", f"s_add_u32 {S_NCH}, {S_ULEN}, 15", f"s_lshr_b32 {S_NCH}, {S_NCH}, 4", f"s_mov_b32 {S_FN}, {S_NCH}"]
    cursors += tile_desc(S_T, uniform, fetch=True, six=six)
    if not uniform:
        cursors += [f"s_mov_b32 {S_LEN2}, s62", f"s_mov_b32 {S_SLOT0}, s63", f"s_mov_b32 {S_NCH}, {S_FN}"]
    prefetch = []  # the first RING-1 chunks of the stream
    for b in range(R - 1):
        prefetch += fetch_glue(f"pre{b}")
        prefetch.append(chunk_load(K, K.bufs[b]))
    if EARLY_FETCH:
        # (round 5) the stream's first chunks are requested BEFORE the pattern table is staged: the two round trips overlap instead of following each
        # other at the head of every workgroup (a wavefront without a tile still stages its share of the table and meets the barrier)
        L += first_tile + ["v_and_b32 v2, 63, v1", off_of_lane("v2", "v2"), f"s_cmp_ge_u32 {S_T}, {S_TEND}", "s_cbranch_scc1 Lnotile_%="]
        L += cursors + prefetch + ["Lnotile_%=:"]
    # ---- stage the pattern table: thread i puts row i at row sigma(i) (the corpus stores renamed symbols)
    if W == 1:
        L += [f"global_load_ubyte v6, v1, {S_SIGMA}", "v_lshlrev_b32 v7, 3, v1",
              f"global_load_dwordx2 v[8:9], v7, {S_PM}" if K.bits == 64 else f"global_load_dword v8, v7, {S_PM}",
              "s_waitcnt vmcnt(0)", f"v_lshlrev_b32 v6, {K.ks}, v6"]
        if getattr(K, "no_partial", False):
            # the 6-bit payload fills a partial last chunk up with the code 63 (flags bit 3: the corpus stores no such symbol): its table row is zero, so that an
            # LCS column over it changes nothing -- whatever original symbol the renaming happens to give that rank
            L += [f"s_bitcmp1_b32 {S_FLAGS}, 3", "s_cbranch_scc0 Lnofill_%=", f"v_cmp_eq_u32 vcc, {63 << K.ks}, v6", "v_cndmask_b32_e64 v8, v8, 0, vcc"]
            L += ["v_cndmask_b32_e64 v9, v9, 0, vcc"] if K.bits == 64 else []
            L += ["Lnofill_%=:"]
        L += ["ds_write_b64 v6, v[8:9]" if K.bits == 64 else "ds_write_b32 v6, v8"]
        if six and uniform:
            # f64 results (flags bit 4: normalized_distance / normalized_similarity / fuzz::ratio): on a single-length corpus the value is a function of the u32
            # distance alone, and the HOST tabulates it (StreamAsmArgs::vtab: the reference's own division and cutoff compare, 256 doubles in the kernarg block) --
            # thread i stages entry i behind the pattern table; the epilogue is one ds_read_b64 and an 8-byte store
            L += [f"s_bitcmp1_b32 {S_FLAGS}, 4", "s_cbranch_scc0 Lnovt_%=", f"global_load_dwordx2 v[12:13], v7, %[kp] offset:{off['vtab']}", "s_waitcnt vmcnt(0)",
                  "ds_write_b64 v7, v[12:13] offset:2048", "Lnovt_%=:"]
    else:  # row i of the host table (W consecutive words) -> word w to plane w, row sigma(i)
        L += [f"global_load_ubyte v6, v1, {S_SIGMA}", f"v_mul_u32_u24 v7, {8 * W}, v1"]
        L += [f"global_load_dwordx2 v[{24 + 2 * w}:{25 + 2 * w}], v7, {S_PM}" + (f" offset:{8 * w}" if w else "") for w in range(W)]
        L += ["s_waitcnt vmcnt(0)", "v_lshlrev_b32 v6, 3, v6"]
        L += [f"ds_write_b64 v6, v[{24 + 2 * w}:{25 + 2 * w}]" + (f" offset:{2048 * w}" if w else "") for w in range(W)]
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    if not EARLY_FETCH:
        L += first_tile
    # ---- lane constants
    L += ["v_and_b32 v1, 63, v1", off_of_lane("v2", "v1"), "v_lshlrev_b32 v3, 2, v1", "v_mov_b32 v5, 0", f"v_mov_b32 {V_KS}, {K.ks}",
          f"s_cmp_ge_u32 {S_T}, {S_TEND}", "s_cbranch_scc1 Lexit_%=", f"s_mov_b32 {S_C}, 0", f"s_mov_b32 {S_AFTER}, 0"]
    if not EARLY_FETCH:
        L += cursors
    L += tile_start(K, uniform, "%=")
    if not EARLY_FETCH:
        L += prefetch
    # ---- the rows of the first chunk's first columns
    L.append(f"s_waitcnt vmcnt({R - 2})")  # buffer 0 has arrived (a partial first chunk gathers its own rows again: harmless)
    for j in range(K.la):
        L += K.gather(j, K.bufs[0], K.bufs[1])
    # ---- the phases
    for P in range(R):
        L.append(f"Lphase{P}_%=:")
        L += fetch_glue(f"ph{P}")
        L += [f"s_lshl_b32 {T0}, {S_C}, 4", f"s_sub_u32 {T0}, {S_LEN2}, {T0}", f"s_mov_b32 {S_K}, 0",  # columns left in this tile
              f"s_cmp_ge_u32 {T0}, 16", f"s_cbranch_scc1 Lkok_{P}_%=",
              f"s_sub_u32 {S_K}, 16, {T0}"]
        if six:  # k positions = 6 k bits of the 96-bit chunk
            L += [f"s_mul_i32 {S_R8}, {S_K}, 6", f"s_lshr_b32 {S_Q}, {S_R8}, 5", f"s_and_b32 {S_R8}, {S_R8}, 31"]
        else:
            L += [f"s_lshr_b32 {S_Q}, {S_K}, 2", f"s_and_b32 {S_R8}, {S_K}, 3", f"s_lshl_b32 {S_R8}, {S_R8}, 3"]
        L += [f"s_sub_u32 {S_SH}, 32, {S_R8}", f"Lkok_{P}_%=:"]
        if W > 1 and BAND:
            L += K.chunk_band()
        L += step(K, P, extra, uniform)
        L += [f"s_add_u32 {S_C}, {S_C}, 1", f"s_cmp_lt_u32 {S_C}, {S_NCH}"]
        if P + 1 < R:
            L += [f"s_cbranch_scc1 Lphase{P + 1}_%=", f"s_mov_b32 {S_NEXT}, {P + 1}", "s_branch Lepi_%="]
        else:
            L += [f"s_cbranch_scc1 Lphase0_%=", f"s_mov_b32 {S_NEXT}, 0"]  # falls into the epilogue
    # ---- tile epilogue
    L.append("Lepi_%=:")
    if W > 1:  # D[len1][len2] = len2 + sum over the words of popcount(VP & valid_w) - popcount(VN & valid_w)
        L += ["v_mov_b32 v6, 0", "v_mov_b32 v7, 0"]
        for w in range(W):
            for h in (0, 1):
                L += [f"v_and_b32 v8, s{72 + 2 * w + h}, v{K.VP[w][h]}", "v_bcnt_u32_b32 v6, v8, v6",
                      f"v_and_b32 v8, s{72 + 2 * w + h}, v{K.VN[w][h]}", "v_bcnt_u32_b32 v7, v8, v7"]
    elif K.name == "lcs32":
        L += ["v_not_b32 v6, v60", f"v_and_b32 v6, {S_VLO}, v6", "v_bcnt_u32_b32 v6, v6, 0"]
    elif K.name == "lcs64":  # the LCS length: the zero bits of S among the query's rows (rf_device.hpp LcsState::result)
        L += ["v_not_b32 v6, v60", f"v_and_b32 v6, {S_VLO}, v6", "v_bcnt_u32_b32 v6, v6, 0", "v_not_b32 v7, v61", f"v_and_b32 v7, {S_VHI}, v7", "v_bcnt_u32_b32 v6, v7, v6"]
    elif K.bits == 64:
        L += [f"v_and_b32 v6, {S_VLO}, v60", "v_bcnt_u32_b32 v6, v6, 0", f"v_and_b32 v7, {S_VHI}, v61", "v_bcnt_u32_b32 v6, v7, v6",
              f"v_and_b32 v7, {S_VLO}, v62", "v_bcnt_u32_b32 v7, v7, 0", f"v_and_b32 v8, {S_VHI}, v63", "v_bcnt_u32_b32 v7, v8, v7"]
    else:
        L += [f"v_and_b32 v6, {S_VLO}, v60", "v_bcnt_u32_b32 v6, v6, 0", f"v_and_b32 v7, {S_VLO}, v61", "v_bcnt_u32_b32 v7, v7, 0"]
    if K.name not in ("lcs64", "lcs32"):
        L += ["v_sub_u32 v6, v6, v7", f"v_add_u32 v6, {S_LEN2}, v6"]                   # raw = len2 + pp - pn
    L += [f"v_mul_lo_u32 v6, v6, {S_VR}", f"v_add_u32 v6, {S_V0}, v6",                  # value = v0 + vR * raw
          f"v_xor_b32 v7, {S_FLIP}, v6", f"v_cmp_ge_u32 vcc, {S_CFLIP}, v7", "v_cndmask_b32 v6, -1, v6, vcc"]  # None unless (value ^ flip) <= cflip
    if not uniform:
        # the index load was issued at the tile's start, BEHIND the ring loads then in flight: all but the newest operation done
        # means it has landed (the over-wait is the chunk fetched one step ago, which the next block needs at its column 8 anyway)
        # (round 5: not vmcnt(1) whatever the ring -- since the index load one chunk per step has been requested, nch of them, and at most R - 1 ring loads are
        # outstanding at a tile's end, all younger than it once nch >= R - 1: "no more than min(nch, R - 1) operations outstanding" is exactly "the index has
        # landed", and a ring of 3 or 4 no longer drains to its newest chunk at the end of every tile)
        for y in range(1, R - 1):
            L += [f"s_cmp_lt_u32 {S_NCH}, {y + 1}", f"s_cbranch_scc0 Lix{y}_%=", f"s_waitcnt vmcnt({y})", "s_branch Lixd_%=", f"Lix{y}_%=:"]
        L += [f"s_waitcnt vmcnt({R - 1})", "Lixd_%=:"]
        # slot store (flags bit 1): index = slot, every lane stores (padding lanes own a slot of the temporary)
        L += [f"s_bitcmp1_b32 {S_FLAGS}, 1", "s_cbranch_scc0 Lnoslot_%=", f"v_add_u32 {V_IDX}, {S_SLOT0}, {V_LANE}", "Lnoslot_%=:"]
    if six and uniform:  # f64 results: the value's double from the staged table (value <= 255: the launcher's condition)
        L += [f"s_bitcmp1_b32 {S_FLAGS}, 4", "s_cbranch_scc0 Lu32_%=", "v_lshlrev_b32 v7, 3, v6", "ds_read_b64 v[6:7], v7 offset:2048",
              f"v_cmp_gt_u32 vcc, {S_N}, {V_IDX}", f"v_lshl_add_u64 v[8:9], v[4:5], 3, {S_OUT}", "s_waitcnt lgkmcnt(0)",
              "s_cmp_eq_u64 vcc, -1", "s_cbranch_scc0 Lpart8_%=",
              "global_store_dwordx2 v[8:9], v[6:7], off", f"s_mov_b32 {S_AFTER}, 1", "s_branch Lstored_%=",
              "Lpart8_%=:", f"s_and_saveexec_b64 {S_EXEC}, vcc", "global_store_dwordx2 v[8:9], v[6:7], off", f"s_mov_b64 exec, {S_EXEC}",
              "s_waitcnt vmcnt(0)", f"s_mov_b32 {S_AFTER}, 0", "s_branch Lstored_%=", "Lu32_%=:"]
    L += [f"v_cmp_gt_u32 vcc, {S_N}, {V_IDX}" if uniform else f"v_cmp_ne_u32 vcc, -1, {V_IDX}",                 # real candidates only
          f"v_lshl_add_u64 v[8:9], v[4:5], 2, {S_OUT}",
          "s_cmp_eq_u64 vcc, -1", "s_cbranch_scc0 Lpart_%=",
          "global_store_dword v[8:9], v6, off", f"s_mov_b32 {S_AFTER}, 1", "s_branch Lstored_%=",
          # a tile with padding lanes (the last of its length): store under the mask and drain, so that the next step's counted waits
          # see no operation of unknown presence
          "Lpart_%=:", f"s_and_saveexec_b64 {S_EXEC}, vcc", "global_store_dword v[8:9], v6, off", f"s_mov_b64 exec, {S_EXEC}",
          "s_waitcnt vmcnt(0)", f"s_mov_b32 {S_AFTER}, 0", "Lstored_%=:",
          f"s_add_u32 {S_T}, {S_T}, {S_STRIDE}", f"s_cmp_ge_u32 {S_T}, {S_TEND}", "s_cbranch_scc1 Lexit_%=", f"s_mov_b32 {S_C}, 0"]
    L += tile_desc(S_T, uniform, fetch=False)
    L += tile_start(K, uniform, "%=")
    for P in range(1, R):
        L += [f"s_cmp_eq_u32 {S_NEXT}, {P}", f"s_cbranch_scc1 Lphase{P}_%="]
    L += ["s_branch Lphase0_%=", "Lexit_%=:"]
    return L


def macro(name, lines):
    return [f"#define {name} \\"] + [f'    "{l}\\n\\t" \\' for l in lines[:-1]] + [f'    "{lines[-1]}\\n"']


RING3 = [18, 22, 26]
KINDS = [
    # name, word bits, look-ahead, ring buffers (first VGPR of each), state registers, s_nop mask over the column's tokens
    Kind("lev64", 64, 8, [14, 18, 22, 26][4 - int(os.environ.get("RF_GEN_RING64", "3")):], range(60, 64), int(os.environ.get("RF_GEN_MASK64", "0x1B3"), 0)),  # (RF_GEN_*: experiment knobs, tools/build_stream_variant.sh)
    Kind("lev32", 32, 8, [42, 46, 22, 26][: int(os.environ.get("RF_GEN_RING32", "4"))], (60, 61), int(os.environ.get("RF_GEN_MASK32", "0x80"), 0)),
    Kind("osa", 64, 7, RING3, range(58, 64), int(os.environ.get("RF_GEN_MASKOSA", "0x613"), 0)),
    # (the LCS column over the 8-bit payload -- Kind("lcs64", 64, 8, [14, 18, 22, 26], (60, 61), 0), any corpus -- was built and measured in round 5: its
    # 18.6-cycle column buys nothing where the compiled scan already waits for HBM (single-length 100 M: 78.6 vs 79.9 Gpairs/s, ragged 20 M: 72.9 vs 73.9):
    # only the 6-bit form below ships, profiles/lcs_cycles_r05.txt)
    Lcs6Kind(64, [14, 18, 22, 26][4 - int(os.environ.get("RF_GEN_RINGLCS6", "4")):], int(os.environ.get("RF_GEN_MASKLCS6", "0"), 0)),
    Lcs6Kind(32, [14, 18, 22, 26], 0),
] + [BlockKind(W, int(os.environ.get("RF_GEN_MASKW", "0x0426"), 0)) for W in (2, 3, 4, 5, 6, 7, 8)]  # s_nop behind a, S, hn, vn: best of 22 placements (profiles/levw_nop_masks_r04.txt)


def main():
    out = ["// GENERATED by tools/gen_stream_asm.py -- do not edit.  See that file for the design; rf_stream_asm.hip for the wrapper.",
           "// kernarg block offsets the asm bodies were generated for (rf_stream_asm.hip static_asserts StreamAsmArgs against them)"]
    o = 0
    for name, size in ARGS:
        out.append(f"#define RF_STREAM_ARG_{name.upper()} {o}")
        o += size
    out.append(f"#define RF_STREAM_ARGS_SIZE {o}")
    out.append('#define RF_STREAM_CLOBBERS ' + ", ".join(f'"v{r}"' for r in range(1, 64)) + ", " + ", ".join(f'"s{r}"' for r in range(8, 92) if r != 32)
               + ', "vcc", "scc", "memory"')  # (exec is restored to all ones before the body ends)
    out.append('#define RF_STREAM_CLOBBERS_WIDE ' + ", ".join(f'"v{r}"' for r in range(1, 104)) + ", " + ", ".join(f'"s{r}"' for r in range(8, 92) if r != 32)
               + ', "vcc", "scc", "memory"')  # (queries of 257 .. 512 symbols: 104 VGPRs = 4 wavefronts per SIMD)
    for K in KINDS:
        for uniform in (True, False):
            out.append(f"// ---- {getattr(K, 'name6', K.name)}, {'single-length corpus (tile t at t * tile_bytes, slot = index)' if uniform else 'tile descriptors + orig[]'}: "
                       f"ring of {K.ring} at v{', v'.join(str(b) for b in K.bufs)}, look-ahead {K.la}")
            out += macro(f"RF_STREAM_{getattr(K, 'name6', K.name).upper()}_{'UNIFORM' if uniform else 'TILES'}_ASM", kernel(K, uniform))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rapidfuzz_rs_amd", "csrc", "rf_stream_asm.inc")
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
