#!/usr/bin/env python3
"""Dumps the 16-column inner block of a scan kernel from the hipcc -S output and tallies its instructions per column:
  make -C rapidfuzz_rs_amd/csrc asm && python tools/isa_tally.py rapidfuzz_rs_amd/csrc/rf_scan.s <mangled-kernel-substring> > profiles/isa_<name>.txt
The inner block is the longest straight-line run of instructions (no label, no branch) inside the kernel."""
import collections
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().endswith(("E:", "E: ; @" + l.split(":")[0])) or (l.startswith("_Z") and needle in l and ":" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = [l for l in lines[start:end + 1] if l.strip() and not l.strip().startswith(";")]
blocks, cur = [], []
for l in body:
    t = l.strip()
    if t.endswith(":") or t.split(":")[0].startswith(".LBB") or t.startswith(("s_cbranch", "s_branch")):
        blocks.append(cur)
        cur = []
    elif not t.startswith("."):
        cur.append(t.split(";")[0].rstrip())
blocks.append(cur)
inner = max(blocks, key=len)
kind = collections.Counter()
mn = collections.Counter()
for ins in inner:
    op = ins.split()[0]
    mn[op] += 1
    k = ("VALU 64-bit / VOP3-only (half rate)" if re.match(r"v_(lshl_add_u64|lshlrev_b64|lshrrev_b64|add_co|addc_co|sub_co|subb_co|alignbit|lshl_or|or3|bfi|cmp_\w+_u64|mul_lo)", op)
         else "VALU SDWA" if op.endswith("_sdwa") else "VALU (full rate)" if op.startswith("v_")
         else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "scratch_"))
         else "s_nop / s_waitcnt" if op in ("s_nop", "s_waitcnt") else "SALU" if op.startswith("s_") else "other")
    kind[k] += 1
cols = sum(1 for i in inner if i.startswith("ds_read"))  # one pattern-table read per column (per word)
print(f"kernel: {lines[start].split(':')[0]}")
print(f"inner block: {len(inner)} instructions, {cols} ds_read per block")
print("per-class tally (count, per 16-column chunk / 16):")
for k, v in sorted(kind.items(), key=lambda kv: -kv[1]):
    print(f"  {k:40s} {v:5d}   {v / 16:6.2f} per column")
print("per-mnemonic tally:")
for k, v in sorted(mn.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v:5d}")
print("\n---- the block ----")
print("\n".join("\t" + i for i in inner))
