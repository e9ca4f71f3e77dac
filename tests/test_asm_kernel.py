"""rf_lev_asm.hip (and jaro_word_asm_kernel in rf_jaro.hip, the same way) keeps its recurrence state and look-ahead table rows in PHYSICAL registers (v34..v49, v60..v63) between asm
blocks (register-asm variables that are in/out operands of every block, so the compiler knows they are live) and uses
v30..v33 / v50..v59 as scratch inside the blocks.  Checked here on the compiler's own output (no GPU needed): nothing outside
the asm statements of that kernel reads or writes the pinned registers -- in particular no copies in and out around the blocks,
which is what any C++ use of those variables produces -- and the resource limits the launch relies on hold."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


CASES = [  # source file, mangled kernel, pinned registers, wavefronts per SIMD the launch is budgeted for, VGPR limit
    ("rf_lev_asm.hip", "_ZN2rf15lev1_asm_kernelENS_10ScanParamsE", r"(3[4-9]|4[0-9]|6[0-3])", 8, 64),
    ("rf_lev_asm.hip", "_ZN2rf16lev32_asm_kernelENS_10ScanParamsE", r"(3[4-9]|4[01]|6[01])", 8, 64),
    ("rf_lev_asm.hip", "_ZN2rf15osa1_asm_kernelENS_10ScanParamsE", r"(3[4-9]|4[0-9]|5[89]|6[0-3])", 8, 64),
    ("rf_jaro.hip", "_ZN2rf20jaro_word_asm_kernelILb0EEEvNS_10ScanParamsE", r"(5[89]|6[0-3])", 7, 72),
    ("rf_jaro.hip", "_ZN2rf20jaro_word_asm_kernelILb1EEEvNS_10ScanParamsE", r"(5[89]|6[0-3])", 7, 72),  # the conflict-free table copy (round 4)
]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("source,kernel,pinned,occupancy,vgprs", CASES)
def test_compiler_never_touches_the_pinned_registers(tmp_path, source, kernel, pinned, occupancy, vgprs):
    src = os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc")
    out = tmp_path / "kernel.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-S", "--cuda-device-only",
                    "-o", str(out), os.path.join(src, source)], check=True, stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    PINNED = re.compile(rf"\bv{pinned}\b|\bv\[{pinned}:|\bv\[\d+:{pinned}\]")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    in_asm, blocks, offenders = False, 0, []
    for l in lines[start:end]:
        if "ASMSTART" in l:
            in_asm, blocks = True, blocks + 1
        elif "ASMEND" in l:
            in_asm = False
        elif not in_asm and not l.strip().startswith(";") and PINNED.search(l):
            offenders.append(l.strip())
    assert blocks >= 4  # prologue, chunk blocks, state init, result
    assert offenders == [], offenders[:5]
    meta = "\n".join(lines[end : end + 80])
    assert re.search(r"; ScratchSize: 0\b", meta), "the asm kernel must not spill"
    assert re.search(rf"; Occupancy: {occupancy}\b", meta), f"the asm kernel is budgeted for {occupancy} wavefronts per SIMD"
    assert int(re.search(r"; NumVgprs: (\d+)", meta).group(1)) <= vgprs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_whole_kernel_asm_scans_resources(tmp_path):
    """rf_stream_asm.hip: the twenty-four whole-kernel asm scans (Levenshtein 64- / 32-bit, OSA, Levenshtein over 2 .. 8 words -- 2 .. 4 since round 4,
    5 .. 8 since round 5 -- each x single-length / tile descriptors; the LCS scans over the 6-bit payload -- 64- and 32-bit words -- x single-length / tiles).  The wrapper hands the asm
    body three operands and nothing else, so everything the launch relies on is visible in the compiler's metadata: no scratch, 64 VGPRs = 8 wavefronts
    per SIMD (104 = 4 for 5 .. 8 words), the pattern table (2 KiB per word) as the only LDS object, and a body that contains no compiler-generated code
    between its first and last instruction (ONE asm statement, then s_endpgm)."""
    src = os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc")
    out = tmp_path / "stream.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-S", "--cuda-device-only",
                    "-o", str(out), os.path.join(src, "rf_stream_asm.hip")], check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    lines = text.splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2rf\d+stream_\w+_kernelENS_13StreamAsmArgsE:", l)]
    assert len(starts) == 24
    for start in starts:
        k = lines[start].split(":")[0]
        words = int(re.search(r"levw(\d)", k).group(1)) if "levw" in k else 1
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        assert sum("ASMSTART" in l for l in body) == 1 and "ASMEND" in lines[end - 1]
        lcs6 = "lcs6" in k and "uniform" in k  # (these also serve f64 results: a second pair of stores, 8 bytes wide, and the value table in LDS behind the pattern table)
        assert sum("global_store_dword" in l for l in body) == (4 if lcs6 else 2)  # the full-tile store and the masked one of a tile with padding lanes
        meta = "\n".join(lines[end : end + 80])
        assert re.search(r"; ScratchSize: 0\b", meta) and re.search(rf"; Occupancy: {4 if words > 4 else 8}\b", meta), k
        assert int(re.search(r"; NumVgprs: (\d+)", meta).group(1)) == (104 if words > 4 else 64)
        assert re.search(rf"\.amdhsa_group_segment_fixed_size {4096 if lcs6 else 2048 * words}\b", meta), k
