"""The measurement table of DESIGN.md section 6 is generated from the committed bench lines (profiles/bench_*.json): it must be
current, so the prose cannot drift from the evidence."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_bench_table_is_generated_from_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "--check"])
    assert r.returncode == 0, "DESIGN.md section 6 is stale: run python tools/design_table.py"


def test_every_profile_the_design_cites_exists():
    import re

    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for name in set(re.findall(r"profiles/([A-Za-z0-9_.]+\.(?:txt|json))", text)):
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), name


def test_asm_chunk_include_is_the_generators_output(tmp_path):
    """rapidfuzz_rs_amd/csrc/rf_lev_chunk_asm.inc (the hand-scheduled 16-column chunk) is generated: the committed file must be
    what tools/gen_lev_chunk_asm.py writes."""
    out = tmp_path / "chunk.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("RF_GEN_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_lev_chunk_asm.py"), str(out)], check=True, env=env)
    assert out.read_text() == open(os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_lev_chunk_asm.inc")).read()


def test_jaro_asm_include_is_the_generators_output(tmp_path):
    out = tmp_path / "jaro.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("RF_GEN_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_jaro_chunk_asm.py"), str(out)], check=True, env=env)
    assert out.read_text() == open(os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_jaro_chunk_asm.inc")).read()


def test_band_asm_include_is_the_generators_output(tmp_path):
    out = tmp_path / "band.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_band_asm.py"), str(out)], check=True)
    assert out.read_text() == open(os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_band_asm.inc")).read()


def test_stream_asm_include_is_the_generators_output(tmp_path):
    """rapidfuzz_rs_amd/csrc/rf_stream_asm.inc (the whole-kernel asm bodies of the no-cutoff Levenshtein / OSA / LCS scans) is generated
    and, since round 6, NOT tracked (100 k lines that tripled the history per generator change): the Makefile writes it from
    tools/gen_stream_asm.py.  The file the library was built from must be what the generator writes now, and its kernarg offsets
    those of the struct the wrapper passes (rf_stream_asm.hip static_asserts them at build time; here: every ARGS entry has a field
    of the same name)."""
    inc = os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_stream_asm.inc")
    if not os.path.exists(inc):  # a fresh checkout that has not been built yet
        subprocess.run(["make", "-C", os.path.dirname(inc), "rf_stream_asm.inc"], check=True)
    out = tmp_path / "stream.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("RF_GEN_")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_stream_asm.py"), str(out)], check=True, env=env)
    assert out.read_text() == open(os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_stream_asm.inc")).read()
    import re

    hip = open(os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_stream_asm.hip")).read()
    struct = hip[hip.index("struct StreamAsmArgs {"): hip.index("};", hip.index("struct StreamAsmArgs {"))]
    for name in re.findall(r"#define RF_STREAM_ARG_(\w+) \d+", out.read_text()):
        assert re.search(rf"\b{name.lower()}\b", struct.lower()), name
