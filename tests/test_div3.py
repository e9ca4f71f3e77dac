"""rf_jaro.hip div3(): x / 3.0 as three fma-class instructions must equal the IEEE quotient bit for bit -- for every value the Jaro table
epilogue can produce (jaro.rs:106-119) and for 50 M random doubles.  Host check (gcc's fma is exact); the device side is held by the
Jaro parity tests (bit-equal f64 against the oracle)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_div3_equals_ieee_division(tmp_path):
    exe = tmp_path / "div3_check"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "div3_check.c"), "-lm"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout
    # the kernel's constant is the one this program checks
    src = open(os.path.join(ROOT, "rapidfuzz_rs_amd", "csrc", "rf_jaro.hip")).read()
    assert "0x1.5555555555555p-2" in src and "__builtin_fma(-3.0, q, x)" in src
