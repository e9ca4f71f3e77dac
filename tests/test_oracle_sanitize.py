"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference fuzzes for crashes with
libFuzzer; the oracle is C, so its known-answer and differential tests are re-run on a sanitizer build).  The instrumented
library is loaded into an ordinary python through LD_PRELOAD of gcc's libasan; any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_file(name):
    r = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(600)
def test_oracle_known_answers_under_asan_ubsan():
    libasan = _gcc_file("libasan.so")
    if libasan is None:
        pytest.skip("gcc's libasan.so is not installed")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "librf_oracle_asan.so"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=libasan, RF_ORACLE_LIB=os.path.join(ROOT, "oracle", "librf_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the reference's own vectors (incl. the 106 k x 107 k OCR pair) and the randomized differential tests against the textbook DPs
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_known_answers.py",
                        "tests/test_oracle_vs_textbook.py"], cwd=ROOT, env=env, capture_output=True, text=True)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert " passed" in out
