"""The header-only C++ facade (include/rapidfuzz_amd.hpp) compiles against the C ABI and behaves."""
import os
import shutil
import subprocess

import pytest

from rapidfuzz_rs_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    N.lib()
    exe = str(tmp_path / "facade_test")
    libdir = os.path.dirname(N.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"),
           "-o", exe, "-L", libdir, "-lrfgpu", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_cpp_facade_compiles_and_runs_cpu(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_facade_on_gpu(tmp_path):
    r = subprocess.run([_build(tmp_path), "gpu"], capture_output=True, text=True)
    assert r.returncode == 0 and "facade ok (gpu)" in r.stdout, r.stdout + r.stderr
