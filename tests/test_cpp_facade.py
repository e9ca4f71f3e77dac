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


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_cpp_host_threads_stress_on_gpu(tmp_path):
    """tests/cpp/stress_threads.cpp: eight host threads, one HIP stream each, a shuffled mix of every call that fills a per-corpus
    cache or takes per-call scratch, on three shared corpora -- every result equal to the single-threaded one.  Round 4: this is what
    found hipMallocAsync handing one thread's live scratch to another (tools/mempool_repro.hip); the library now has its own
    stream-ordered scratch allocator (rf_scratch.hip).  Python threads cannot stand in for it: they start too far apart."""
    N.lib()
    exe = str(tmp_path / "stress_threads")
    libdir = os.path.dirname(N.LIB_PATH)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "stress_threads.cpp"), "-o", exe,
                    "-L", libdir, "-lrfgpu", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"], check=True, capture_output=True)
    for _ in range(3):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and ", 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
