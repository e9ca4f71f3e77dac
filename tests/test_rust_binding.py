"""The Rust companion crate (rust/rapidfuzz-gpu) cannot be built here (no cargo), so its raw binding is held to the header
mechanically: sys.rs must be exactly what tools/gen_rust_sys.py generates from include/rfgpu.h TODAY, and -- parsed
independently of the generator -- must declare the same symbols with the same argument names in the same order, and
RfArgs must have the field order of the C struct (which is also what the ctypes binding uses)."""
import os
import re
import subprocess
import sys

from rapidfuzz_rs_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS_RS = os.path.join(ROOT, "rust", "rapidfuzz-gpu", "src", "sys.rs")


def _header_prototypes():
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rfgpu.h")).read(), flags=re.S)
    protos = {}
    for name, args in re.findall(r"\b(rf_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        args = " ".join(args.split())
        protos[name] = [] if args in ("", "void") else [re.search(r"(\w+)$", a.strip()).group(1) for a in args.split(",")]
    return protos


def test_sys_rs_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"])
    assert r.returncode == 0, "rust/rapidfuzz-gpu/src/sys.rs is stale: run python tools/gen_rust_sys.py"


def test_sys_rs_declares_the_header():
    rs = open(SYS_RS).read()
    block = rs[rs.index('extern "C" {'):]
    rust = {name: [a.split(":")[0].strip() for a in args.split(",") if a.strip()] for name, args in re.findall(r"pub fn (rf_\w+)\((.*?)\)", block)}
    c = _header_prototypes()
    assert set(rust) == set(c) == set(N.SYMBOLS), set(rust) ^ set(c)
    for name in c:
        assert rust[name] == c[name], (name, rust[name], c[name])


def test_rf_args_field_order():
    rs = open(SYS_RS).read()
    body = re.search(r"pub struct RfArgs \{(.*?)\}", rs, flags=re.S).group(1)
    rust_fields = re.findall(r"pub (\w+):", body)
    assert rust_fields == [f[0] for f in N.RfArgs._fields_]
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rfgpu.h")).read(), flags=re.S)
    cbody = re.search(r"typedef struct rf_args \{(.*?)\} rf_args;", hdr, flags=re.S).group(1)
    c_fields = [n.strip() for decl in cbody.split(";") if decl.strip() for n in re.sub(r"^\s*\w+\s+", "", decl.strip()).split(",")]
    assert c_fields == rust_fields


def test_wrapper_only_calls_declared_symbols():
    """every rf_* the safe wrapper modules call exists in sys.rs (a renamed entry point must not survive in the wrapper)"""
    declared = set(re.findall(r"pub fn (rf_\w+)\(", open(SYS_RS).read()))
    src_dir = os.path.dirname(SYS_RS)
    for f in os.listdir(src_dir):
        if f.endswith(".rs") and f != "sys.rs":
            used = set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", open(os.path.join(src_dir, f)).read()))
            assert used <= declared, (f, used - declared)
