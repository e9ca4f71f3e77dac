// Locates librfgpu.so: RFGPU_LIB_DIR, or <repo>/rapidfuzz_rs_amd next to this crate (built by `make -C rapidfuzz_rs_amd/csrc`
// with hipcc --offload-arch=gfx950).  At run time the loader additionally needs libamdhip64.so (ROCm >= 7.0).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("RFGPU_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../rapidfuzz_rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rfgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RFGPU_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/rfgpu.h");
}
