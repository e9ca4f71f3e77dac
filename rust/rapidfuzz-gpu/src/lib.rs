//! `rapidfuzz-gpu`: the batch path of rapidfuzz-rs (`distance::*::BatchComparator`, `fuzz::RatioBatchComparator`) served by
//! hand-written gfx950 kernels through the C ABI of `include/rfgpu.h`.
//!
//! The reference crate is `#![forbid(unsafe_code)]` (src/lib.rs:78) and has no FFI, so the binding lives in this companion
//! crate.  Module and method names follow the reference (`levenshtein::BatchComparator::new(s1).distance(s2)`); the one
//! addition is the `*_many` family, which is the user's `for candidate in corpus { scorer.distance(candidate) }` loop
//! (rapidfuzz-benches/benches/bench_levenshtein.rs:51-60) as a single call over a [`Corpus`] kept in HBM.
//!
//! **Status: unbuilt and untested in the repository's image (no cargo / rustc).**  `sys.rs` is generated from the header
//! and checked against it by `tests/test_rust_binding.py`.
pub mod sys;

mod corpus;
mod metric;

pub use corpus::Corpus;
pub use metric::{Args, Error, TopK};

/// `rapidfuzz::distance::*` -- one module per metric with a bit-parallel batch path.
pub mod distance {
    crate::metric::usize_metric!(levenshtein, RF_LEVENSHTEIN, "levenshtein::BatchComparator (src/distance/levenshtein.rs:1636-1818)");
    crate::metric::usize_metric!(indel, RF_INDEL, "indel::BatchComparator (src/distance/indel.rs:375-521)");
    crate::metric::usize_metric!(lcs_seq, RF_LCS_SEQ, "lcs_seq::BatchComparator (src/distance/lcs_seq.rs:800-949)");
    crate::metric::usize_metric!(osa, RF_OSA, "osa::BatchComparator (src/distance/osa.rs:431-461)");
    crate::metric::f64_metric!(jaro, RF_JARO, "jaro::BatchComparator (src/distance/jaro.rs:830-977)");
    crate::metric::f64_metric!(jaro_winkler, RF_JARO_WINKLER, "jaro_winkler::BatchComparator (src/distance/jaro_winkler.rs:413-575)");
}

/// `rapidfuzz::fuzz::RatioBatchComparator` (src/fuzz.rs:98-150).
pub mod fuzz {
    crate::metric::f64_metric!(ratio, RF_FUZZ_RATIO, "fuzz::RatioBatchComparator (src/fuzz.rs:98-150); reproduces fuzz.rs:141 (quirk Q1) unless Args::ratio_indel_normalization()");
}
