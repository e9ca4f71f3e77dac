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

/// `rf_release_caches`: gives back what the library keeps per process between calls (the streamed scans' pinned + device buffer sets, parked scratch
/// blocks).  No reference analogue; a long-lived service calls it after a burst.  What a [`Corpus`] keeps goes with the corpus.
pub fn release_caches() {
    // (always RF_OK)
    let _ = unsafe { sys::rf_release_caches() };
}
pub use metric::{Args, DistanceCutoff, Element, Error, NoScoreCutoff, SimilarityCutoff, TopK, WeightTable, WithScoreCutoff};

/// `rapidfuzz::distance::*` -- one module per metric with a bit-parallel batch path.  Each has `BatchComparator<Elem1>` with the
/// reference's eight methods (`distance`, `distance_with_args`, `similarity`, ..., `normalized_similarity_with_args`), the eight free
/// functions of the same names, and the `*_many` family (the caller's loop over candidates as one scan).
pub mod distance {
    crate::metric::usize_metric!(levenshtein, RF_LEVENSHTEIN, "levenshtein::BatchComparator (src/distance/levenshtein.rs:1636-1818)");
    crate::metric::usize_metric!(indel, RF_INDEL, "indel::BatchComparator (src/distance/indel.rs:375-521)");
    crate::metric::usize_metric!(lcs_seq, RF_LCS_SEQ, "lcs_seq::BatchComparator (src/distance/lcs_seq.rs:800-949)");
    crate::metric::usize_metric!(osa, RF_OSA, "osa::BatchComparator (src/distance/osa.rs:431-461)");
    crate::metric::f64_metric!(jaro, RF_JARO, "jaro::BatchComparator (src/distance/jaro.rs:830-977)");
    crate::metric::f64_metric!(jaro_winkler, RF_JARO_WINKLER, "jaro_winkler::BatchComparator (src/distance/jaro_winkler.rs:404-575)");
}

/// `rapidfuzz::fuzz` (src/fuzz.rs): `ratio`, `ratio_with_args`, `RatioBatchComparator`.  Reproduces fuzz.rs:141 (quirk Q1: the batch
/// comparator normalises by max(len1, len2) through its inner LCS comparator) unless `Args::ratio_indel_normalization()`.
pub mod fuzz {
    use crate::metric::*;
    use crate::sys::*;
    use crate::Corpus;
    pub use crate::metric::Args;
    crate::metric::comparator_core!(RatioBatchComparator, RF_FUZZ_RATIO);
    impl<Elem1: Element> RatioBatchComparator<Elem1> {
        /// fuzz.rs:117-125
        pub fn similarity<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
            self.similarity_with_args(s2, &Args::default())
        }
        /// fuzz.rs:127-150
        pub fn similarity_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
        where
            Iter2: IntoIterator<Item = Elem1>,
            CutoffType: SimilarityCutoff<f64>,
        {
            let s2: Vec<Elem1> = s2.into_iter().collect();
            args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_SIMILARITY, &args.lower(args.score_cutoff.cutoff())))
        }
        /// `for c in corpus { self.similarity_with_args(c, args) }` as one scan
        pub fn similarity_many<C: SimilarityCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
            Ok(many_f64(self.h, corpus, RF_OP_SIMILARITY, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
        }
    }
    /// fuzz.rs:48-58
    pub fn ratio<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
        ratio_with_args(s1, s2, &Args::default())
    }
    /// fuzz.rs:60-96: the free function normalises the documented way (2 * lcs / (len1 + len2)); only the batch comparator has quirk Q1
    pub fn ratio_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: SimilarityCutoff<f64>>(
        s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
        RatioBatchComparator::new(s1).similarity_with_args(s2, &args.ratio_indel_normalization())
    }
}
