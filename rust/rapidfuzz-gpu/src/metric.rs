//! Shared plumbing of the metric modules: `Args` (the reference's builder structs flattened, rfgpu.h `rf_args`), errors,
//! and the two macros that stamp out one module per metric with the reference's method names.
use crate::sys::*;
pub use rapidfuzz::distance::levenshtein::WeightTable; // levenshtein.rs:128-148

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

/// `levenshtein::Args` / `jaro_winkler::Args` / ... (levenshtein.rs:86-126, jaro_winkler.rs:25-62) in one builder.
#[derive(Clone, Copy)]
pub struct Args(pub(crate) RfArgs);
impl Default for Args {
    fn default() -> Self {
        let mut a = std::mem::MaybeUninit::<RfArgs>::uninit();
        unsafe {
            rf_args_default(a.as_mut_ptr());
            Args(a.assume_init())
        }
    }
}
impl Args {
    pub fn score_cutoff(mut self, c: usize) -> Self { self.0.cutoff_usize = c as u64; self }
    pub fn score_cutoff_f64(mut self, c: f64) -> Self { self.0.cutoff_f64 = c; self }
    /// accepted and ignored, like the reference's results (levenshtein.rs:2153-2160)
    pub fn score_hint(mut self, h: usize) -> Self { self.0.score_hint_usize = h as u64; self }
    pub fn weights(mut self, w: &WeightTable) -> Self {
        self.0.insertion_cost = w.insertion_cost as u64;
        self.0.deletion_cost = w.deletion_cost as u64;
        self.0.substitution_cost = w.substitution_cost as u64;
        self
    }
    pub fn prefix_weight(mut self, w: f64) -> Self { self.0.prefix_weight = w; self }
    pub fn ratio_indel_normalization(mut self) -> Self { self.0.flags |= RF_FLAG_RATIO_INDEL_NORMALIZATION; self }
}

/// k best candidates under (score, index); the engine's own reduction (the reference has no extract API).
pub struct TopK { pub scores: Vec<u32>, pub indices: Vec<u64> }

pub(crate) fn many_u32(c: *const RfComparator, corpus: &crate::Corpus, op: std::os::raw::c_int, a: &Args) -> Result<Vec<Option<usize>>, Error> {
    let mut out = vec![0u32; corpus.len()];
    check(unsafe { rf_many_u32(c, corpus.0, op, &a.0, out.as_mut_ptr(), RF_MEM_HOST, std::ptr::null_mut()) })?;
    Ok(out.into_iter().map(|d| (d != RF_NONE_U32).then_some(d as usize)).collect())
}
pub(crate) fn many_f64(c: *const RfComparator, corpus: &crate::Corpus, op: std::os::raw::c_int, a: &Args) -> Result<Vec<Option<f64>>, Error> {
    let mut out = vec![0f64; corpus.len()];
    check(unsafe { rf_many_f64(c, corpus.0, op, &a.0, out.as_mut_ptr(), RF_MEM_HOST, std::ptr::null_mut()) })?;
    Ok(out.into_iter().map(|v| (!v.is_nan()).then_some(v)).collect())
}
pub(crate) fn one_u32(c: *const RfComparator, s2: &[u8], op: std::os::raw::c_int, a: &Args, device: i32) -> Result<Option<usize>, Error> {
    let (mut v, mut some) = (0u32, 0);
    check(unsafe { rf_one_u32(c, s2.as_ptr(), s2.len(), op, &a.0, device, &mut v, &mut some) })?;
    Ok((some != 0).then_some(v as usize))
}
pub(crate) fn one_f64(c: *const RfComparator, s2: &[u8], op: std::os::raw::c_int, a: &Args, device: i32) -> Result<Option<f64>, Error> {
    let (mut v, mut some) = (0f64, 0);
    check(unsafe { rf_one_f64(c, s2.as_ptr(), s2.len(), op, &a.0, device, &mut v, &mut some) })?;
    Ok((some != 0).then_some(v))
}

/// The handle every BatchComparator wraps: `new`, `Clone`, `Drop`.
macro_rules! comparator_core {
    ($metric:ident) => {
        pub struct BatchComparator(pub(crate) *mut RfComparator);
        unsafe impl Send for BatchComparator {}
        unsafe impl Sync for BatchComparator {}
        impl Clone for BatchComparator {
            fn clone(&self) -> Self {
                let mut h = std::ptr::null_mut();
                // (the reference's constructors cannot fail; an allocation failure here panics with the library's message
                // instead of handing out a null handle)
                check(unsafe { rf_comparator_clone(self.0, &mut h) }).expect("rf_comparator_clone");
                Self(h)
            }
        }
        impl Drop for BatchComparator {
            fn drop(&mut self) {
                unsafe { rf_comparator_free(self.0) }
            }
        }
        impl BatchComparator {
            /// The multi-GPU step for ANY metric / op / k (16-byte entries: order-preserving key + 64-bit global index, `rf_topk_entry`):
            /// scan this rank's shard, all-gather k entries per rank over the caller's RCCL communicator, merge -- on the device.
            ///
            /// # Safety
            /// `d_local` (k entries), `d_all` (world * k) and `d_merged` (k) must be device pointers valid on `stream`.
            #[allow(clippy::too_many_arguments)]
            pub unsafe fn topk_sharded_entries_device(&self, shard: &crate::Corpus, op: std::os::raw::c_int, k: u64, args: &Args, shard_start: u64,
                                                      nccl_comm: *mut std::os::raw::c_void, world: u32, d_local: *mut RfTopkEntry, d_all: *mut RfTopkEntry,
                                                      d_merged: *mut RfTopkEntry, stream: *mut std::os::raw::c_void) -> Result<(), Error> {
                check(rf_topk_entries_device(self.0, shard.0, op, &args.0, k, shard_start, d_local, stream))?;
                check(rf_topk_allgather_merge_entries(d_local, k, nccl_comm, world, d_all, d_merged, shard.device(), stream))
            }
            /// `BatchComparator::new(s1)` over `u8` elements.
            pub fn new<I: IntoIterator<Item = u8>>(s1: I) -> Self {
                let s1: Vec<u8> = s1.into_iter().collect();
                let mut h = std::ptr::null_mut();
                check(unsafe { rf_comparator_new($metric, s1.as_ptr(), s1.len(), &mut h) }).expect("rf_comparator_new");
                Self(h)
            }
            /// `BatchComparator::new(s1.chars())`: searched in corpora built by `Corpus::from_chars`.
            pub fn from_chars<I: IntoIterator<Item = char>>(s1: I) -> Self {
                let s1: Vec<u32> = s1.into_iter().map(|c| c as u32).collect();
                let mut h = std::ptr::null_mut();
                check(unsafe { rf_comparator_new_u32($metric, s1.as_ptr(), s1.len(), &mut h) }).expect("rf_comparator_new_u32");
                Self(h)
            }
        }
    };
}
pub(crate) use comparator_core;

/// usize-valued metrics: distance / similarity are `usize`, normalized_* are `f64`.
macro_rules! usize_metric {
    ($name:ident, $metric:ident, $doc:literal) => {
        #[doc = $doc]
        pub mod $name {
            use crate::metric::*;
            use crate::sys::*;
            use crate::Corpus;
            crate::metric::comparator_core!($metric);
            impl BatchComparator {
                /// The reference's per-candidate method (one kernel launch for one pair: ~3 orders slower than the CPU crate's
                /// 0.18 us; it exists so call sites compile unchanged -- loops belong in `distance_many`).
                pub fn distance<I: IntoIterator<Item = u8>>(&self, s2: I) -> usize {
                    self.distance_with_args(s2, &Args::default()).expect("no cutoff")
                }
                pub fn distance_with_args<I: IntoIterator<Item = u8>>(&self, s2: I, args: &Args) -> Option<usize> {
                    let s2: Vec<u8> = s2.into_iter().collect();
                    one_u32(self.0, &s2, RF_OP_DISTANCE, args, 0).expect("gpu")
                }
                pub fn similarity_with_args<I: IntoIterator<Item = u8>>(&self, s2: I, args: &Args) -> Option<usize> {
                    let s2: Vec<u8> = s2.into_iter().collect();
                    one_u32(self.0, &s2, RF_OP_SIMILARITY, args, 0).expect("gpu")
                }
                pub fn normalized_distance_with_args<I: IntoIterator<Item = u8>>(&self, s2: I, args: &Args) -> Option<f64> {
                    let s2: Vec<u8> = s2.into_iter().collect();
                    one_f64(self.0, &s2, RF_OP_NORMALIZED_DISTANCE, args, 0).expect("gpu")
                }
                pub fn normalized_similarity_with_args<I: IntoIterator<Item = u8>>(&self, s2: I, args: &Args) -> Option<f64> {
                    let s2: Vec<u8> = s2.into_iter().collect();
                    one_f64(self.0, &s2, RF_OP_NORMALIZED_SIMILARITY, args, 0).expect("gpu")
                }
                /// `for c in corpus { self.distance_with_args(c, args) }` as one scan; `None` where the reference returns `None`.
                pub fn distance_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<usize>>, Error> { many_u32(self.0, corpus, RF_OP_DISTANCE, args) }
                pub fn similarity_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<usize>>, Error> { many_u32(self.0, corpus, RF_OP_SIMILARITY, args) }
                pub fn normalized_distance_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<f64>>, Error> { many_f64(self.0, corpus, RF_OP_NORMALIZED_DISTANCE, args) }
                pub fn normalized_similarity_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<f64>>, Error> { many_f64(self.0, corpus, RF_OP_NORMALIZED_SIMILARITY, args) }
                /// k best candidates by (distance, index); `index_base` makes shards of one logical corpus comparable.
                pub fn topk(&self, corpus: &Corpus, k: u32, args: &Args, index_base: u64) -> Result<TopK, Error> {
                    let (mut s, mut i, mut n) = (vec![0u32; k as usize], vec![0u64; k as usize], 0u32);
                    check(unsafe { rf_topk_u32(self.0, corpus.0, RF_OP_DISTANCE, &args.0, k, index_base, s.as_mut_ptr(), i.as_mut_ptr(), &mut n,
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
                pub unsafe fn topk_sharded_device(&self, shard: &Corpus, k: u32, args: &Args, shard_start: u32, nccl_comm: *mut std::os::raw::c_void,
                                                  world: u32, d_local: *mut u64, d_all: *mut u64, d_merged: *mut u64, stream: *mut std::os::raw::c_void) -> Result<(), Error> {
                    check(rf_topk_keys_device(self.0, shard.0, RF_OP_DISTANCE, &args.0, k, shard_start, d_local, std::ptr::null_mut(), RF_MEM_DEVICE, stream))?;
                    check(rf_topk_allgather_merge(d_local, k, nccl_comm, world, d_all, d_merged, shard.device(), stream))
                }
            }
            // free functions: same values as the comparator (the reference's tests assert exactly that, levenshtein.rs:1847-1875)
            pub fn distance<I1: IntoIterator<Item = u8>, I2: IntoIterator<Item = u8>>(s1: I1, s2: I2) -> usize { BatchComparator::new(s1).distance(s2) }
            pub fn distance_with_args<I1: IntoIterator<Item = u8>, I2: IntoIterator<Item = u8>>(s1: I1, s2: I2, args: &Args) -> Option<usize> { BatchComparator::new(s1).distance_with_args(s2, args) }
        }
    };
}
pub(crate) use usize_metric;

/// f64-valued metrics (jaro, jaro_winkler, fuzz::ratio): every method returns `f64`.
macro_rules! f64_metric {
    ($name:ident, $metric:ident, $doc:literal) => {
        #[doc = $doc]
        pub mod $name {
            use crate::metric::*;
            use crate::sys::*;
            use crate::Corpus;
            crate::metric::comparator_core!($metric);
            impl BatchComparator {
                pub fn similarity_with_args<I: IntoIterator<Item = u8>>(&self, s2: I, args: &Args) -> Option<f64> {
                    let s2: Vec<u8> = s2.into_iter().collect();
                    one_f64(self.0, &s2, RF_OP_SIMILARITY, args, 0).expect("gpu")
                }
                pub fn distance_with_args<I: IntoIterator<Item = u8>>(&self, s2: I, args: &Args) -> Option<f64> {
                    let s2: Vec<u8> = s2.into_iter().collect();
                    one_f64(self.0, &s2, RF_OP_DISTANCE, args, 0).expect("gpu")
                }
                pub fn similarity_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<f64>>, Error> { many_f64(self.0, corpus, RF_OP_SIMILARITY, args) }
                pub fn distance_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<f64>>, Error> { many_f64(self.0, corpus, RF_OP_DISTANCE, args) }
                pub fn normalized_similarity_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<f64>>, Error> { many_f64(self.0, corpus, RF_OP_NORMALIZED_SIMILARITY, args) }
                pub fn normalized_distance_many(&self, corpus: &Corpus, args: &Args) -> Result<Vec<Option<f64>>, Error> { many_f64(self.0, corpus, RF_OP_NORMALIZED_DISTANCE, args) }
            }
        }
    };
}
pub(crate) use f64_metric;
