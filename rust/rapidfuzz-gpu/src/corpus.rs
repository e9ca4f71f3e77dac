//! The candidates of the user's `for` loop, packed once (length-bucketed tiles of 64, rfgpu.h "corpus") and kept in HBM.
use crate::metric::{check, Error};
use crate::sys::*;
use std::ffi::CString;

pub struct Corpus(pub(crate) *mut RfCorpus);
// handles are immutable after creation and may be shared between threads (rfgpu.h "Threading")
unsafe impl Send for Corpus {}
unsafe impl Sync for Corpus {}

impl Corpus {
    /// Byte candidates (`u8` elements, the reference's `HashableChar` case `u8`, details/common.rs:34).
    pub fn new<'a, I: IntoIterator<Item = &'a [u8]>>(candidates: I, device: i32) -> Result<Self, Error> {
        let (mut bytes, mut offsets) = (Vec::new(), vec![0u64]);
        for c in candidates {
            bytes.extend_from_slice(c);
            offsets.push(bytes.len() as u64);
        }
        let mut h = std::ptr::null_mut();
        check(unsafe { rf_corpus_pack(bytes.as_ptr(), offsets.as_ptr(), offsets.len() - 1, device, &mut h) })?;
        Ok(Corpus(h))
    }
    /// Entries of a slot-ordered result vector (`RF_FLAG_SLOT_ORDER`): `len()` for a single-length corpus, 64 per tile otherwise.
    pub fn slot_count(&self) -> usize {
        unsafe { rf_corpus_slot_count(self.0) }
    }
    /// The original candidate index of every slot (`u32::MAX`: the slot holds no candidate): kept once by a caller that takes its results in slot order.
    pub fn slot_index(&self) -> Result<Vec<u32>, Error> {
        let mut out = vec![0u32; self.slot_count()];
        check(unsafe { rf_corpus_slot_index(self.0, out.as_mut_ptr(), RF_MEM_HOST) })?;
        Ok(out)
    }
    /// `&str` candidates compared as `char`s (`BatchComparator::new(s.chars())`): one u32 per char, the corpus keeps its
    /// own alphabet.
    pub fn from_chars<'a, I: IntoIterator<Item = &'a str>>(candidates: I, device: i32) -> Result<Self, Error> {
        let (mut elems, mut offsets) = (Vec::<u32>::new(), vec![0u64]);
        for c in candidates {
            elems.extend(c.chars().map(|ch| ch as u32));
            offsets.push(elems.len() as u64);
        }
        let mut h = std::ptr::null_mut();
        check(unsafe { rf_corpus_pack_u32(elems.as_ptr(), offsets.as_ptr(), offsets.len() - 1, device, &mut h) })?;
        Ok(Corpus(h))
    }
    /// A packed corpus written by [`Corpus::save`]; validated on load.
    pub fn load(path: &str, device: i32) -> Result<Self, Error> {
        let p = CString::new(path).map_err(|_| Error(RF_ERR_INVALID_ARG, "path contains NUL".into()))?;
        let mut h = std::ptr::null_mut();
        check(unsafe { rf_corpus_load(p.as_ptr(), device, &mut h) })?;
        Ok(Corpus(h))
    }
    pub fn save(&self, path: &str) -> Result<(), Error> {
        let p = CString::new(path).map_err(|_| Error(RF_ERR_INVALID_ARG, "path contains NUL".into()))?;
        check(unsafe { rf_corpus_save(self.0, p.as_ptr()) })
    }
    pub fn len(&self) -> usize {
        unsafe { rf_corpus_count(self.0) }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    pub fn device(&self) -> i32 {
        unsafe { rf_corpus_device(self.0) }
    }
}
impl Drop for Corpus {
    fn drop(&mut self) {
        unsafe { rf_corpus_free(self.0) }
    }
}
