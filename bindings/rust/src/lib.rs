//! Rust host side of the MI355X search path: the reference's `Searcher<P>` surface
//! (`/root/reference/src/search.rs:227-784`: `new_fwd`, `new_rc`, `search`, `search_all`,
//! `encode_patterns`, `search_encoded_patterns`) forwarded through the C-ABI of `include/sassy_hip.h`.
//!
//! SOURCE ONLY: the image this repository is built in has no `rustc` / `cargo`; nothing in the test
//! suite compiles this file.  Every `extern "C"` item below is a declaration of `include/sassy_hip.h`;
//! the C client `tests/c/dropin_client.c` and the Python `ctypes` layer exercise the same symbols.
use std::ffi::{c_char, c_int, CStr, CString};
use std::marker::PhantomData;

#[repr(C)]
struct RawMatch {
    // include/sassy_hip.h: sassy_hip_Match (64 bytes)
    pattern_idx: u64,
    text_idx: u64,
    text_start: u64,
    text_end: u64,
    pattern_start: u64,
    pattern_end: u64,
    cost: i32,
    strand: u8,
    _pad: [u8; 3],
    cigar_off: u32,
    cigar_len: u32,
}
#[repr(C)]
struct RawSearcher {
    _p: [u8; 0],
}
#[repr(C)]
struct RawMultiTicket {
    _p: [u8; 0],
}
#[repr(C)]
struct RawResult {
    _p: [u8; 0],
}
#[repr(C)]
struct RawEncoded {
    _p: [u8; 0],
}
#[repr(C)]
struct RawTicket {
    _p: [u8; 0],
}

pub const ALL_MINIMA: u32 = 1;
pub const WITHOUT_TRACE: u32 = 2;
pub const TEXT_ON_DEVICE: u32 = 4;

extern "C" {
    fn sassy_hip_searcher_new(alphabet: *const c_char, rc: bool, alpha: f32) -> *mut RawSearcher;
    fn sassy_searcher_free(s: *mut RawSearcher);
    fn sassy_hip_last_error() -> *const c_char;
    fn sassy_hip_search(s: *mut RawSearcher, pattern: *const u8, pattern_len: usize, text: *const u8,
                        text_len: usize, k: usize, flags: u32, out: *mut *mut RawResult) -> c_int;
    fn sassy_hip_search_shard_begin(s: *mut RawSearcher, pattern: *const u8, pattern_len: usize, d_text: *const u8,
                                    halo_len: u64, shard_len: u64, global_offset: u64, total_len: u64, k: usize,
                                    flags: u32, out: *mut *mut RawTicket) -> c_int;
    fn sassy_hip_search_finish(s: *mut RawSearcher, t: *mut RawTicket, out: *mut *mut RawResult) -> c_int;
    fn sassy_hip_encode_patterns(s: *mut RawSearcher, patterns: *const u8, npat: usize, plen: usize) -> *mut RawEncoded;
    fn sassy_hip_encoded_free(e: *mut RawEncoded);
    fn sassy_hip_search_encoded(s: *mut RawSearcher, e: *const RawEncoded, text: *const u8, text_len: usize,
                                k: usize, flags: u32, out: *mut *mut RawResult) -> c_int;
    fn sassy_hip_result_len(r: *const RawResult) -> usize;
    fn sassy_hip_result_matches(r: *const RawResult) -> *const RawMatch;
    fn sassy_hip_result_cigars(r: *const RawResult) -> *const c_char;
    fn sassy_hip_result_free(r: *mut RawResult);
    fn sassy_hip_set_only_best_match(s: *mut RawSearcher, on: c_int) -> c_int;
    fn sassy_hip_set_max_n_frac(s: *mut RawSearcher, f: f32) -> c_int;
    fn sassy_hip_set_device(s: *mut RawSearcher, device: c_int) -> c_int;
    fn sassy_hip_merge_shards(results: *const *const RawResult, n: usize, incoming_state: c_int,
                              out: *mut *mut RawResult) -> c_int;
    fn sassy_hip_multi_new(alphabet: *const c_char, alpha: f32, devices: *const c_int, n_devices: usize) -> *mut RawMulti;
    fn sassy_hip_multi_set_text(m: *mut RawMulti, text: *const u8, len: usize, max_pattern_len: usize, max_k: usize) -> c_int;
    fn sassy_hip_multi_search(m: *mut RawMulti, pattern: *const u8, pattern_len: usize, k: usize, flags: u32,
                              out: *mut *mut RawResult) -> c_int;
    fn sassy_hip_multi_free(m: *mut RawMulti);
    // searches in flight over several devices (include/sassy_hip.h)
    fn sassy_hip_multi_set_pipe_depth(m: *mut RawMulti, depth: c_int) -> c_int;
    fn sassy_hip_multi_search_begin(m: *mut RawMulti, pattern: *const u8, pattern_len: usize, k: usize, flags: u32,
                                    out: *mut *mut RawMultiTicket) -> c_int;
    fn sassy_hip_multi_search_finish(m: *mut RawMulti, ticket: *mut RawMultiTicket, out: *mut *mut RawResult) -> c_int;
}

#[repr(C)]
pub struct RawMulti {
    _p: [u8; 0],
}

/// One text over several GPUs inside one process (include/sassy_hip.h: sassy_hip_multi_*): the reference's thread
/// fan-out (bin/grep.rs:476-503) with a device per thread.  Forward strand, like the shard calls.
pub struct MultiSearcher {
    raw: *mut RawMulti,
}

impl MultiSearcher {
    /// `devices`: empty = every visible device.
    pub fn new(alphabet: &str, devices: &[i32]) -> Self {
        let a = std::ffi::CString::new(alphabet).unwrap();
        let raw = unsafe { sassy_hip_multi_new(a.as_ptr(), f32::NAN, if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() }, devices.len()) };
        assert!(!raw.is_null(), "{}", last_error());
        MultiSearcher { raw }
    }
    pub fn set_text(&mut self, text: &[u8], max_pattern_len: usize, max_k: usize) {
        let rc = unsafe { sassy_hip_multi_set_text(self.raw, text.as_ptr(), text.len(), max_pattern_len, max_k) };
        assert_eq!(rc, 0, "{}", last_error());
    }
    pub fn search(&mut self, pattern: &[u8], k: usize) -> Vec<Match> {
        let mut res = std::ptr::null_mut();
        let rc = unsafe { sassy_hip_multi_search(self.raw, pattern.as_ptr(), pattern.len(), k, 0, &mut res) };
        assert_eq!(rc, 0, "{}", last_error());
        unsafe { collect(res) }
    }
}

impl Drop for MultiSearcher {
    fn drop(&mut self) {
        unsafe { sassy_hip_multi_free(self.raw) }
    }
}

/// The reference's `Strand` (src/search.rs:107-119).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Strand {
    Fwd,
    Rc,
}

/// The reference's `Match` (src/search.rs:35-62); `cigar` in SAM text form (`pa_types::Cigar::to_string`).
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Match {
    pub pattern_idx: usize,
    pub text_idx: usize,
    pub text_start: usize,
    pub text_end: usize,
    pub pattern_start: usize,
    pub pattern_end: usize,
    pub cost: i32,
    pub strand: Strand,
    pub cigar: String,
}

/// Alphabet marker types, as the reference's `profiles::{Dna, Iupac, Ascii}`.
pub trait Profile {
    const NAME: &'static str;
}
pub struct Dna;
pub struct Iupac;
pub struct Ascii;
impl Profile for Dna {
    const NAME: &'static str = "dna";
}
impl Profile for Iupac {
    const NAME: &'static str = "iupac";
}
impl Profile for Ascii {
    const NAME: &'static str = "ascii";
}

pub struct Searcher<P: Profile> {
    raw: *mut RawSearcher,
    _p: PhantomData<P>,
}

/// `EncodedPatterns<P>` (src/pattern_tiling/general.rs:132-150): opaque on both sides.
pub struct EncodedPatterns<P: Profile> {
    raw: *mut RawEncoded,
    _p: PhantomData<P>,
}
impl<P: Profile> Drop for EncodedPatterns<P> {
    fn drop(&mut self) {
        unsafe { sassy_hip_encoded_free(self.raw) }
    }
}

/// A search in flight (`include/sassy_hip.h`: sassy_hip_search_shard_begin).
pub struct Ticket(*mut RawTicket);

fn last_error() -> String {
    unsafe { CStr::from_ptr(sassy_hip_last_error()) }.to_string_lossy().into_owned()
}

unsafe fn collect(res: *mut RawResult) -> Vec<Match> {
    let n = sassy_hip_result_len(res);
    let rows = std::slice::from_raw_parts(sassy_hip_result_matches(res), n);
    let pool = sassy_hip_result_cigars(res);
    let out = rows
        .iter()
        .map(|m| Match {
            pattern_idx: m.pattern_idx as usize,
            text_idx: m.text_idx as usize,
            text_start: m.text_start as usize,
            text_end: m.text_end as usize,
            pattern_start: m.pattern_start as usize,
            pattern_end: m.pattern_end as usize,
            cost: m.cost,
            strand: if m.strand == 0 { Strand::Fwd } else { Strand::Rc },
            cigar: CStr::from_ptr(pool.add(m.cigar_off as usize)).to_string_lossy().into_owned(),
        })
        .collect();
    sassy_hip_result_free(res);
    out
}

impl<P: Profile> Searcher<P> {
    /// `Searcher::new(rc, alpha)` (src/search.rs:486-503); panics like the reference on an invalid combination.
    pub fn new(rc: bool, alpha: Option<f32>) -> Self {
        let name = CString::new(P::NAME).unwrap();
        let raw = unsafe { sassy_hip_searcher_new(name.as_ptr(), rc, alpha.unwrap_or(f32::NAN)) };
        assert!(!raw.is_null(), "{}", last_error());
        Searcher { raw, _p: PhantomData }
    }
    /// `Searcher::new_fwd()` / `new_rc()` (src/search.rs:261-283).
    pub fn new_fwd() -> Self {
        Self::new(false, None)
    }
    pub fn new_rc() -> Self {
        Self::new(true, None)
    }
    pub fn only_best_match(self) -> Self {
        unsafe { sassy_hip_set_only_best_match(self.raw, 1) };
        self
    }
    pub fn with_max_n_frac(self, f: f32) -> Self {
        unsafe { sassy_hip_set_max_n_frac(self.raw, f) };
        self
    }

    fn run(&mut self, pattern: &[u8], text: &[u8], k: usize, flags: u32) -> Vec<Match> {
        let mut res = std::ptr::null_mut();
        let rc = unsafe {
            sassy_hip_search(self.raw, pattern.as_ptr(), pattern.len(), text.as_ptr(), text.len(), k, flags, &mut res)
        };
        assert_eq!(rc, 0, "{}", last_error()); // the reference panics on its errors
        unsafe { collect(res) }
    }
    /// `Searcher::search` (src/search.rs:510-525): rightmost local minima with cost <= k.
    pub fn search(&mut self, pattern: &[u8], text: &[u8], k: usize) -> Vec<Match> {
        self.run(pattern, text, k, 0)
    }
    /// `Searcher::search_all` (src/search.rs:685-700).
    pub fn search_all(&mut self, pattern: &[u8], text: &[u8], k: usize) -> Vec<Match> {
        self.run(pattern, text, k, ALL_MINIMA)
    }
    /// `Searcher::encode_patterns` (src/search.rs:404-406): equal-length patterns (<= 64).
    pub fn encode_patterns(&mut self, patterns: &[Vec<u8>]) -> EncodedPatterns<P> {
        assert!(!patterns.is_empty(), "No queries provided");
        let plen = patterns[0].len();
        assert!(patterns.iter().all(|p| p.len() == plen), "All pattern must have the same length");
        let flat: Vec<u8> = patterns.iter().flatten().copied().collect();
        let raw = unsafe { sassy_hip_encode_patterns(self.raw, flat.as_ptr(), patterns.len(), plen) };
        assert!(!raw.is_null(), "{}", last_error());
        EncodedPatterns { raw, _p: PhantomData }
    }
    /// `Searcher::search_encoded_patterns` (src/search.rs:415-423); owned `Vec` instead of a borrowed slice.
    pub fn search_encoded_patterns(&mut self, encoded: &EncodedPatterns<P>, text: &[u8], k: usize) -> Vec<Match> {
        let mut res = std::ptr::null_mut();
        let rc = unsafe { sassy_hip_search_encoded(self.raw, encoded.raw, text.as_ptr(), text.len(), k, 0, &mut res) };
        assert_eq!(rc, 0, "{}", last_error());
        unsafe { collect(res) }
    }

    /// A stream of searches over a text that lives in HBM: queue one and go on (up to two in flight).
    ///
    /// # Safety
    /// `d_text` must be a device pointer to `total_len` resident bytes that stay unchanged until `finish`.
    pub unsafe fn begin_on_device(&mut self, pattern: &[u8], d_text: *const u8, total_len: u64, k: usize) -> Ticket {
        let mut t = std::ptr::null_mut();
        let rc = sassy_hip_search_shard_begin(self.raw, pattern.as_ptr(), pattern.len(), d_text, 0, total_len, 0, total_len,
                                              k, 0, &mut t);
        assert_eq!(rc, 0, "{}", last_error());
        Ticket(t)
    }
    pub fn finish(&mut self, ticket: Ticket) -> Vec<Match> {
        let mut res = std::ptr::null_mut();
        let rc = unsafe { sassy_hip_search_finish(self.raw, ticket.0, &mut res) };
        assert_eq!(rc, 0, "{}", last_error());
        unsafe { collect(res) }
    }
}

impl<P: Profile> Drop for Searcher<P> {
    fn drop(&mut self) {
        unsafe { sassy_searcher_free(self.raw) }
    }
}
