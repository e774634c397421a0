//! bitnuc-hip: the public surface of `bitnuc` (src/lib.rs:214-220 of the reference)
//! re-implemented over libbitnuc_hip.so.  Swap `use bitnuc::{..}` for
//! `use bitnuc_hip::{..}`; names, argument meaning, Vec clear/append conventions and
//! error values are the reference's.
//!
//! NOT COMPILED IN THIS REPO's CI: the build image has no Rust toolchain.  Every function
//! below is a mechanical wrapper of a C-ABI entry point that is compiled and tested
//! (see tests/ and include/bitnuc.hpp, the same layer in C++).
pub mod ffi;

use std::cell::RefCell;
use std::fmt;

/// src/error.rs:3-18 of the reference.
#[derive(Debug, PartialEq, Eq)]
pub enum NucleotideError {
    InvalidBase(u8),
    SequenceTooLong(usize),
    InvalidLength(usize),
    IndexOutOfBounds { index: usize, length: usize },
    InvalidRange { start: usize, end: usize, length: usize },
    Unsupported,
}

impl fmt::Display for NucleotideError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            NucleotideError::InvalidBase(b) => write!(f, "Invalid nucleotide base: {}", b),
            NucleotideError::SequenceTooLong(len) => write!(f, "Sequence length {} exceeds maximum", len),
            NucleotideError::InvalidLength(len) => write!(f, "Invalid length: {}", len),
            NucleotideError::IndexOutOfBounds { index, length } => {
                write!(f, "Index {} out of bounds for sequence of length {}", index, length)
            }
            NucleotideError::InvalidRange { start, end, length } => {
                write!(f, "Invalid range {}..{} for sequence of length {}", start, end, length)
            }
            NucleotideError::Unsupported => write!(f, "Unsupported architecture"),
        }
    }
}
impl std::error::Error for NucleotideError {}

fn to_err(e: &ffi::bitnuc_err) -> NucleotideError {
    match e.status {
        ffi::BITNUC_INVALID_BASE => NucleotideError::InvalidBase(e.byte),
        ffi::BITNUC_SEQUENCE_TOO_LONG => NucleotideError::SequenceTooLong(e.value as usize),
        ffi::BITNUC_INVALID_LENGTH => NucleotideError::InvalidLength(e.value as usize),
        ffi::BITNUC_INDEX_OUT_OF_BOUNDS => NucleotideError::IndexOutOfBounds { index: e.index as usize, length: e.value as usize },
        // a HIP failure has no reference counterpart; there is no CPU fallback to fall to
        ffi::BITNUC_BACKEND_ERROR => panic!("bitnuc-hip: HIP backend error {}", e.backend_code),
        _ => NucleotideError::Unsupported,
    }
}

/// One device + stream + scratch.  `!Sync`: use one per thread.
pub struct Context {
    raw: *mut ffi::bitnuc_ctx,
}

impl Context {
    pub fn new(device: i32) -> Self {
        let mut raw = std::ptr::null_mut();
        let mut e = ffi::bitnuc_err::default();
        let st = unsafe { ffi::bitnuc_ctx_create(device, &mut raw, &mut e) };
        assert!(st == ffi::BITNUC_OK, "bitnuc-hip: no usable HIP device (hipError {})", e.backend_code);
        Context { raw }
    }
    pub fn raw(&self) -> *mut ffi::bitnuc_ctx {
        self.raw
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ffi::bitnuc_ctx_destroy(self.raw) }
    }
}

thread_local! {
    static CTX: RefCell<Option<Context>> = RefCell::new(None);
}
fn with_ctx<R>(f: impl FnOnce(*mut ffi::bitnuc_ctx) -> R) -> R {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        let ctx = c.get_or_insert_with(|| Context::new(0));
        f(ctx.raw)
    })
}

/// `bitnuc::as_2bit` (src/utils/packing/mod.rs:80-110).
pub fn as_2bit(seq: &[u8]) -> Result<u64, NucleotideError> {
    let mut out = 0u64;
    let mut e = ffi::bitnuc_err::default();
    // context-free: single words are host code inside the library (include/bitnuc_hip.h, "Size dispatch"), like the reference's #[inline(always)] function
    let st = unsafe { ffi::bitnuc_as_2bit(std::ptr::null_mut(), seq.as_ptr(), seq.len(), &mut out, &mut e) };
    if st == ffi::BITNUC_OK { Ok(out) } else { Err(to_err(&e)) }
}

/// `bitnuc::from_2bit` (src/utils/unpacking/mod.rs:119-147): APPENDS `expected_size` bases.
pub fn from_2bit(packed: u64, expected_size: usize, sequence: &mut Vec<u8>) -> Result<(), NucleotideError> {
    let mut tmp = [0u8; 32];
    let mut e = ffi::bitnuc_err::default();
    let st = unsafe { ffi::bitnuc_from_2bit(std::ptr::null_mut(), packed, expected_size, tmp.as_mut_ptr(), &mut e) };
    if st != ffi::BITNUC_OK {
        return Err(to_err(&e));
    }
    sequence.extend_from_slice(&tmp[..expected_size]);
    Ok(())
}

/// `bitnuc::from_2bit_alloc` (src/utils/unpacking/mod.rs:178-182).
pub fn from_2bit_alloc(packed: u64, expected_size: usize) -> Result<Vec<u8>, NucleotideError> {
    let mut sequence = Vec::with_capacity(expected_size.min(32));
    from_2bit(packed, expected_size, &mut sequence)?;
    Ok(sequence)
}

/// `bitnuc::encode` (src/utils/mod.rs:22-25): CLEARS `ebuf`, then fills it.  On
/// `InvalidBase` the Vec keeps the words of the chunks before the failing one, like the
/// reference (packing/avx.rs:142-143).  Empty input panics, like the reference (:138).
pub fn encode(sequence: &[u8], ebuf: &mut Vec<u64>) -> Result<(), NucleotideError> {
    ebuf.clear();
    let n_chunks = sequence.len().div_ceil(32);
    let _ = n_chunks - 1; // same overflow panic as `for _ in 0..n_chunks - 1` in the reference
    ebuf.resize(n_chunks, 0);
    let mut n_words = 0usize;
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_encode(c, sequence.as_ptr(), sequence.len(), ebuf.as_mut_ptr(), &mut n_words, &mut e)
    });
    ebuf.truncate(n_words);
    if st == ffi::BITNUC_OK { Ok(()) } else { Err(to_err(&e)) }
}

/// `bitnuc::encode_alloc` (src/utils/mod.rs:38-42).
pub fn encode_alloc(sequence: &[u8]) -> Result<Vec<u64>, NucleotideError> {
    let mut ebuf = Vec::new();
    encode(sequence, &mut ebuf)?;
    Ok(ebuf)
}

/// `bitnuc::decode` (src/utils/mod.rs:60-62): APPENDS `n_bases` bases to `dbuf`.
pub fn decode(ebuf: &[u64], n_bases: usize, dbuf: &mut Vec<u8>) -> Result<(), NucleotideError> {
    let old = dbuf.len();
    dbuf.resize(old + n_bases, 0);
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_decode(c, ebuf.as_ptr(), ebuf.len(), n_bases, dbuf.as_mut_ptr().add(old), &mut e)
    });
    if st != ffi::BITNUC_OK {
        dbuf.truncate(old);
        return Err(to_err(&e));
    }
    Ok(())
}

/// `bitnuc::hdist_scalar` (src/utils/functions/hamming/scalar.rs:11-48).
pub fn hdist_scalar(u: u64, v: u64, len: usize) -> Result<u32, NucleotideError> {
    let mut out = 0u32;
    let mut e = ffi::bitnuc_err::default();
    let st = unsafe { ffi::bitnuc_hdist_scalar(std::ptr::null_mut(), u, v, len, &mut out, &mut e) };
    if st == ffi::BITNUC_OK { Ok(out) } else { Err(to_err(&e)) }
}

/// `bitnuc::hdist` (src/utils/functions/hamming/multi.rs:121-160).
pub fn hdist(ebuf1: &[u64], ebuf2: &[u64], n_bases: usize) -> Result<u32, NucleotideError> {
    let mut out = 0u32;
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_hdist(c, ebuf1.as_ptr(), ebuf1.len(), ebuf2.as_ptr(), ebuf2.len(), n_bases, &mut out, &mut e)
    });
    if st == ffi::BITNUC_OK { Ok(out) } else { Err(to_err(&e)) }
}

/// `bitnuc::split_packed` (src/utils/functions/split.rs:15-99), word for word
/// (`BITNUC_SPLIT_AS_WRITTEN`): validates, clears `lbuf`/`rbuf`, fills them.
pub fn split_packed(ebuf: &[u64], slen: usize, idx: usize, lbuf: &mut Vec<u64>, rbuf: &mut Vec<u64>) -> Result<(), NucleotideError> {
    split_packed_with(ebuf, slen, idx, lbuf, rbuf, ffi::BITNUC_SPLIT_AS_WRITTEN)
}

/// The funnel-shift split: `lbuf == encode(seq[..idx])`, `rbuf == encode(seq[idx..])`.
pub fn split_packed_canonical(ebuf: &[u64], slen: usize, idx: usize, lbuf: &mut Vec<u64>, rbuf: &mut Vec<u64>) -> Result<(), NucleotideError> {
    split_packed_with(ebuf, slen, idx, lbuf, rbuf, ffi::BITNUC_SPLIT_CANONICAL)
}

fn split_packed_with(ebuf: &[u64], slen: usize, idx: usize, lbuf: &mut Vec<u64>, rbuf: &mut Vec<u64>, flags: std::os::raw::c_int) -> Result<(), NucleotideError> {
    let (mut nl, mut nr) = (0usize, 0usize);
    let mut e = ffi::bitnuc_err::default();
    if unsafe { ffi::bitnuc_split_packed_sizes(ebuf.len(), slen, idx, flags, &mut nl, &mut nr, &mut e) } != ffi::BITNUC_OK {
        return Err(to_err(&e)); // before the buffers are cleared, like split.rs:23-32
    }
    lbuf.clear();
    rbuf.clear();
    lbuf.resize(nl, 0);
    rbuf.resize(nr, 0);
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_split_packed(c, ebuf.as_ptr(), ebuf.len(), slen, idx, flags, lbuf.as_mut_ptr(), &mut nl, rbuf.as_mut_ptr(), &mut nr, &mut e)
    });
    if st == ffi::BITNUC_OK { Ok(()) } else { Err(to_err(&e)) }
}

/// Batched form of the `for kmer in kmers { as_2bit(kmer)? }` idiom (README.md:52-56 of
/// the reference): `count` k-mers of length `k`, k-mer j at `kmers[j*stride..]`.
pub fn as_2bit_batch(kmers: &[u8], k: usize, stride: usize, count: usize) -> Result<Vec<u64>, NucleotideError> {
    assert!(count == 0 || (count - 1) * stride + k <= kmers.len());
    let mut out = vec![0u64; count];
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe { ffi::bitnuc_as_2bit_batch(c, kmers.as_ptr(), k, stride, count, out.as_mut_ptr(), &mut e) });
    if st == ffi::BITNUC_OK { Ok(out) } else { Err(to_err(&e)) }
}

/// `reference.windows(k).map(|w| hdist_scalar(as_2bit(w)?, query, k))` in one launch.
pub fn kmer_hdist_scan(reference: &[u8], k: usize, query: u64) -> Result<Vec<u8>, NucleotideError> {
    let nwin = if k > 0 && reference.len() >= k { reference.len() - k + 1 } else { 0 };
    let mut out = vec![0u8; nwin];
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_kmer_hdist_scan(c, reference.as_ptr(), reference.len(), k, query, out.as_mut_ptr(), &mut e)
    });
    if st == ffi::BITNUC_OK { Ok(out) } else { Err(to_err(&e)) }
}

/// `for s in seqs { encode(s, &mut ebuf)? }` in one launch: sequence i =
/// `seq[offsets[i]..offsets[i+1]]`; returns (concatenated words, word_offsets).
pub fn encode_batch(seq: &[u8], offsets: &[u64]) -> Result<(Vec<u64>, Vec<u64>), NucleotideError> {
    let count = offsets.len().saturating_sub(1);
    let cap = if count > 0 { ((offsets[count] - offsets[0]) / 32) as usize + count } else { 0 };
    let mut out = vec![0u64; cap];
    let mut word_offsets = vec![0u64; count + 1];
    let mut n_words = 0usize;
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_encode_batch(c, seq.as_ptr(), offsets.as_ptr(), count, out.as_mut_ptr(), cap,
                                 word_offsets.as_mut_ptr(), &mut n_words, &mut e)
    });
    if st != ffi::BITNUC_OK {
        return Err(to_err(&e));
    }
    out.truncate(n_words);
    Ok((out, word_offsets))
}

/// Inverse of `encode_batch`: sequence i's bases land at `out[offsets[i]..offsets[i+1]]`.
pub fn decode_batch(words: &[u64], word_offsets: &[u64], offsets: &[u64]) -> Result<Vec<u8>, NucleotideError> {
    let count = offsets.len().saturating_sub(1);
    let mut out = vec![0u8; if count > 0 { offsets[count] as usize } else { 0 }];
    let mut e = ffi::bitnuc_err::default();
    let st = with_ctx(|c| unsafe {
        ffi::bitnuc_decode_batch(c, words.as_ptr(), word_offsets.as_ptr(), offsets.as_ptr(), count, out.as_mut_ptr(), &mut e)
    });
    if st == ffi::BITNUC_OK { Ok(out) } else { Err(to_err(&e)) }
}

/// `bitnuc::PackedSequence` (src/sequence.rs:5-262) over GPU-encoded data.
#[derive(Debug, PartialEq, Eq, Clone, Hash)]
pub struct PackedSequence {
    data: Vec<u64>,
    length: usize,
}

impl PackedSequence {
    pub fn new(seq: &[u8]) -> Result<Self, NucleotideError> {
        let mut data = Vec::new();
        if !seq.is_empty() {
            encode(seq, &mut data)?;
        }
        Ok(Self { data, length: seq.len() })
    }
    pub fn len(&self) -> usize {
        self.length
    }
    pub fn is_empty(&self) -> bool {
        self.length == 0
    }
    pub fn get(&self, index: usize) -> Result<u8, NucleotideError> {
        if index >= self.length {
            return Err(NucleotideError::IndexOutOfBounds { index, length: self.length });
        }
        // the reference's shift + mask + match (sequence.rs:121-134): no call into the library
        Ok(b"ACGT"[((self.data[index / 32] >> ((index % 32) * 2)) & 3) as usize])
    }
    pub fn slice(&self, range: std::ops::Range<usize>) -> Result<Vec<u8>, NucleotideError> {
        if range.start > range.end || range.end > self.length {
            return Err(NucleotideError::InvalidRange { start: range.start, end: range.end, length: self.length });
        }
        if range.start == range.end {
            return Ok(Vec::new());
        }
        let (w0, w1) = (range.start / 32, range.end.div_ceil(32));
        let n = self.length.min(w1 * 32) - w0 * 32;
        let mut chunk = Vec::with_capacity(n);
        decode(&self.data[w0..w1], n, &mut chunk)?; // only the words the range touches
        Ok(chunk[range.start - w0 * 32..range.end - w0 * 32].to_vec())
    }
    pub fn to_vec(&self) -> Result<Vec<u8>, NucleotideError> {
        self.slice(0..self.length)
    }
}

/// src/utils/analysis.rs:3-39, evaluated on the packed words on the GPU.
pub trait GCContent {
    fn gc_content(&self) -> f64;
}
pub trait BaseCount {
    fn base_counts(&self) -> [usize; 4];
}
impl BaseCount for PackedSequence {
    fn base_counts(&self) -> [usize; 4] {
        let mut c = [0u64; 4];
        let mut e = ffi::bitnuc_err::default();
        let st = with_ctx(|ctx| unsafe {
            ffi::bitnuc_base_counts(ctx, self.data.as_ptr(), self.data.len(), self.length, c.as_mut_ptr(), &mut e)
        });
        assert!(st == ffi::BITNUC_OK);
        [c[0] as usize, c[1] as usize, c[2] as usize, c[3] as usize]
    }
}
impl GCContent for PackedSequence {
    fn gc_content(&self) -> f64 {
        if self.length == 0 {
            return 0.0;
        }
        let c = self.base_counts();
        ((c[1] + c[2]) as f64 / self.length as f64) * 100.0
    }
}

#[cfg(test)]
mod testing {
    // the reference's own unit tests (src/utils/packing/mod.rs:144-198 etc.) run unchanged
    // against this crate; tests/cpp/test_bitnuc_hpp.cpp holds the compiled C++ twin.
    use super::*;

    #[test]
    fn test_as_2bit_valid_sequence() {
        assert_eq!(as_2bit(b"ACGT").unwrap(), 0b11100100);
        assert!(matches!(as_2bit(b"ACGN"), Err(NucleotideError::InvalidBase(b'N'))));
        assert!(matches!(as_2bit(&vec![b'A'; 33]), Err(NucleotideError::SequenceTooLong(33))));
    }

    #[test]
    fn test_round_trip() {
        let seq = b"ACTGACTGACTGACTGACTGACTGACTGACTGACTGA";
        let mut ebuf = Vec::new();
        encode(seq, &mut ebuf).unwrap();
        let mut out = Vec::new();
        decode(&ebuf, seq.len(), &mut out).unwrap();
        assert_eq!(&out, seq);
    }
}
