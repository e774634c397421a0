"""Host-side mirror of bitnuc's public API over the C ABI (Python flavour).

Same names, argument meaning and error behaviour as the reference's
`pub use` list (src/lib.rs:214-220):

    as_2bit, from_2bit, from_2bit_alloc, encode, encode_alloc, decode,
    hdist_scalar, hdist, NucleotideError

`Vec<u64>` / `Vec<u8>` arguments map to Python mutable sequences:
  * `encode(seq, ebuf)` CLEARS `ebuf` then fills it     (packing/avx.rs:132)
  * `decode(ebuf, n, dbuf)` / `from_2bit(p, n, seq)` APPEND (unpacking/avx.rs:122,140)
`ebuf` may be a list or array('Q'); `dbuf` a bytearray.

All bulk arithmetic runs on the GPU through libbitnuc_hip.so; single words and host-pointer calls
below the library's host cutoff are the library's own host code (include/bitnuc_hip.h, "Size dispatch").
The module-level single-word functions need no GPU context, like the reference's.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class NucleotideError(Exception):
    """src/error.rs:3-18.  `kind` is the variant name; payload fields as in Rust."""

    def __init__(self, kind, **payload):
        self.kind = kind
        self.payload = payload
        for k, v in payload.items():
            setattr(self, k, v)
        super().__init__(self._display())

    def _display(self):  # Display impl, src/error.rs:20-45
        p = self.payload
        if self.kind == "InvalidBase":
            return f"Invalid nucleotide base: {p['byte']}"
        if self.kind == "SequenceTooLong":
            return f"Sequence length {p['len']} exceeds maximum"
        if self.kind == "InvalidLength":
            return f"Invalid length: {p['len']}"
        if self.kind == "IndexOutOfBounds":
            return f"Index {p['index']} out of bounds for sequence of length {p['length']}"
        if self.kind == "InvalidRange" and "start" in p:
            return f"Invalid range {p['start']}..{p['end']} for sequence of length {p['length']}"
        if self.kind == "Unsupported":
            return "Unsupported architecture"
        return self.kind

    def __eq__(self, other):  # derive(PartialEq, Eq), src/error.rs:3
        return isinstance(other, NucleotideError) and self.kind == other.kind and \
            self._cmp_payload() == other._cmp_payload()

    def _cmp_payload(self):  # InvalidBase carries an extra byte index that Rust's variant does not have
        return {k: v for k, v in self.payload.items() if not (self.kind == "InvalidBase" and k == "index")}

    __hash__ = Exception.__hash__


class BackendError(RuntimeError):
    """HIP runtime failure (no reference counterpart)."""


def _raise(err):
    st = err.status
    if st == L.INVALID_BASE:
        raise NucleotideError("InvalidBase", byte=int(err.byte), index=int(err.index))
    if st == L.SEQUENCE_TOO_LONG:
        raise NucleotideError("SequenceTooLong", len=int(err.value))
    if st == L.INVALID_LENGTH:
        raise NucleotideError("InvalidLength", len=int(err.value))
    if st == L.INDEX_OUT_OF_BOUNDS:
        raise NucleotideError("IndexOutOfBounds", index=int(err.index), length=int(err.value))
    if st == L.INVALID_RANGE:  # decreasing offsets in a ragged batch: err.value = the sequence whose end lies before its start
        raise NucleotideError("InvalidRange", index=int(err.value))
    if st == L.UNSUPPORTED:
        raise NucleotideError("Unsupported")
    if st == L.BACKEND_ERROR:
        raise BackendError(f"HIP error {err.backend_code} (is a gfx950 device visible? bitnuc_amd has no CPU fallback)")
    raise RuntimeError(f"bitnuc status {st}")


def _as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.dtype != np.uint8:
        raise TypeError("sequence must be bytes-like / uint8")
    return np.ascontiguousarray(a)


def _as_u64(buf):
    if isinstance(buf, np.ndarray):
        if buf.dtype != np.uint64:
            raise TypeError("ebuf must be uint64")
        return np.ascontiguousarray(buf)
    return np.asarray(list(buf), dtype=np.uint64)


def _ptr(a):
    return C.c_void_p(a.ctypes.data if a.size else 0)


def _dev_ptr(x):
    """int device address, or anything with .data_ptr() (torch tensor)."""
    if x is None:
        return C.c_void_p(0)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Context:
    """One device + stream + scratch (bitnuc_ctx).  Not thread-safe; make one per thread."""

    def __init__(self, device=0, stream=None, lib_path=None):
        self._lib = L.load(lib_path)
        self._h = C.c_void_p()
        err = L.BitnucErr()
        if stream is None:
            st = self._lib.bitnuc_ctx_create(device, C.byref(self._h), C.byref(err))
        else:
            st = self._lib.bitnuc_ctx_create_on_stream(device, C.c_void_p(int(stream)), C.byref(self._h), C.byref(err))
        if st != L.OK:
            self._h = C.c_void_p()
            _raise(err)
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not getattr(self, "_borrowed", False):  # CommGroup.context(): the handle is the group's
                self._lib.bitnuc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- tuning --------------------------------------------------------------------
    def set_variant(self, key, value, strict=False):
        """Previous value, or -2 when this build does not hold `value` (the product refuses every formulation but the shipped one).
        strict=True raises instead of returning -2 / leaving the knob where it was: an A/B that ignores a refusal times a kernel against
        itself, so tools/ and the variant tests use require_variant()."""
        prev = self._lib.bitnuc_ctx_set_variant(self._h, key.encode(), int(value))
        if strict and int(value) >= 0 and (prev == -2 or self.get(key) != int(value)):
            raise ValueError(f"bitnuc_ctx_set_variant({key!r}, {value}): refused by this build ({prev}); the knob stays at {self.get(key)}"
                             + ("" if self.get("sweep_build") == 1 else " -- alternative formulations live in the evidence build (build.ensure_built(sweep=True))"))
        return prev

    def require_variant(self, key, value):
        return self.set_variant(key, value, strict=True)

    def get(self, key):
        return self._lib.bitnuc_ctx_set_variant(self._h, key.encode(), -1)

    @property
    def stream(self):
        return self._lib.bitnuc_ctx_stream(self._h)

    def sync(self):
        err = L.BitnucErr()
        if self._lib.bitnuc_ctx_sync(self._h, C.byref(err)) != L.OK:
            _raise(err)

    # -- host-pointer API -------------------------------------------------------------
    def as_2bit(self, seq):
        s = _as_u8(seq)
        out = C.c_uint64(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_as_2bit(self._h, _ptr(s), s.size, C.byref(out), C.byref(err)) != L.OK:
            _raise(err)
        return out.value

    def from_2bit_alloc(self, packed, expected_size):
        out = np.empty(min(int(expected_size), 32), dtype=np.uint8)
        err = L.BitnucErr()
        if self._lib.bitnuc_from_2bit(self._h, C.c_uint64(packed), int(expected_size), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out.tobytes()

    def from_2bit(self, packed, expected_size, sequence):
        sequence += self.from_2bit_alloc(packed, expected_size)  # append (unpacking/avx.rs:59,73)

    def encode_array(self, sequence):
        """-> (words ndarray[uint64]) or raises; on InvalidBase the exception carries
        `.words`, the words the reference's Vec holds at that point."""
        s = _as_u8(sequence)
        if s.size == 0:
            # packing/avx.rs:138: `0..n_chunks - 1` underflows -> the reference panics
            raise RuntimeError("encode of an empty sequence panics in the reference (attempt to subtract with overflow)")
        out = np.empty((s.size + 31) // 32, dtype=np.uint64)
        nw = C.c_size_t(0)
        err = L.BitnucErr()
        st = self._lib.bitnuc_encode(self._h, _ptr(s), s.size, _ptr(out), C.byref(nw), C.byref(err))
        if st != L.OK:
            try:
                _raise(err)
            except NucleotideError as e:
                e.words = out[: nw.value].copy()
                raise
        return out

    def encode_into(self, sequence, out):
        """encode into a caller-owned uint64 array of >= ceil(len/32) words (no allocation: a pipeline reuses its buffers,
        and a freshly allocated output would be timed by its page faults, not by the codec).  -> number of words."""
        s = _as_u8(sequence)
        if not (isinstance(out, np.ndarray) and out.dtype == np.uint64 and out.flags.c_contiguous and out.size >= (s.size + 31) // 32):
            raise ValueError("out must be a contiguous uint64 array of at least ceil(len/32) words")
        nw = C.c_size_t(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_encode(self._h, _ptr(s), s.size, _ptr(out), C.byref(nw), C.byref(err)) != L.OK:
            _raise(err)
        return nw.value

    def decode_into(self, ebuf, n_bases, out):
        """decode into a caller-owned uint8 array of >= n_bases bytes (see encode_into)."""
        e = _as_u64(ebuf)
        if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous and out.size >= n_bases):
            raise ValueError("out must be a contiguous uint8 array of at least n_bases bytes")
        err = L.BitnucErr()
        if self._lib.bitnuc_decode(self._h, _ptr(e), e.size, int(n_bases), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)

    def encode(self, sequence, ebuf):
        del ebuf[:]  # ebuf.clear(), packing/avx.rs:132
        try:
            words = self.encode_array(sequence)
        except NucleotideError as e:
            ebuf.extend(int(w) for w in getattr(e, "words", ()))
            raise
        ebuf.extend(words.tolist())

    def encode_alloc(self, sequence):
        ebuf = []
        self.encode(sequence, ebuf)
        return ebuf

    def decode_array(self, ebuf, n_bases):
        e = _as_u64(ebuf)
        out = np.empty(int(n_bases), dtype=np.uint8)
        err = L.BitnucErr()
        if self._lib.bitnuc_decode(self._h, _ptr(e), e.size, int(n_bases), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    def decode(self, ebuf, n_bases, dbuf):
        dbuf += self.decode_array(ebuf, n_bases).tobytes()  # append (unpacking/avx.rs:122,140)

    def hdist_scalar(self, u, v, length):
        out = C.c_uint32(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_hdist_scalar(self._h, C.c_uint64(u), C.c_uint64(v), int(length), C.byref(out), C.byref(err)) != L.OK:
            _raise(err)
        return out.value

    def hdist(self, ebuf1, ebuf2, n_bases):
        a, b = _as_u64(ebuf1), _as_u64(ebuf2)
        out = C.c_uint32(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_hdist(self._h, _ptr(a), a.size, _ptr(b), b.size, int(n_bases), C.byref(out), C.byref(err)) != L.OK:
            _raise(err)
        return out.value

    def as_2bit_batch(self, kmers, k, stride=None, count=None):
        s = _as_u8(kmers)
        stride = k if stride is None else stride
        if count is None:
            count = 0 if s.size < k else (s.size - k) // stride + 1
        elif count and 0 < k <= 32 and (count - 1) * stride + k > s.size:  # the C ABI trusts the caller's sizes
            raise ValueError("kmers holds fewer than (count-1)*stride + k bytes")
        out = np.empty(count, dtype=np.uint64)
        err = L.BitnucErr()
        if self._lib.bitnuc_as_2bit_batch(self._h, _ptr(s), int(k), int(stride), int(count), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    def kmer_hdist_scan(self, ref, k, query):
        s = _as_u8(ref)
        nwin = s.size - k + 1 if (s.size >= k and k > 0) else 0
        out = np.empty(nwin, dtype=np.uint8)
        err = L.BitnucErr()
        if self._lib.bitnuc_kmer_hdist_scan(self._h, _ptr(s), s.size, int(k), C.c_uint64(query), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    # -- analysis on packed words (src/utils/analysis.rs, hamming/scalar.rs) ------------------
    def base_counts(self, words, n_bases):
        """[A, C, G, T] counts of a packed sequence (BaseCount::base_counts, analysis.rs:23-39)."""
        w = _as_u64(words)
        out = np.zeros(4, dtype=np.uint64)
        err = L.BitnucErr()
        if self._lib.bitnuc_base_counts(self._h, _ptr(w), w.size, int(n_bases), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return [int(x) for x in out]

    def gc_content(self, words, n_bases):
        """GCContent::gc_content (analysis.rs:7-16): percent, 0.0 for an empty sequence."""
        if n_bases == 0:
            return 0.0
        c = self.base_counts(words, n_bases)
        return (float(c[1] + c[2]) / float(n_bases)) * 100.0

    def hdist_pairs(self, a, b, length):
        a, b = _as_u64(a), _as_u64(b)
        if a.size != b.size:
            raise ValueError("a and b must hold the same number of words")
        out = np.empty(a.size, dtype=np.uint8)
        err = L.BitnucErr()
        if self._lib.bitnuc_hdist_pairs(self._h, _ptr(a), _ptr(b), a.size, int(length), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    def hdist_query(self, query, targets, length):
        t = _as_u64(targets)
        out = np.empty(t.size, dtype=np.uint8)
        err = L.BitnucErr()
        if self._lib.bitnuc_hdist_query(self._h, C.c_uint64(query), _ptr(t), t.size, int(length), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    def split_packed(self, ebuf, slen, idx, lbuf, rbuf, canonical=False):
        """split_packed(ebuf, slen, idx, &mut lbuf, &mut rbuf) (functions/split.rs:15-99): clears both
        lists, then fills them.  canonical=False reproduces the reference word for word (see
        include/bitnuc_hip.h); canonical=True gives encode(seq[:idx]) / encode(seq[idx:])."""
        e = _as_u64(ebuf)
        flags = L.SPLIT_CANONICAL if canonical else L.SPLIT_AS_WRITTEN
        nl, nr = C.c_size_t(0), C.c_size_t(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_split_packed_sizes(e.size, int(slen), int(idx), flags, C.byref(nl), C.byref(nr), C.byref(err)) != L.OK:
            _raise(err)  # the reference validates before it clears (split.rs:23-32)
        lo, ro = np.empty(nl.value, dtype=np.uint64), np.empty(nr.value, dtype=np.uint64)
        if self._lib.bitnuc_split_packed(self._h, _ptr(e), e.size, int(slen), int(idx), flags, _ptr(lo), C.byref(nl),
                                         _ptr(ro), C.byref(nr), C.byref(err)) != L.OK:
            _raise(err)
        del lbuf[:]
        del rbuf[:]
        lbuf.extend(int(x) for x in lo[: nl.value])
        rbuf.extend(int(x) for x in ro[: nr.value])

    def split_packed_sizes(self, n_words, slen, idx, canonical=False):
        nl, nr = C.c_size_t(0), C.c_size_t(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_split_packed_sizes(int(n_words), int(slen), int(idx), L.SPLIT_CANONICAL if canonical else L.SPLIT_AS_WRITTEN,
                                               C.byref(nl), C.byref(nr), C.byref(err)) != L.OK:
            _raise(err)
        return nl.value, nr.value

    def split_packed_dev(self, d_ebuf, n_words, slen, idx, d_lbuf, d_rbuf, canonical=False):
        self._call_dev(self._lib.bitnuc_split_packed_dev, _dev_ptr(d_ebuf), int(n_words), int(slen), int(idx),
                       L.SPLIT_CANONICAL if canonical else L.SPLIT_AS_WRITTEN, _dev_ptr(d_lbuf), _dev_ptr(d_rbuf))

    def base_counts_dev(self, d_words, n_words, n_bases, d_counts):
        self._call_dev(self._lib.bitnuc_base_counts_dev, _dev_ptr(d_words), int(n_words), int(n_bases), _dev_ptr(d_counts))

    def hdist_pairs_dev(self, d_a, d_b, count, length, d_dist):
        self._call_dev(self._lib.bitnuc_hdist_pairs_dev, _dev_ptr(d_a), _dev_ptr(d_b), int(count), int(length), _dev_ptr(d_dist))

    def hdist_query_dev(self, query, d_targets, count, length, d_dist):
        err = L.BitnucErr()
        if self._lib.bitnuc_hdist_query_dev(self._h, C.c_uint64(query), _dev_ptr(d_targets), int(count), int(length), _dev_ptr(d_dist), C.byref(err)) != L.OK:
            _raise(err)

    # -- ragged batches of independent sequences ---------------------------------------------
    def encode_batch(self, seq, offsets):
        """Encode `count` back-to-back sequences (sequence i = seq[offsets[i]:offsets[i+1]]).
        -> (words ndarray[uint64], word_offsets ndarray[uint64] of count+1 entries)."""
        s = _as_u8(seq)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        count = off.size - 1
        if count < 0:
            raise ValueError("offsets needs count+1 entries")
        if count and int(off.max()) > s.size:
            raise ValueError("offsets point past the end of seq")
        cap = int((int(off[-1]) - int(off[0])) // 32 + count) if count else 0
        out = np.empty(cap, dtype=np.uint64)
        wo = np.zeros(count + 1, dtype=np.uint64)
        nw = C.c_size_t(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_encode_batch(self._h, _ptr(s), _ptr(off), count, _ptr(out), cap, _ptr(wo), C.byref(nw), C.byref(err)) != L.OK:
            if err.status == L.INVALID_RANGE:
                raise NucleotideError("InvalidRange", index=int(err.value))
            _raise(err)
        return out[: nw.value], wo

    def encode_fixed(self, seq, read_len, stride=None, count=None):
        """Encode `count` reads of `read_len` bases, read r at seq[r*stride:]. -> ndarray[uint64] of
        shape (count, ceil(read_len/32)): row r == encode(read r)."""
        s = _as_u8(seq)
        stride = read_len if stride is None else stride
        if count is None:
            count = 0 if s.size < read_len else (s.size - read_len) // stride + 1
        elif count and (count - 1) * stride + read_len > s.size:
            raise ValueError("seq holds fewer than (count-1)*stride + read_len bytes")
        wpr = (read_len + 31) // 32
        out = np.empty((count, wpr), dtype=np.uint64)
        err = L.BitnucErr()
        if self._lib.bitnuc_encode_fixed(self._h, _ptr(s), int(read_len), int(stride), int(count), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    def decode_fixed(self, words, read_len, stride=None, out=None):
        """Inverse of encode_fixed: -> ndarray[uint8]; read r at [r*stride, r*stride+read_len).
        With stride > read_len pass `out` to keep its separator bytes."""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        wpr = (read_len + 31) // 32
        count = w.size // wpr if wpr else 0
        stride = read_len if stride is None else stride
        nbytes = (count - 1) * stride + read_len if count else 0
        if out is None:
            out = np.zeros(nbytes, dtype=np.uint8)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous and out.size >= nbytes):
            raise ValueError("out must be a contiguous uint8 array of at least (count-1)*stride + read_len bytes")
        err = L.BitnucErr()
        if self._lib.bitnuc_decode_fixed(self._h, _ptr(w), int(read_len), int(stride), int(count), _ptr(out), C.byref(err)) != L.OK:
            _raise(err)
        return out

    def encode_fixed_dev(self, d_seq, read_len, stride, count, d_out):
        self._call_dev(self._lib.bitnuc_encode_fixed_dev, _dev_ptr(d_seq), int(read_len), int(stride), int(count), _dev_ptr(d_out))

    def decode_fixed_dev(self, d_words, read_len, stride, count, d_out):
        self._call_dev(self._lib.bitnuc_decode_fixed_dev, _dev_ptr(d_words), int(read_len), int(stride), int(count), _dev_ptr(d_out))

    def encode_many(self, seqs):
        """`[encode_alloc(s) for s in seqs]` in one launch: seqs is an iterable of bytes-like
        sequences.  -> list of uint64 arrays (views into one buffer), one per sequence."""
        seqs = [bytes(s) for s in seqs]
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        words, wo = self.encode_batch(b"".join(seqs), off)
        return [words[int(wo[i]):int(wo[i + 1])] for i in range(len(seqs))]

    def decode_batch(self, words, word_offsets, offsets):
        """Inverse of encode_batch: -> ndarray[uint8] of offsets[-1] bytes, sequence i at
        [offsets[i], offsets[i+1]) (bytes before offsets[0] are zero)."""
        w = _as_u64(words)
        wo = np.ascontiguousarray(word_offsets, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        count = off.size - 1
        if count < 0 or wo.size != off.size:
            raise ValueError("offsets and word_offsets need count+1 entries each")
        if w.size < int(wo[-1]):
            raise ValueError("words holds fewer than word_offsets[-1] entries")
        out = np.zeros(int(off[-1]), dtype=np.uint8)
        err = L.BitnucErr()
        if self._lib.bitnuc_decode_batch(self._h, _ptr(w), _ptr(wo), _ptr(off), count, _ptr(out), C.byref(err)) != L.OK:
            if err.status == L.INVALID_RANGE:
                raise NucleotideError("InvalidRange", index=int(err.value))
            _raise(err)
        return out

    def batch_word_offsets_dev(self, d_offsets, count, d_word_offsets):
        total = C.c_size_t(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_batch_word_offsets_dev(self._h, _dev_ptr(d_offsets), int(count), _dev_ptr(d_word_offsets), C.byref(total), C.byref(err)) != L.OK:
            _raise(err)
        return total.value

    def encode_batch_dev(self, d_seq, d_offsets, d_word_offsets, count, total_words, d_out):
        self._call_dev(self._lib.bitnuc_encode_batch_dev, _dev_ptr(d_seq), _dev_ptr(d_offsets), _dev_ptr(d_word_offsets), int(count), int(total_words), _dev_ptr(d_out))

    def decode_batch_dev(self, d_words, d_word_offsets, d_offsets, count, total_words, d_out):
        self._call_dev(self._lib.bitnuc_decode_batch_dev, _dev_ptr(d_words), _dev_ptr(d_word_offsets), _dev_ptr(d_offsets), int(count), int(total_words), _dev_ptr(d_out))

    # -- device-pointer API (async; data errors surface at sync()) -----------------------
    def _call_dev(self, fn, *args):
        err = L.BitnucErr()
        if fn(self._h, *args, C.byref(err)) != L.OK:
            _raise(err)

    def encode_dev(self, d_seq, length, d_out):
        self._call_dev(self._lib.bitnuc_encode_dev, _dev_ptr(d_seq), int(length), _dev_ptr(d_out))

    def decode_dev(self, d_ebuf, n_words, n_bases, d_out):
        self._call_dev(self._lib.bitnuc_decode_dev, _dev_ptr(d_ebuf), int(n_words), int(n_bases), _dev_ptr(d_out))

    def as_2bit_batch_dev(self, d_kmers, k, stride, count, d_out):
        self._call_dev(self._lib.bitnuc_as_2bit_batch_dev, _dev_ptr(d_kmers), int(k), int(stride), int(count), _dev_ptr(d_out))

    def kmer_hdist_scan_dev(self, d_ref, n, k, query, d_dist):
        self._call_dev(self._lib.bitnuc_kmer_hdist_scan_dev, _dev_ptr(d_ref), int(n), int(k), C.c_uint64(query), _dev_ptr(d_dist))

    def kmer_hdist_count_dev(self, d_ref, n, k, query, tau, d_count):
        """Number of windows with Hamming distance <= tau to the query (fused scan, no distance bytes written) -> *d_count (u64)."""
        self._call_dev(self._lib.bitnuc_kmer_hdist_count_dev, _dev_ptr(d_ref), int(n), int(k), C.c_uint64(query), int(tau), _dev_ptr(d_count))

    def hdist_dev(self, d_a, na, d_b, nb, n_bases, d_result):
        self._call_dev(self._lib.bitnuc_hdist_dev, _dev_ptr(d_a), int(na), _dev_ptr(d_b), int(nb), int(n_bases), _dev_ptr(d_result))

    def nucgen_dev(self, d_out, length, seed, first=0, flags=0):
        self._call_dev(self._lib.bitnuc_nucgen_dev, _dev_ptr(d_out), int(length), C.c_uint64(seed), C.c_uint64(first), int(flags))

    def host_pipe_info(self):
        """Configuration and creation-time measurements of the pipelined host-pointer path (creates it if needed)."""
        names = ["cores_visible", "cores_quota", "cores_usable", "chunk_bases", "depth", "encode_stage_in_threads", "encode_hand_back_threads",
                 "decode_stage_in_threads", "decode_hand_back_threads", "heavy_side_thread_cap", "gpu_numa_node", "workers_bound_to_cpus", "direct_engine"]
        out = (C.c_double * len(names))()
        err = L.BitnucErr()
        if self._lib.bitnuc_host_pipe_info(self._h, out, len(names), C.byref(err)) != L.OK:
            _raise(err)
        return {k: int(v) for k, v in zip(names, out)}

    def stream_probe_dev(self, mode, d_src, d_dst, nbytes):
        self._call_dev(self._lib.bitnuc_stream_probe_dev, int(mode), _dev_ptr(d_src), _dev_ptr(d_dst), int(nbytes))


class BatchPlan:
    """Layout plan of a ragged batch (bitnuc_batch_plan): built once from the device offsets table, used by every
    encode / decode of that layout.  `build` may be called again for the next batch (memory is reused)."""

    def __init__(self, ctx, d_offsets=None, count=0):
        self._ctx, self._lib = ctx, ctx._lib
        self._h = C.c_void_p()
        err = L.BitnucErr()
        if self._lib.bitnuc_batch_plan_create(ctx._h, C.byref(self._h), C.byref(err)) != L.OK:
            self._h = C.c_void_p()
            _raise(err)
        self.total_words = 0
        if d_offsets is not None:
            self.build(d_offsets, count)

    def build(self, d_offsets, count):
        total = C.c_size_t(0)
        err = L.BitnucErr()
        if self._lib.bitnuc_batch_plan_build_dev(self._ctx._h, self._h, _dev_ptr(d_offsets), int(count), C.byref(total), C.byref(err)) != L.OK:
            _raise(err)
        self.total_words = total.value
        return total.value

    @property
    def word_offsets_ptr(self):
        """Device address of the plan's word-offsets table (count + 1 u64 entries)."""
        return self._lib.bitnuc_batch_plan_word_offsets_dev(self._h)

    def encode_dev(self, d_seq, d_out):
        err = L.BitnucErr()
        if self._lib.bitnuc_encode_batch_plan_dev(self._ctx._h, self._h, _dev_ptr(d_seq), _dev_ptr(d_out), C.byref(err)) != L.OK:
            _raise(err)

    def decode_dev(self, d_words, d_out):
        err = L.BitnucErr()
        if self._lib.bitnuc_decode_batch_plan_dev(self._ctx._h, self._h, _dev_ptr(d_words), _dev_ptr(d_out), C.byref(err)) != L.OK:
            _raise(err)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.bitnuc_batch_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class Comm:
    """One rank of an RCCL communicator bound to a Context (config 4's concatenation).
    One process per GPU: rank 0 makes `Comm.unique_id()`, shares the 128 bytes, every rank
    calls `Comm(ctx, nranks, rank, uid)`."""

    def __init__(self, ctx, nranks, rank, uid):
        self._lib = L.load()
        self._ctx = ctx
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(uid))
        err = L.BitnucErr()
        if self._lib.bitnuc_comm_init_rank(ctx._h, int(nranks), int(rank), buf, C.byref(self._h), C.byref(err)) != L.OK:
            self._h = C.c_void_p()
            _raise(err)

    @staticmethod
    def unique_id():
        lib = L.load()
        buf = (C.c_uint8 * 128)()
        err = L.BitnucErr()
        if lib.bitnuc_comm_get_unique_id(buf, C.byref(err)) != L.OK:
            _raise(err)
        return bytes(buf)

    @property
    def nranks(self):
        return self._lib.bitnuc_comm_nranks(self._h)

    @property
    def rank(self):
        return self._lib.bitnuc_comm_rank(self._h)

    def allgather_words_dev(self, d_local, count, d_all):
        err = L.BitnucErr()
        if self._lib.bitnuc_allgather_words_dev(self._ctx._h, self._h, _dev_ptr(d_local), int(count), _dev_ptr(d_all), C.byref(err)) != L.OK:
            _raise(err)

    def allgatherv_words_dev(self, counts, d_all):
        """In-place all-gather of UNEQUAL word counts (a ragged batch sharded by whole sequences, dist.batch_shard_ranges): rank r's
        counts[r] words already sit at d_all + sum(counts[:r]); afterwards every rank holds all of them."""
        arr = (C.c_size_t * len(counts))(*[int(x) for x in counts])
        err = L.BitnucErr()
        if self._lib.bitnuc_allgatherv_words_dev(self._ctx._h, self._h, arr, _dev_ptr(d_all), C.byref(err)) != L.OK:
            _raise(err)

    def set_threaded(self, threaded=True):
        return self._lib.bitnuc_comm_set_threaded(self._h, int(bool(threaded)))

    def encode_sharded_allgather_dev(self, d_seq_shard, shard_len, d_all):
        err = L.BitnucErr()
        if self._lib.bitnuc_encode_sharded_allgather_dev(self._ctx._h, self._h, _dev_ptr(d_seq_shard), int(shard_len), _dev_ptr(d_all), C.byref(err)) != L.OK:
            _raise(err)

    def encode_sharded_allgather_overlapped_dev(self, d_seq_shard, shard_len, n_chunks, d_all):
        """Same result as encode_sharded_allgather_dev, the exchange hidden behind the encode: n_chunks pieces, each moved in
        place by a second stream as soon as it is encoded (include/bitnuc_hip.h)."""
        err = L.BitnucErr()
        if self._lib.bitnuc_encode_sharded_allgather_overlapped_dev(self._ctx._h, self._h, _dev_ptr(d_seq_shard), int(shard_len), int(n_chunks), _dev_ptr(d_all), C.byref(err)) != L.OK:
            _raise(err)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.bitnuc_comm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class CommGroup:
    """All ranks of a communicator in ONE thread (bitnuc_comm_init_all[_devices]): n contexts + communicators, driven together by the
    _all entry points, which issue every rank's part of an exchange inside one RCCL group and synchronise every stream before they
    return.  The per-rank entry points refuse these communicators (they would wait for peers this thread has not issued yet)."""

    def __init__(self, n_gpus, devices=None, lib_path=None):
        self._lib = L.load(lib_path)
        self.n = int(n_gpus)
        self._ctxs = (C.c_void_p * self.n)()
        self._comms = (C.c_void_p * self.n)()
        devs = (C.c_int * self.n)(*devices) if devices is not None else None
        err = L.BitnucErr()
        if self._lib.bitnuc_comm_init_all_devices(self.n, devs, self._ctxs, self._comms, C.byref(err)) != L.OK:
            self.n = 0
            _raise(err)

    def encode_sharded_allgather(self, d_seq_shards, shard_len, d_alls, n_chunks=0):
        """d_seq_shards[r]: rank r's shard (shard_len bases, a multiple of 32, on rank r's device); d_alls[r]: rank r's output of
        n * shard_len / 32 words.  n_chunks = 0: one grouped ncclAllGather; >= 1: the chunked in-place exchange behind the encode.
        On InvalidBase the error carries the shard-relative index and, in .value, the rank."""
        seqs = (C.c_void_p * self.n)(*[_dev_ptr(t) for t in d_seq_shards])
        alls = (C.c_void_p * self.n)(*[_dev_ptr(t) for t in d_alls])
        err = L.BitnucErr()
        if n_chunks:
            st = self._lib.bitnuc_encode_sharded_allgather_overlapped_all(self.n, self._ctxs, self._comms, seqs, int(shard_len), int(n_chunks), alls, C.byref(err))
        else:
            st = self._lib.bitnuc_encode_sharded_allgather_all(self.n, self._ctxs, self._comms, seqs, int(shard_len), alls, C.byref(err))
        if st != L.OK:
            try:
                _raise(err)
            except NucleotideError as e:
                e.rank = int(err.value) if st == L.INVALID_BASE else None  # whose shard holds the byte
                raise

    def context(self, rank):
        """Rank `rank`'s context as a Context object that does NOT own the handle (the group destroys it): for the calls a rank makes
        on its own device between the group's exchanges -- building a BatchPlan, encoding its run of a ragged batch into its slot."""
        c = Context.__new__(Context)
        c._lib, c._h, c.device, c._borrowed = self._lib, C.c_void_p(self._ctxs[rank]), None, True
        return c

    def allgatherv_words(self, counts, d_alls):
        """bitnuc_allgatherv_words_all: every rank's counts[r] words (already at d_alls[r] + sum(counts[:r])) reach every rank's buffer."""
        arr = (C.c_size_t * self.n)(*[int(x) for x in counts])
        alls = (C.c_void_p * self.n)(*[_dev_ptr(t) for t in d_alls])
        err = L.BitnucErr()
        if self._lib.bitnuc_allgatherv_words_all(self.n, self._ctxs, self._comms, arr, alls, C.byref(err)) != L.OK:
            _raise(err)

    def close(self):
        for i in range(getattr(self, "n", 0)):
            if self._comms[i]:
                self._lib.bitnuc_comm_destroy(self._comms[i])
                self._comms[i] = None
            if self._ctxs[i]:
                self._lib.bitnuc_ctx_destroy(self._ctxs[i])
                self._ctxs[i] = None
        self.n = 0

    __del__ = close


def batch_shard_ranges(offsets, nranks, lib_path=None):
    """bitnuc_batch_shard_ranges (host arithmetic, no device): the partition of a ragged batch by WHOLE sequences, balanced by word
    count.  Returns (seq_first[nranks+1], word_first[nranks+1]) as numpy arrays; bitnuc_amd.dist.batch_shard_ranges is the same rule
    in numpy (tests hold the two against each other)."""
    lib = L.load(lib_path)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    count = off.size - 1
    seq_first = np.zeros(nranks + 1, dtype=np.uint64)
    word_first = np.zeros(nranks + 1, dtype=np.uint64)
    err = L.BitnucErr()
    if lib.bitnuc_batch_shard_ranges(_ptr(off), count, int(nranks), _ptr(seq_first), _ptr(word_first), C.byref(err)) != L.OK:
        _raise(err)
    return seq_first, word_first


def peer_link_probe(src_device, dst_devices, nbytes=256 << 20, reps=3, lib_path=None):
    """xGMI link probe: hipMemcpyPeerAsync from src_device to each of dst_devices, one link at a time and all at once.
    Returns {"gb_s_each": [...], "gb_s_all": x}; raises NucleotideError('Unsupported') with fewer than two devices."""
    lib = L.load(lib_path)
    n = len(dst_devices)
    dst = (C.c_int * max(n, 1))(*dst_devices)
    each = (C.c_double * max(n, 1))()
    allv = C.c_double(0)
    err = L.BitnucErr()
    if lib.bitnuc_peer_link_probe(int(src_device), dst, n, int(nbytes), int(reps), each, C.byref(allv), C.byref(err)) != L.OK:
        _raise(err)
    return {"gb_s_each": [round(each[i], 1) for i in range(n)], "gb_s_all": round(allv.value, 1)}


# ---- module-level functions with the reference's names (default context, device 0) ----------
_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context(0)
    return _default


_ctxfree = None


def context_free():
    """A handle with a NULL bitnuc_ctx: the single-word functions (and bulk host-pointer calls below the
    library's host cutoff) are host code and need no device, like the reference's free functions."""
    global _ctxfree
    if _ctxfree is None:
        c = Context.__new__(Context)
        c._lib, c._h, c.device = L.load(), C.c_void_p(0), None
        _ctxfree = c
    return _ctxfree


def as_2bit(seq):
    return context_free().as_2bit(seq)


def from_2bit(packed, expected_size, sequence):
    return context_free().from_2bit(packed, expected_size, sequence)


def from_2bit_alloc(packed, expected_size):
    return context_free().from_2bit_alloc(packed, expected_size)


def encode(sequence, ebuf):
    return default_context().encode(sequence, ebuf)


def encode_alloc(sequence):
    return default_context().encode_alloc(sequence)


def decode(ebuf, n_bases, dbuf):
    return default_context().decode(ebuf, n_bases, dbuf)


def hdist_scalar(u, v, length):
    return context_free().hdist_scalar(u, v, length)


def hdist(ebuf1, ebuf2, n_bases):
    return default_context().hdist(ebuf1, ebuf2, n_bases)


def as_2bit_batch(kmers, k, stride=None, count=None):
    return default_context().as_2bit_batch(kmers, k, stride, count)


def kmer_hdist_scan(ref, k, query):
    return default_context().kmer_hdist_scan(ref, k, query)


def split_packed(ebuf, slen, idx, lbuf, rbuf, canonical=False):
    return default_context().split_packed(ebuf, slen, idx, lbuf, rbuf, canonical=canonical)
