// bitnuc.hpp -- compiled host layer over the C ABI (include/bitnuc_hip.h) that mirrors
// bitnuc's public Rust API (src/lib.rs:214-220) in C++: same names, argument meaning,
// Vec clear/append conventions and error values, so that code (and tests) written
// against the reference read the same here.  north_star asks for this layer in Rust;
// the image has no rustc/cargo, so the Rust shim is source-only (rust/) and this
// header is the layer that is compiled and tested.  Header-only; link -lbitnuc_hip.
//
//   Rust                                               C++
//   as_2bit(&[u8]) -> Result<u64, NucleotideError>     Result<uint64_t> as_2bit(bytes)
//   from_2bit(u64, usize, &mut Vec<u8>) -> Result<()>  Result<void> from_2bit(p, n, std::vector<uint8_t>&)
//   from_2bit_alloc(u64, usize) -> Result<Vec<u8>>     Result<std::vector<uint8_t>> from_2bit_alloc(p, n)
//   encode(&[u8], &mut Vec<u64>) -> Result<()>         Result<void> encode(bytes, std::vector<uint64_t>&)
//   encode_alloc(&[u8]) -> Result<Vec<u64>>            Result<std::vector<uint64_t>> encode_alloc(bytes)
//   decode(&[u64], usize, &mut Vec<u8>) -> Result<()>  Result<void> decode(words, n, std::vector<uint8_t>&)
//   hdist_scalar(u64, u64, usize) -> Result<u32>       Result<uint32_t> hdist_scalar(u, v, len)
//   hdist(&[u64], &[u64], usize) -> Result<u32>        Result<uint32_t> hdist(a, b, n)
//   split_packed(&[u64], usize, usize, &mut Vec<u64>, &mut Vec<u64>) -> Result<()>
//                                                      Result<void> split_packed(words, slen, idx, lbuf&, rbuf&)
//
// Every function computes on the GPU through libbitnuc_hip.so; there is no CPU path.
#pragma once

#include "bitnuc_hip.h"

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace bitnuc {

// src/error.rs:3-18
struct NucleotideError {
    enum Kind { InvalidBase, SequenceTooLong, InvalidLength, IndexOutOfBounds, InvalidRange, Unsupported, Backend };
    Kind kind = Unsupported;
    uint8_t base = 0;       // InvalidBase(u8)
    size_t len = 0;         // SequenceTooLong(usize) / InvalidLength(usize)
    uint64_t index = 0;     // extra: absolute index of the invalid base
    int backend_code = 0;   // extra: hipError_t for Backend
    size_t start = 0, end = 0, length = 0, oob_index = 0; // IndexOutOfBounds{index,length} / InvalidRange{start,end,length}

    static NucleotideError invalid_base(uint8_t b) { NucleotideError e; e.kind = InvalidBase; e.base = b; return e; }
    static NucleotideError sequence_too_long(size_t n) { NucleotideError e; e.kind = SequenceTooLong; e.len = n; return e; }
    static NucleotideError invalid_length(size_t n) { NucleotideError e; e.kind = InvalidLength; e.len = n; return e; }
    static NucleotideError unsupported() { NucleotideError e; e.kind = Unsupported; return e; }

    // derive(PartialEq, Eq): variant + payload
    bool operator==(const NucleotideError &o) const {
        if (kind != o.kind) return false;
        if (kind == InvalidBase) return base == o.base;
        if (kind == SequenceTooLong || kind == InvalidLength) return len == o.len;
        if (kind == IndexOutOfBounds) return oob_index == o.oob_index && length == o.length;
        if (kind == InvalidRange) return start == o.start && end == o.end && length == o.length;
        return true;
    }
    bool operator!=(const NucleotideError &o) const { return !(*this == o); }

    std::string to_string() const { // Display, src/error.rs:20-45
        switch (kind) {
        case InvalidBase: return "Invalid nucleotide base: " + std::to_string(base);
        case SequenceTooLong: return "Sequence length " + std::to_string(len) + " exceeds maximum";
        case InvalidLength: return "Invalid length: " + std::to_string(len);
        case IndexOutOfBounds: return "Index " + std::to_string(oob_index) + " out of bounds for sequence of length " + std::to_string(length);
        case InvalidRange: return "Invalid range " + std::to_string(start) + ".." + std::to_string(end) + " for sequence of length " + std::to_string(length);
        case Unsupported: return "Unsupported architecture";
        case Backend: return "HIP backend error " + std::to_string(backend_code);
        default: return "NucleotideError";
        }
    }

    static NucleotideError from_c(const bitnuc_err &e) {
        NucleotideError r;
        switch (e.status) {
        case BITNUC_INVALID_BASE: r.kind = InvalidBase; r.base = e.byte; r.index = e.index; break;
        case BITNUC_SEQUENCE_TOO_LONG: r.kind = SequenceTooLong; r.len = (size_t)e.value; break;
        case BITNUC_INVALID_LENGTH: r.kind = InvalidLength; r.len = (size_t)e.value; break;
        case BITNUC_INDEX_OUT_OF_BOUNDS: r.kind = IndexOutOfBounds; r.oob_index = (size_t)e.index; r.length = (size_t)e.value; break;
        case BITNUC_INVALID_RANGE: r.kind = InvalidRange; break;
        case BITNUC_BACKEND_ERROR: r.kind = Backend; r.backend_code = e.backend_code; break;
        default: r.kind = Unsupported; break;
        }
        return r;
    }
};

// A small Result<T, NucleotideError>.
template <class T> class Result {
  public:
    Result(T v) : ok_(true), val_(std::move(v)) {}
    Result(NucleotideError e) : ok_(false), err_(e) {}
    bool is_ok() const { return ok_; }
    bool is_err() const { return !ok_; }
    T &unwrap() { if (!ok_) throw std::runtime_error("called unwrap() on an Err value: " + err_.to_string()); return val_; }
    const NucleotideError &unwrap_err() const { if (ok_) throw std::runtime_error("called unwrap_err() on an Ok value"); return err_; }
    bool operator==(const Result &o) const { return ok_ == o.ok_ && (ok_ ? val_ == o.val_ : err_ == o.err_); }
  private:
    bool ok_;
    T val_{};
    NucleotideError err_{};
};
template <> class Result<void> {
  public:
    Result() : ok_(true) {}
    Result(NucleotideError e) : ok_(false), err_(e) {}
    bool is_ok() const { return ok_; }
    bool is_err() const { return !ok_; }
    void unwrap() const { if (!ok_) throw std::runtime_error("called unwrap() on an Err value: " + err_.to_string()); }
    const NucleotideError &unwrap_err() const { if (ok_) throw std::runtime_error("called unwrap_err() on an Ok value"); return err_; }
  private:
    bool ok_;
    NucleotideError err_{};
};
template <class T> Result<T> Ok(T v) { return Result<T>(std::move(v)); }
inline Result<void> Ok() { return Result<void>(); }

// &[u8]
struct Bytes {
    const uint8_t *ptr;
    size_t len;
    Bytes(const uint8_t *p, size_t n) : ptr(p), len(n) {}
    Bytes(const std::vector<uint8_t> &v) : ptr(v.data()), len(v.size()) {}
    Bytes(std::string_view s) : ptr(reinterpret_cast<const uint8_t *>(s.data())), len(s.size()) {}
    Bytes(const char *s) : Bytes(std::string_view(s)) {}
    Bytes(const std::string &s) : Bytes(std::string_view(s)) {}
};
// &[u64]
struct Words {
    const uint64_t *ptr;
    size_t len;
    Words(const uint64_t *p, size_t n) : ptr(p), len(n) {}
    Words(const std::vector<uint64_t> &v) : ptr(v.data()), len(v.size()) {}
};

// One device + stream + scratch.  Not shareable between threads concurrently.
class Context {
  public:
    explicit Context(int device = 0) {
        bitnuc_err e;
        if (bitnuc_ctx_create(device, &ctx_, &e) != BITNUC_OK)
            throw std::runtime_error("bitnuc: cannot create a HIP context (" + NucleotideError::from_c(e).to_string() +
                                     "); there is no CPU fallback");
    }
    ~Context() { bitnuc_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    bitnuc_ctx *raw() const { return ctx_; }

    Result<uint64_t> as_2bit(Bytes seq) const { // src/utils/packing/mod.rs:80-110
        uint64_t out = 0;
        bitnuc_err e;
        if (bitnuc_as_2bit(ctx_, seq.ptr, seq.len, &out, &e) != BITNUC_OK) return NucleotideError::from_c(e);
        return out;
    }
    Result<void> from_2bit(uint64_t packed, size_t expected_size, std::vector<uint8_t> &sequence) const { // unpacking/mod.rs:119-147
        uint8_t tmp[32];
        bitnuc_err e;
        if (bitnuc_from_2bit(ctx_, packed, expected_size, tmp, &e) != BITNUC_OK) return NucleotideError::from_c(e);
        sequence.insert(sequence.end(), tmp, tmp + expected_size); // appends (unpacking/avx.rs:59,73)
        return Ok();
    }
    Result<std::vector<uint8_t>> from_2bit_alloc(uint64_t packed, size_t expected_size) const { // unpacking/mod.rs:178-182
        std::vector<uint8_t> v;
        v.reserve(expected_size <= 32 ? expected_size : 0);
        Result<void> r = from_2bit(packed, expected_size, v);
        if (r.is_err()) return r.unwrap_err();
        return v;
    }
    Result<void> encode(Bytes sequence, std::vector<uint64_t> &ebuf) const { // src/utils/mod.rs:22-25
        ebuf.clear();                                                         // packing/avx.rs:132
        if (sequence.len == 0) // packing/avx.rs:138: `0..n_chunks - 1` underflows -> panic
            throw std::logic_error("attempt to subtract with overflow (encode of an empty sequence panics in the reference)");
        ebuf.resize((sequence.len + 31) / 32);
        size_t nw = 0;
        bitnuc_err e;
        int st = bitnuc_encode(ctx_, sequence.ptr, sequence.len, ebuf.data(), &nw, &e);
        ebuf.resize(nw); // on Err: the words pushed before the failing chunk (avx.rs:142-143)
        if (st != BITNUC_OK) return NucleotideError::from_c(e);
        return Ok();
    }
    Result<std::vector<uint64_t>> encode_alloc(Bytes sequence) const { // src/utils/mod.rs:38-42
        std::vector<uint64_t> ebuf;
        Result<void> r = encode(sequence, ebuf);
        if (r.is_err()) return r.unwrap_err();
        return ebuf;
    }
    Result<void> decode(Words ebuf, size_t n_bases, std::vector<uint8_t> &dbuf) const { // src/utils/mod.rs:60-62
        const size_t old = dbuf.size();
        dbuf.resize(old + n_bases); // appends (unpacking/avx.rs:122,140)
        bitnuc_err e;
        if (bitnuc_decode(ctx_, ebuf.ptr, ebuf.len, n_bases, dbuf.data() + old, &e) != BITNUC_OK) {
            dbuf.resize(old);
            return NucleotideError::from_c(e);
        }
        return Ok();
    }
    Result<uint32_t> hdist_scalar(uint64_t u, uint64_t v, size_t len) const { // hamming/scalar.rs:11-48
        uint32_t out = 0;
        bitnuc_err e;
        if (bitnuc_hdist_scalar(ctx_, u, v, len, &out, &e) != BITNUC_OK) return NucleotideError::from_c(e);
        return out;
    }
    Result<uint32_t> hdist(Words a, Words b, size_t n_bases) const { // hamming/multi.rs:121-160
        uint32_t out = 0;
        bitnuc_err e;
        if (bitnuc_hdist(ctx_, a.ptr, a.len, b.ptr, b.len, n_bases, &out, &e) != BITNUC_OK) return NucleotideError::from_c(e);
        return out;
    }
    // split_packed(ebuf, slen, idx, &mut lbuf, &mut rbuf)  functions/split.rs:15-99: validates, clears both
    // vectors, fills them.  canonical = false: the reference's words as written; true: encode(seq[..idx]) /
    // encode(seq[idx..]) (see include/bitnuc_hip.h).
    Result<void> split_packed(Words ebuf, size_t slen, size_t idx, std::vector<uint64_t> &lbuf, std::vector<uint64_t> &rbuf,
                              bool canonical = false) const {
        const int flags = canonical ? BITNUC_SPLIT_CANONICAL : BITNUC_SPLIT_AS_WRITTEN;
        size_t nl = 0, nr = 0;
        bitnuc_err e;
        if (bitnuc_split_packed_sizes(ebuf.len, slen, idx, flags, &nl, &nr, &e) != BITNUC_OK) return NucleotideError::from_c(e);
        lbuf.assign(nl, 0);
        rbuf.assign(nr, 0);
        if (bitnuc_split_packed(ctx_, ebuf.ptr, ebuf.len, slen, idx, flags, lbuf.data(), &nl, rbuf.data(), &nr, &e) != BITNUC_OK)
            return NucleotideError::from_c(e);
        return Result<void>();
    }
    // batched forms of the README.md:52-56 / src/lib.rs:170-173 host loops
    Result<std::vector<uint64_t>> as_2bit_batch(Bytes kmers, size_t k, size_t stride, size_t count) const {
        std::vector<uint64_t> out(count);
        bitnuc_err e;
        if (count && k && k <= 32 && (count - 1) * stride + k > kmers.len) return NucleotideError::unsupported(); // slice too short for the batch
        if (bitnuc_as_2bit_batch(ctx_, kmers.ptr, k, stride, count, out.data(), &e) != BITNUC_OK) return NucleotideError::from_c(e);
        return out;
    }
    Result<std::vector<uint8_t>> kmer_hdist_scan(Bytes ref, size_t k, uint64_t query) const {
        std::vector<uint8_t> out((k && ref.len >= k) ? ref.len - k + 1 : 0);
        bitnuc_err e;
        if (bitnuc_kmer_hdist_scan(ctx_, ref.ptr, ref.len, k, query, out.data(), &e) != BITNUC_OK) return NucleotideError::from_c(e);
        return out;
    }

    // Ragged batch: `for s in seqs { encode(s, &mut ebuf)? }` in one launch.  Sequence i =
    // seq[offsets[i] .. offsets[i+1]); returns the concatenated words and fills word_offsets
    // (count+1 entries, word_offsets[i] = first word of sequence i).
    Result<std::vector<uint64_t>> encode_batch(Bytes seq, const std::vector<uint64_t> &offsets,
                                               std::vector<uint64_t> &word_offsets) const {
        const size_t count = offsets.empty() ? 0 : offsets.size() - 1;
        word_offsets.assign(count + 1, 0);
        std::vector<uint64_t> out(count ? (size_t)((offsets[count] - offsets[0]) / 32 + count) : 0);
        size_t nw = 0;
        bitnuc_err e;
        if (count && seq.len < offsets[count]) return NucleotideError::unsupported(); // offsets point past the slice
        int st = bitnuc_encode_batch(ctx_, seq.ptr, offsets.data(), count, out.data(), out.size(), word_offsets.data(), &nw, &e);
        if (st != BITNUC_OK) return NucleotideError::from_c(e);
        out.resize(nw);
        return out;
    }
    // Fixed-length reads: `count` reads of read_len bases, read r at seq[r*stride ..]; returns
    // count*ceil(read_len/32) words, row r == encode_alloc(read r).
    Result<std::vector<uint64_t>> encode_fixed(Bytes seq, size_t read_len, size_t stride, size_t count) const {
        std::vector<uint64_t> out(count * ((read_len + 31) / 32));
        bitnuc_err e;
        if (count && (stride < read_len || (count - 1) * stride + read_len > seq.len)) return NucleotideError::unsupported();
        if (bitnuc_encode_fixed(ctx_, seq.ptr, read_len, stride, count, out.data(), &e) != BITNUC_OK) return NucleotideError::from_c(e);
        return out;
    }
    // Inverse: sequence i's bases are written at out[offsets[i] .. offsets[i+1]).
    Result<std::vector<uint8_t>> decode_batch(Words words, const std::vector<uint64_t> &word_offsets,
                                              const std::vector<uint64_t> &offsets) const {
        const size_t count = offsets.empty() ? 0 : offsets.size() - 1;
        std::vector<uint8_t> out(count ? (size_t)offsets[count] : 0);
        bitnuc_err e;
        if (word_offsets.size() != offsets.size() || (count && words.len < word_offsets[count])) return NucleotideError::unsupported();
        if (bitnuc_decode_batch(ctx_, words.ptr, word_offsets.data(), offsets.data(), count, out.data(), &e) != BITNUC_OK)
            return NucleotideError::from_c(e);
        return out;
    }

  private:
    bitnuc_ctx *ctx_ = nullptr;
};

// Free functions with the reference's names, on a per-thread default context (device 0).
inline Context &default_context() {
    thread_local Context ctx(0);
    return ctx;
}

// PackedSequence (src/sequence.rs:5-262) over GPU-encoded data, with the GCContent / BaseCount
// traits of src/utils/analysis.rs:3-39.  `new_` encodes on the GPU; slice / to_vec / get decode
// only the words the range touches, on the GPU; the traits count on the packed words.
class PackedSequence {
  public:
    static Result<PackedSequence> new_(Bytes seq) { // sequence.rs:40-52
        PackedSequence p;
        if (seq.len != 0) { // sequence.rs:42-44: skip encoding for empty sequences
            Result<void> r = default_context().encode(seq, p.data_);
            if (r.is_err()) return r.unwrap_err();
        }
        p.length_ = seq.len;
        return p;
    }
    size_t len() const { return length_; }
    bool is_empty() const { return length_ == 0; }
    const std::vector<uint64_t> &data() const { return data_; }
    Result<uint8_t> get(size_t index) const { // sequence.rs:116-135
        if (index >= length_) {
            NucleotideError e; e.kind = NucleotideError::IndexOutOfBounds; e.oob_index = index; e.length = length_;
            return e;
        }
        // the reference's shift + mask + match (sequence.rs:121-134): no call into the library
        return static_cast<uint8_t>("ACGT"[(data_[index / 32] >> (2 * (index % 32))) & 3]);
    }
    Result<std::vector<uint8_t>> slice(size_t start, size_t end) const { // sequence.rs:198-212
        if (start > end || end > length_) {
            NucleotideError e; e.kind = NucleotideError::InvalidRange; e.start = start; e.end = end; e.length = length_;
            return e;
        }
        std::vector<uint8_t> out;
        if (start == end) return out;
        const size_t w0 = start / 32, w1 = (end + 31) / 32;
        const size_t n = (length_ < w1 * 32 ? length_ : w1 * 32) - w0 * 32;
        std::vector<uint8_t> chunk;
        Result<void> r = default_context().decode(Words(data_.data() + w0, w1 - w0), n, chunk);
        if (r.is_err()) return r.unwrap_err();
        out.assign(chunk.begin() + (start - w0 * 32), chunk.begin() + (end - w0 * 32));
        return out;
    }
    Result<std::vector<uint8_t>> to_vec() const { return slice(0, length_); } // sequence.rs:260-262
    bool operator==(const PackedSequence &o) const { return length_ == o.length_ && data_ == o.data_; }
    bool operator!=(const PackedSequence &o) const { return !(*this == o); }
    // traits (analysis.rs)
    std::array<size_t, 4> base_counts() const {
        uint64_t c[4] = {0, 0, 0, 0};
        bitnuc_err e;
        if (bitnuc_base_counts(default_context().raw(), data_.data(), data_.size(), length_, c, &e) != BITNUC_OK)
            throw std::runtime_error("bitnuc: base_counts failed: " + NucleotideError::from_c(e).to_string());
        return {(size_t)c[0], (size_t)c[1], (size_t)c[2], (size_t)c[3]};
    }
    double gc_content() const { // analysis.rs:7-16
        if (length_ == 0) return 0.0;
        const auto c = base_counts();
        return ((double)(c[1] + c[2]) / (double)length_) * 100.0;
    }

  private:
    std::vector<uint64_t> data_;
    size_t length_ = 0;
};
inline Result<uint64_t> as_2bit(Bytes seq) { return default_context().as_2bit(seq); }
inline Result<void> from_2bit(uint64_t packed, size_t n, std::vector<uint8_t> &sequence) { return default_context().from_2bit(packed, n, sequence); }
inline Result<std::vector<uint8_t>> from_2bit_alloc(uint64_t packed, size_t n) { return default_context().from_2bit_alloc(packed, n); }
inline Result<void> encode(Bytes sequence, std::vector<uint64_t> &ebuf) { return default_context().encode(sequence, ebuf); }
inline Result<std::vector<uint64_t>> encode_alloc(Bytes sequence) { return default_context().encode_alloc(sequence); }
inline Result<void> decode(Words ebuf, size_t n_bases, std::vector<uint8_t> &dbuf) { return default_context().decode(ebuf, n_bases, dbuf); }
inline Result<uint32_t> hdist_scalar(uint64_t u, uint64_t v, size_t len) { return default_context().hdist_scalar(u, v, len); }
inline Result<uint32_t> hdist(Words a, Words b, size_t n_bases) { return default_context().hdist(a, b, n_bases); }
inline Result<void> split_packed(Words ebuf, size_t slen, size_t idx, std::vector<uint64_t> &lbuf, std::vector<uint64_t> &rbuf,
                                 bool canonical = false) {
    return default_context().split_packed(ebuf, slen, idx, lbuf, rbuf, canonical);
}

} // namespace bitnuc
