/*
 * bitnuc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's 2-bit pack/unpack hot path,
 * used as the parity checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under bitnuc_amd/ (the product) may
 * include, link, import or call anything in oracle/.
 *
 * Parity status: PINNED.  Every function here is checked against the
 * reference's own known-answer vectors (tests/golden/golden.json, generated
 * from the reference's in-file unit tests and doc-tests; see
 * tests/golden/README.md).  The reference itself is a Rust crate and this image
 * has no rustc/cargo, so oracle/_ref cannot be built.
 *
 * Citations are file:line under /root/reference.
 */
#ifndef BITNUC_ORACLE_H
#define BITNUC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status vocabulary == NucleotideError, src/error.rs:3-18 */
enum {
    ORC_OK = 0,
    ORC_INVALID_BASE = 1,      /* InvalidBase(u8)        */
    ORC_SEQUENCE_TOO_LONG = 2, /* SequenceTooLong(usize) */
    ORC_INVALID_LENGTH = 3,    /* InvalidLength(usize)   */
    ORC_INDEX_OUT_OF_BOUNDS = 4,
    ORC_INVALID_RANGE = 5,
    ORC_UNSUPPORTED = 6,
    ORC_PANIC = 99 /* the reference panics here (documented edge cases) */
};

typedef struct {
    int32_t status;
    uint8_t byte;   /* offending base for INVALID_BASE */
    uint64_t value; /* len for TOO_LONG / INVALID_LENGTH */
    uint64_t index; /* absolute index of the offending base (oracle extra) */
} orc_err;

/* src/utils/packing/naive.rs:3-20 (canonical) == avx.rs:76-128 on valid input */
int orc_as_2bit(const uint8_t *seq, size_t len, uint64_t *out, orc_err *err);

/* src/utils/unpacking/naive.rs:3-25; writes exactly n bytes (caller appends) */
int orc_from_2bit(uint64_t packed, size_t n, uint8_t *out, orc_err *err);

/* src/utils/packing/avx.rs:130-151 (== naive.rs:22-43).  out must hold
 * ceil(len/32) words.  *n_words = words pushed before return (on error: the
 * words of the chunks before the failing one, as Vec state in the reference).
 * len == 0 -> ORC_PANIC (n_chunks-1 underflow, avx.rs:138). */
int orc_encode(const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words,
               orc_err *err);

/* src/utils/unpacking/avx.rs:116-153 with the short-buffer rule of
 * src/utils/unpacking/mod.rs:29-47 (InvalidLength(n_bases)).  Writes n_bases
 * bytes at out. */
int orc_decode(const uint64_t *ebuf, size_t n_words, size_t n_bases,
               uint8_t *out, orc_err *err);

/* src/utils/functions/hamming/scalar.rs:11-48 */
int orc_hdist_scalar(uint64_t u, uint64_t v, size_t len, uint32_t *out,
                     orc_err *err);

/* src/utils/functions/hamming/multi.rs:121-160 */
int orc_hdist(const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
              size_t n_bases, uint32_t *out, orc_err *err);

/* Host loop over as_2bit for `count` k-mers at byte stride `stride`
 * (README.md:52-56 idiom).  First failing k-mer (lowest index) decides the
 * error; err->index = absolute byte offset of the bad base. */
int orc_as_2bit_batch(const uint8_t *kmers, size_t k, size_t stride,
                      size_t count, uint64_t *out, orc_err *err);

/* Composition as_2bit(window) o hdist_scalar(., query, k) over
 * ref.windows(k) (src/lib.rs:170-173 idiom + hamming/scalar.rs:11-48);
 * dist must hold n-k+1 bytes.  n < k -> 0 windows, OK. */
int orc_kmer_hdist_scan(const uint8_t *ref, size_t n, size_t k, uint64_t query,
                        uint8_t *dist, orc_err *err);

/* src/utils/analysis.rs:23-39 on PackedSequence::to_vec(): counts = {A,C,G,T} of the first
 * n_bases bases (decode, then count bytes). */
int orc_base_counts(const uint64_t *words, size_t n_words, size_t n_bases, uint64_t counts[4], orc_err *err);
/* src/utils/analysis.rs:7-16 */
double orc_gc_content(const uint64_t *words, size_t n_words, size_t n_bases);
/* loop of hdist_scalar over word pairs / one query (hamming/scalar.rs:11-48) */
int orc_hdist_pairs(const uint64_t *a, const uint64_t *b, size_t count, size_t len, uint8_t *dist, orc_err *err);

/* src/utils/functions/split.rs:15-99, as written (including the previous-word carry of :84-94
 * and the conditional final push of :97-99).  lbuf and rbuf must each hold n_words + 1 words.
 * idx > slen -> ORC_INDEX_OUT_OF_BOUNDS {index = idx, value = slen}; a buffer that does not reach
 * the split word -> ORC_PANIC (ebuf[chunk_idx], :78).  Parity: the reference's tests (split.rs:108-224)
 * cover one-word right parts and idx % 32 == 0; multi-word right parts with a shift are pinned
 * by this restatement only. */
int orc_split_packed(const uint64_t *ebuf, size_t n_words, size_t slen, size_t idx, uint64_t *lbuf, size_t *n_left,
                     uint64_t *rbuf, size_t *n_right, orc_err *err);

/* Synthetic "nucgen-like" generator shared with the device generator:
 * base i = "ACGT"[(mix(seed, i/32) >> 2*(i%32)) & 3], mix = splitmix64
 * finaliser of seed + (i/32+1)*0x9E3779B97F4A7C15. `first` = absolute index of
 * out[0]. flags bit0: cyclic "ACGT"[i%4] instead (benches/simd_comparison.rs:4-7) */
void orc_nucgen(uint8_t *out, size_t len, uint64_t seed, uint64_t first,
                int flags);

#ifdef __cplusplus
}
#endif
#endif
