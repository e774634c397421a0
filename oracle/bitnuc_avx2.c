/*
 * bitnuc_avx2.c -- TEST INFRASTRUCTURE ONLY (see bitnuc_oracle.h).
 *
 * "Reference-algorithm restatement" of the reference's x86_64 AVX2 bulk
 * encode/decode AS WRITTEN (same Intel intrinsics, same scalar pre-validation,
 * same scalar shift-OR compaction, same per-word Vec::push / extend_from_slice),
 * used only as (1) a second oracle cross-checked against bitnuc_oracle.c and
 * (2) the timed cpu_baseline ("kind": "port") in bench.py.  It is deliberately
 * NOT improved (no pext / movemask / multiply compaction): it stands in for the
 * reference binary, which cannot be built here (no rustc in the image).
 *
 * Follows: src/utils/packing/avx.rs:13-151, src/utils/unpacking/avx.rs:25-33,116-153.
 */
#include "bitnuc_oracle.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

/* ---- minimal Vec<T> with Rust's amortised-doubling push -------------------- */
typedef struct { uint64_t *ptr; size_t len, cap; } vec_u64;
typedef struct { uint8_t *ptr; size_t len, cap; } vec_u8;

static inline void vec_u64_push(vec_u64 *v, uint64_t x) { /* Vec::push: capacity check per word */
    if (v->len == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 4;
        v->ptr = (uint64_t *)realloc(v->ptr, nc * sizeof(uint64_t));
        v->cap = nc;
    }
    v->ptr[v->len++] = x;
}
static inline void vec_u8_reserve(vec_u8 *v, size_t add) {
    if (v->cap - v->len < add) {
        size_t nc = v->len + add;
        if (nc < v->cap * 2) nc = v->cap * 2;
        v->ptr = (uint8_t *)realloc(v->ptr, nc);
        v->cap = nc;
    }
}
static inline void vec_u8_extend(vec_u8 *v, const uint8_t *src, size_t n) { /* extend_from_slice */
    vec_u8_reserve(v, n);
    memcpy(v->ptr + v->len, src, n);
    v->len += n;
}

/* ---- packing/avx.rs:33-74 -------------------------------------------------- */
static inline __m256i dual_mask(__m256i chunk, char upper, char lower) { /* :33-39 */
    return _mm256_or_si256(_mm256_cmpeq_epi8(chunk, _mm256_set1_epi8(upper)),
                           _mm256_cmpeq_epi8(chunk, _mm256_set1_epi8(lower)));
}
static inline __m256i process_simd_chunk(__m256i chunk) { /* :41-74 */
    const __m256i ones = _mm256_set1_epi8(1), twos = _mm256_set1_epi8(2),
                  threes = _mm256_set1_epi8(3);
    __m256i c = dual_mask(chunk, 'C', 'c'), g = dual_mask(chunk, 'G', 'g'),
            t = dual_mask(chunk, 'T', 't');
    __m256i r = _mm256_setzero_si256();
    r = _mm256_or_si256(_mm256_and_si256(c, ones), _mm256_andnot_si256(c, r));
    r = _mm256_or_si256(_mm256_and_si256(g, twos), _mm256_andnot_si256(g, r));
    r = _mm256_or_si256(_mm256_and_si256(t, threes), _mm256_andnot_si256(t, r));
    return r;
}

static inline int is_valid_base(uint8_t b) { /* matches!() at avx.rs:88 */
    switch (b) {
    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return 1;
    default: return 0;
    }
}

/* packing/avx.rs:76-128.  `avail` = bytes readable from seq (>= len); the
 * reference's 256-bit load at :103 reads 32 bytes from a slice that may hold
 * only 16 (UB it gets away with); here the same load is used only when 32 bytes
 * are really readable, else a 128-bit load of the 16 bytes actually consumed. */
static int avx2_as_2bit(const uint8_t *seq, size_t len, size_t avail,
                        uint64_t *out, orc_err *err, size_t base_index) {
    if (len > 32) { /* :77-79 */
        if (err) { err->status = ORC_SEQUENCE_TOO_LONG; err->value = len; }
        return ORC_SEQUENCE_TOO_LONG;
    }
    if (len < 16) { /* :82-84 -> naive::as_2bit */
        orc_err e;
        int st = orc_as_2bit(seq, len, out, &e);
        if (st != ORC_OK && err) { *err = e; err->index += base_index; }
        return st;
    }
    for (size_t i = 0; i < len; i++) { /* :86-91 scalar pre-validation */
        if (!is_valid_base(seq[i])) {
            if (err) { err->status = ORC_INVALID_BASE; err->byte = seq[i]; err->value = 0; err->index = base_index + i; }
            return ORC_INVALID_BASE;
        }
    }
    uint64_t packed = 0;
    size_t simd_len = len - (len % 16); /* :96 */
    for (size_t ci = 0; ci < simd_len; ci += 16) { /* :101-112 */
        __m256i chunk;
        if (ci + 32 <= avail)
            chunk = _mm256_loadu_si256((const __m256i *)(seq + ci)); /* :103 */
        else
            chunk = _mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)(seq + ci)));
        __m256i result = process_simd_chunk(chunk); /* :104 */
        uint8_t temp[32];
        _mm256_storeu_si256((__m256i *)temp, result); /* :106-107 */
        for (size_t i = 0; i < 16; i++)               /* :109-111 scalar shift-OR */
            packed |= (uint64_t)temp[i] << ((ci + i) * 2);
    }
    for (size_t i = simd_len; i < len; i++) { /* :115-124 */
        uint64_t bits;
        switch (seq[i]) {
        case 'A': case 'a': bits = 0; break;
        case 'C': case 'c': bits = 1; break;
        case 'G': case 'g': bits = 2; break;
        default: bits = 3; break;
        }
        packed |= bits << (i * 2);
    }
    *out = packed;
    return ORC_OK;
}

/* packing/avx.rs:130-151.  Returns a malloc'd buffer in *out_words (caller frees). */
int orc_avx2_encode(const uint8_t *seq, size_t len, uint64_t **out_words,
                    size_t *n_words, orc_err *err) {
    vec_u64 ebuf = {0, 0, 0}; /* ebuf.clear() on a fresh Vec, :132 */
    if (err) memset(err, 0, sizeof *err);
    *out_words = NULL;
    *n_words = 0;
    if (len == 0) { if (err) err->status = ORC_PANIC; return ORC_PANIC; } /* :138 underflow */
    size_t n_chunks = (len + 31) / 32; /* :135 */
    size_t l = 0;
    for (size_t c = 0; c + 1 < n_chunks; c++) { /* :138-145 */
        uint64_t bits;
        int st = avx2_as_2bit(seq + l, 32, len - l, &bits, err, l);
        if (st != ORC_OK) { *out_words = ebuf.ptr; *n_words = ebuf.len; return st; }
        vec_u64_push(&ebuf, bits); /* :143 */
        l += 32;
    }
    uint64_t bits; /* :147-148 */
    int st = avx2_as_2bit(seq + l, len - l, len - l, &bits, err, l);
    if (st != ORC_OK) { *out_words = ebuf.ptr; *n_words = ebuf.len; return st; }
    vec_u64_push(&ebuf, bits);
    *out_words = ebuf.ptr;
    *n_words = ebuf.len;
    return ORC_OK;
}

/* unpacking/avx.rs:25-33 */
static inline __m256i unpack_32_bases(uint64_t packed, __m256i lookup) {
    uint8_t indices[32];
    for (int i = 0; i < 32; i++) indices[i] = (uint8_t)((packed >> (i * 2)) & 3); /* :27-30 */
    __m256i idx = _mm256_loadu_si256((const __m256i *)indices);                   /* :31 */
    return _mm256_shuffle_epi8(lookup, idx);                                      /* :32 */
}

/* unpacking/avx.rs:116-153.  Returns a malloc'd buffer in *out (caller frees). */
int orc_avx2_decode(const uint64_t *ebuf, size_t n_words, size_t n_bases,
                    uint8_t **out, size_t *out_len, orc_err *err) {
    vec_u8 seq = {0, 0, 0};
    if (err) memset(err, 0, sizeof *err);
    *out = NULL;
    *out_len = 0;
    if (n_words < (n_bases + 31) / 32) { /* adopted short-buffer rule, mod.rs:40-45 */
        if (err) { err->status = ORC_INVALID_LENGTH; err->value = n_bases; }
        return ORC_INVALID_LENGTH;
    }
    vec_u8_reserve(&seq, n_bases); /* :122 */
    const __m256i lookup = _mm256_setr_epi8( /* :125-131 */
        'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T',
        'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T');
    size_t full = n_bases / 32; /* :134 */
    uint8_t temp[32];
    for (size_t w = 0; w < full; w++) { /* :137-141 */
        __m256i r = unpack_32_bases(ebuf[w], lookup);
        _mm256_storeu_si256((__m256i *)temp, r);
        vec_u8_extend(&seq, temp, 32);
    }
    size_t rem = n_bases % 32; /* :144-150 */
    if (rem) {
        __m256i r = unpack_32_bases(ebuf[full], lookup);
        _mm256_storeu_si256((__m256i *)temp, r);
        vec_u8_extend(&seq, temp, rem);
    }
    *out = seq.ptr;
    *out_len = seq.len;
    return ORC_OK;
}

void orc_free(void *p) { free(p); }

/* ---- timing harness for bench.py's cpu_baseline -----------------------------
 * Encodes then decodes `len` bases `reps` times on the calling thread;
 * returns seconds for encode and decode separately (best-of not taken here;
 * bench.py takes the median).  Checks the round trip on the last rep. */
#include <time.h>
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int orc_avx2_time_roundtrip(const uint8_t *seq, size_t len, double *enc_s,
                            double *dec_s) {
    uint64_t *words = NULL; size_t nw = 0; orc_err e;
    double t0 = now_s();
    int st = orc_avx2_encode(seq, len, &words, &nw, &e);
    double t1 = now_s();
    if (st != ORC_OK) { free(words); return st; }
    uint8_t *back = NULL; size_t bl = 0;
    double t2 = now_s();
    st = orc_avx2_decode(words, nw, len, &back, &bl, &e);
    double t3 = now_s();
    int ok = (st == ORC_OK) && bl == len;
    /* decode emits uppercase only: compare case-insensitively is not needed for
     * the uppercase synthetic inputs the bench uses */
    if (ok && memcmp(back, seq, len) != 0) ok = 0;
    free(words); free(back);
    *enc_s = t1 - t0; *dec_s = t3 - t2;
    return ok ? ORC_OK : ORC_UNSUPPORTED;
}
