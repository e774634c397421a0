/*
 * bitnuc_oracle.c -- TEST INFRASTRUCTURE ONLY (see bitnuc_oracle.h).
 *
 * Scalar C restatement of the reference's hot path.  Written for auditability
 * against the reference's naive modules, with the x86_64 path's edge
 * behaviours.  Parity: PINNED by tests/golden/golden.json.
 */
#include "bitnuc_oracle.h"

#include <string.h>

static void set_err(orc_err *err, int status, uint8_t byte, uint64_t value,
                    uint64_t index) {
    if (err) {
        err->status = status;
        err->byte = byte;
        err->value = value;
        err->index = index;
    }
}

/* match arms of src/utils/packing/naive.rs:10-16 */
static int base_code(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

int orc_as_2bit(const uint8_t *seq, size_t len, uint64_t *out, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    /* naive.rs:5-7 / avx.rs:77-79: length is checked before any base */
    if (len > 32) {
        set_err(err, ORC_SEQUENCE_TOO_LONG, 0, len, 0);
        return ORC_SEQUENCE_TOO_LONG;
    }
    uint64_t packed = 0;
    for (size_t i = 0; i < len; i++) {
        int c = base_code(seq[i]);
        if (c < 0) {
            /* naive.rs:15 and avx.rs:86-91 both report the FIRST bad byte */
            set_err(err, ORC_INVALID_BASE, seq[i], 0, i);
            return ORC_INVALID_BASE;
        }
        packed |= (uint64_t)c << (2 * i); /* naive.rs:17 */
    }
    *out = packed;
    return ORC_OK;
}

int orc_from_2bit(uint64_t packed, size_t n, uint8_t *out, orc_err *err) {
    static const uint8_t lut[4] = {'A', 'C', 'G', 'T'}; /* unpacking/naive.rs:14-20 */
    set_err(err, ORC_OK, 0, 0, 0);
    if (n > 32) { /* unpacking/naive.rs:8-10, avx.rs:55-57 */
        set_err(err, ORC_INVALID_LENGTH, 0, n, 0);
        return ORC_INVALID_LENGTH;
    }
    for (size_t i = 0; i < n; i++)
        out[i] = lut[(packed >> (2 * i)) & 3]; /* unpacking/naive.rs:12-22 */
    return ORC_OK;
}

int orc_encode(const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words,
               orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    size_t pushed = 0; /* ebuf.clear(), avx.rs:132 */
    if (n_words) *n_words = 0;
    if (len == 0) {
        /* avx.rs:135-138: n_chunks = 0, `0..n_chunks - 1` underflows -> panic */
        set_err(err, ORC_PANIC, 0, 0, 0);
        return ORC_PANIC;
    }
    size_t n_chunks = (len + 31) / 32; /* div_ceil, avx.rs:135 */
    size_t l = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        size_t clen = (c + 1 < n_chunks) ? 32 : len - l; /* avx.rs:138-148 */
        uint64_t bits;
        orc_err e;
        int st = orc_as_2bit(seq + l, clen, &bits, &e);
        if (st != ORC_OK) {
            /* `?` at avx.rs:142/147: ebuf keeps the words pushed so far */
            set_err(err, st, e.byte, e.value, l + e.index);
            if (n_words) *n_words = pushed;
            return st;
        }
        out[pushed++] = bits; /* ebuf.push, avx.rs:143/148 */
        l += clen;
    }
    if (n_words) *n_words = pushed;
    return ORC_OK;
}

int orc_decode(const uint64_t *ebuf, size_t n_words, size_t n_bases,
               uint8_t *out, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    size_t need = (n_bases + 31) / 32;
    if (n_words < need) {
        /* unpacking/mod.rs:40-45 (fallback) -> InvalidLength(n_bases); the AVX2
         * path would panic at avx.rs:146 or stop early at :137 -- the build
         * adopts the fallback's defined behaviour. */
        set_err(err, ORC_INVALID_LENGTH, 0, n_bases, 0);
        return ORC_INVALID_LENGTH;
    }
    size_t full = n_bases / 32; /* avx.rs:134 */
    for (size_t w = 0; w < full; w++)
        orc_from_2bit(ebuf[w], 32, out + 32 * w, NULL); /* avx.rs:137-141 */
    size_t rem = n_bases % 32; /* avx.rs:144 */
    if (rem) orc_from_2bit(ebuf[full], rem, out + 32 * full, NULL); /* :145-150 */
    return ORC_OK;
}

int orc_hdist_scalar(uint64_t u, uint64_t v, size_t len, uint32_t *out,
                     orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    if (len > 32) { /* scalar.rs:13-15 */
        set_err(err, ORC_INVALID_LENGTH, 0, len, 0);
        return ORC_INVALID_LENGTH;
    }
    if (len == 0 || u == v) { /* scalar.rs:18-20 */
        *out = 0;
        return ORC_OK;
    }
    size_t valid_bits = 2 * len;                                   /* :23 */
    uint64_t mask = valid_bits == 64 ? ~0ull : ((1ull << valid_bits) - 1); /* :26-30 */
    uint64_t diff = (u ^ v) & mask;                                /* :33 */
    uint64_t lower = diff & 0x5555555555555555ull & mask;          /* :40 */
    uint64_t upper = (diff & 0xAAAAAAAAAAAAAAAAull & mask) >> 1;   /* :41 */
    *out = (uint32_t)__builtin_popcountll(lower | upper);          /* :44-47 */
    return ORC_OK;
}

int orc_hdist(const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
              size_t n_bases, uint32_t *out, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    size_t expected = (n_bases + 31) / 32; /* multi.rs:124 */
    if (na < expected || nb < expected) {  /* multi.rs:125-127 */
        set_err(err, ORC_INVALID_LENGTH, 0, n_bases, 0);
        return ORC_INVALID_LENGTH;
    }
    size_t full = n_bases / 32; /* multi.rs:129 */
    uint32_t total = 0;         /* u32 accumulator, wraps like release Rust */
    for (size_t w = 0; w < full; w++) { /* multi.rs:147-151 (value == AVX2 path :10-67) */
        uint32_t d;
        orc_hdist_scalar(a[w], b[w], 32, &d, NULL);
        total += d;
    }
    size_t rem = n_bases % 32; /* multi.rs:154-157 */
    if (rem) {
        uint32_t d;
        orc_hdist_scalar(a[full], b[full], rem, &d, NULL);
        total += d;
    }
    *out = total;
    return ORC_OK;
}

int orc_as_2bit_batch(const uint8_t *kmers, size_t k, size_t stride,
                      size_t count, uint64_t *out, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    for (size_t j = 0; j < count; j++) {
        orc_err e;
        int st = orc_as_2bit(kmers + j * stride, k, &out[j], &e);
        if (st != ORC_OK) {
            set_err(err, st, e.byte, e.value, j * stride + e.index);
            return st;
        }
    }
    return ORC_OK;
}

int orc_kmer_hdist_scan(const uint8_t *ref, size_t n, size_t k, uint64_t query,
                        uint8_t *dist, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    if (k > 32) { /* as_2bit's length check fires on the first window */
        set_err(err, ORC_SEQUENCE_TOO_LONG, 0, k, 0);
        return ORC_SEQUENCE_TOO_LONG;
    }
    if (n < k || k == 0) return ORC_OK; /* slice::windows: no windows (k==0 panics in Rust; defined as empty) */
    for (size_t i = 0; i + k <= n; i++) {
        uint64_t w;
        orc_err e;
        int st = orc_as_2bit(ref + i, k, &w, &e);
        if (st != ORC_OK) {
            set_err(err, st, e.byte, e.value, i + e.index);
            return st;
        }
        uint32_t d;
        orc_hdist_scalar(w, query, k, &d, NULL);
        dist[i] = (uint8_t)d;
    }
    return ORC_OK;
}

int orc_base_counts(const uint64_t *words, size_t n_words, size_t n_bases, uint64_t counts[4], orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (n_words < (n_bases + 31) / 32) {
        set_err(err, ORC_INVALID_LENGTH, 0, n_bases, 0);
        return ORC_INVALID_LENGTH;
    }
    for (size_t i = 0; i < n_bases; i++) { /* to_vec() = get(i) per base (sequence.rs:116-135), then analysis.rs:27-36 */
        uint8_t base;
        orc_from_2bit(words[i / 32] >> (2 * (i % 32)), 1, &base, NULL);
        switch (base) {
        case 'A': counts[0]++; break;
        case 'C': counts[1]++; break;
        case 'G': counts[2]++; break;
        case 'T': counts[3]++; break;
        default: break;
        }
    }
    return ORC_OK;
}

double orc_gc_content(const uint64_t *words, size_t n_words, size_t n_bases) {
    uint64_t c[4];
    if (n_bases == 0 || orc_base_counts(words, n_words, n_bases, c, NULL) != ORC_OK) return 0.0; /* analysis.rs:10-11 */
    return ((double)(c[1] + c[2]) / (double)n_bases) * 100.0;                                   /* analysis.rs:13-14 */
}

int orc_hdist_pairs(const uint64_t *a, const uint64_t *b, size_t count, size_t len, uint8_t *dist, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    for (size_t i = 0; i < count; i++) {
        uint32_t d;
        int st = orc_hdist_scalar(a[i], b[i], len, &d, err);
        if (st != ORC_OK) return st;
        dist[i] = (uint8_t)d;
    }
    return ORC_OK;
}

int orc_split_packed(const uint64_t *ebuf, size_t n_words, size_t slen, size_t idx, uint64_t *lbuf, size_t *n_left,
                     uint64_t *rbuf, size_t *n_right, orc_err *err) {
    set_err(err, ORC_OK, 0, 0, 0);
    size_t nl = 0, nr = 0; /* lbuf.clear(); rbuf.clear(), split.rs:31-32 (after the bounds check) */
    if (idx > slen) {      /* split.rs:23-28 */
        set_err(err, ORC_INDEX_OUT_OF_BOUNDS, 0, slen, idx);
        return ORC_INDEX_OUT_OF_BOUNDS;
    }
    *n_left = *n_right = 0;
    if (idx == 0) { /* split.rs:35-39 */
        memcpy(rbuf, ebuf, n_words * 8);
        *n_right = n_words;
        return ORC_OK;
    }
    if (idx == slen) { /* split.rs:40-44 */
        memcpy(lbuf, ebuf, n_words * 8);
        *n_left = n_words;
        return ORC_OK;
    }
    if (n_words == 0) return ORC_OK; /* split.rs:47-49 */
    size_t right_chunks = (slen - idx + 31) / 32; /* split.rs:53-57 */
    size_t chunk_idx = idx / 32;                  /* split.rs:64 */
    unsigned bit_idx = (unsigned)(idx % 32) * 2;  /* split.rs:65 */
    if (chunk_idx > 0 && chunk_idx <= n_words) {  /* split.rs:68-70 */
        memcpy(lbuf, ebuf, chunk_idx * 8);
        nl = chunk_idx;
    }
    if (chunk_idx >= n_words) { /* ebuf[chunk_idx], split.rs:78: index out of bounds -> panic */
        set_err(err, ORC_PANIC, 0, 0, 0);
        return ORC_PANIC;
    }
    uint64_t split_mask = bit_idx == 0 ? 0 : (1ull << bit_idx) - 1; /* split.rs:73-77 */
    lbuf[nl++] = ebuf[chunk_idx] & split_mask;                     /* split.rs:78 */
    unsigned right_shift = bit_idx;                                /* split.rs:81 */
    uint64_t carry = 0;                                            /* split.rs:82 */
    for (size_t i = chunk_idx; i < n_words; i++) {                 /* split.rs:84 */
        uint64_t curr = ebuf[i];
        rbuf[nr++] = carry | (curr >> right_shift);                /* split.rs:86-87 */
        carry = right_shift == 0 ? 0 : curr << (64 - right_shift); /* split.rs:90-94 */
    }
    if (carry != 0 && nr < right_chunks) rbuf[nr++] = carry; /* split.rs:97-99 */
    *n_left = nl;
    *n_right = nr;
    return ORC_OK;
}

static uint64_t mix64(uint64_t z) { /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_nucgen(uint8_t *out, size_t len, uint64_t seed, uint64_t first,
                int flags) {
    static const uint8_t lut[4] = {'A', 'C', 'G', 'T'};
    for (size_t j = 0; j < len; j++) {
        uint64_t i = first + j;
        if (flags & 1) {
            out[j] = lut[i & 3]; /* benches/simd_comparison.rs:4-7 */
        } else {
            uint64_t w = mix64(seed + (i / 32 + 1) * 0x9E3779B97F4A7C15ull);
            out[j] = lut[(w >> (2 * (i % 32))) & 3];
        }
        if (flags & 2) { /* lower-case mix, p = 0.25 (SURVEY 8d parity variant): a second word stream decides the case */
            uint64_t c = mix64((seed ^ 0xC0FFEE5EEDC0DE55ull) + (i / 32 + 1) * 0x9E3779B97F4A7C15ull);
            if (((c >> (2 * (i % 32))) & 3) == 0) out[j] |= 0x20;
        }
    }
}
