// host_word.h -- the library's own HOST code for single words and tiny inputs.
//
// The reference's `as_2bit` / `from_2bit` / `hdist_scalar` are #[inline(always)] functions that
// cost nanoseconds (src/utils/packing/mod.rs:80-110, unpacking/mod.rs:119-147,
// functions/hamming/scalar.rs:11-48), and every size its own benches use (4..1024 bases,
// benches/simd_comparison.rs:19-89) is far below the size at which a kernel launch (~35 us with
// its copies) pays off.  SURVEY 8(b) therefore puts these entry points, and bulk calls below a
// cutoff, on the host.  This is a size dispatch, not a fallback: without a HIP device every call
// at or above the cutoff and every *_dev call still fails with BITNUC_BACKEND_ERROR, bench.py and
// the roofline numbers never come through here, and BITNUC_FORCE_GPU=1 / set_variant("force_gpu", 1)
// sends every call to the kernels (what the GPU parity tests do).
//
// Formulation: 8 bases per step in one 64-bit register (SWAR), not the reference's per-byte match
// and not the oracle's loops:
//   code      = ((b >> 1) ^ (b >> 2)) & 3 on all 8 bytes at once (A/a 0, C/c 1, G/g 2, T/t 3;
//               the values of packing/naive.rs:10-15)
//   validity  = exact zero-byte test of (b & 0xDF) ^ {'A','C','G','T'} (the accepted set is exactly
//               those 8 byte values), first bad byte by count-trailing-zeros
//   compaction= three shift/OR/mask rounds (8 two-bit fields -> 16 contiguous bits)
//   decode    = the inverse spread, then ASCII from the two code bit-planes:
//               0x40 | !(hi&lo) | (hi^lo)<<1 | hi<<2 | (hi&lo)<<4  ->  A C G T
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace bitnuc_host {

constexpr uint64_t k01 = 0x0101010101010101ull, k7F = 0x7F7F7F7F7F7F7F7Full, k80 = 0x8080808080808080ull;

// 0x80 in every byte of v that is zero, 0x00 elsewhere (exact: no borrow between bytes)
static inline uint64_t zero_bytes(uint64_t v) { return ~(((v & k7F) + k7F) | v | k7F); }

static inline uint64_t load_le(const uint8_t *p, unsigned n) { // n <= 8 bytes, missing bytes read as 'A'
    uint64_t x = 0x4141414141414141ull;
    memcpy(&x, p, n);
    return x;
}

// 0x80 in every byte of x that is one of ACGTacgt
static inline uint64_t valid_mask(uint64_t x) {
    const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;
    return zero_bytes(u ^ (k01 * 'A')) | zero_bytes(u ^ (k01 * 'C')) | zero_bytes(u ^ (k01 * 'G')) | zero_bytes(u ^ (k01 * 'T'));
}

// 8 ASCII bytes -> 16 bits of codes (byte 0 in bits 0-1)
static inline uint32_t pack8(uint64_t x) {
    uint64_t c = ((x >> 1) ^ (x >> 2)) & (k01 * 3);
    c = (c | (c >> 6)) & 0x000F000F000F000Full;
    c = (c | (c >> 12)) & 0x000000FF000000FFull;
    c = (c | (c >> 24)) & 0xFFFFull;
    return (uint32_t)c;
}

// 16 bits of codes -> 8 ASCII bytes
static inline uint64_t unpack8(uint32_t w) {
    uint64_t s = w & 0xFFFFu;
    s = (s | (s << 24)) & 0x000000FF000000FFull;
    s = (s | (s << 12)) & 0x000F000F000F000Full;
    s = (s | (s << 6)) & (k01 * 3);
    const uint64_t lo = s & k01, hi = (s >> 1) & k01, t = hi & lo;
    return (k01 * 0x40) | (t ^ k01) | ((lo ^ hi) << 1) | (hi << 2) | (t << 4);
}

// as_2bit of len <= 32 bytes.  Returns -1, or the index of the first invalid byte.
static inline int pack_word(const uint8_t *seq, size_t len, uint64_t *out) {
    uint64_t w = 0;
    for (size_t i = 0; i < len; i += 8) {
        const unsigned n = len - i < 8 ? (unsigned)(len - i) : 8u;
        const uint64_t x = load_le(seq + i, n);
        const uint64_t bad = ~valid_mask(x) & k80; // padding bytes are 'A': valid
        if (bad) return (int)i + (__builtin_ctzll(bad) >> 3);
        uint64_t c = pack8(x);
        if (n < 8) c &= (1ull << (2 * n)) - 1; // padding 'A' is code 0 anyway; keep the invariant explicit
        w |= c << (2 * i);
    }
    *out = w;
    return -1;
}

// from_2bit: exactly n <= 32 bytes at out
static inline void unpack_word(uint64_t w, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i += 8) {
        const uint64_t a = unpack8((uint32_t)(w >> (2 * i)));
        const size_t m = n - i < 8 ? n - i : 8;
        memcpy(out + i, &a, m);
    }
}

// hdist_scalar (hamming/scalar.rs:33-47): number of differing 2-bit fields among the first len <= 32
static inline uint32_t hdist_word(uint64_t u, uint64_t v, size_t len) {
    const uint64_t mask = len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1);
    const uint64_t d = (u ^ v) & mask;
    return (uint32_t)__builtin_popcountll((d | (d >> 1)) & 0x5555555555555555ull);
}

// bulk encode below the cutoff: words of ceil(len/32).  Returns -1 or the index of the first invalid byte
// (the words before the failing 32-base chunk are written: packing/avx.rs:142-143).
static inline long long encode_small(const uint8_t *seq, size_t len, uint64_t *out) {
    const size_t full = len / 32;
    for (size_t w = 0; w < full; ++w) {
        const uint8_t *p = seq + 32 * w;
        uint64_t x[4];
        memcpy(x, p, 32);
        const uint64_t ok = valid_mask(x[0]) & valid_mask(x[1]) & valid_mask(x[2]) & valid_mask(x[3]);
        if (ok != k80) {
            uint64_t dummy;
            return (long long)(32 * w) + pack_word(p, 32, &dummy);
        }
        out[w] = (uint64_t)pack8(x[0]) | ((uint64_t)pack8(x[1]) << 16) | ((uint64_t)pack8(x[2]) << 32) | ((uint64_t)pack8(x[3]) << 48);
    }
    if (len % 32) {
        const int bad = pack_word(seq + 32 * full, len % 32, out + full);
        if (bad >= 0) return (long long)(32 * full) + bad;
    }
    return -1;
}

static inline void decode_small(const uint64_t *words, size_t n_bases, uint8_t *out) {
    const size_t full = n_bases / 32;
    for (size_t w = 0; w < full; ++w) {
        const uint64_t v = words[w];
        const uint64_t a[4] = {unpack8((uint32_t)v), unpack8((uint32_t)(v >> 16)), unpack8((uint32_t)(v >> 32)), unpack8((uint32_t)(v >> 48))};
        memcpy(out + 32 * w, a, 32);
    }
    if (n_bases % 32) unpack_word(words[full], n_bases % 32, out + 32 * full);
}

static inline uint32_t hdist_small(const uint64_t *a, const uint64_t *b, size_t n_bases) {
    uint32_t total = 0; // u32 accumulator like the reference's (multi.rs:130)
    const size_t full = n_bases / 32;
    for (size_t w = 0; w < full; ++w) total += hdist_word(a[w], b[w], 32);
    if (n_bases % 32) total += hdist_word(a[full], b[full], n_bases % 32);
    return total;
}

} // namespace bitnuc_host
