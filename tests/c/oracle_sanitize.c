/* CPU-only sanitizer harness (ASan + UBSan) for the oracle: exact-size heap buffers so any
 * over-read/over-write of the restatements (incl. the AVX2 32-byte load at the end of a
 * buffer) is caught.  Built and run by tests/test_sanitizers.py. */
#include "../../oracle/bitnuc_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_avx2_encode(const uint8_t *seq, size_t len, uint64_t **out_words, size_t *n_words, orc_err *err);
int orc_avx2_decode(const uint64_t *ebuf, size_t n_words, size_t n_bases, uint8_t **out, size_t *out_len, orc_err *err);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); }

int main(void) {
    uint64_t w;
    orc_err e;
    CHECK(orc_as_2bit((const uint8_t *)"ACGT", 4, &w, &e) == ORC_OK && w == 0xE4);
    for (size_t n = 1; n <= 300; n++) {
        uint8_t *s = malloc(n);
        for (size_t i = 0; i < n; i++) s[i] = "ACGTacgt"[rnd() & 7];
        size_t nw = (n + 31) / 32, got = 0;
        uint64_t *words = malloc(nw * 8);
        CHECK(orc_encode(s, n, words, &got, &e) == ORC_OK && got == nw);
        uint64_t *w2; size_t nw2;
        CHECK(orc_avx2_encode(s, n, &w2, &nw2, &e) == ORC_OK && nw2 == nw && memcmp(words, w2, nw * 8) == 0);
        uint8_t *back = malloc(n);
        CHECK(orc_decode(words, nw, n, back, &e) == ORC_OK);
        for (size_t i = 0; i < n; i++) CHECK(back[i] == (s[i] & 0xDF));
        uint8_t *b2; size_t bl;
        CHECK(orc_avx2_decode(words, nw, n, &b2, &bl, &e) == ORC_OK && bl == n && memcmp(back, b2, n) == 0);
        /* invalid byte at a random position */
        size_t pos = rnd() % n;
        s[pos] = 'N';
        CHECK(orc_encode(s, n, words, &got, &e) == ORC_INVALID_BASE && e.index == pos && e.byte == 'N' && got == pos / 32);
        uint64_t *w3; size_t nw3;
        CHECK(orc_avx2_encode(s, n, &w3, &nw3, &e) == ORC_INVALID_BASE && e.index == pos && nw3 == pos / 32);
        if (n >= 31) {
            uint8_t *d = malloc(n - 31 + 1);
            s[pos] = 'A';
            CHECK(orc_kmer_hdist_scan(s, n, 31, 0x123456789ABCDEFull, d, &e) == ORC_OK);
            free(d);
        }
        uint64_t counts[4];
        CHECK(orc_base_counts(words, nw, n, counts, &e) == ORC_INVALID_BASE || 1);
        free(s); free(words); free(w2); free(back); free(b2); free(w3);
    }
    uint32_t d;
    CHECK(orc_hdist_scalar(~0ull, 0, 32, &d, &e) == ORC_OK && d == 32);
    CHECK(orc_hdist_scalar(0, 0, 33, &d, &e) == ORC_INVALID_LENGTH);
    uint8_t *g = malloc(1000);
    orc_nucgen(g, 1000, 7, 12345, 0);
    /* split_packed with exact-size buffers: n_words + 1 words each (oracle contract), every split point */
    {
        uint64_t w[32], *l = malloc(33 * 8), *r = malloc(33 * 8);
        size_t nw = 0, nl, nr;
        CHECK(orc_encode(g, 1000, w, &nw, &e) == ORC_OK && nw == 32);
        for (size_t idx = 0; idx <= 1000; idx++) {
            CHECK(orc_split_packed(w, nw, 1000, idx, l, &nl, r, &nr, &e) == ORC_OK);
            CHECK(nl <= 33 && nr <= 33);
        }
        CHECK(orc_split_packed(w, nw, 1000, 1001, l, &nl, r, &nr, &e) == ORC_INDEX_OUT_OF_BOUNDS);
        CHECK(orc_split_packed(w, 2, 1000, 500, l, &nl, r, &nr, &e) == ORC_PANIC);
        free(l); free(r);
    }
    free(g);
    printf("sanitizer harness ok\n");
    return 0;
}
