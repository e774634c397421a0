// CPU-only sanitizer harness for the PRODUCT's own host code (sanitizers used to stop at the oracle):
//   * csrc/host_word.h   -- the SWAR single-word / below-cutoff codec (what bitnuc_as_2bit, bitnuc_from_2bit, bitnuc_hdist_scalar
//                           and small host-pointer bitnuc_encode / bitnuc_decode / bitnuc_hdist run), on exact-size heap buffers
//                           so that any over-read / over-write is caught, checked against the oracle;
//   * csrc/host_pool.h   -- the staging pool of the pipelined host-pointer path (mutex + two condition variables, a blocking
//                           and an asynchronous job form), hammered in the call pattern of encode_pipelined / decode_pipelined:
//                           sizes around the 1 MiB serial threshold and the 4096-byte slice edges, 1..9 threads.
// Built twice by tests/test_sanitizers.py: -fsanitize=address,undefined and -fsanitize=thread.  Neither header needs HIP.
// The reference's single-thread contract is src/utils/unpacking/avx.rs:37 (its only static); these threads only move bytes.
#include "../../bitnuc_amd/csrc/host_pool.h"
#include "../../bitnuc_amd/csrc/host_word.h"
#include "../../oracle/bitnuc_oracle.h"

#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

static void host_word_checks() {
    orc_err e;
    // single words: every length 0..32, every byte value at every position (validity + first-invalid-byte index)
    for (size_t len = 0; len <= 32; ++len) {
        uint8_t *s = static_cast<uint8_t *>(malloc(len ? len : 1));
        for (size_t i = 0; i < len; ++i) s[i] = "ACGTacgt"[rnd() & 7];
        uint64_t w = 0, ow = 0;
        CHECK(bitnuc_host::pack_word(s, len, &w) == -1);
        CHECK(orc_as_2bit(s, len, &ow, &e) == ORC_OK && w == ow);
        uint8_t *back = static_cast<uint8_t *>(malloc(len ? len : 1)), *oback = static_cast<uint8_t *>(malloc(len ? len : 1));
        bitnuc_host::unpack_word(w, len, back);
        CHECK(orc_from_2bit(w, len, oback, &e) == ORC_OK && memcmp(back, oback, len) == 0);
        uint32_t d = 0;
        const uint64_t v = w ^ (((uint64_t)rnd() << 32) | rnd());
        CHECK(orc_hdist_scalar(w, v, len, &d, &e) == ORC_OK && d == bitnuc_host::hdist_word(w, v, len));
        for (size_t pos = 0; pos < len; ++pos)
            for (unsigned b = 0; b < 256; ++b) {
                const uint8_t keep = s[pos];
                s[pos] = (uint8_t)b;
                const int st = orc_as_2bit(s, len, &ow, &e);
                const int bad = bitnuc_host::pack_word(s, len, &w);
                if (st == ORC_OK) CHECK(bad == -1 && w == ow);
                else CHECK(bad == (int)e.index && s[bad] == e.byte);
                s[pos] = keep;
            }
        free(s); free(back); free(oback);
    }
    // bulk below the cutoff: every length 1..=1000 round trip (src/utils/mod.rs:113-133) + a few larger, exact-size buffers
    for (size_t n = 1; n <= 1400; n = n < 1000 ? n + 1 : n + 97) {
        uint8_t *s = static_cast<uint8_t *>(malloc(n));
        for (size_t i = 0; i < n; ++i) s[i] = "ACGTacgt"[rnd() & 7];
        const size_t nw = (n + 31) / 32;
        uint64_t *words = static_cast<uint64_t *>(malloc(nw * 8)), *owords = static_cast<uint64_t *>(malloc(nw * 8));
        size_t got = 0;
        CHECK(bitnuc_host::encode_small(s, n, words) == -1);
        CHECK(orc_encode(s, n, owords, &got, &e) == ORC_OK && got == nw && memcmp(words, owords, nw * 8) == 0);
        uint8_t *back = static_cast<uint8_t *>(malloc(n)), *oback = static_cast<uint8_t *>(malloc(n));
        bitnuc_host::decode_small(words, n, back);
        CHECK(orc_decode(words, nw, n, oback, &e) == ORC_OK && memcmp(back, oback, n) == 0);
        uint32_t hd = 0;
        for (size_t i = 0; i < nw; ++i) owords[i] = words[i] ^ (((uint64_t)rnd() << 32) | rnd());
        CHECK(orc_hdist(words, nw, owords, nw, n, &hd, &e) == ORC_OK && hd == bitnuc_host::hdist_small(words, owords, n));
        // first invalid byte in sequence order, and the words before the failing chunk (packing/avx.rs:86-91,142-143)
        const size_t p1 = rnd() % n, p2 = p1 + (rnd() % (n - p1));
        s[p2] = 'X';
        s[p1] = 'N';
        memset(words, 0xEE, nw * 8);
        const long long bad = bitnuc_host::encode_small(s, n, words);
        CHECK(orc_encode(s, n, owords, &got, &e) == ORC_INVALID_BASE);
        CHECK(bad == (long long)e.index && (size_t)bad == p1 && got == p1 / 32 && memcmp(words, owords, got * 8) == 0);
        free(s); free(words); free(owords); free(back); free(oback);
    }
}

// the call pattern of encode_pipelined / decode_pipelined (codec.hip): per chunk a blocking stage-in copy by `pool`, then -- before
// the chunk's "D2H" may overwrite a pinned output -- a wait for the previous hand-back, then the asynchronous hand-back of an
// older chunk by `pool_out`, which overlaps the next chunk's stage-in.  Buffers are exact-size heap blocks.
static void pool_checks(int threads_in, int threads_out) {
    using bitnuc_host::CopyPool;
    CopyPool pool(threads_in), pool_out(threads_out + 1);
    const size_t sizes[] = {0, 1, 4095, 4096, 4097, (1u << 20) - 1, 1u << 20, (1u << 20) + 1, 3 * 4096 * 7 + 5, (5u << 20) + 4097};
    for (size_t n : sizes) {
        const int depth = 3, nchunks = 7;
        uint8_t *src = static_cast<uint8_t *>(malloc(n * nchunks + 1)), *dst = static_cast<uint8_t *>(malloc(n * nchunks + 1));
        for (size_t i = 0; i < n * nchunks; ++i) src[i] = (uint8_t)(i * 131 + (i >> 9));
        memset(dst, 0, n * nchunks + 1);
        uint8_t *stage[3];
        for (int b = 0; b < depth; ++b) stage[b] = static_cast<uint8_t *>(malloc(n + 1));
        for (int use_in = 1; use_in <= threads_in; use_in += (threads_in > 4 ? 3 : 1))
            for (int ci = 0; ci < nchunks + depth - 1; ++ci) {
                const int b = ci % depth;
                if (ci < nchunks) {
                    pool_out.wait(); // the staging buffer b may still be read by the hand-back of chunk ci - depth
                    pool.copy(stage[b], src + (size_t)ci * n, n, use_in);
                }
                if (ci >= depth - 1) {
                    const int j = ci - (depth - 1);
                    pool_out.start(dst + (size_t)j * n, stage[j % depth], n, 1 + (ci % threads_out));
                }
            }
        pool_out.wait();
        CHECK(memcmp(src, dst, n * nchunks) == 0);
        for (int b = 0; b < depth; ++b) free(stage[b]);
        free(src); free(dst);
    }
    // back-to-back jobs of both forms on one pool, and destruction with an asynchronous job outstanding
    uint8_t *a = static_cast<uint8_t *>(malloc(3u << 20)), *b2 = static_cast<uint8_t *>(malloc(3u << 20));
    memset(a, 7, 3u << 20);
    for (int rep = 0; rep < 20; ++rep) {
        pool_out.start(b2, a, (3u << 20) - rep * 4099, 1 + rep % (threads_out));
        if (rep & 1) pool_out.copy(b2, a, 2u << 20, 2 + rep % 3); // copy() waits for the outstanding job first
    }
    pool_out.wait();
    CHECK(b2[12345] == 7);
    {
        CopyPool tmp(4);
        tmp.start(b2, a, 3u << 20, 3);
    } // ~CopyPool waits for the job before joining
    free(a); free(b2);
}

// the call pattern of pipe_run_direct (csrc/host_pipe.h): the poster fills device-buffer stand-in b = ci % depth itself ("H2D"), posts the
// chunk's hand-back to the mover, and before it refills b waits for the ticket of the chunk that used b; an error code travels
// back through an atomic the tasks write; leaving early drains; destruction runs what is still queued.
static void task_thread_checks() {
    using bitnuc_host::TaskThread;
    const size_t sizes[] = {0, 1, 4097, (1u << 20) + 1, (3u << 20) + 5};
    TaskThread w;
    for (size_t n : sizes) {
        const int depth = 3, nchunks = 11;
        uint8_t *src = static_cast<uint8_t *>(malloc(n * nchunks + 1)), *dst = static_cast<uint8_t *>(malloc(n * nchunks + 1));
        for (size_t i = 0; i < n * nchunks; ++i) src[i] = (uint8_t)(i * 29 + (i >> 7));
        memset(dst, 0, n * nchunks + 1);
        uint8_t *devbuf[3];
        for (int b = 0; b < depth; ++b) devbuf[b] = static_cast<uint8_t *>(malloc(n + 1));
        std::atomic<int> rc{0};
        const uint64_t base = w.tickets();
        for (int ci = 0; ci < nchunks; ++ci) {
            const int b = ci % depth;
            if (ci >= depth) w.wait_done(base + (uint64_t)(ci - depth) + 1);
            memcpy(devbuf[b], src + (size_t)ci * n, n);
            uint8_t *from = devbuf[b], *to = dst + (size_t)ci * n;
            const uint64_t t = w.post([=, &rc] { memcpy(to, from, n); if (ci == 7) rc.store(7); });
            CHECK(t == base + (uint64_t)ci + 1);
        }
        w.drain();
        CHECK(rc.load() == 7 && memcmp(src, dst, n * nchunks) == 0);
        for (int b = 0; b < depth; ++b) free(devbuf[b]);
        free(src); free(dst);
    }
    w.drain(); // nothing outstanding: returns at once
    std::atomic<int> ran{0};
    {
        TaskThread tmp;
        for (int i = 0; i < 50; ++i) tmp.post([&ran] { ++ran; });
    } // ~TaskThread runs what is queued, then joins
    CHECK(ran.load() == 50);
    { TaskThread idle; } // never used
}

int main() {
    host_word_checks();
    task_thread_checks();
    for (int t = 1; t <= 9; t += 2) pool_checks(t, t);
    pool_checks(8, 3);
    pool_checks(2, 9);
    CHECK(bitnuc_host::cores_visible() >= 1 && bitnuc_host::cores_usable() >= 1 && bitnuc_host::cores_usable() <= bitnuc_host::cores_visible());
    printf("host sanitizer harness ok\n");
    return 0;
}
