// analysis.hip -- the callers just above the codec behind the C ABI: base counts / GC content on packed words
// (src/utils/analysis.rs:7-39), many-pair and one-query hdist_scalar (hamming/scalar.rs:11-48), split_packed
// (src/utils/functions/split.rs:15-99).  Kernels: analysis_device.h.
#include "runtime.h"
#include "analysis_device.h"

using namespace bitnuc_dev;
using namespace bitnuc_rt;

extern "C" {

// ---- analysis on packed words --------------------------------------------------------------
int bitnuc_base_counts_dev(bitnuc_ctx *c, const uint64_t *d_words, size_t n_words, size_t n_bases, uint64_t *d_counts, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (n_words < words_for(n_bases)) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (!d_counts || (n_bases && (!d_words || (reinterpret_cast<uintptr_t>(d_words) & 7)))) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (n_bases == 0) {
        HIPCHK(hipMemsetAsync(d_counts, 0, 4 * sizeof(uint64_t), c->stream));
        return BITNUC_OK;
    }
    const unsigned long long tiles = (n_bases / 32) / (kBlock * 2) + 1;
    const unsigned grid = (unsigned)(tiles < c->reduce_blocks ? tiles : c->reduce_blocks);
    base_counts_kernel<<<grid, kBlock, 0, c->stream>>>(reinterpret_cast<const unsigned long long *>(d_words), n_bases,
                                                       reinterpret_cast<unsigned long long *>(d_counts), c->d_acc, c->d_tickets);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_base_counts(bitnuc_ctx *c, const uint64_t *words, size_t n_words, size_t n_bases, uint64_t counts[4], bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    const size_t need = words_for(n_bases);
    if (n_words < need) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (!counts || (n_bases && !words)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = ensure_scratch(c, 1, need * 8 + 16, err)) return st;
    if (int st = ensure_scratch(c, 2, 64, err)) return st;
    if (need) HIPCHK(hipMemcpyAsync(c->scratch[1], words, need * 8, hipMemcpyHostToDevice, c->stream));
    if (int st = bitnuc_base_counts_dev(c, reinterpret_cast<const uint64_t *>(c->scratch[1]), need, n_bases,
                                        reinterpret_cast<uint64_t *>(c->scratch[2]), err)) return st;
    HIPCHK(hipMemcpyAsync(counts, c->scratch[2], 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BITNUC_OK;
}

static int hdist_words_launch(bitnuc_ctx *c, bool query_mode, const uint64_t *d_a, const uint64_t *d_b, uint64_t query,
                              size_t count, size_t len, uint8_t *d_dist, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (len > 32) return fail(err, BITNUC_INVALID_LENGTH, len); // hamming/scalar.rs:13-15
    if (count == 0) return BITNUC_OK;
    if (!d_a || (!query_mode && !d_b) || !d_dist || (reinterpret_cast<uintptr_t>(d_a) & 7) ||
        (!query_mode && (reinterpret_cast<uintptr_t>(d_b) & 7))) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned long long *a = reinterpret_cast<const unsigned long long *>(d_a), *b = reinterpret_cast<const unsigned long long *>(d_b);
    size_t done = 0;
    if (knobs(c).hdist_words_impl == 1 && aligned16(d_a) && (query_mode || aligned16(d_b)) && (reinterpret_cast<uintptr_t>(d_dist) & 3) == 0 && count >= 256) {
        // whole 256-word wave tiles through the coalesced kernel (one tile per wave: the hardware dispatcher walks them)
        const unsigned long long tiles = count / 256;
        const unsigned grid = grid_for(c, (tiles + kBlock / 64 - 1) / (kBlock / 64));
        if (query_mode) hdist_words_coalesced_kernel<true><<<grid, kBlock, 0, c->stream>>>(a, nullptr, query, tiles, (unsigned)len, d_dist);
        else hdist_words_coalesced_kernel<false><<<grid, kBlock, 0, c->stream>>>(a, b, 0, tiles, (unsigned)len, d_dist);
        HIPCHK(hipGetLastError());
        done = (size_t)tiles * 256;
        if (done == count) return BITNUC_OK;
    }
    const size_t rest = count - done;
    const unsigned grid = grid_for(c, (rest / 4 + kBlock - 1) / kBlock + 1);
    if (query_mode) hdist_words_kernel<true><<<grid, kBlock, 0, c->stream>>>(a + done, nullptr, query, rest, (unsigned)len, d_dist + done);
    else hdist_words_kernel<false><<<grid, kBlock, 0, c->stream>>>(a + done, b + done, 0, rest, (unsigned)len, d_dist + done);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_hdist_pairs_dev(bitnuc_ctx *c, const uint64_t *d_a, const uint64_t *d_b, size_t count, size_t len, uint8_t *d_dist, bitnuc_err *err) {
    return hdist_words_launch(c, false, d_a, d_b, 0, count, len, d_dist, err);
}

int bitnuc_hdist_query_dev(bitnuc_ctx *c, uint64_t query, const uint64_t *d_targets, size_t count, size_t len, uint8_t *d_dist, bitnuc_err *err) {
    return hdist_words_launch(c, true, d_targets, nullptr, query, count, len, d_dist, err);
}

static int hdist_words_host(bitnuc_ctx *c, bool query_mode, const uint64_t *a, const uint64_t *b, uint64_t query, size_t count,
                            size_t len, uint8_t *dist, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (len > 32) return fail(err, BITNUC_INVALID_LENGTH, len);
    if (count == 0) return BITNUC_OK;
    if (!a || (!query_mode && !b) || !dist) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const size_t per = kHostChunk / 8;
    const size_t m0 = count < per ? count : per;
    if (int st = ensure_scratch(c, 0, m0 * 8, err)) return st;
    if (!query_mode) if (int st = ensure_scratch(c, 1, m0 * 8, err)) return st;
    if (int st = ensure_scratch(c, 2, m0 + 16, err)) return st;
    for (size_t i0 = 0; i0 < count; i0 += per) {
        const size_t m = count - i0 < per ? count - i0 : per;
        HIPCHK(hipMemcpyAsync(c->scratch[0], a + i0, m * 8, hipMemcpyHostToDevice, c->stream));
        if (!query_mode) HIPCHK(hipMemcpyAsync(c->scratch[1], b + i0, m * 8, hipMemcpyHostToDevice, c->stream));
        if (int st = hdist_words_launch(c, query_mode, reinterpret_cast<const uint64_t *>(c->scratch[0]),
                                        reinterpret_cast<const uint64_t *>(c->scratch[1]), query, m, len, c->scratch[2], err)) return st;
        HIPCHK(hipMemcpyAsync(dist + i0, c->scratch[2], m, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return BITNUC_OK;
}

int bitnuc_hdist_pairs(bitnuc_ctx *c, const uint64_t *a, const uint64_t *b, size_t count, size_t len, uint8_t *dist, bitnuc_err *err) {
    return hdist_words_host(c, false, a, b, 0, count, len, dist, err);
}

int bitnuc_hdist_query(bitnuc_ctx *c, uint64_t query, const uint64_t *targets, size_t count, size_t len, uint8_t *dist, bitnuc_err *err) {
    return hdist_words_host(c, true, targets, nullptr, query, count, len, dist, err);
}

// ---- split_packed (src/utils/functions/split.rs:15-99) -----------------------------------------
extern "C++" {
namespace {
struct SplitPlan {
    size_t n_left = 0, n_right = 0, c = 0, src_words = 0;
    unsigned s = 0;
    uint64_t lmask = ~0ull, rmask = ~0ull;
    int kind = 0; // 0: kernel, 1: right = ebuf (idx == 0), 2: left = ebuf (idx == slen), 3: nothing to write
};

int split_plan(size_t n_words, size_t slen, size_t idx, int flags, SplitPlan *p, bitnuc_err *err) {
    if (flags != BITNUC_SPLIT_AS_WRITTEN && flags != BITNUC_SPLIT_CANONICAL) return fail(err, BITNUC_UNSUPPORTED);
    if (idx > slen) { // split.rs:23-28
        fail(err, BITNUC_INDEX_OUT_OF_BOUNDS, slen);
        if (err) err->index = idx;
        return BITNUC_INDEX_OUT_OF_BOUNDS;
    }
    const size_t need = words_for(slen);
    p->c = idx / 32;
    p->s = (unsigned)(idx % 32) * 2;
    if (flags == BITNUC_SPLIT_CANONICAL) {
        if (n_words < need) return fail(err, BITNUC_INVALID_LENGTH, slen);
        const size_t rem = (slen - idx) % 32;
        p->n_left = p->c + (p->s != 0);
        p->n_right = (slen - idx) / 32 + (rem != 0);
        p->lmask = p->s ? (1ull << p->s) - 1 : ~0ull;
        p->rmask = rem ? (1ull << (2 * rem)) - 1 : ~0ull;
        p->src_words = need;
        p->kind = p->n_left + p->n_right ? 0 : 3;
        return BITNUC_OK;
    }
    p->src_words = n_words;
    if (idx == 0) { p->kind = 1; p->n_right = n_words; return BITNUC_OK; }    // split.rs:35-39
    if (idx == slen) { p->kind = 2; p->n_left = n_words; return BITNUC_OK; }  // split.rs:40-44
    if (n_words == 0) { p->kind = 3; return BITNUC_OK; }                      // split.rs:47-49
    // ebuf[chunk_idx] (split.rs:78) panics on a buffer that does not reach the split word, and a buffer
    // shorter than ceil(slen/32) makes the output length depend on the data (split.rs:97-99): both are
    // defined here as InvalidLength(slen), the rule decode() uses for short buffers.
    if (n_words < need) return fail(err, BITNUC_INVALID_LENGTH, slen);
    p->n_left = p->c + 1;                               // split.rs:73-78: full chunks, then the masked split chunk
    p->n_right = n_words - p->c;                        // split.rs:84: one word per input word from chunk_idx on
    p->lmask = p->s ? (1ull << p->s) - 1 : 0;           // split.rs:73-77
    p->kind = 0;
    return BITNUC_OK;
}
} // namespace
} // extern "C++"

int bitnuc_split_packed_sizes(size_t n_words, size_t slen, size_t idx, int flags, size_t *n_left, size_t *n_right, bitnuc_err *err) {
    clear_err(err);
    SplitPlan p;
    if (int st = split_plan(n_words, slen, idx, flags, &p, err)) return st;
    if (n_left) *n_left = p.n_left;
    if (n_right) *n_right = p.n_right;
    return BITNUC_OK;
}

int bitnuc_split_packed_dev(bitnuc_ctx *c, const uint64_t *d_ebuf, size_t n_words, size_t slen, size_t idx, int flags,
                            uint64_t *d_lbuf, uint64_t *d_rbuf, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    SplitPlan p;
    if (int st = split_plan(n_words, slen, idx, flags, &p, err)) return st;
    if (p.kind == 3) return BITNUC_OK;
    if (!d_ebuf || (p.n_left && !d_lbuf) || (p.n_right && !d_rbuf) ||
        ((reinterpret_cast<uintptr_t>(d_ebuf) | reinterpret_cast<uintptr_t>(d_lbuf) | reinterpret_cast<uintptr_t>(d_rbuf)) & 7))
        return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (p.kind == 1 || p.kind == 2) { // a plain copy of the whole buffer
        if (n_words) HIPCHK(hipMemcpyAsync(p.kind == 1 ? d_rbuf : d_lbuf, d_ebuf, n_words * 8, hipMemcpyDeviceToDevice, c->stream));
        return BITNUC_OK;
    }
    const unsigned long long items = (unsigned long long)p.n_left + p.n_right;
    // a resident grid-stride grid (4 workgroups per CU) measured 15-20 % faster here than one tile per workgroup
    // (tools/sweep_small_grids.py); the codec-wide grid_mult knob still overrides it
    const unsigned long long tiles = (items + kBlock - 1) / kBlock, cap = (unsigned long long)c->num_cu * 4;
    const unsigned grid = c->grid_mult > 0 ? grid_for(c, tiles) : (unsigned)(tiles < cap ? tiles : cap);
    const unsigned long long *e = reinterpret_cast<const unsigned long long *>(d_ebuf);
    unsigned long long *l = reinterpret_cast<unsigned long long *>(d_lbuf), *r = reinterpret_cast<unsigned long long *>(d_rbuf);
    if (flags == BITNUC_SPLIT_CANONICAL)
        split_packed_kernel<true><<<grid, kBlock, 0, c->stream>>>(e, p.src_words, p.c, p.s, p.n_left, p.n_right, p.lmask, p.rmask, l, r);
    else
        split_packed_kernel<false><<<grid, kBlock, 0, c->stream>>>(e, p.src_words, p.c, p.s, p.n_left, p.n_right, p.lmask, p.rmask, l, r);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_split_packed(bitnuc_ctx *c, const uint64_t *ebuf, size_t n_words, size_t slen, size_t idx, int flags,
                        uint64_t *lbuf, size_t *n_left, uint64_t *rbuf, size_t *n_right, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    SplitPlan p;
    if (int st = split_plan(n_words, slen, idx, flags, &p, err)) return st;
    if (n_left) *n_left = p.n_left;
    if (n_right) *n_right = p.n_right;
    if (p.kind == 3) return BITNUC_OK;
    if ((p.src_words && !ebuf) || (p.n_left && !lbuf) || (p.n_right && !rbuf)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const size_t src = p.kind == 0 ? p.src_words : n_words;
    if (int st = ensure_scratch(c, 1, src * 8 + 16, err)) return st;
    if (int st = ensure_scratch(c, 0, p.n_left * 8 + 16, err)) return st;
    if (int st = ensure_scratch(c, 2, p.n_right * 8 + 16, err)) return st;
    if (src) HIPCHK(hipMemcpyAsync(c->scratch[1], ebuf, src * 8, hipMemcpyHostToDevice, c->stream));
    if (int st = bitnuc_split_packed_dev(c, reinterpret_cast<const uint64_t *>(c->scratch[1]), n_words, slen, idx, flags,
                                         reinterpret_cast<uint64_t *>(c->scratch[0]), reinterpret_cast<uint64_t *>(c->scratch[2]), err)) return st;
    if (p.n_left) HIPCHK(hipMemcpyAsync(lbuf, c->scratch[0], p.n_left * 8, hipMemcpyDeviceToHost, c->stream));
    if (p.n_right) HIPCHK(hipMemcpyAsync(rbuf, c->scratch[2], p.n_right * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BITNUC_OK;
}

} // extern "C"
