// batch.hip -- batches of independent sequences (reads, contigs) behind the C ABI (include/bitnuc_hip.h): ragged batches with
// offset tables, with a layout plan, and fixed-length reads.  The reference's user loops encode / decode over many short
// sequences, each call padding its own last word (src/utils/mod.rs:22-25,60-62, packing/avx.rs:147-148).  Kernels: batch_device.h.
#include "runtime.h"
#include "batch_device.h"
#include "host_pipe.h"

using namespace bitnuc_dev;
using namespace bitnuc_rt;

extern "C" {

#ifdef BITNUC_SWEEP_VARIANTS // round 2's table-driven form (tile records by a search pre-kernel): evidence build only
// ---- ragged batches ---------------------------------------------------------------------------
// rec[b] = {owner, first byte} of every 64-word wave tile, into context scratch (enqueued on the stream)
static int batch_owners(bitnuc_ctx *c, const uint64_t *d_offsets, const uint64_t *d_word_offsets, size_t count, size_t total_words,
                        const TileRec **recs, bitnuc_err *err) {
    const size_t ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    if (int st = ensure_scratch(c, 3, ntiles * sizeof(TileRec), err)) return st;
    TileRec *o = reinterpret_cast<TileRec *>(c->scratch[3]);
    const unsigned og = (unsigned)((ntiles + kBlock - 1) / kBlock);
    const unsigned long long *po = reinterpret_cast<const unsigned long long *>(d_offsets), *pw = reinterpret_cast<const unsigned long long *>(d_word_offsets);
    // count / total_words as a 0.64 fixed-point number (count <= total_words unless sequences are empty; saturate then)
    const unsigned long long ratio64 = count >= total_words ? ~0ull : (unsigned long long)((((unsigned __int128)count) << 64) / total_words);
    // measured (profiles/r01_ab_owner_estimate.txt): the multiply-high guess wins by 9 us of 17 for long sequences, the
    // 128-bit division by 6 of 28 for read-sized ones (same loads either way; the slower arithmetic spreads them out)
    const int est_mode = knobs(c).owner_est < 3 ? knobs(c).owner_est : (total_words >= 16 * (unsigned long long)count ? 2 : 0);
    if (est_mode == 0) block_owner_kernel<0><<<og, kBlock, 0, c->stream>>>(po, pw, count, total_words, ntiles, ratio64, o);
    else if (est_mode == 1) block_owner_kernel<1><<<og, kBlock, 0, c->stream>>>(po, pw, count, total_words, ntiles, ratio64, o);
    else block_owner_kernel<2><<<og, kBlock, 0, c->stream>>>(po, pw, count, total_words, ntiles, ratio64, o);
    HIPCHK(hipGetLastError());
    *recs = o;
    return BITNUC_OK;
}
#endif

static int check_offsets(const uint64_t *offsets, size_t count, bitnuc_err *err) {
    for (size_t i = 0; i < count; ++i)
        if (offsets[i + 1] < offsets[i]) { // argument check only (not codec arithmetic)
            if (err) { memset(err, 0, sizeof *err); err->status = BITNUC_INVALID_RANGE; err->value = i + 1; }
            return BITNUC_INVALID_RANGE;
        }
    return BITNUC_OK;
}

int bitnuc_batch_word_offsets_dev(bitnuc_ctx *c, const uint64_t *d_offsets, size_t count, uint64_t *d_word_offsets, size_t *total_words, bitnuc_err *err) {
    clear_err(err);
    if (total_words) *total_words = 0;
    if (int st = check_ctx(c, err)) return st;
    if (!d_word_offsets || (count && !d_offsets)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (count == 0) {
        HIPCHK(hipMemsetAsync(d_word_offsets, 0, sizeof(uint64_t), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return BITNUC_OK;
    }
    const size_t per_block = (size_t)kScanTile;
    const size_t nblocks = (count + per_block - 1) / per_block;
    if (int st = ensure_scratch(c, 3, (nblocks + 3) * sizeof(uint64_t), err)) return st;
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(c->scratch[3]);
    const unsigned long long *off = reinterpret_cast<const unsigned long long *>(d_offsets);
    unsigned long long *wo = reinterpret_cast<unsigned long long *>(d_word_offsets);
    word_offsets_block_sums<<<(unsigned)nblocks, kBlock, 0, c->stream>>>(off, count, sums);
    word_offsets_scan_sums<<<1, kBlock, 0, c->stream>>>(sums, nblocks, off, count);
    word_offsets_finish<<<(unsigned)nblocks, kBlock, 0, c->stream>>>(off, count, sums, wo);
    HIPCHK(hipGetLastError());
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_word_offsets + count, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (total_words) *total_words = (size_t)total;
    return BITNUC_OK;
}

// The table-driven forms.  Everything the layout plan depends on is in the caller's two tables and `total_words`, so ONE
// asynchronous launch (plan_emit_kernel: no scan, no host synchronisation, no memset) emits the plan's pad bytes, tile bases and
// buffer bounds into context scratch and the plan kernels do the work.  The scratch plan lives until the next table-driven call
// on this context (stream order).  Scratch grows on the first call / a larger batch (an allocation, which waits for the stream).
static int emit_scratch_plan(bitnuc_ctx *c, const uint64_t *d_offsets, const uint64_t *d_word_offsets, size_t count, size_t total_words,
                             uint8_t **P, unsigned long long **tile_base, unsigned long long **bounds, bitnuc_err *err) {
    const size_t ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    if (int st = ensure_scratch(c, 6, total_words + 2 + kBatchTile, err)) return st;
    if (int st = ensure_scratch(c, 7, (ntiles + 1 + 2) * sizeof(unsigned long long), err)) return st;
    *P = c->scratch[6];
    *tile_base = reinterpret_cast<unsigned long long *>(c->scratch[7]);
    *bounds = *tile_base + ntiles + 1;
    const unsigned grid = (unsigned)((count + kScanTile - 1) / kScanTile);
    // word_offsets[i] is by definition the prefix sum of ceil(len / 32): the kernel reads one entry per workgroup and scans the rest
    plan_emit_kernel<true, false><<<grid, kBlock, 0, c->stream>>>(reinterpret_cast<const unsigned long long *>(d_offsets),
                                                                   reinterpret_cast<const unsigned long long *>(d_word_offsets), count, *P, *tile_base, *bounds, nullptr);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

static int launch_plan_encode(bitnuc_ctx *c, const unsigned long long *d_base, const uint8_t *d_P, size_t total_words, const unsigned long long *d_bounds,
                              const uint8_t *d_seq, uint64_t *d_out, bitnuc_err *err) {
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    const int threads = knobs(c).plan_enc_block; // 64, 128 or 256 threads: a wave owns a tile, so any number of waves per workgroup works
    const size_t per_block = (size_t)kBatchTile * (size_t)(threads / 64);
    const unsigned long long blocks = (total_words + per_block - 1) / per_block;
    unsigned long long *o = reinterpret_cast<unsigned long long *>(d_out);
    const unsigned long long U = (unsigned long long)knobs(c).plan_enc_tiles;
    const unsigned grid = grid_for(c, (blocks + U - 1) / U, threads);
#define PLAN_ENC(UU) encode_batch_plan_kernel<UU><<<grid, threads, 0, c->stream>>>(d_seq, d_base, d_P, total_words, d_bounds, o, slot)
    if constexpr (kEvidenceBuild) {
#define PLAN_ENC_ABL(A) encode_batch_plan_kernel<1, A><<<grid, threads, 0, c->stream>>>(d_seq, d_base, d_P, total_words, d_bounds, o, slot)
        if (U == 2) PLAN_ENC(2);
        else if (U == 4) PLAN_ENC(4);
        else switch (knobs(c).plan_enc_abl) { // timing-only ablations, right only for 32-base reads (tools/ab_plan_enc_ablate.py)
        case 1: PLAN_ENC_ABL(1); break;
        case 2: PLAN_ENC_ABL(2); break;
        case 4: PLAN_ENC_ABL(4); break;
        case 6: PLAN_ENC_ABL(6); break;
        case 7: PLAN_ENC_ABL(7); break;
        default: PLAN_ENC(1); break;
        }
#undef PLAN_ENC_ABL
    } else {
        PLAN_ENC(1);
    }
#undef PLAN_ENC
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

static int launch_plan_decode(bitnuc_ctx *c, const unsigned long long *d_base, const uint8_t *d_P, size_t total_words, const uint64_t *d_words,
                              uint8_t *d_out, bitnuc_err *err) {
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(d_words);
    const int tiles_per_wave = knobs(c).plan_tiles;
    const size_t per_block2 = (size_t)kBatchTile * kBatchWaves * (size_t)tiles_per_wave;
    const unsigned grid2 = grid_for(c, (total_words + per_block2 - 1) / per_block2);
#ifdef BITNUC_SWEEP_VARIANTS
    {
        if (knobs(c).plan_dec_lines) { // line-owning tiles (evidence/batch_evidence.h; lost its A/B, profiles/r04_ab_plan_lines.txt): one tile per wave trip, the shipped store policy
            const size_t per_block1 = (size_t)kBatchTile * kBatchWaves;
            if (knobs(c).plan_dec_lines == 2) decode_batch_plan_lines_kernel<2, 16><<<grid_for(c, (total_words + per_block1 - 1) / per_block1), kBlock, 0, c->stream>>>(w, d_base, d_P, total_words, d_out);
            else decode_batch_plan_lines_kernel<2, 128><<<grid_for(c, (total_words + per_block1 - 1) / per_block1), kBlock, 0, c->stream>>>(w, d_base, d_P, total_words, d_out);
            HIPCHK(hipGetLastError());
            return BITNUC_OK;
        }
    }
#endif
#define PLAN_DEC(POL, U) decode_batch_plan_kernel<POL, U><<<grid2, kBlock, 0, c->stream>>>(w, d_base, d_P, total_words, d_out)
#define PLAN_DEC_U(POL) do { if (tiles_per_wave == 1) PLAN_DEC(POL, 1); else if (tiles_per_wave == 2) PLAN_DEC(POL, 2); else PLAN_DEC(POL, 4); } while (0)
    if constexpr (kEvidenceBuild) {
        if (knobs(c).plan_store == 0) PLAN_DEC_U(0);
        else if (knobs(c).plan_store == 1) PLAN_DEC_U(1);
        else PLAN_DEC_U(2);
    } else {
        PLAN_DEC(2, 1);
    }
#undef PLAN_DEC_U
#undef PLAN_DEC
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_encode_batch_dev(bitnuc_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, const uint64_t *d_word_offsets, size_t count, size_t total_words, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || total_words == 0) return BITNUC_OK;
    if (!d_seq || !d_offsets || !d_word_offsets || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (knobs(c).batch_tables_impl == 1 || !kEvidenceBuild) {
        uint8_t *P;
        unsigned long long *tile_base, *bounds;
        if (int st = emit_scratch_plan(c, d_offsets, d_word_offsets, count, total_words, &P, &tile_base, &bounds, err)) return st;
        return launch_plan_encode(c, tile_base, P, total_words, bounds, d_seq, d_out, err);
    }
#ifdef BITNUC_SWEEP_VARIANTS
    { // round 2's form: tile records by a search pre-kernel + O(1) window lookup inside the main kernel
        const unsigned long long *po = reinterpret_cast<const unsigned long long *>(d_offsets), *pw = reinterpret_cast<const unsigned long long *>(d_word_offsets);
        unsigned long long *o = reinterpret_cast<unsigned long long *>(d_out);
        const TileRec *recs;
        if (int st = batch_owners(c, d_offsets, d_word_offsets, count, total_words, &recs, err)) return st;
        unsigned long long *slot;
        if (int st = take_slot(c, 0, &slot, err)) return st;
        const size_t per_block = (size_t)kBatchTile * kBatchWaves;
        const unsigned grid = grid_for(c, (total_words + per_block - 1) / per_block);
        switch (knobs(c).batch_abl) { // timing-only ablations (tools/ab_batch_ablate.py): anything but 0 produces wrong words
#define ABL_CASE(A) case A: encode_batch2_kernel<A><<<grid, kBlock, 0, c->stream>>>(d_seq, po, pw, count, total_words, recs, o, slot); break;
        ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(8) ABL_CASE(9) ABL_CASE(11)
#undef ABL_CASE
        default: encode_batch2_kernel<0><<<grid, kBlock, 0, c->stream>>>(d_seq, po, pw, count, total_words, recs, o, slot);
        }
        HIPCHK(hipGetLastError());
    }
#endif
    return BITNUC_OK;
}

int bitnuc_decode_batch_dev(bitnuc_ctx *c, const uint64_t *d_words, const uint64_t *d_word_offsets, const uint64_t *d_offsets, size_t count, size_t total_words, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || total_words == 0) return BITNUC_OK;
    if (!d_words || !d_offsets || !d_word_offsets || !d_out || (reinterpret_cast<uintptr_t>(d_words) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (knobs(c).batch_tables_impl == 1 || !kEvidenceBuild) {
        uint8_t *P;
        unsigned long long *tile_base, *bounds;
        if (int st = emit_scratch_plan(c, d_offsets, d_word_offsets, count, total_words, &P, &tile_base, &bounds, err)) return st;
        return launch_plan_decode(c, tile_base, P, total_words, d_words, d_out, err);
    }
#ifdef BITNUC_SWEEP_VARIANTS
    {
        const unsigned long long *po = reinterpret_cast<const unsigned long long *>(d_offsets), *pw = reinterpret_cast<const unsigned long long *>(d_word_offsets);
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(d_words);
        const TileRec *recs;
        if (int st = batch_owners(c, d_offsets, d_word_offsets, count, total_words, &recs, err)) return st;
        const size_t per_block = (size_t)kBatchTile * kBatchWaves;
        const unsigned grid = grid_for(c, (total_words + per_block - 1) / per_block);
        switch (knobs(c).batch_abl) {
#define ABL_CASE(A) case A: decode_batch2_kernel<A><<<grid, kBlock, 0, c->stream>>>(w, pw, po, count, total_words, recs, d_out); break;
        ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(9) ABL_CASE(11) ABL_CASE(15)
#undef ABL_CASE
        default: decode_batch2_kernel<0><<<grid, kBlock, 0, c->stream>>>(w, pw, po, count, total_words, recs, d_out);
        }
        HIPCHK(hipGetLastError());
    }
#endif
    return BITNUC_OK;
}

int bitnuc_encode_batch(bitnuc_ctx *c, const uint8_t *seq, const uint64_t *offsets, size_t count, uint64_t *out, size_t out_cap_words, uint64_t *word_offsets, size_t *n_words, bitnuc_err *err) {
    clear_err(err);
    if (n_words) *n_words = 0;
    if (int st = check_ctx(c, err)) return st;
    if (count == 0) { if (word_offsets) word_offsets[0] = 0; return BITNUC_OK; }
    if (!offsets || !word_offsets) return fail(err, BITNUC_UNSUPPORTED);
    if (int st = check_offsets(offsets, count, err)) return st;
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    const uint64_t b0 = offsets[0], nbytes = offsets[count] - b0;
    if (int st = ensure_scratch(c, 0, nbytes + 16, err)) return st;
    if (int st = ensure_scratch(c, 4, (count + 1) * 8, err)) return st;
    uint64_t *d_off = reinterpret_cast<uint64_t *>(c->scratch[4]);
    if (nbytes && !seq) return fail(err, BITNUC_UNSUPPORTED);
    if (nbytes) HIPCHK(hipMemcpyAsync(c->scratch[0], seq + b0, nbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_off, offsets, (count + 1) * 8, hipMemcpyHostToDevice, c->stream));
    size_t total = 0;
    const uint64_t *d_wo;
    if (knobs(c).batch_host_plan) { // the layout plan: word offsets + tile bases + pad bytes in one go (the context keeps one for its host calls)
        if (!c->host_plan) if (int st = bitnuc_batch_plan_create(c, &c->host_plan, err)) return st;
        if (int st = bitnuc_batch_plan_build_dev(c, c->host_plan, d_off, count, &total, err)) return st;
        d_wo = bitnuc_batch_plan_word_offsets_dev(c->host_plan);
    } else {
        if (int st = ensure_scratch(c, 5, (count + 1) * 8, err)) return st;
        if (int st = bitnuc_batch_word_offsets_dev(c, d_off, count, reinterpret_cast<uint64_t *>(c->scratch[5]), &total, err)) return st;
        d_wo = reinterpret_cast<const uint64_t *>(c->scratch[5]);
    }
    HIPCHK(hipMemcpyAsync(word_offsets, d_wo, (count + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (total > out_cap_words || (total && !out)) { HIPCHK(hipStreamSynchronize(c->stream)); return fail(err, BITNUC_INVALID_LENGTH, total); }
    if (int st = ensure_scratch(c, 1, total * 8 + 16, err)) return st;
    // the kernels index the sequence buffer with the caller's offsets: rebase the device pointer
    const uint8_t *d_seq = c->scratch[0] - b0;
    if (total) {
        if (knobs(c).batch_host_plan) { if (int st = bitnuc_encode_batch_plan_dev(c, c->host_plan, d_seq, reinterpret_cast<uint64_t *>(c->scratch[1]), err)) return st; }
        else if (int st = bitnuc_encode_batch_dev(c, d_seq, d_off, d_wo, count, total, reinterpret_cast<uint64_t *>(c->scratch[1]), err)) return st;
        HIPCHK(hipMemcpyAsync(out, c->scratch[1], total * 8, hipMemcpyDeviceToHost, c->stream));
    }
    bitnuc_err e;
    int st = drain(c, &e);
    if (st != BITNUC_OK) { if (err) *err = e; return st; }
    if (n_words) *n_words = total;
    return BITNUC_OK;
}

int bitnuc_decode_batch(bitnuc_ctx *c, const uint64_t *words, const uint64_t *word_offsets, const uint64_t *offsets, size_t count, uint8_t *out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0) return BITNUC_OK;
    if (!offsets || !word_offsets) return fail(err, BITNUC_UNSUPPORTED);
    if (int st = check_offsets(offsets, count, err)) return st;
    // the kernels index both buffers through these tables: a table that does not match the offsets
    // (word_offsets[i+1] - word_offsets[i] == ceil(len_i / 32), starting at 0) is refused here rather
    // than turned into an out-of-bounds device access (argument check only, not codec arithmetic)
    if (word_offsets[0] != 0) return fail(err, BITNUC_INVALID_RANGE, 0);
    for (size_t i = 0; i < count; ++i)
        if (word_offsets[i + 1] - word_offsets[i] != words_for((size_t)(offsets[i + 1] - offsets[i])) || word_offsets[i + 1] < word_offsets[i])
            return fail(err, BITNUC_INVALID_RANGE, i + 1);
    const uint64_t b0 = offsets[0], nbytes = offsets[count] - b0, total = word_offsets[count];
    if (total == 0) return BITNUC_OK;
    if (!words || !out) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = ensure_scratch(c, 0, nbytes + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, total * 8 + 16, err)) return st;
    if (int st = ensure_scratch(c, 4, (count + 1) * 8, err)) return st;
    if (int st = ensure_scratch(c, 5, (count + 1) * 8, err)) return st;
    uint64_t *d_off = reinterpret_cast<uint64_t *>(c->scratch[4]), *d_wo = reinterpret_cast<uint64_t *>(c->scratch[5]);
    HIPCHK(hipMemcpyAsync(c->scratch[1], words, total * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_off, offsets, (count + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_wo, word_offsets, (count + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (knobs(c).batch_host_plan) { // the caller's word_offsets were checked against the offsets above; the plan rebuilds them on the device
        size_t ptotal = 0;
        if (!c->host_plan) if (int st = bitnuc_batch_plan_create(c, &c->host_plan, err)) return st;
        if (int st = bitnuc_batch_plan_build_dev(c, c->host_plan, d_off, count, &ptotal, err)) return st;
        if (ptotal != total) return fail(err, BITNUC_INVALID_RANGE, count);
        if (int st = bitnuc_decode_batch_plan_dev(c, c->host_plan, reinterpret_cast<const uint64_t *>(c->scratch[1]), c->scratch[0] - b0, err)) return st;
    } else if (int st = bitnuc_decode_batch_dev(c, reinterpret_cast<const uint64_t *>(c->scratch[1]), d_wo, d_off, count, total, c->scratch[0] - b0, err)) return st;
    HIPCHK(hipMemcpyAsync(out + b0, c->scratch[0], nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BITNUC_OK;
}

// ---- ragged batches with a layout plan --------------------------------------------------------------
struct bitnuc_batch_plan {
    int device = 0;
    size_t count = 0, total_words = 0;
    unsigned long long seq_begin = 0, seq_end = 0;
    unsigned long long *d_wo = nullptr;   // count + 1 word offsets
    unsigned long long *d_base = nullptr; // one byte offset per 64-word tile
    uint8_t *d_P = nullptr;               // total_words + 1 pad bytes
    unsigned long long *d_bounds = nullptr; // offsets[0], offsets[count] (what the plan encode clips its loads to)
    size_t cap_wo = 0, cap_base = 0, cap_P = 0, cap_bounds = 0;
    bool built = false;
};

int bitnuc_batch_plan_create(bitnuc_ctx *c, bitnuc_batch_plan **out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    bitnuc_batch_plan *p = new bitnuc_batch_plan();
    p->device = c->device;
    *out = p;
    return BITNUC_OK;
}

void bitnuc_batch_plan_destroy(bitnuc_batch_plan *p) {
    if (!p) return;
    DeviceGuard g(p->device);
    if (p->d_wo) (void)hipFree(p->d_wo);
    if (p->d_base) (void)hipFree(p->d_base);
    if (p->d_P) (void)hipFree(p->d_P);
    if (p->d_bounds) (void)hipFree(p->d_bounds);
    delete p;
}

size_t bitnuc_batch_plan_total_words(const bitnuc_batch_plan *p) { return p && p->built ? p->total_words : 0; }
size_t bitnuc_batch_plan_count(const bitnuc_batch_plan *p) { return p && p->built ? p->count : 0; }
const uint64_t *bitnuc_batch_plan_word_offsets_dev(const bitnuc_batch_plan *p) { return p && p->built ? reinterpret_cast<const uint64_t *>(p->d_wo) : nullptr; }

extern "C++" {
namespace {
template <class T> int plan_reserve(T **buf, size_t *cap, size_t need_elems, hipStream_t stream, bitnuc_err *err) {
    if (need_elems <= *cap) return BITNUC_OK;
    if (*buf) {
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    size_t want = need_elems + need_elems / 4 + 64; // head-room: a stream of batches of similar size reuses the plan's memory
    if (hipMalloc(reinterpret_cast<void **>(buf), want * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        want = need_elems;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(buf), want * sizeof(T)));
    }
    *cap = want;
    return BITNUC_OK;
}
} // namespace
} // extern "C++"

int bitnuc_batch_plan_build_dev(bitnuc_ctx *c, bitnuc_batch_plan *p, const uint64_t *d_offsets, size_t count, size_t *total_words, bitnuc_err *err) {
    clear_err(err);
    if (total_words) *total_words = 0;
    if (int st = check_ctx(c, err)) return st;
    if (!p || p->device != c->device || (count && !d_offsets)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    p->built = false;
    if (int st = plan_reserve(&p->d_wo, &p->cap_wo, count + 1, c->stream, err)) return st;
    // sums -> scan of the sums -> (host learns the total and sizes the plan) -> offsets + pad bytes + tile bases + bounds in one pass
    unsigned long long ends[3] = {0, 0, 0}; // total words, offsets[0], offsets[count]
    const size_t nblocks = (count + kScanTile - 1) / kScanTile;
    const unsigned long long *off = reinterpret_cast<const unsigned long long *>(d_offsets);
    unsigned long long *sums = nullptr;
    if (count) {
        if (int st = ensure_scratch(c, 3, (nblocks + 3) * sizeof(uint64_t), err)) return st;
        sums = reinterpret_cast<unsigned long long *>(c->scratch[3]);
        word_offsets_block_sums<<<(unsigned)nblocks, kBlock, 0, c->stream>>>(off, count, sums);
        word_offsets_scan_sums<<<1, kBlock, 0, c->stream>>>(sums, nblocks, off, count);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ends, sums + nblocks, sizeof ends, hipMemcpyDeviceToHost, c->stream)); // total, offsets[0], offsets[count]
        HIPCHK(hipStreamSynchronize(c->stream));
    } else {
        HIPCHK(hipMemsetAsync(p->d_wo, 0, sizeof(uint64_t), c->stream));
    }
    const size_t total = (size_t)ends[0];
    p->count = count;
    p->total_words = total;
    const size_t ntiles = (total + kBatchTile - 1) / kBatchTile;
    if (int st = plan_reserve(&p->d_base, &p->cap_base, ntiles + 1, c->stream, err)) return st;
    if (int st = plan_reserve(&p->d_P, &p->cap_P, total + 2 + kBatchTile, c->stream, err)) return st;
    if (int st = plan_reserve(&p->d_bounds, &p->cap_bounds, 2, c->stream, err)) return st;
    if (count) { // one pass: word offsets + pad bytes + tile bases + buffer bounds (every pad byte is written: no memset)
        plan_emit_kernel<false, true><<<(unsigned)nblocks, kBlock, 0, c->stream>>>(off, sums, count, p->d_P, p->d_base, p->d_bounds, p->d_wo);
        HIPCHK(hipGetLastError());
    }
    p->seq_begin = ends[1];
    p->seq_end = ends[2];
    HIPCHK(hipStreamSynchronize(c->stream)); // the build is synchronous: the plan's tables may be read on any stream afterwards
    p->built = true;
    if (total_words) *total_words = total;
    return BITNUC_OK;
}

int bitnuc_encode_batch_plan_dev(bitnuc_ctx *c, const bitnuc_batch_plan *p, const uint8_t *d_seq, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!p || !p->built || p->device != c->device) return fail(err, BITNUC_UNSUPPORTED);
    if (p->total_words == 0) return BITNUC_OK;
    if (!d_seq || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    return launch_plan_encode(c, p->d_base, p->d_P, p->total_words, p->d_bounds, d_seq, d_out, err);
}

int bitnuc_decode_batch_plan_dev(bitnuc_ctx *c, const bitnuc_batch_plan *p, const uint64_t *d_words, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!p || !p->built || p->device != c->device) return fail(err, BITNUC_UNSUPPORTED);
    if (p->total_words == 0) return BITNUC_OK;
    if (!d_words || !d_out || (reinterpret_cast<uintptr_t>(d_words) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    return launch_plan_decode(c, p->d_base, p->d_P, p->total_words, d_words, d_out, err);
}

// ---- fixed-length reads ------------------------------------------------------------------------
int bitnuc_encode_fixed_dev(bitnuc_ctx *c, const uint8_t *d_seq, size_t read_len, size_t stride, size_t count, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || read_len == 0) return BITNUC_OK; // zero-length reads produce no words
    if (stride < read_len || read_len > 0xFFFFFFFFull - 64) return fail(err, BITNUC_UNSUPPORTED);
    if (!d_seq || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned wpr = (unsigned)words_for(read_len);
    const unsigned long long total = (unsigned long long)count * wpr;
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    const size_t per_block = (size_t)kBatchTile * kBatchWaves;
    const unsigned grid = grid_for(c, (total + per_block - 1) / per_block);
    unsigned long long *o = reinterpret_cast<unsigned long long *>(d_out);
    const unsigned long long seq_end = (unsigned long long)(count - 1) * stride + read_len;
    const unsigned magic = (unsigned)((0x100000000ull + wpr - 1) / wpr); // ceil(2^32 / wpr): exact floor(t / wpr) for t < 2^16
    const unsigned long long magic64 = wpr == 1 ? 0ull : ~0ull / wpr + 1;  // floor(2^64 / wpr) + 1: exact floor(w / wpr) by multiply-high while w * wpr < 2^64
    if (stride == read_len) encode_fixed_kernel<false><<<grid, kBlock, 0, c->stream>>>(d_seq, (unsigned)read_len, stride, wpr, magic, magic64, total, seq_end, knobs(c).fixed_stream, o, slot);
    else encode_fixed_kernel<true><<<grid, kBlock, 0, c->stream>>>(d_seq, (unsigned)read_len, stride, wpr, magic, magic64, total, seq_end, 0, o, slot);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_decode_fixed_dev(bitnuc_ctx *c, const uint64_t *d_words, size_t read_len, size_t stride, size_t count, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || read_len == 0) return BITNUC_OK;
    if (stride < read_len || read_len > 0xFFFFFFFFull - 64) return fail(err, BITNUC_UNSUPPORTED);
    if (!d_words || !d_out || (reinterpret_cast<uintptr_t>(d_words) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned wpr = (unsigned)words_for(read_len);
    const unsigned long long total = (unsigned long long)count * wpr;
    const size_t per_block = (size_t)kBatchTile * kBatchWaves;
    const unsigned grid = grid_for(c, (total + per_block - 1) / per_block);
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(d_words);
    const unsigned magic = (unsigned)((0x100000000ull + wpr - 1) / wpr);
    const unsigned long long magic64 = wpr == 1 ? 0ull : ~0ull / wpr + 1;
    if (stride != read_len) decode_fixed_kernel<false><<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, stride, wpr, magic, magic64, total, d_out);
    else if (knobs(c).fixed_dec_strip == 2 || !kEvidenceBuild) decode_fixed_tile_kernel<2><<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, wpr, magic, magic64, total, d_out);
#ifdef BITNUC_SWEEP_VARIANTS
    else if (knobs(c).fixed_dec_strip) decode_fixed_strip_kernel<<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, wpr, magic, magic64, total, d_out);
    else decode_fixed_kernel<true><<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, stride, wpr, magic, magic64, total, d_out);
#endif
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

// Host-pointer fixed-length reads through the pipelined staging engine (host_pipe.h): a chunk is as many whole reads as fit both
// pinned buffers (their bytes in an A buffer, their words in a B buffer; decode the other way round).
constexpr int kNotPipelined = -1; // not a bitnuc_status: "this shape does not fit the engine, use the staged loop"
static int fixed_pipelined(bitnuc_ctx *c, bool encode, const uint8_t *seq, const uint64_t *words, size_t read_len, size_t stride, size_t count,
                           uint64_t *out_words, uint8_t *out_seq, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    const size_t wpr = words_for(read_len);
    size_t per = (p->chunk / 4) / (8 * wpr);
    const size_t by_bytes = p->chunk > read_len ? (p->chunk - read_len) / stride + 1 : 0;
    if (by_bytes < per) per = by_bytes;
    // a read larger than a chunk (today unreachable: the callers admit read_len, stride < 1 Mi and a chunk is at least 1 MiB): nothing
    // has been started, the pipe is intact, the caller goes on to its staged-scratch loop
    if (per == 0) return kNotPipelined;
    PipeAbort guard{c, p};
    struct Job {
        bitnuc_ctx *c; bool encode; const uint8_t *seq; const uint64_t *words; size_t read_len, stride, count, per, wpr; uint64_t *out_words; uint8_t *out_seq;
        size_t nchunks; int in_kind, out_kind, in_threads, out_threads;
        size_t items(size_t ci) const { return count - ci * per < per ? count - ci * per : per; }
        size_t seq_bytes(size_t ci) const { return (items(ci) - 1) * stride + read_len; }
        const void *in_src(size_t ci) const { return encode ? static_cast<const void *>(seq + ci * per * stride) : static_cast<const void *>(words + ci * per * wpr); }
        size_t in_bytes(size_t ci) const { return encode ? seq_bytes(ci) : items(ci) * wpr * 8; }
        void *out_dst(size_t ci) const { return encode ? static_cast<void *>(out_words + ci * per * wpr) : static_cast<void *>(out_seq + ci * per * stride); }
        size_t out_bytes(size_t ci) const { return encode ? items(ci) * wpr * 8 : seq_bytes(ci); }
        int launch(size_t ci, const uint8_t *d_in, uint8_t *d_out, bitnuc_err *err) const {
            if (encode) {
                if (int st = bitnuc_encode_fixed_dev(c, d_in, read_len, stride, items(ci), reinterpret_cast<uint64_t *>(d_out), err)) return st;
                set_last_slot_base(c, (unsigned long long)(ci * per) * stride); // report the index in the caller's buffer
                return BITNUC_OK;
            }
            return bitnuc_decode_fixed_dev(c, reinterpret_cast<const uint64_t *>(d_in), read_len, stride, items(ci), d_out, err);
        }
    } job{c, encode, seq, words, read_len, stride, count, per, wpr, out_words, out_seq, (count + per - 1) / per};
    job.in_kind = encode ? kBufA : kBufB;
    job.out_kind = encode ? kBufB : kBufA;
    job.in_threads = encode ? p->enc_in : p->dec_in;
    job.out_threads = encode ? p->enc_out : p->dec_out;
    if (int st = pipe_run(c, p, job, err)) return st;
    bitnuc_err e;
    const int st = drain(c, &e);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    guard.dismissed = true;
    if (st != BITNUC_OK) { if (err) *err = e; return st; }
    return BITNUC_OK;
}

int bitnuc_encode_fixed(bitnuc_ctx *c, const uint8_t *seq, size_t read_len, size_t stride, size_t count, uint64_t *out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || read_len == 0) return BITNUC_OK;
    if (stride < read_len || !seq || !out) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    if (c->host_pipeline && (count - 1) * stride + read_len >= kPipeMin && stride < ((size_t)1 << 20)) {
        const int st = fixed_pipelined(c, true, seq, nullptr, read_len, stride, count, out, nullptr, err);
        if (st != kNotPipelined) return st;
    }
    const size_t wpr = words_for(read_len);
    size_t per = kHostChunk / stride; // reads per staged chunk
    if (per == 0) per = 1;
    if (per > count) per = count;
    if (int st = ensure_scratch(c, 0, (per - 1) * stride + read_len + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, per * wpr * 8 + 16, err)) return st;
    for (size_t r0 = 0; r0 < count; r0 += per) {
        const size_t m = count - r0 < per ? count - r0 : per;
        const size_t bytes = (m - 1) * stride + read_len;
        HIPCHK(hipMemcpyAsync(c->scratch[0], seq + r0 * stride, bytes, hipMemcpyHostToDevice, c->stream));
        bitnuc_err e;
        if (int st = bitnuc_encode_fixed_dev(c, c->scratch[0], read_len, stride, m, reinterpret_cast<uint64_t *>(c->scratch[1]), &e)) { if (err) *err = e; return st; }
        set_last_slot_base(c, (unsigned long long)r0 * stride); // report the index in the caller's buffer
        HIPCHK(hipMemcpyAsync(out + r0 * wpr, c->scratch[1], m * wpr * 8, hipMemcpyDeviceToHost, c->stream));
        int st = drain(c, &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
    }
    return BITNUC_OK;
}

int bitnuc_decode_fixed(bitnuc_ctx *c, const uint64_t *words, size_t read_len, size_t stride, size_t count, uint8_t *out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || read_len == 0) return BITNUC_OK;
    if (stride < read_len || !words || !out) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    // back-to-back reads only: with separators the bytes between reads are the caller's and would have to travel both ways
    if (c->host_pipeline && stride == read_len && count * read_len >= kPipeMin && read_len < ((size_t)1 << 20)) {
        if (int st = flush_pending(c, err)) return st;
        const int st = fixed_pipelined(c, false, nullptr, words, read_len, stride, count, nullptr, out, err);
        if (st != kNotPipelined) return st;
    }
    const size_t wpr = words_for(read_len);
    size_t per = kHostChunk / stride;
    if (per == 0) per = 1;
    if (per > count) per = count;
    if (int st = ensure_scratch(c, 0, (per - 1) * stride + read_len + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, per * wpr * 8 + 16, err)) return st;
    for (size_t r0 = 0; r0 < count; r0 += per) {
        const size_t m = count - r0 < per ? count - r0 : per;
        const size_t bytes = (m - 1) * stride + read_len;
        HIPCHK(hipMemcpyAsync(c->scratch[1], words + r0 * wpr, m * wpr * 8, hipMemcpyHostToDevice, c->stream));
        if (stride != read_len) // separator bytes are the caller's: bring them in so they go back unchanged
            HIPCHK(hipMemcpyAsync(c->scratch[0], out + r0 * stride, bytes, hipMemcpyHostToDevice, c->stream));
        if (int st = bitnuc_decode_fixed_dev(c, reinterpret_cast<const uint64_t *>(c->scratch[1]), read_len, stride, m, c->scratch[0], err)) return st;
        HIPCHK(hipMemcpyAsync(out + r0 * stride, c->scratch[0], bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return BITNUC_OK;
}

} // extern "C"
