// runtime.hip -- the context of libbitnuc_hip.so (include/bitnuc_hip.h): device + stream, the per-launch data-error slots
// and their lifetime (ordinary launches, launches recorded into a hipGraph), staging scratch, tuning knobs.
// No kernels here; the arithmetic of the path lives in the kernel units (runtime.h lists them) and, for single words and
// calls below the host cutoff, in host_word.h.
#include "runtime.h"

#include <stdlib.h>

using namespace bitnuc_rt;

namespace {

constexpr unsigned long long kNoBad = ~0ull; // == bitnuc_dev::kNoBad (device_prims.h)

hipError_t slot_block_alloc(SlotBlock *b, int cap, hipStream_t stream) {
    hipError_t rc = hipMalloc(reinterpret_cast<void **>(&b->d), sizeof(unsigned long long) * (size_t)cap);
    if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void **>(&b->h), sizeof(unsigned long long) * (size_t)cap, hipHostMallocDefault);
    if (rc == hipSuccess) rc = hipMemsetAsync(b->d, 0xFF, sizeof(unsigned long long) * (size_t)cap, stream); // stream-ordered before the launches that use it
    if (rc != hipSuccess) {
        if (b->d) (void)hipFree(b->d);
        if (b->h) (void)hipHostFree(b->h);
        b->d = b->h = nullptr;
        return rc;
    }
    b->cap = cap;
    b->used = 0;
    b->base.clear();
    b->base.reserve((size_t)cap < 65536 ? (size_t)cap : 65536);
    return hipSuccess;
}

void slot_block_free(SlotBlock *b) {
    if (b->d) (void)hipFree(b->d);
    if (b->h) (void)hipHostFree(b->h);
    b->d = b->h = nullptr;
    b->cap = b->used = 0;
}

bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { // e.g. the legacy NULL stream while another stream captures
        (void)hipGetLastError();
        return false;
    }
    return st == hipStreamCaptureStatusActive;
}

// scope of a drain: host-pointer calls look at their own (ordinary) launches only; bitnuc_ctx_sync also at the captured ones
enum { kDrainOrdinary = 0, kDrainAll = 1 };

int drain_scope(bitnuc_ctx *c, bitnuc_err *err, int scope) {
    const int n_cap = scope == kDrainAll ? c->n_cap : 0;
    // the slot read-back is stream-ordered behind the launches it reports on: one wait covers both
    for (SlotBlock &b : c->slots)
        if (b.used > 0) HIPCHK(hipMemcpyAsync(b.h, b.d, sizeof(unsigned long long) * (size_t)b.used, hipMemcpyDeviceToHost, c->stream));
    if (n_cap > 0) HIPCHK(hipMemcpyAsync(c->h_cap, c->d_cap, sizeof(unsigned long long) * (size_t)n_cap, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    bitnuc_err found;
    memset(&found, 0, sizeof found);
    auto report = [&](unsigned long long v, unsigned long long base) { // slot = (byte index << 8) | byte, device_prims.h latch_bad
        if (found.status != BITNUC_OK) return;
        found.status = BITNUC_INVALID_BASE;
        found.byte = (uint8_t)(v & 0xFF);
        found.index = base + (v >> 8);
    };
    // ordinary launches, in launch order
    const size_t nblocks = c->slots.size();
    for (size_t bi = 0; bi < nblocks; ++bi) {
        SlotBlock &b = c->slots[bi];
        bool fired = false;
        for (int i = 0; i < b.used; ++i)
            if (b.h[i] != kNoBad) { fired = true; report(b.h[i], b.base[(size_t)i]); }
        // re-arm what fired -- only in the block that stays (the others are freed below: the stream is idle)
        if (fired && bi + 1 == nblocks) HIPCHK(hipMemsetAsync(b.d, 0xFF, sizeof(unsigned long long) * (size_t)b.used, c->stream));
        b.used = 0;
        b.base.clear();
    }
    if (nblocks > 1) { // the ring grew: keep the newest (largest) block
        for (size_t bi = 0; bi + 1 < nblocks; ++bi) slot_block_free(&c->slots[bi]);
        SlotBlock keep = c->slots.back();
        c->slots.clear();
        c->slots.push_back(keep);
    }
    // launches recorded into a hipGraph, in capture order: their slots live on and are re-armed for the next replay
    bool cap_fired = false;
    for (int i = 0; i < n_cap; ++i)
        if (c->h_cap[i] != kNoBad) { cap_fired = true; report(c->h_cap[i], c->cap_base[i]); }
    if (cap_fired) HIPCHK(hipMemsetAsync(c->d_cap, 0xFF, sizeof(unsigned long long) * (size_t)n_cap, c->stream));
    if (err) *err = found;
    return found.status;
}

// A FIFO, not one slot: error A, a host-pointer call (its implicit drain defers A), error B, another host-pointer call (defers B): the next
// two syncs report A, then B (B used to be dropped).  Bounded: a caller that never syncs keeps the 64 oldest.
void defer(bitnuc_ctx *c, const bitnuc_err &e) {
    if (c->deferred.size() < 64) c->deferred.push_back(e);
}

} // namespace

namespace bitnuc_rt {

int ensure_scratch(bitnuc_ctx *c, int which, size_t bytes, bitnuc_err *err) {
    const bool capturing = stream_is_capturing(c->stream);
    if (bytes <= c->scratch_cap[which]) {
        // the launches that follow bake this buffer's address into the graph being recorded: it must outlive every replay
        if (capturing) c->scratch_in_graph[which] = true;
        return BITNUC_OK;
    }
    // Growth is an allocation (and, for a buffer no graph holds, a wait for the stream + a free): impossible while the stream is being
    // captured -- refused before anything is touched, the capture stays valid.  Warm up with the largest batch first, or use
    // bitnuc_batch_plan (its tables are the plan's own, sized once).
    if (capturing) return fail(err, BITNUC_UNSUPPORTED, bytes);
    const size_t old_cap = c->scratch_cap[which];
    if (c->scratch[which]) {
        if (c->scratch_in_graph[which]) {
            // a recorded graph will write through the old address at its next replay: keep the old buffer alive until the context goes
            c->retired_scratch.push_back(c->scratch[which]);
            c->scratch_in_graph[which] = false;
        } else {
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipFree(c->scratch[which]));
        }
        c->scratch[which] = nullptr;
        c->scratch_cap[which] = 0;
    }
    // grow geometrically (a caller whose batches creep up in size should not reallocate every call), 4 KiB granules
    size_t want = old_cap + old_cap / 2;
    if (want < bytes) want = bytes;
    size_t cap = (want + 4095) & ~(size_t)4095;
    if (hipMalloc(&c->scratch[which], cap) != hipSuccess) { // not enough for the head-room: take exactly what is needed
        (void)hipGetLastError();
        cap = (bytes + 4095) & ~(size_t)4095;
        HIPCHK(hipMalloc(&c->scratch[which], cap));
    }
    c->scratch_cap[which] = cap;
    return BITNUC_OK;
}

bool slots_outstanding(const bitnuc_ctx *c) {
    for (const SlotBlock &b : c->slots)
        if (b.used > 0) return true;
    return false;
}

int drain(bitnuc_ctx *c, bitnuc_err *err) { return drain_scope(c, err, kDrainOrdinary); }

int flush_pending(bitnuc_ctx *c, bitnuc_err *err) {
    if (!slots_outstanding(c)) return BITNUC_OK; // nothing asynchronous outstanding: stream order is enough
    bitnuc_err e;
    const int st = drain_scope(c, &e, kDrainOrdinary);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    if (st != BITNUC_OK) defer(c, e);
    return BITNUC_OK;
}

int take_slot(bitnuc_ctx *c, unsigned long long base, unsigned long long **slot, bitnuc_err *err) {
    if (stream_is_capturing(c->stream)) {
        // The launch is being recorded: it runs again at every replay of the graph, long after the next sync has emptied
        // the ring.  It gets a slot of its own for the life of the context; every bitnuc_ctx_sync examines and re-arms it.
        if (c->n_cap == kCapturedSlots) return fail(err, BITNUC_UNSUPPORTED, (uint64_t)kCapturedSlots);
        c->cap_base[c->n_cap] = base;
        *slot = c->d_cap + c->n_cap;
        c->n_cap++;
        return BITNUC_OK;
    }
    SlotBlock *b = &c->slots.back();
    if (b->used == b->cap) {
        if (b->cap < kSlotBlockMax) { // grow: a new block of twice the size, the full ones stay until the next drain
            SlotBlock nb;
            HIPCHK(slot_block_alloc(&nb, b->cap * 2, c->stream));
            c->slots.push_back(nb);
            b = &c->slots.back();
        } else { // > 2 M launches without a sync: an implicit drain (documented in include/bitnuc_hip.h); its error is kept for the next sync
            bitnuc_err e;
            const int st = drain_scope(c, &e, kDrainOrdinary);
            if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
            if (st != BITNUC_OK) defer(c, e);
            b = &c->slots.back();
        }
    }
    b->base.push_back(base);
    *slot = b->d + b->used;
    b->used++;
    return BITNUC_OK;
}

void set_last_slot_base(bitnuc_ctx *c, unsigned long long base) {
    if (stream_is_capturing(c->stream)) { if (c->n_cap > 0) c->cap_base[c->n_cap - 1] = base; return; }
    SlotBlock &b = c->slots.back();
    if (b.used > 0) b.base[(size_t)b.used - 1] = base;
}

} // namespace bitnuc_rt

// =====================================================================================
// C ABI: context
// =====================================================================================
extern "C" {

#ifndef BITNUC_CSRC_SHA
#define BITNUC_CSRC_SHA "unknown" // bitnuc_amd/build.py passes the hash of csrc/ + include/bitnuc_hip.h; a build without it cannot be held against its sources
#endif
const char *bitnuc_version(void) { return kEvidenceBuild ? "bitnuc_hip 0.4.0 gfx950 csrc:" BITNUC_CSRC_SHA " sweep" : "bitnuc_hip 0.4.0 gfx950 csrc:" BITNUC_CSRC_SHA; }

int bitnuc_ctx_create_on_stream(int device, void *hip_stream, bitnuc_ctx **out, bitnuc_err *err) {
    clear_err(err);
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail_hip(err, hipErrorInvalidDevice);
    DeviceGuard g(device);
    bitnuc_ctx *c = new bitnuc_ctx();
    c->device = device;
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    hipDeviceProp_t prop;
    hipError_t rc = hipGetDeviceProperties(&prop, device);
    if (rc == hipSuccess) c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (rc == hipSuccess) {
        SlotBlock b;
        rc = slot_block_alloc(&b, kSlotBlock0, c->stream);
        if (rc == hipSuccess) c->slots.push_back(b);
    }
    if (rc == hipSuccess) rc = hipMalloc(reinterpret_cast<void **>(&c->d_cap), sizeof(unsigned long long) * kCapturedSlots);
    if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void **>(&c->h_cap), sizeof(unsigned long long) * kCapturedSlots, hipHostMallocDefault);
    if (rc == hipSuccess) rc = hipMemset(c->d_cap, 0xFF, sizeof(unsigned long long) * kCapturedSlots);
    if (rc == hipSuccess) rc = hipMalloc(&c->d_sink, 64);
    if (rc == hipSuccess) rc = hipMemset(c->d_sink, 0, 64);
    if (const char *e = getenv("BITNUC_FORCE_GPU")) c->force_gpu = atoi(e) != 0;
    if (const char *e = getenv("BITNUC_PIPE_IMPL")) { if (!strcmp(e, "direct")) c->pipe_impl = 1; else if (!strcmp(e, "staged")) c->pipe_impl = 0; }
    if (const char *e = getenv("BITNUC_HOST_CUTOFF")) { const long long v = atoll(e); if (v >= 0) c->host_cutoff = c->host_cutoff_decode = (size_t)v; }
    c->hdist_blocks = (unsigned)c->num_cu;     // 1 per CU: 74.3 us for two 250 MB operands where 2 per CU take 77.2 and 4 per CU 84.4 (profiles/r03_ab_hdist_grid.txt)
    c->reduce_blocks = (unsigned)c->num_cu * 2; // a resident grid of 2 workgroups of 256 threads per CU (profiles/r01_sweep13_reduce_grid.txt: the tail of atomics + ticket grows with the grid)
    if (rc == hipSuccess) rc = hipMalloc(&c->d_acc, 64 + 8 * 1024); // [0..7] the single-launch reductions' accumulators, [8..8+1024) the fused count's partials (scan_mfma_device.h kScanPartials)
    if (rc == hipSuccess) rc = hipMemset(c->d_acc, 0, 64 + 8 * 1024);
    if (rc == hipSuccess) rc = hipMalloc(&c->d_tickets, 64);
    if (rc == hipSuccess) rc = hipMemset(c->d_tickets, 0, 64);
    if (rc == hipSuccess) rc = hipStreamSynchronize(c->stream); // the slot block's memset
    if (rc != hipSuccess) {
        bitnuc_ctx_destroy(c);
        return fail_hip(err, rc);
    }
    *out = c;
    return BITNUC_OK;
}

int bitnuc_ctx_create(int device, bitnuc_ctx **out, bitnuc_err *err) {
    clear_err(err);
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail_hip(err, hipErrorInvalidDevice);
    DeviceGuard g(device);
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int st = bitnuc_ctx_create_on_stream(device, s, out, err);
    if (st != BITNUC_OK) { (void)hipStreamDestroy(s); return st; }
    (*out)->own_stream = true;
    return BITNUC_OK;
}

void bitnuc_ctx_destroy(bitnuc_ctx *c) {
    if (!c) return;
    DeviceGuard g(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 8; ++i)
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    for (uint8_t *p : c->retired_scratch) (void)hipFree(p);
    for (SlotBlock &b : c->slots) slot_block_free(&b);
    if (c->d_cap) (void)hipFree(c->d_cap);
    if (c->h_cap) (void)hipHostFree(c->h_cap);
    if (c->d_sink) (void)hipFree(c->d_sink);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->d_tickets) (void)hipFree(c->d_tickets);
    pipe_destroy(c->pipe);
    bitnuc_batch_plan_destroy(c->host_plan);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void *bitnuc_ctx_stream(bitnuc_ctx *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

int bitnuc_ctx_sync(bitnuc_ctx *c, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    bitnuc_err e;
    int st = drain_scope(c, &e, kDrainAll);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    if (!c->deferred.empty()) { // earlier implicit drains saw errors first: the oldest is this sync's; what this drain found queues up behind them
        if (st != BITNUC_OK) defer(c, e);
        e = c->deferred.front();
        c->deferred.erase(c->deferred.begin());
        st = e.status;
    }
    if (err) *err = e;
    return st;
}

namespace {

// The selectors of alternative formulations (runtime.h SweepKnobs): key, field, and the values the evidence build accepts.
struct SweepKey {
    const char *key;
    int bitnuc_rt::SweepKnobs::*field;
    int lo, hi;          // accepted range ...
    unsigned long long only; // ... or, when non-zero, a set: bit v set = value v accepted (v < 64), plus `also`
    int also[3];         // values >= 64 of the set (0 = none)
};
constexpr SweepKey kSweepKeys[] = {
    {"dyn_lds", &bitnuc_rt::SweepKnobs::dyn_lds, 0, 64 * 1024, 0, {0, 0, 0}},
    {"batch_dense", &bitnuc_rt::SweepKnobs::batch_dense, 0, 1, 0, {0, 0, 0}},
    {"batch_slide", &bitnuc_rt::SweepKnobs::batch_slide, 0, 1, 0, {0, 0, 0}},
    {"slide_impl", &bitnuc_rt::SweepKnobs::slide_impl, 0, 1, 0, {0, 0, 0}},
    {"slide_rounds", &bitnuc_rt::SweepKnobs::slide_rounds, 0, 0, 1ull << 1 | 1ull << 2 | 1ull << 4 | 1ull << 8, {0, 0, 0}},
    {"slide2_rounds", &bitnuc_rt::SweepKnobs::slide2_rounds, 0, 0, 1ull << 1 | 1ull << 2 | 1ull << 4, {0, 0, 0}},
    {"dense_policy", &bitnuc_rt::SweepKnobs::dense_policy, 0, 3, 0, {0, 0, 0}},
    {"scan_policy", &bitnuc_rt::SweepKnobs::scan_policy, 0, 3, 0, {0, 0, 0}},
    {"kmer_block", &bitnuc_rt::SweepKnobs::kmer_block, 0, 0, 0, {64, 128, 256}},
    {"dense_unroll", &bitnuc_rt::SweepKnobs::dense_unroll, 0, 0, 1ull << 1 | 1ull << 2 | 1ull << 4, {0, 0, 0}},
    {"scan_unroll", &bitnuc_rt::SweepKnobs::scan_unroll, 0, 0, 1ull << 1 | 1ull << 2 | 1ull << 4, {0, 0, 0}},
    {"scan_impl", &bitnuc_rt::SweepKnobs::scan_impl, 0, 8, 0, {0, 0, 0}},
    {"scan_mfma_unroll", &bitnuc_rt::SweepKnobs::scan_mfma_unroll, 0, 0, 1ull << 2 | 1ull << 3 | 1ull << 4, {0, 0, 0}},
    {"scan_mfma_shift", &bitnuc_rt::SweepKnobs::scan_mfma_shift, 0, 6, 0, {0, 0, 0}},
    {"scan_mfma_count_form", &bitnuc_rt::SweepKnobs::scan_mfma_count_form, 0, 2, 0, {0, 0, 0}},
    {"scan_mfma_block", &bitnuc_rt::SweepKnobs::scan_mfma_block, 0, 0, 0, {64, 128, 256}},
    {"scan_mfma_ch3", &bitnuc_rt::SweepKnobs::scan_mfma_ch3, 0, 1, 0, {0, 0, 0}},
    {"scan_mfma_match", &bitnuc_rt::SweepKnobs::scan_mfma_match, 0, 1, 0, {0, 0, 0}},
    {"scan_mfma_count_emit", &bitnuc_rt::SweepKnobs::scan_mfma_count_emit, 0, 2, 0, {0, 0, 0}},
    {"scan_mfma_count_rounds", &bitnuc_rt::SweepKnobs::scan_mfma_count_rounds, 2, 4, 0, {0, 0, 0}},
    {"scan_mfma_count_grid", &bitnuc_rt::SweepKnobs::scan_mfma_count_grid, 1, 64, 0, {0, 0, 0}},
    {"scan_mfma_count_persist", &bitnuc_rt::SweepKnobs::scan_mfma_count_persist, 0, 1, 0, {0, 0, 0}},
    {"scan_mfma_persist", &bitnuc_rt::SweepKnobs::scan_mfma_persist, 0, 1, 0, {0, 0, 0}},
    {"scan_mfma_grid", &bitnuc_rt::SweepKnobs::scan_mfma_grid, 1, 256, 0, {0, 0, 0}},
    {"scan_mfma_pack", &bitnuc_rt::SweepKnobs::scan_mfma_pack, 0, 2, 0, {0, 0, 0}},
    {"scan_mfma_policy", &bitnuc_rt::SweepKnobs::scan_mfma_policy, 0, 3, 0, {0, 0, 0}},
    {"hdist_tiled", &bitnuc_rt::SweepKnobs::hdist_tiled, 0, 1, 0, {0, 0, 0}},
    {"hdist_words_impl", &bitnuc_rt::SweepKnobs::hdist_words_impl, 0, 1, 0, {0, 0, 0}},
    {"fixed_stream", &bitnuc_rt::SweepKnobs::fixed_stream, 0, 1, 0, {0, 0, 0}},
    {"fixed_dec_strip", &bitnuc_rt::SweepKnobs::fixed_dec_strip, 0, 2, 0, {0, 0, 0}},
    {"owner_est", &bitnuc_rt::SweepKnobs::owner_est, 0, 3, 0, {0, 0, 0}},
    {"batch_tables_impl", &bitnuc_rt::SweepKnobs::batch_tables_impl, 0, 1, 0, {0, 0, 0}},
    {"batch_host_plan", &bitnuc_rt::SweepKnobs::batch_host_plan, 0, 1, 0, {0, 0, 0}},
    {"plan_dec_lines", &bitnuc_rt::SweepKnobs::plan_dec_lines, 0, 2, 0, {0, 0, 0}},
    {"plan_tiles", &bitnuc_rt::SweepKnobs::plan_tiles, 0, 0, 1ull << 1 | 1ull << 2 | 1ull << 4, {0, 0, 0}},
    {"plan_store", &bitnuc_rt::SweepKnobs::plan_store, 0, 2, 0, {0, 0, 0}},
    {"plan_enc_block", &bitnuc_rt::SweepKnobs::plan_enc_block, 0, 0, 0, {64, 128, 256}},
    {"plan_enc_tiles", &bitnuc_rt::SweepKnobs::plan_enc_tiles, 0, 0, 1ull << 1 | 1ull << 2 | 1ull << 4, {0, 0, 0}},
    {"plan_enc_abl", &bitnuc_rt::SweepKnobs::plan_enc_abl, 0, 7, 0, {0, 0, 0}},
    {"batch_abl", &bitnuc_rt::SweepKnobs::batch_abl, 0, 15, 0, {0, 0, 0}},
};

// -3: not one of these keys; otherwise the previous value, or -2 for a value this build does not hold
int set_sweep_key(bitnuc_ctx *c, const char *key, int value) {
    for (const SweepKey &k : kSweepKeys) {
        if (strcmp(key, k.key)) continue;
        const int prev = bitnuc_rt::knobs(c).*(k.field);
        if (value < 0) return prev; // query
#ifdef BITNUC_SWEEP_VARIANTS
        bool ok = k.only ? (value < 64 && (k.only >> value & 1)) : (k.also[0] ? false : value >= k.lo && value <= k.hi);
        for (int a : k.also) ok = ok || (a && a == value);
        if (ok) c->sweep.*(k.field) = value; // (an unaccepted value changes nothing, as before)
        return prev;
#else
        return value == prev ? prev : -2; // the product holds the shipped formulation only
#endif
    }
    return -3;
}

} // namespace

int bitnuc_ctx_set_variant(bitnuc_ctx *c, const char *key, int value) {
    if (!c || !key) return -1;
    if (const int r = set_sweep_key(c, key, value); r != -3) return r;
    int prev = -1;
    if (!strcmp(key, "encode")) {
        prev = c->enc_variant;
        if (value >= 0) { // a variant this build does not hold is refused: -2, nothing changes
            if (value != codec_ballot_variant() && !codec_variant_built(value)) return -2;
            c->enc_variant = value;
        }
    }
    else if (!strcmp(key, "decode")) { prev = c->dec_variant; if (value >= 0) { if (!codec_decode_variant_ok(value)) return -2; c->dec_variant = value; } }
    else if (!strcmp(key, "force_gpu")) { prev = c->force_gpu; if (value == 0 || value == 1) c->force_gpu = value; }
    else if (!strcmp(key, "host_cutoff")) { prev = (int)(c->host_cutoff > 0x7FFFFFFF ? 0x7FFFFFFF : c->host_cutoff); if (value >= 0) c->host_cutoff = c->host_cutoff_decode = (size_t)value; } // sets both
    else if (!strcmp(key, "host_cutoff_decode")) { prev = (int)(c->host_cutoff_decode > 0x7FFFFFFF ? 0x7FFFFFFF : c->host_cutoff_decode); if (value >= 0) c->host_cutoff_decode = (size_t)value; }
    else if (!strcmp(key, "pipe_impl")) { prev = c->pipe_impl; if (value == 0 || value == 1) c->pipe_impl = value; }
    else if (!strcmp(key, "host_pipeline")) { prev = c->host_pipeline; if (value == 0 || value == 1) c->host_pipeline = value; }
    else if (!strcmp(key, "sweep_build")) { prev = kEvidenceBuild ? 1 : 0; }
    else if (!strcmp(key, "grid_mult")) { prev = c->grid_mult; if (value >= 0 && value <= 64) c->grid_mult = value; }
    else if (!strcmp(key, "reduce_mult")) { prev = (int)(c->reduce_blocks / (unsigned)c->num_cu); if (value >= 1 && value <= 32) c->reduce_blocks = (unsigned)c->num_cu * (unsigned)value; }
    else if (!strcmp(key, "hdist_mult")) { prev = (int)(c->hdist_blocks / (unsigned)c->num_cu); if (value >= 1 && value <= 32) c->hdist_blocks = (unsigned)c->num_cu * (unsigned)value; }
    else if (!strcmp(key, "num_variants")) { prev = codec_num_variants(); }
    else if (!strcmp(key, "num_cu")) { prev = c->num_cu; }
    else if (!strcmp(key, "captured_slots")) { prev = c->n_cap; }
    return prev;
}

} // extern "C"
