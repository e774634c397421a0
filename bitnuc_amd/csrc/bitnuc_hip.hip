// bitnuc_hip.hip -- host runtime + C ABI of libbitnuc_hip.so (see include/bitnuc_hip.h).
//
// Owns: the context (device, stream, per-launch error slots, staging scratch),
// kernel-variant dispatch, argument validation with the reference's error
// vocabulary (src/error.rs:3-18), and the host-pointer convenience paths.
// All arithmetic of the path runs in the kernels of codec_device.h /
// kmer_device.h; there is deliberately no CPU implementation in this library.
#include "../../include/bitnuc_hip.h"
#include "codec_device.h"
#include "kmer_device.h"
#include "batch_device.h"
#include "analysis_device.h"
#include "host_word.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

using namespace bitnuc_dev;

namespace {

constexpr int kSlots = 4096;                    // data-error slots between two syncs
constexpr size_t kHostChunk = (size_t)128 << 20; // bases per staged chunk on the simple host-pointer path
constexpr size_t kPipeChunkDefault = (size_t)32 << 20; // bases per chunk of the pipelined host-pointer path (encode / decode); BITNUC_PIPE_CHUNK_MB overrides
constexpr size_t kPipeMin = (size_t)8 << 20;     // inputs below this stay on the simple path (latency, not bandwidth, matters there)
// Bulk host-pointer calls below this many bases run on the host (host_word.h): a launch with its two copies
// costs ~35 us, the SWAR loop moves ~2-4 GB/s, so the crossover sits near 10^5 bases (tools/latency.py).
// Size dispatch of the host-pointer bulk calls: the measured crossover between the library's host SWAR code (one
// thread, 9-10 Gbases/s) and the GPU path (stage in, launch, stage out, one wait: 33-40 us + PCIe) on the GPU box, tools/host_cutoff.py,
// profiles/r02_host_cutoff.txt: encode 1 Mi bases (101 vs 104 us), decode 512 Ki bases (62 vs 63 us).
constexpr size_t kDefaultHostCutoff = (size_t)1 << 20;       // encode, hdist
constexpr size_t kDefaultHostCutoffDecode = (size_t)1 << 19; // decode

struct Pending {
    unsigned long long base;   // added to the slot's index (host path chunk offset)
};

} // namespace

struct bitnuc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    unsigned long long *d_slots = nullptr;
    unsigned long long *h_slots = nullptr; // pinned mirror
    Pending pending[kSlots];
    int n_pending = 0;
    bool have_deferred = false; // an error found by an implicit drain, reported at next sync
    bitnuc_err deferred;
    uint8_t *scratch[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_cap[6] = {0, 0, 0, 0, 0, 0};
    uint32_t *d_sink = nullptr;
    unsigned long long *d_acc = nullptr; // accumulators of the single-launch reductions, zero between launches: [0..2] base_counts C,G,T; [4] hdist (u32)
    unsigned *d_tickets = nullptr;       // [0] base_counts, [1] hdist: arrival counters, zero between launches
    unsigned reduce_blocks = 512;
    int enc_variant = 39, dec_variant = 22; // kDefaultEnc / kDefaultDec
    int grid_mult = 0;                   // see grid_for()
    int batch_dense = 1;                 // stride == k batches use kmer_dense_kernel
    int slide_rounds = 1;                // kmer_slide_kernel: consecutive 992-base rounds per wave trip (1, 2 or 4)
    int batch_slide = 1;                 // stride 1 (every window of a sequence), 2, 4, 8, 16 batches use kmer_slide_kernel
    int dense_policy = 3, scan_policy = 3; // bit0: nt loads, bit1: nt stores
    int fixed_stream = 1;                  // encode_fixed (back-to-back reads): 1 = cut the tile's 2-bit stream
    int fixed_dec_strip = 2;               // decode_fixed (back-to-back reads): 0 = byte scatter, 1 = bit strip with per-lane 64-bit positions, 2 = the plan decode's tile body with arithmetic lookups (tools/ab_fixed_dec.py)
    int owner_est = 3;                     // block_owner_kernel's first guess: 0 = 128-bit division, 1 = double, 2 = exact 0.64 fixed-point multiply-high, 3 = 2 or 0 by average sequence length
    int batch_host_plan = 1;               // host-pointer ragged-batch calls build a layout plan (bitnuc_batch_plan) and use the plan kernels
    bitnuc_batch_plan *host_plan = nullptr; // ... kept by the context
    int plan_tiles = 1;                    // decode_batch_plan_kernel: consecutive tiles per wave trip (1, 2 or 4)
    int plan_enc_abl = 0;                  // evidence build: timing-only ablations of the plan encode's loads (see plan_enc_issue)
    int plan_enc_block = 256;              // threads per workgroup of the plan encode (64, 128, 256)
    int plan_enc_tiles = 1;                // encode_batch_plan_kernel: consecutive tiles per wave trip (1, 2 or 4)
    int plan_store = 2;                    // decode_batch_plan_kernel's whole-chunk store policy: 0 nt, 1 plain, 2 plain on the shared edge lines + nt elsewhere
    int batch_abl = 0;                     // timing-only ablation mask of the second formulation (tools/ab_batch_ablate.py); 0 in normal use
    int kmer_block = 256;                  // threads per workgroup of the dense-batch and scan kernels: 64, 128 or 256
    int dense_unroll = 1;                  // items (64 k-mers = 2 dwordx4 per lane) in flight per wave: 1, 2 or 4
    int scan_unroll = 4;                   // rounds (1 KiB loads) in flight per wave: 1, 2 or 4
    int scan_impl = 1;                     // 1 = line-aligned rounds of 1024 windows (kmer_scan2_kernel), 0 = rounds of 992 windows (kmer_scan_kernel)
    int force_gpu = 0;                     // 1: single-word and below-cutoff calls launch kernels too (GPU parity tests, BITNUC_FORCE_GPU=1)
    size_t host_cutoff = kDefaultHostCutoff; // bulk host-pointer encode / hdist below this many bases run on the host (host_word.h)
    size_t host_cutoff_decode = kDefaultHostCutoffDecode; // ... decode
    int host_pipeline = 1;                 // large host-pointer encode / decode: pinned double buffers + overlapped H2D / kernel / D2H
    struct HostPipe *pipe = nullptr;       // created on the first large host-pointer call
};

namespace {

void clear_err(bitnuc_err *e) {
    if (e) memset(e, 0, sizeof *e);
}
int fail(bitnuc_err *e, int status, uint64_t value = 0) {
    if (e) { memset(e, 0, sizeof *e); e->status = status; e->value = value; }
    return status;
}
int fail_hip(bitnuc_err *e, hipError_t rc) {
    if (e) { memset(e, 0, sizeof *e); e->status = BITNUC_BACKEND_ERROR; e->backend_code = (int32_t)rc; }
    return BITNUC_BACKEND_ERROR;
}
#define HIPCHK(expr)                                       \
    do {                                                   \
        hipError_t rc__ = (expr);                          \
        if (rc__ != hipSuccess) return fail_hip(err, rc__); \
    } while (0)

struct DeviceGuard { // hipSetDevice is per-thread state: every entry point selects the context's device and restores the caller's
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

int ensure_scratch(bitnuc_ctx *c, int which, size_t bytes, bitnuc_err *err) {
    if (bytes <= c->scratch_cap[which]) return BITNUC_OK;
    const size_t old_cap = c->scratch_cap[which];
    if (c->scratch[which]) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(c->scratch[which]));
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

} // namespace

// ---- pipelined host-pointer path --------------------------------------------------------------------
// A caller's buffers are pageable.  Handing them to hipMemcpyAsync makes the runtime stage them through its own
// pinned bounce buffers on the calling thread, serialising copy-in, kernel and copy-out.  Here the library owns the
// staging: a small worker pool copies chunk c+1 from the caller's memory into one of two pinned input buffers while
// the DMA engines move chunk c (H2D on one stream, D2H on another) and the kernel runs on the context's stream; events
// order the three streams and guard buffer reuse; the host waits only when a pinned buffer is about to be overwritten.
struct CopyPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    uint8_t *dst = nullptr;
    const uint8_t *src = nullptr;
    size_t bytes = 0, slice = 0;
    unsigned generation = 0, pending = 0;
    bool stop = false;
    int n = 1; // workers + the calling thread

    explicit CopyPool(int nthreads) : n(nthreads < 1 ? 1 : nthreads) {
        for (int i = 1; i < n; ++i) threads.emplace_back([this, i] { run(i); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
    int skip = 0; // 1 while an asynchronous copy runs: worker i then owns slice i - 1 (the caller takes none)
    void copy_slice(int i) const {
        const size_t a = slice * (size_t)(i - skip);
        if (a >= bytes) return;
        const size_t m = bytes - a < slice ? bytes - a : slice;
        memcpy(dst + a, src + a, m);
    }
    void run(int i) {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv_work.wait(g, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            copy_slice(i);
            {
                std::lock_guard<std::mutex> g(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    // asynchronous form: the workers copy, the caller goes on (wait() before touching either buffer again).  Pools used this way
    // are created with one thread more than the copies should use: the caller's slice index 0 is then an empty slice.
    bool busy = false;
    void start(void *d, const void *s_, size_t nbytes) {
        wait();
        if (n == 1 || nbytes == 0) { if (nbytes) memcpy(d, s_, nbytes); return; }
        {
            std::lock_guard<std::mutex> g(mu);
            dst = static_cast<uint8_t *>(d);
            src = static_cast<const uint8_t *>(s_);
            bytes = nbytes;
            slice = ((nbytes + (n - 1) - 1) / (n - 1) + 4095) & ~(size_t)4095; // n - 1 workers
            skip = 1;
            pending = (unsigned)(n - 1);
            ++generation;
            busy = true;
        }
        cv_work.notify_all();
    }
    void wait() {
        if (!busy) return;
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return pending == 0; });
        busy = false;
    }
    void copy(void *d, const void *s_, size_t nbytes) { // blocking parallel memcpy
        if (n == 1 || nbytes < ((size_t)1 << 20)) { memcpy(d, s_, nbytes); return; }
        {
            std::lock_guard<std::mutex> g(mu);
            dst = static_cast<uint8_t *>(d);
            src = static_cast<const uint8_t *>(s_);
            bytes = nbytes;
            slice = ((nbytes + n - 1) / n + 4095) & ~(size_t)4095;
            skip = 0;
            pending = (unsigned)(n - 1);
            ++generation;
        }
        cv_work.notify_all();
        copy_slice(0);
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return pending == 0; });
    }
};

constexpr int kPipeDepth = 3; // buffer sets in flight: the host hands chunk c-2 to the caller while chunk c-1 is on the DMA engines and chunk c is staged
struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[kPipeDepth] = {}, ev_k[kPipeDepth] = {}, ev_out[kPipeDepth] = {};
    uint8_t *pin_a[kPipeDepth] = {}, *pin_b[kPipeDepth] = {}; // a: ASCII-sized (chunk + 64), b: word-sized (chunk / 4 + 64)
    uint8_t *dev_a[kPipeDepth] = {}, *dev_b[kPipeDepth] = {};
    CopyPool *pool = nullptr;     // stage-in: blocking, the calling thread takes a slice
    CopyPool *pool_out = nullptr; // hand-back to the caller: asynchronous, overlaps the next chunk's stage-in
    size_t chunk = kPipeChunkDefault; // bases per chunk (a multiple of 32)
    bool ok = false;
};

namespace {

int host_threads() {
    if (const char *e = getenv("BITNUC_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) return v;
    }
    cpu_set_t set;
    int avail = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) avail = CPU_COUNT(&set);
    return avail < 8 ? (avail < 1 ? 1 : avail) : 8;
}

void pipe_destroy(HostPipe *p) {
    if (!p) return;
    for (int i = 0; i < kPipeDepth; ++i) {
        if (p->pin_a[i]) (void)hipHostFree(p->pin_a[i]);
        if (p->pin_b[i]) (void)hipHostFree(p->pin_b[i]);
        if (p->dev_a[i]) (void)hipFree(p->dev_a[i]);
        if (p->dev_b[i]) (void)hipFree(p->dev_b[i]);
        if (p->ev_in[i]) (void)hipEventDestroy(p->ev_in[i]);
        if (p->ev_k[i]) (void)hipEventDestroy(p->ev_k[i]);
        if (p->ev_out[i]) (void)hipEventDestroy(p->ev_out[i]);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p->pool;
    delete p->pool_out;
    delete p;
}

int pipe_get(bitnuc_ctx *c, HostPipe **out, bitnuc_err *err) {
    if (c->pipe && c->pipe->ok) { *out = c->pipe; return BITNUC_OK; }
    HostPipe *p = new HostPipe();
    hipError_t rc = hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking);
    if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking);
    if (const char *e = getenv("BITNUC_PIPE_CHUNK_MB")) {
        const long v = atol(e);
        if (v >= 1 && v <= 1024) p->chunk = (size_t)v << 20;
    }
    const size_t na = p->chunk + 64, nb = p->chunk / 4 + 64;
    for (int i = 0; i < kPipeDepth && rc == hipSuccess; ++i) {
        rc = hipEventCreateWithFlags(&p->ev_in[i], hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&p->ev_k[i], hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&p->ev_out[i], hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void **>(&p->pin_a[i]), na, hipHostMallocDefault);
        if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void **>(&p->pin_b[i]), nb, hipHostMallocDefault);
        if (rc == hipSuccess) rc = hipMalloc(&p->dev_a[i], na);
        if (rc == hipSuccess) rc = hipMalloc(&p->dev_b[i], nb);
    }
    if (rc != hipSuccess) { pipe_destroy(p); return fail_hip(err, rc); }
    p->pool = new CopyPool(host_threads());
    p->pool_out = new CopyPool(host_threads() / 2 + 1); // its workers only: the calling thread is busy staging the next chunk in
    p->ok = true;
    c->pipe = p;
    *out = p;
    return BITNUC_OK;
}

} // namespace

namespace {

// Drain: wait for the stream, find the first latched data error among the pending
// launches (launch order), reset the slots.
int drain(bitnuc_ctx *c, bitnuc_err *err) {
    const int n = c->n_pending;
    // the slot read-back is stream-ordered behind the launches it reports on: one wait covers both
    if (n > 0) HIPCHK(hipMemcpyAsync(c->h_slots, c->d_slots, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n == 0) { clear_err(err); return BITNUC_OK; }
    int hit = -1;
    for (int i = 0; i < n; ++i)
        if (c->h_slots[i] != kNoBad) { hit = i; break; }
    bitnuc_err found;
    memset(&found, 0, sizeof found);
    if (hit >= 0) { // slot = (byte index << 8) | byte, codec_device.h latch_bad
        found.status = BITNUC_INVALID_BASE;
        found.byte = (uint8_t)(c->h_slots[hit] & 0xFF);
        found.index = c->pending[hit].base + (c->h_slots[hit] >> 8);
        HIPCHK(hipMemsetAsync(c->d_slots, 0xFF, sizeof(unsigned long long) * n, c->stream));
    }
    c->n_pending = 0;
    if (err) *err = found;
    return found.status;
}

// Host-pointer (synchronous) calls start from an empty slot ring so that the error they
// return is their own; an InvalidBase latched by earlier asynchronous launches is kept for
// the next bitnuc_ctx_sync().
int flush_pending(bitnuc_ctx *c, bitnuc_err *err) {
    if (c->n_pending == 0) return BITNUC_OK; // nothing asynchronous outstanding: stream order is enough
    bitnuc_err e;
    const int st = drain(c, &e);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    if (st != BITNUC_OK && !c->have_deferred) { c->have_deferred = true; c->deferred = e; }
    return BITNUC_OK;
}

// Reserve the error slot of the next launch (drains implicitly when the ring is full).
int take_slot(bitnuc_ctx *c, unsigned long long base, unsigned long long **slot, bitnuc_err *err) {
    if (c->n_pending == kSlots) {
        bitnuc_err e;
        int st = drain(c, &e);
        if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
        if (st != BITNUC_OK && !c->have_deferred) { c->have_deferred = true; c->deferred = e; }
    }
    c->pending[c->n_pending] = Pending{base};
    *slot = c->d_slots + c->n_pending;
    c->n_pending++;
    return BITNUC_OK;
}

// grid_mult = resident 256-thread workgroups per CU for grid-stride launches (scaled
// for other block sizes); 0 = one tile per workgroup, the hardware dispatcher walks the
// tiles (fastest for the streaming codec: no tail imbalance -- profiles/).
unsigned grid_for(const bitnuc_ctx *c, unsigned long long tiles, int block = kBlock) {
    if (tiles == 0) return 1;
    if (c->grid_mult <= 0) return (unsigned)(tiles < 0x7FFFFFFFull ? tiles : 0x7FFFFFFFull);
    unsigned long long cap = (unsigned long long)c->num_cu * c->grid_mult * kBlock / block;
    if (cap == 0) cap = 1;
    return (unsigned)(tiles < cap ? tiles : cap);
}

inline size_t words_for(size_t n_bases) { return n_bases / 32 + (n_bases % 32 != 0); } // ceil(n/32) without overflow

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Alternative formulations that lost their A/B (profiles/) stay in the source as evidence, but only the evidence build
// (-DBITNUC_SWEEP_VARIANTS, libbitnuc_hip_sweep.so: tools/ and the variant tests) instantiates them; in the product a
// set_variant() to anything but the shipped value returns -2 and changes nothing.
#ifdef BITNUC_SWEEP_VARIANTS
constexpr bool kEvidenceBuild = true;
#else
constexpr bool kEvidenceBuild = false;
#endif

// ---- kernel-variant tables -------------------------------------------------------
// The product library ships the variants that are in use: the tuned defaults (encode 14, decode 22), the plain
// reference shape (0) and the previous default (3).  The other 43 and the lane-per-base ballot formulation are
// measurement evidence (profiles/): they are compiled only with -DBITNUC_SWEEP_VARIANTS, into
// libbitnuc_hip_sweep.so, which tools/sweep*.py and the all-variants parity test load.
//              id  UNROLL BLOCK NTLD   NTST   XPOSE  XCD
#ifdef BITNUC_SWEEP_VARIANTS
#define BITNUC_VARIANTS(X)                            \
    X(0, 4, 256, false, false, false, false)          \
    X(1, 4, 256, true, true, false, false)            \
    X(2, 2, 256, true, true, false, false)            \
    X(3, 2, 256, true, false, false, false)           \
    X(4, 2, 256, false, false, false, false)          \
    X(5, 4, 256, true, false, false, false)           \
    X(6, 1, 256, true, false, false, false)           \
    X(7, 2, 512, true, false, false, false)           \
    X(8, 2, 1024, true, false, false, false)          \
    X(9, 4, 256, false, false, true, false)           \
    X(10, 4, 256, true, true, true, false)            \
    X(11, 4, 256, true, false, true, false)           \
    X(12, 2, 256, true, false, false, true)           \
    X(13, 4, 256, false, false, false, true)          \
    X(14, 2, 128, true, false, false, false)          \
    X(15, 8, 256, true, false, false, false)          \
    X(16, 4, 512, false, false, false, false)         \
    X(17, 2, 512, false, false, false, false)         \
    X(18, 4, 256, false, true, false, false)          \
    X(19, 4, 512, true, false, true, false)           \
    X(20, 1, 256, false, false, false, false)         \
    X(21, 1, 512, true, false, false, false)          \
    X(22, 2, 256, false, true, false, false)          \
    X(23, 4, 1024, false, false, false, false)        \
    X(24, 2, 256, false, false, false, true)          \
    X(25, 4, 256, true, false, false, true)           \
    X(26, 8, 256, false, false, false, true)          \
    X(27, 4, 512, false, false, false, true)          \
    X(28, 4, 256, false, false, true, true)           \
    X(29, 1, 256, false, false, false, true)          \
    X(30, 2, 256, true, true, false, true)            \
    X(31, 4, 256, true, true, false, true)            \
    X(32, 8, 256, true, true, false, false)           \
    X(33, 4, 512, true, true, false, false)           \
    X(34, 4, 256, false, true, false, true)           \
    X(35, 4, 128, true, true, false, false)           \
    X(36, 2, 64, true, false, false, false)           \
    X(37, 4, 128, true, false, false, false)          \
    X(38, 1, 128, true, false, false, false)          \
    X(39, 2, 128, true, true, false, true)            \
    X(40, 4, 128, true, true, false, true)            \
    X(41, 2, 512, true, true, false, true)            \
    X(42, 1, 256, true, true, false, true)            \
    X(43, 2, 128, true, false, false, true)           \
    X(44, 1, 128, true, true, false, true)            \
    X(45, 1, 512, true, true, false, true)            \
    X(46, 1, 1024, true, true, false, true)
#else
#define BITNUC_VARIANTS(X)                            \
    X(0, 4, 256, false, false, false, false)          \
    X(3, 2, 256, true, false, false, false)           \
    X(22, 2, 256, false, true, false, false)          \
    X(39, 2, 128, true, true, false, true)
#endif
constexpr int kNumVariants = 47;    // ids 0..46; which of them this build holds: variant_info(id).built
[[maybe_unused]] constexpr int kBallotVariant = 100; // encode only: lane-per-base + ballot (sweep build; set_variant("encode", 100))
// defaults from the sustained (back-to-back) pair sweeps in profiles/ (10^9 bases, one tile per
// workgroup, interleaved rounds in one process, decode reading words written two steps
// earlier so that none of its input is Infinity-Cache resident -- what bench.py times):
//   encode 39: nt loads + nt stores, 2 groups in flight per lane, 128-thread workgroups, XCD-contiguous tile order
//   decode 22: plain loads + nt stores, 2 groups per lane
// The pair is tuned, not each kernel, and every good pair lands on the same plateau of ~0.40 ms per step = 6.3 TB/s of
// mixed read/write HBM traffic (all 47 x 47 pairs: profiles/r02_sweep_pairs_all_cold.txt; the ten best are within 0.6 %).
// What differs is how the step divides: with encode 14 (plain, allocating stores -- the round-1 default) the 250 MB of
// packed words sit dirty in the 256 MiB Infinity Cache and are written back while the DECODE runs: encode 0.182 ms, decode
// 0.214 ms.  With nt stores the encode pays for its own writes: 0.199-0.204 / 0.194 ms; the step is the same (five processes each:
// 0.4019 vs 0.4024 ms, profiles/r02_encode_variant_stability.txt): a choice of attribution, not of speed.
constexpr int kDefaultEnc = 39, kDefaultDec = 22;

struct VariantInfo { int unroll, block; bool ntld, ntst, xpose, xcd, built; };
constexpr VariantInfo variant_info(int id) {
    switch (id) {
#define X(vid, U, B, NL, NS, XP, XC) case vid: return VariantInfo{U, B, NL, NS, XP, XC, true};
        BITNUC_VARIANTS(X)
#undef X
    default: return VariantInfo{0, 0, false, false, false, false, false};
    }
}

template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool XPOSE, bool XCD>
hipError_t launch_encode_t(bitnuc_ctx *c, const uint8_t *seq, uint32_t *out32, unsigned long long len,
                           unsigned long long *slot, bool al) {
    const unsigned long long tile = (unsigned long long)BLOCK * UNROLL;
    const unsigned grid = grid_for(c, (len >> 4) / tile + 1, BLOCK);
    if (al) encode_kernel<UNROLL, BLOCK, NTLD, NTST, true, XPOSE, XCD><<<grid, BLOCK, 0, c->stream>>>(seq, out32, len, slot);
    else if constexpr (!XPOSE) encode_kernel<UNROLL, BLOCK, NTLD, NTST, false, false, XCD><<<grid, BLOCK, 0, c->stream>>>(seq, out32, len, slot);
    return hipGetLastError();
}

hipError_t launch_encode(bitnuc_ctx *c, const uint8_t *seq, uint64_t *out, unsigned long long len, unsigned long long *slot) {
    uint32_t *o = reinterpret_cast<uint32_t *>(out);
    const bool in_al = aligned16(seq), out_al = aligned16(out);
#ifdef BITNUC_SWEEP_VARIANTS
    if (c->enc_variant == kBallotVariant) { // lane-per-base + ballot formulation (evidence variant)
        const unsigned grid = grid_for(c, ((len + 63) / 64 + (kBlock / 64) * 4 - 1) / ((kBlock / 64) * 4));
        encode_ballot_kernel<4><<<grid, kBlock, 0, c->stream>>>(seq, reinterpret_cast<unsigned long long *>(out), len, slot);
        return hipGetLastError();
    }
#endif
    int v = c->enc_variant;
    // the LDS-transpose variant needs 16-byte aligned buffers on both sides
    if (variant_info(v).xpose && !(in_al && out_al)) v = kDefaultEnc;
    switch (v) {
#define X(id, U, B, NL, NS, XP, XC) \
    case id: return launch_encode_t<U, B, NL, NS, XP, XC>(c, seq, o, len, slot, XP ? true : in_al);
        BITNUC_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool XPOSE, bool XCD>
hipError_t launch_decode_t(bitnuc_ctx *c, const uint32_t *in32, uint8_t *out, unsigned long long n_bases, bool al) {
    const unsigned long long tile = (unsigned long long)BLOCK * UNROLL;
    const unsigned grid = grid_for(c, (n_bases >> 4) / tile + 1, BLOCK);
    if (al) decode_kernel<UNROLL, BLOCK, NTLD, NTST, true, XPOSE, XCD><<<grid, BLOCK, 0, c->stream>>>(in32, out, n_bases);
    else decode_kernel<UNROLL, BLOCK, NTLD, NTST, false, XPOSE, XCD><<<grid, BLOCK, 0, c->stream>>>(in32, out, n_bases);
    return hipGetLastError();
}

// decode variants 47..54: decode_x2_kernel (8-byte loads + LDS transpose) for the whole 2 KiB wave tiles, the default
// decode_kernel for what is left.  id - 47: bit 0 = nt loads, bit 1 = plain (not nt) stores, bit 2 = 2 words in flight per lane.
constexpr int kX2First = 47, kX2Last = 54;
template <int UNROLL>
hipError_t launch_decode_x2_t(bitnuc_ctx *c, int mode, const unsigned long long *w, uint8_t *out, unsigned long long tiles) {
    constexpr int B = 256;
    const unsigned long long per = (unsigned long long)(B / 64) * UNROLL;
    const unsigned grid = (unsigned)((tiles + per - 1) / per);
    switch (mode & 3) {
    case 0: decode_x2_kernel<B, UNROLL, false, true><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    case 1: decode_x2_kernel<B, UNROLL, true, true><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    case 2: decode_x2_kernel<B, UNROLL, false, false><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    default: decode_x2_kernel<B, UNROLL, true, false><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    }
    return hipGetLastError();
}

hipError_t launch_decode(bitnuc_ctx *c, const uint64_t *ebuf, uint8_t *out, unsigned long long n_bases) {
    const bool in_al = aligned16(ebuf), out_al = aligned16(out);
    if constexpr (kEvidenceBuild) if (c->dec_variant >= kX2First && c->dec_variant <= kX2Last && out_al) {
        const unsigned long long tiles = n_bases >> 11; // whole 2 KiB (64-word) wave tiles
        if (tiles) {
            const int mode = c->dec_variant - kX2First;
            const unsigned long long *w = reinterpret_cast<const unsigned long long *>(ebuf);
            const hipError_t rc = (mode & 4) ? launch_decode_x2_t<2>(c, mode, w, out, tiles) : launch_decode_x2_t<1>(c, mode, w, out, tiles);
            if (rc != hipSuccess) return rc;
        }
        const unsigned long long done = tiles << 11;
        if (done == n_bases) return hipSuccess;
        return launch_decode_t<2, 256, false, true, false, false>(c, reinterpret_cast<const uint32_t *>(ebuf) + (done >> 4), out + done, n_bases - done, true);
    }
    const uint32_t *i = reinterpret_cast<const uint32_t *>(ebuf);
    int v = c->dec_variant;
    if (v >= kX2First) v = kDefaultDec; // x2 asked for an unaligned output: the default kernel handles any alignment
    if (variant_info(v).xpose && !in_al) v = kDefaultDec;
    switch (v) {
#define X(id, U, B, NL, NS, XP, XC) \
    case id: return launch_decode_t<U, B, NL, NS, XP, XC>(c, i, out, n_bases, out_al);
        BITNUC_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_batch(bitnuc_ctx *c, const uint8_t *kmers, size_t k, size_t stride, size_t count, uint64_t *out,
                        unsigned long long *slot) {
    unsigned long long *o = reinterpret_cast<unsigned long long *>(out);
    size_t done = 0;
    if (stride == k && count >= 64 && c->batch_dense) {
        // dense layout: whole waves of 64 k-mers go through the bulk-encode-shaped kernel
        const unsigned long long items = count / 64;
        const int un = c->dense_unroll, kb = c->kmer_block;
        const unsigned grid = grid_for(c, (items + (kb / 64) * un - 1) / ((kb / 64) * un), kb);
#define DENSE_LAUNCH(AL, NL, NS, U) kmer_dense_kernel<AL, NL, NS, U><<<grid, kb, 0, c->stream>>>(kmers, (unsigned)k, items, o, slot)
#define DENSE_POLICY(U)                                              \
    switch (c->dense_policy) { /* bit0: nt loads, bit1: nt stores */ \
    case 0: DENSE_LAUNCH(true, false, false, U); break;              \
    case 1: DENSE_LAUNCH(true, true, false, U); break;               \
    case 2: DENSE_LAUNCH(true, false, true, U); break;               \
    default: DENSE_LAUNCH(true, true, true, U); break;               \
    }
        if (!aligned16(kmers)) { DENSE_LAUNCH(false, false, false, 1); }
        else if constexpr (!kEvidenceBuild) { DENSE_LAUNCH(true, true, true, 1); } // the shipped form: dense_policy 3, dense_unroll 1
        else if (un == 1) { DENSE_POLICY(1) }
        else if (un == 2) { DENSE_POLICY(2) }
        else { DENSE_POLICY(4) }
#undef DENSE_POLICY
#undef DENSE_LAUNCH
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = items * 64;
        if (done == count) return hipSuccess;
    }
    // (k >= stride: every byte of the span belongs to some k-mer, so validating whole 16-byte groups examines no byte the
    // reference's loop would not; with gaps between k-mers the general kernel looks at each k-mer's own bytes only)
    if ((stride == 1 || stride == 2 || stride == 4 || stride == 8 || stride == 16) && k >= stride && done == 0 && c->batch_slide &&
        aligned16(kmers) && aligned16(out) && (count - 1) * stride + k >= 1024) {
        // windows at a small power-of-two stride (1 = every window of a sequence): whole 1 KiB wave rounds through the
        // sliding kernel, 992 / stride windows each; the round that would read past the batch's last byte is left over
        const unsigned long long rounds = ((count - 1) * stride + k - 1024) / kScanWaveWindows + 1;
        const unsigned long long per_wave = (unsigned long long)c->slide_rounds;
        const unsigned grid = grid_for(c, (rounds + per_wave * (kBlock / 64) - 1) / (per_wave * (kBlock / 64)));
        const bool nts = (c->dense_policy & 2) != 0;
#define SLIDE_U(S, NT) do { if constexpr (kEvidenceBuild) { \
                              if (per_wave == 2) { kmer_slide_kernel<S, NT, 2><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); break; } \
                              if (per_wave == 4) { kmer_slide_kernel<S, NT, 4><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); break; } \
                              if (per_wave == 8) { kmer_slide_kernel<S, NT, 8><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); break; } } \
                            kmer_slide_kernel<S, NT, 1><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); } while (0)
#define SLIDE(S) do { if (nts) SLIDE_U(S, true); else SLIDE_U(S, false); } while (0)
        switch (stride) {
        case 1: SLIDE(1); break;
        case 2: SLIDE(2); break;
        case 4: SLIDE(4); break;
        case 8: SLIDE(8); break;
        default: SLIDE(16); break;
        }
#undef SLIDE
#undef SLIDE_U
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = (size_t)(rounds * (kScanWaveWindows / stride));
        if (done >= count) return hipSuccess;
    }
    if (stride >= 3 && stride < 32 && k >= stride && done == 0 && c->batch_slide && aligned16(kmers) &&
        (count - 1) * stride + k >= 1024) {
        // any other small stride with overlapping k-mers: the sliding round with per-lane window selection
        const unsigned long long rounds = ((count - 1) * stride + k - 1024) / kScanWaveWindows + 1;
        const unsigned grid = grid_for(c, (rounds + kBlock / 64 - 1) / (kBlock / 64));
        const unsigned magic = (unsigned)((0x100000000ull + stride - 1) / stride); // exact floor(t / stride) for t < 2^16
        const unsigned long long magic64 = ~0ull / stride + 1; // stride >= 3: no overflow
        kmer_slide_any_kernel<<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, (unsigned)stride, magic, magic64, rounds, o, slot);
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = (size_t)((rounds * kScanWaveWindows + stride - 1) / stride); // k-mers that start before the last round's end
        if (done >= count) return hipSuccess;
    }
    // general strides, and the < 64 k-mers a dense batch leaves over.  The error slot holds
    // byte offsets relative to `kmers`, so the leftover launch passes the offset it starts at.
    const size_t rest = count - done;
    const unsigned grid = grid_for(c, (rest + kBlock - 1) / kBlock);
    if (stride <= (size_t)kStagedMaxStride)
        kmer_batch_kernel<true><<<grid, kBlock, 0, c->stream>>>(kmers + done * stride, (unsigned)k, stride, rest, o + done, slot, done * stride);
    else
        kmer_batch_kernel<false><<<grid, kBlock, 0, c->stream>>>(kmers + done * stride, (unsigned)k, stride, rest, o + done, slot, done * stride);
    return hipGetLastError();
}

// de-interleave a packed query into its two bit-planes (bit i = low / high code bit of base i)
void query_planes(uint64_t query, size_t k, uint32_t *ql, uint32_t *qh) {
    *ql = *qh = 0;
    for (unsigned i = 0; i < k; ++i) {
        *ql |= (uint32_t)((query >> (2 * i)) & 1) << i;
        *qh |= (uint32_t)((query >> (2 * i + 1)) & 1) << i;
    }
}

hipError_t launch_scan(bitnuc_ctx *c, const uint8_t *ref, size_t n, size_t k, uint64_t query, uint8_t *dist,
                       unsigned long long *slot) {
    uint32_t ql, qh;
    query_planes(query, k, &ql, &qh);
    const int unroll = c->scan_unroll, kb = c->kmer_block;
    const bool al = aligned16(ref) && aligned16(dist);
    if (c->scan_impl == 1 && al) { // line-aligned rounds of 1024 windows
        const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
        const unsigned grid = grid_for(c, rounds / ((kb / 64) * unroll) + 1, kb);
#define SCAN2(NL, NS, U) kmer_scan2_kernel<true, NL, NS, U, false><<<grid, kb, 0, c->stream>>>(ref, n, (unsigned)k, query, ql, qh, 0u, dist, nullptr, nullptr, nullptr, slot)
#define SCAN2_POLICY(U)                                              \
    switch (c->scan_policy) { /* bit0: nt loads, bit1: nt stores */ \
    case 0: SCAN2(false, false, U); break;                           \
    case 1: SCAN2(true, false, U); break;                            \
    case 2: SCAN2(false, true, U); break;                            \
    default: SCAN2(true, true, U); break;                            \
    }
        if constexpr (!kEvidenceBuild) { SCAN2(true, true, 4); } // the shipped form: scan_policy 3, scan_unroll 4
        else if (unroll == 1) { SCAN2_POLICY(1) } else if (unroll == 2) { SCAN2_POLICY(2) } else { SCAN2_POLICY(4) }
#undef SCAN2_POLICY
#undef SCAN2
        return hipGetLastError();
    }
    const unsigned long long rounds = n >= 1024 ? (n - 1024) / kScanWaveWindows + 1 : 0;
    const unsigned grid = grid_for(c, rounds / ((kb / 64) * unroll) + 1, kb);
#define SCAN_LAUNCH(AL, NL, NS, U) \
    kmer_scan_kernel<AL, NL, NS, U><<<grid, kb, 0, c->stream>>>(ref, n, (unsigned)k, query, ql, qh, dist, slot)
#define SCAN_POLICY(U)                                             \
    switch (c->scan_policy) { /* bit0: nt loads, bit1: nt stores */ \
    case 0: SCAN_LAUNCH(true, false, false, U); break;             \
    case 1: SCAN_LAUNCH(true, true, false, U); break;              \
    case 2: SCAN_LAUNCH(true, false, true, U); break;              \
    default: SCAN_LAUNCH(true, true, true, U); break;              \
    }
    if (!al || !kEvidenceBuild) { // the product reaches this only for unaligned pointers
        SCAN_LAUNCH(false, false, false, 1);
    } else if constexpr (kEvidenceBuild) {
        if (unroll == 1) { SCAN_POLICY(1) } else if (unroll == 2) { SCAN_POLICY(2) } else { SCAN_POLICY(4) }
    }
#undef SCAN_POLICY
#undef SCAN_LAUNCH
    return hipGetLastError();
}

// shared argument checks --------------------------------------------------------------
int check_ctx(bitnuc_ctx *c, bitnuc_err *err) {
    if (!c) return fail(err, BITNUC_UNSUPPORTED);
    return BITNUC_OK;
}

} // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char *bitnuc_version(void) { return "bitnuc_hip 0.1.0 gfx950"; }

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
    if (rc == hipSuccess) rc = hipMalloc(&c->d_slots, sizeof(unsigned long long) * kSlots);
    if (rc == hipSuccess) rc = hipHostMalloc(reinterpret_cast<void **>(&c->h_slots), sizeof(unsigned long long) * kSlots, hipHostMallocDefault);
    if (rc == hipSuccess) rc = hipMalloc(&c->d_sink, 64);
    if (rc == hipSuccess) rc = hipMemset(c->d_slots, 0xFF, sizeof(unsigned long long) * kSlots);
    if (rc == hipSuccess) rc = hipMemset(c->d_sink, 0, 64);
    if (const char *e = getenv("BITNUC_FORCE_GPU")) c->force_gpu = atoi(e) != 0;
    if (const char *e = getenv("BITNUC_HOST_CUTOFF")) { const long long v = atoll(e); if (v >= 0) c->host_cutoff = c->host_cutoff_decode = (size_t)v; }
    c->reduce_blocks = (unsigned)c->num_cu * 2; // a resident grid of 2 workgroups of 256 threads per CU (profiles/r01_sweep13_reduce_grid.txt: the tail of atomics + ticket grows with the grid)
    if (rc == hipSuccess) rc = hipMalloc(&c->d_acc, 64);
    if (rc == hipSuccess) rc = hipMemset(c->d_acc, 0, 64);
    if (rc == hipSuccess) rc = hipMalloc(&c->d_tickets, 64);
    if (rc == hipSuccess) rc = hipMemset(c->d_tickets, 0, 64);
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
    for (int i = 0; i < 6; ++i)
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    if (c->d_slots) (void)hipFree(c->d_slots);
    if (c->h_slots) (void)hipHostFree(c->h_slots);
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
    int st = drain(c, &e);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    if (c->have_deferred) { // an earlier implicit drain saw an error first
        c->have_deferred = false;
        e = c->deferred;
        st = e.status;
    }
    if (err) *err = e;
    return st;
}

int bitnuc_ctx_set_variant(bitnuc_ctx *c, const char *key, int value) {
    if (!c || !key) return -1;
    int prev = -1;
    if (!strcmp(key, "encode")) {
        prev = c->enc_variant;
        if (value >= 0) { // a variant this build does not hold is refused: -2, nothing changes
#ifdef BITNUC_SWEEP_VARIANTS
            if (value == kBallotVariant) { c->enc_variant = value; return prev; }
#endif
            if (!variant_info(value).built) return -2;
            c->enc_variant = value;
        }
    }
    else if (!strcmp(key, "decode")) { prev = c->dec_variant; if (value >= 0) { if (!(kEvidenceBuild && value >= kX2First && value <= kX2Last) && !variant_info(value).built) return -2; c->dec_variant = value; } }
    else if (!strcmp(key, "force_gpu")) { prev = c->force_gpu; if (value == 0 || value == 1) c->force_gpu = value; }
    else if (!strcmp(key, "host_cutoff")) { prev = (int)(c->host_cutoff > 0x7FFFFFFF ? 0x7FFFFFFF : c->host_cutoff); if (value >= 0) c->host_cutoff = c->host_cutoff_decode = (size_t)value; } // sets both
    else if (!strcmp(key, "host_cutoff_decode")) { prev = (int)(c->host_cutoff_decode > 0x7FFFFFFF ? 0x7FFFFFFF : c->host_cutoff_decode); if (value >= 0) c->host_cutoff_decode = (size_t)value; }
    else if (!strcmp(key, "host_pipeline")) { prev = c->host_pipeline; if (value == 0 || value == 1) c->host_pipeline = value; }
    else if (!strcmp(key, "sweep_build")) {
#ifdef BITNUC_SWEEP_VARIANTS
        prev = 1;
#else
        prev = 0;
#endif
    }
    else if (!strcmp(key, "grid_mult")) { prev = c->grid_mult; if (value >= 0 && value <= 64) c->grid_mult = value; }
    else if (!strcmp(key, "batch_dense")) { prev = c->batch_dense; if (value >= 0 && value <= 1) c->batch_dense = value; }
    else if (!strcmp(key, "batch_slide")) { prev = c->batch_slide; if (value >= 0 && value <= 1) c->batch_slide = value; }
    else if (!strcmp(key, "dense_policy")) { prev = c->dense_policy; if (value >= 0 && !kEvidenceBuild && value != 3) return -2; if (value >= 0 && value <= 3) c->dense_policy = value; }
    else if (!strcmp(key, "scan_policy")) { prev = c->scan_policy; if (value >= 0 && !kEvidenceBuild && value != 3) return -2; if (value >= 0 && value <= 3) c->scan_policy = value; }
    else if (!strcmp(key, "fixed_stream")) { prev = c->fixed_stream; if (value == 0 || value == 1) c->fixed_stream = value; }
    else if (!strcmp(key, "fixed_dec_strip")) { prev = c->fixed_dec_strip; if (value >= 0 && !kEvidenceBuild && value != 2) return -2; if (value >= 0 && value <= 2) c->fixed_dec_strip = value; }
    else if (!strcmp(key, "owner_est")) { prev = c->owner_est; if (value >= 0 && value <= 3) c->owner_est = value; }
    else if (!strcmp(key, "batch_host_plan")) { prev = c->batch_host_plan; if (value == 0 || value == 1) c->batch_host_plan = value; }
    else if (!strcmp(key, "plan_tiles")) { prev = c->plan_tiles; if (value >= 0 && !kEvidenceBuild && value != 1) return -2; if (value == 1 || value == 2 || value == 4) c->plan_tiles = value; }
    else if (!strcmp(key, "plan_enc_abl")) { prev = c->plan_enc_abl; if (value >= 0 && !kEvidenceBuild && value != 0) return -2; if (value >= 0 && value <= 7) c->plan_enc_abl = value; }
    else if (!strcmp(key, "plan_enc_block")) { prev = c->plan_enc_block; if (value == 64 || value == 128 || value == 256) c->plan_enc_block = value; }
    else if (!strcmp(key, "plan_enc_tiles")) { prev = c->plan_enc_tiles; if (value >= 0 && !kEvidenceBuild && value != 1) return -2; if (value == 1 || value == 2 || value == 4) c->plan_enc_tiles = value; }
    else if (!strcmp(key, "slide_rounds")) { prev = c->slide_rounds; if (value >= 0 && !kEvidenceBuild && value != 1) return -2; if (value == 1 || value == 2 || value == 4 || value == 8) c->slide_rounds = value; }
    else if (!strcmp(key, "plan_store")) { prev = c->plan_store; if (value >= 0 && !kEvidenceBuild && value != 2) return -2; if (value >= 0 && value <= 2) c->plan_store = value; }
    else if (!strcmp(key, "batch_abl")) {
        prev = c->batch_abl;
#ifdef BITNUC_SWEEP_VARIANTS
        if (value >= 0 && value <= 15) c->batch_abl = value;
#else
        if (value > 0) return -2; // ablated kernels exist in the evidence build only
#endif
    }
    else if (!strcmp(key, "kmer_block")) { prev = c->kmer_block; if (value == 64 || value == 128 || value == 256) c->kmer_block = value; }
    else if (!strcmp(key, "dense_unroll")) { prev = c->dense_unroll; if (value >= 0 && !kEvidenceBuild && value != 1) return -2; if (value == 1 || value == 2 || value == 4) c->dense_unroll = value; }
    else if (!strcmp(key, "scan_impl")) { prev = c->scan_impl; if (value >= 0 && !kEvidenceBuild && value != 1) return -2; if (value == 0 || value == 1) c->scan_impl = value; }
    else if (!strcmp(key, "scan_unroll")) { prev = c->scan_unroll; if (value >= 0 && !kEvidenceBuild && value != 4) return -2; if (value == 1 || value == 2 || value == 4) c->scan_unroll = value; }
    else if (!strcmp(key, "reduce_mult")) { prev = (int)(c->reduce_blocks / (unsigned)c->num_cu); if (value >= 1 && value <= 32) c->reduce_blocks = (unsigned)c->num_cu * (unsigned)value; }
    else if (!strcmp(key, "num_variants")) { prev = kNumVariants; }
    else if (!strcmp(key, "num_cu")) { prev = c->num_cu; }
    return prev;
}

// ---- device-pointer entry points ---------------------------------------------------------
int bitnuc_encode_dev(bitnuc_ctx *c, const uint8_t *d_seq, size_t len, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (len == 0) return BITNUC_OK; // 0 words (the reference panics: packing/avx.rs:138)
    if (!d_seq || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    HIPCHK(launch_encode(c, d_seq, d_out, len, slot));
    return BITNUC_OK;
}

int bitnuc_decode_dev(bitnuc_ctx *c, const uint64_t *d_ebuf, size_t n_words, size_t n_bases, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    // unpacking/mod.rs:40-45: missing words -> InvalidLength(n_bases)
    if (n_words < words_for(n_bases)) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (n_bases == 0) return BITNUC_OK; // unpacking/avx.rs:134-145: nothing appended
    if (!d_ebuf || !d_out || (reinterpret_cast<uintptr_t>(d_ebuf) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    HIPCHK(launch_decode(c, d_ebuf, d_out, n_bases));
    return BITNUC_OK;
}

int bitnuc_as_2bit_batch_dev(bitnuc_ctx *c, const uint8_t *d_kmers, size_t k, size_t stride, size_t count, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0) return BITNUC_OK;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k); // packing/naive.rs:5-7, before any base
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 7) || stride == 0) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (k == 0) { // as_2bit(b"") == Ok(0)
        HIPCHK(hipMemsetAsync(d_out, 0, sizeof(uint64_t) * count, c->stream));
        return BITNUC_OK;
    }
    if (!d_kmers) return fail(err, BITNUC_UNSUPPORTED);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    HIPCHK(launch_batch(c, d_kmers, k, stride, count, d_out, slot));
    return BITNUC_OK;
}

int bitnuc_kmer_hdist_scan_dev(bitnuc_ctx *c, const uint8_t *d_ref, size_t n, size_t k, uint64_t query, uint8_t *d_dist, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (k == 0 || n < k) return BITNUC_OK; // no windows
    if (!d_ref || !d_dist) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    HIPCHK(launch_scan(c, d_ref, n, k, query, d_dist, slot));
    return BITNUC_OK;
}

int bitnuc_kmer_hdist_count_dev(bitnuc_ctx *c, const uint8_t *d_ref, size_t n, size_t k, uint64_t query, unsigned tau, uint64_t *d_count, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (!d_count || (reinterpret_cast<uintptr_t>(d_count) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (k == 0 || n < k) { // no windows
        HIPCHK(hipMemsetAsync(d_count, 0, sizeof(uint64_t), c->stream));
        return BITNUC_OK;
    }
    if (!d_ref) return fail(err, BITNUC_UNSUPPORTED);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    uint32_t ql, qh;
    query_planes(query, k, &ql, &qh);
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
    // a resident grid (the accumulator's ticket needs every workgroup to arrive; 4 trips of 4 rounds per wave keep the tail short)
    const unsigned long long want = rounds / ((kBlock / 64) * 4) + 1, cap = (unsigned long long)c->num_cu * 8;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    unsigned long long *res = reinterpret_cast<unsigned long long *>(d_count);
    if (aligned16(d_ref)) kmer_scan2_kernel<true, true, false, 4, true><<<grid, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, ql, qh, tau, nullptr, res, c->d_acc + 5, c->d_tickets + 2, slot);
    else kmer_scan2_kernel<false, false, false, 1, true><<<grid, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, ql, qh, tau, nullptr, res, c->d_acc + 5, c->d_tickets + 2, slot);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_hdist_dev(bitnuc_ctx *c, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, size_t n_bases, uint32_t *d_result, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    const size_t need = words_for(n_bases);
    if (na < need || nb < need) return fail(err, BITNUC_INVALID_LENGTH, n_bases); // hamming/multi.rs:124-127
    if (!d_result) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (n_bases == 0) {
        HIPCHK(hipMemsetAsync(d_result, 0, sizeof(uint32_t), c->stream));
        return BITNUC_OK;
    }
    if (!d_a || !d_b) return fail(err, BITNUC_UNSUPPORTED);
    const unsigned long long tiles = (n_bases / 32) / (kBlock * 2) + 1;
    const unsigned grid = (unsigned)(tiles < c->reduce_blocks ? tiles : c->reduce_blocks);
    hdist_kernel<<<grid, kBlock, 0, c->stream>>>(reinterpret_cast<const unsigned long long *>(d_a),
                                                 reinterpret_cast<const unsigned long long *>(d_b), n_bases, d_result, reinterpret_cast<unsigned *>(c->d_acc + 4), c->d_tickets + 1);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_nucgen_dev(bitnuc_ctx *c, uint8_t *d_out, size_t len, uint64_t seed, uint64_t first, int flags, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (len == 0) return BITNUC_OK;
    if (!d_out) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned grid = grid_for(c, ((len + 15) / 16 + kBlock - 1) / kBlock);
    nucgen_kernel<<<grid, kBlock, 0, c->stream>>>(d_out, len, seed, first, flags);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_stream_probe_dev(bitnuc_ctx *c, int mode, const void *d_src, void *d_dst, size_t bytes, bitnuc_err *err) {
    // mode: bits 0-2 = 0 read / 1 copy / 2 fill; bit 3 = nt loads; bit 4 = nt stores; bit 5 = 2 (not 4) groups per lane
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    const unsigned long long n16 = bytes / 16;
    const bool ntl = (mode & 8) != 0, nts = (mode & 16) != 0, u2 = (mode & 32) != 0;
    const unsigned grid = grid_for(c, n16 / (kBlock * (u2 ? 2 : 4)) + 1);
    const u32x4 *src = static_cast<const u32x4 *>(d_src);
    u32x4 *dst = static_cast<u32x4 *>(d_dst);
#define PROBE(K, ...) K<<<grid, kBlock, 0, c->stream>>>(__VA_ARGS__)
    switch (mode & 7) {
    case 0:
        if (!d_src || !aligned16(d_src)) return fail(err, BITNUC_UNSUPPORTED);
        if (u2) { if (ntl) PROBE((probe_read_kernel<2, true>), src, n16, c->d_sink); else PROBE((probe_read_kernel<2, false>), src, n16, c->d_sink); }
        else { if (ntl) PROBE((probe_read_kernel<4, true>), src, n16, c->d_sink); else PROBE((probe_read_kernel<4, false>), src, n16, c->d_sink); }
        break;
    case 1:
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        if (ntl && nts) PROBE((probe_copy_kernel<4, true, true>), src, dst, n16);
        else if (ntl) PROBE((probe_copy_kernel<4, true, false>), src, dst, n16);
        else if (nts) PROBE((probe_copy_kernel<4, false, true>), src, dst, n16);
        else PROBE((probe_copy_kernel<4, false, false>), src, dst, n16);
        break;
    case 2:
        if (!d_dst || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        if (nts) PROBE((probe_fill_kernel<4, true>), dst, n16); else PROBE((probe_fill_kernel<4, false>), dst, n16);
        break;
    case 3: { // encode_kernel's shape (variant 39: 2 rounds, 128 threads, nt loads, nt stores, XCD-contiguous tiles): `bytes` of ASCII-side input
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        const unsigned g3 = grid_for(c, n16 / (128 * 2) + 1, 128);
        probe_enc_shape_kernel<2, 128, true, true, true><<<g3, 128, 0, c->stream>>>(src, static_cast<uint32_t *>(d_dst), n16);
        break;
    }
    case 4: { // decode_kernel's shape (variant 22: 2 rounds, 256 threads, plain loads, nt stores): `bytes` of ASCII-side output
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        const unsigned g4 = grid_for(c, n16 / (256 * 2) + 1, 256);
        probe_dec_shape_kernel<2, 256, false, true><<<g4, 256, 0, c->stream>>>(static_cast<const uint32_t *>(d_src), dst, n16);
        break;
    }
    default:
        return fail(err, BITNUC_UNSUPPORTED);
    }
#undef PROBE
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

// ---- host-pointer entry points (synchronous, staged through device scratch) -------------------
// true when a bulk host-pointer call of n bases belongs on the host: below the cutoff and not forced to the GPU.
// A NULL context is accepted for such calls (the reference's functions need no context either).
static inline bool on_host(const bitnuc_ctx *c, size_t n, bool decode = false) {
    if (c) return !c->force_gpu && n < (decode ? c->host_cutoff_decode : c->host_cutoff);
    return n < (decode ? kDefaultHostCutoffDecode : kDefaultHostCutoff);
}

// encode / decode of a large pageable buffer: pinned double buffers, three streams (see HostPipe)
static int encode_pipelined(bitnuc_ctx *c, const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    struct HandBackDone { CopyPool *q; ~HandBackDone() { q->wait(); } } hand_back_done{p->pool_out}; // no return leaves workers writing into `out`
    const size_t kPipeChunk = p->chunk;
    const size_t nchunks = (len + kPipeChunk - 1) / kPipeChunk;
    auto chunk_len = [&](size_t ci) { return len - ci * kPipeChunk < kPipeChunk ? len - ci * kPipeChunk : kPipeChunk; };
    constexpr int D = kPipeDepth, LAG = kPipeDepth - 1;
    for (size_t ci = 0; ci < nchunks + LAG; ++ci) {
        const int b = (int)(ci % D);
        if (ci < nchunks) {
            const size_t n = chunk_len(ci), nw = words_for(n);
            if (ci >= (size_t)D) HIPCHK(hipEventSynchronize(p->ev_in[b])); // pinned input b: its previous H2D has left
            p->pool->copy(p->pin_a[b], seq + ci * kPipeChunk, n);
            if (ci >= (size_t)D) HIPCHK(hipStreamWaitEvent(p->s_in, p->ev_k[b], 0)); // device input b: the kernel of chunk ci-D has read it
            HIPCHK(hipMemcpyAsync(p->dev_a[b], p->pin_a[b], n, hipMemcpyHostToDevice, p->s_in));
            HIPCHK(hipEventRecord(p->ev_in[b], p->s_in));
            HIPCHK(hipStreamWaitEvent(c->stream, p->ev_in[b], 0));
            if (ci >= (size_t)D) HIPCHK(hipStreamWaitEvent(c->stream, p->ev_out[b], 0)); // device output b: its D2H of chunk ci-D is done
            unsigned long long *slot;
            if (int st = take_slot(c, ci * kPipeChunk, &slot, err)) return st;
            HIPCHK(launch_encode(c, p->dev_a[b], reinterpret_cast<uint64_t *>(p->dev_b[b]), n, slot));
            HIPCHK(hipEventRecord(p->ev_k[b], c->stream));
            HIPCHK(hipStreamWaitEvent(p->s_out, p->ev_k[b], 0));
            p->pool_out->wait(); // pinned output b is being handed to the caller since the previous iteration
            HIPCHK(hipMemcpyAsync(p->pin_b[b], p->dev_b[b], nw * 8, hipMemcpyDeviceToHost, p->s_out));
            HIPCHK(hipEventRecord(p->ev_out[b], p->s_out));
        }
        if (ci >= (size_t)LAG) { // hand chunk ci-LAG's words to the caller: its D2H finished long ago, the DMA queues stay full meanwhile
            const size_t j = ci - LAG;
            const int pb = (int)(j % D);
            HIPCHK(hipEventSynchronize(p->ev_out[pb]));
            p->pool_out->start(out + j * (kPipeChunk / 32), p->pin_b[pb], words_for(chunk_len(j)) * 8); // overlaps the next chunk's stage-in
        }
    }
    p->pool_out->wait();
    bitnuc_err e;
    const int st = drain(c, &e); // one drain at the end: slots are examined in launch order = sequence order
    if (st != BITNUC_OK) {
        if (err) *err = e;
        if (st == BITNUC_INVALID_BASE && n_words) *n_words = (size_t)(e.index / 32);
        return st;
    }
    if (n_words) *n_words = words_for(len);
    return BITNUC_OK;
}

static int decode_pipelined(bitnuc_ctx *c, const uint64_t *ebuf, size_t n_bases, uint8_t *out, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    struct HandBackDone { CopyPool *q; ~HandBackDone() { q->wait(); } } hand_back_done{p->pool_out}; // no return leaves workers writing into `out`
    const size_t kPipeChunk = p->chunk;
    const size_t nchunks = (n_bases + kPipeChunk - 1) / kPipeChunk;
    auto chunk_len = [&](size_t ci) { return n_bases - ci * kPipeChunk < kPipeChunk ? n_bases - ci * kPipeChunk : kPipeChunk; };
    constexpr int D = kPipeDepth, LAG = kPipeDepth - 1;
    for (size_t ci = 0; ci < nchunks + LAG; ++ci) {
        const int b = (int)(ci % D);
        if (ci < nchunks) {
            const size_t n = chunk_len(ci), nw = words_for(n);
            if (ci >= (size_t)D) HIPCHK(hipEventSynchronize(p->ev_in[b]));
            p->pool->copy(p->pin_b[b], ebuf + ci * (kPipeChunk / 32), nw * 8);
            if (ci >= (size_t)D) HIPCHK(hipStreamWaitEvent(p->s_in, p->ev_k[b], 0));
            HIPCHK(hipMemcpyAsync(p->dev_b[b], p->pin_b[b], nw * 8, hipMemcpyHostToDevice, p->s_in));
            HIPCHK(hipEventRecord(p->ev_in[b], p->s_in));
            HIPCHK(hipStreamWaitEvent(c->stream, p->ev_in[b], 0));
            if (ci >= (size_t)D) HIPCHK(hipStreamWaitEvent(c->stream, p->ev_out[b], 0));
            HIPCHK(launch_decode(c, reinterpret_cast<const uint64_t *>(p->dev_b[b]), p->dev_a[b], n));
            HIPCHK(hipEventRecord(p->ev_k[b], c->stream));
            HIPCHK(hipStreamWaitEvent(p->s_out, p->ev_k[b], 0));
            p->pool_out->wait(); // pinned output b is being handed to the caller since the previous iteration
            HIPCHK(hipMemcpyAsync(p->pin_a[b], p->dev_a[b], n, hipMemcpyDeviceToHost, p->s_out));
            HIPCHK(hipEventRecord(p->ev_out[b], p->s_out));
        }
        if (ci >= (size_t)LAG) {
            const size_t j = ci - LAG;
            const int pb = (int)(j % D);
            HIPCHK(hipEventSynchronize(p->ev_out[pb]));
            p->pool_out->start(out + j * kPipeChunk, p->pin_a[pb], chunk_len(j)); // overlaps the next chunk's stage-in
        }
    }
    p->pool_out->wait();
    HIPCHK(hipStreamSynchronize(c->stream));
    return BITNUC_OK;
}

int bitnuc_encode(bitnuc_ctx *c, const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words, bitnuc_err *err) {
    clear_err(err);
    if (n_words) *n_words = 0;
    if (len == 0) return BITNUC_OK; // 0 words (the reference panics there: packing/avx.rs:138)
    if (!seq || !out) return fail(err, BITNUC_UNSUPPORTED);
    if (on_host(c, len)) { // host_word.h: same words, same first-invalid-byte rule, no launch
        const long long bad = bitnuc_host::encode_small(seq, len, out);
        if (bad >= 0) {
            if (err) { memset(err, 0, sizeof *err); err->status = BITNUC_INVALID_BASE; err->byte = seq[bad]; err->index = (uint64_t)bad; }
            if (n_words) *n_words = (size_t)bad / 32;
            return BITNUC_INVALID_BASE;
        }
        if (n_words) *n_words = words_for(len);
        return BITNUC_OK;
    }
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    if (c->host_pipeline && len >= kPipeMin) return encode_pipelined(c, seq, len, out, n_words, err);
    const size_t chunk = len < kHostChunk ? len : kHostChunk;
    if (int st = ensure_scratch(c, 0, chunk + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, words_for(chunk) * 8 + 16, err)) return st;
    for (size_t off = 0; off < len; off += chunk) {
        const size_t n = len - off < chunk ? len - off : chunk;
        const size_t nw = words_for(n);
        HIPCHK(hipMemcpyAsync(c->scratch[0], seq + off, n, hipMemcpyHostToDevice, c->stream));
        unsigned long long *slot;
        if (int st = take_slot(c, off, &slot, err)) return st;
        HIPCHK(launch_encode(c, c->scratch[0], reinterpret_cast<uint64_t *>(c->scratch[1]), n, slot));
        HIPCHK(hipMemcpyAsync(out + off / 32, c->scratch[1], nw * 8, hipMemcpyDeviceToHost, c->stream));
        // the call is synchronous and stops at the first failing chunk (the words before it are the caller's)
        bitnuc_err e;
        int st = drain(c, &e);
        if (st != BITNUC_OK) {
            if (err) *err = e;
            if (st == BITNUC_INVALID_BASE && n_words) *n_words = (size_t)(e.index / 32);
            return st;
        }
    }
    if (n_words) *n_words = words_for(len);
    return BITNUC_OK;
}

int bitnuc_decode(bitnuc_ctx *c, const uint64_t *ebuf, size_t n_words, size_t n_bases, uint8_t *out, bitnuc_err *err) {
    clear_err(err);
    if (n_words < words_for(n_bases)) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (n_bases == 0) return BITNUC_OK;
    if (!ebuf || !out) return fail(err, BITNUC_UNSUPPORTED);
    if (on_host(c, n_bases, true)) {
        bitnuc_host::decode_small(ebuf, n_bases, out);
        return BITNUC_OK;
    }
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    if (c->host_pipeline && n_bases >= kPipeMin) return decode_pipelined(c, ebuf, n_bases, out, err);
    const size_t chunk = n_bases < kHostChunk ? n_bases : kHostChunk;
    if (int st = ensure_scratch(c, 0, chunk + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, words_for(chunk) * 8 + 16, err)) return st;
    for (size_t off = 0; off < n_bases; off += chunk) {
        const size_t n = n_bases - off < chunk ? n_bases - off : chunk;
        const size_t nw = words_for(n);
        HIPCHK(hipMemcpyAsync(c->scratch[1], ebuf + off / 32, nw * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(launch_decode(c, reinterpret_cast<const uint64_t *>(c->scratch[1]), c->scratch[0], n));
        HIPCHK(hipMemcpyAsync(out + off, c->scratch[0], n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return BITNUC_OK;
}

int bitnuc_as_2bit_batch(bitnuc_ctx *c, const uint8_t *kmers, size_t k, size_t stride, size_t count, uint64_t *out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0) return BITNUC_OK;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (!out || stride == 0) return fail(err, BITNUC_UNSUPPORTED);
    if (k == 0) { memset(out, 0, sizeof(uint64_t) * count); return BITNUC_OK; }
    if (!kmers) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    // chunk by k-mers so a staged chunk stays <= kHostChunk bytes
    size_t per = kHostChunk / stride;
    if (per == 0) per = 1;
    if (per > count) per = count;
    if (int st = ensure_scratch(c, 0, (per - 1) * stride + k + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, per * 8, err)) return st;
    for (size_t j0 = 0; j0 < count; j0 += per) {
        const size_t m = count - j0 < per ? count - j0 : per;
        const size_t bytes = (m - 1) * stride + k;
        HIPCHK(hipMemcpyAsync(c->scratch[0], kmers + j0 * stride, bytes, hipMemcpyHostToDevice, c->stream));
        unsigned long long *slot;
        if (int st = take_slot(c, (unsigned long long)j0 * stride, &slot, err)) return st;
        HIPCHK(launch_batch(c, c->scratch[0], k, stride, m, reinterpret_cast<uint64_t *>(c->scratch[1]), slot));
        HIPCHK(hipMemcpyAsync(out + j0, c->scratch[1], m * 8, hipMemcpyDeviceToHost, c->stream));
        bitnuc_err e;
        int st = drain(c, &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
    }
    return BITNUC_OK;
}

int bitnuc_kmer_hdist_scan(bitnuc_ctx *c, const uint8_t *ref, size_t n, size_t k, uint64_t query, uint8_t *dist, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (k == 0 || n < k) return BITNUC_OK;
    if (!ref || !dist) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    const size_t nwin = n - k + 1;
    const size_t chunk = nwin < kHostChunk ? nwin : kHostChunk; // windows per staged chunk
    if (int st = ensure_scratch(c, 0, chunk + k + 16, err)) return st;
    if (int st = ensure_scratch(c, 2, chunk + 16, err)) return st;
    for (size_t off = 0; off < nwin; off += chunk) {
        const size_t w = nwin - off < chunk ? nwin - off : chunk;
        const size_t bytes = w + k - 1;
        HIPCHK(hipMemcpyAsync(c->scratch[0], ref + off, bytes, hipMemcpyHostToDevice, c->stream));
        unsigned long long *slot;
        if (int st = take_slot(c, off, &slot, err)) return st;
        HIPCHK(launch_scan(c, c->scratch[0], bytes, k, query, c->scratch[2], slot));
        HIPCHK(hipMemcpyAsync(dist + off, c->scratch[2], w, hipMemcpyDeviceToHost, c->stream));
        bitnuc_err e;
        int st = drain(c, &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
    }
    return BITNUC_OK;
}

int bitnuc_hdist(bitnuc_ctx *c, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, size_t n_bases, uint32_t *out, bitnuc_err *err) {
    clear_err(err);
    const size_t need = words_for(n_bases);
    if (na < need || nb < need) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    if (n_bases == 0) { *out = 0; return BITNUC_OK; }
    if (!a || !b) return fail(err, BITNUC_UNSUPPORTED);
    if (on_host(c, n_bases)) { *out = bitnuc_host::hdist_small(a, b, n_bases); return BITNUC_OK; }
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    const size_t chunk_words = kHostChunk / 8;
    const size_t cw = need < chunk_words ? need : chunk_words;
    if (int st = ensure_scratch(c, 0, cw * 8, err)) return st;
    if (int st = ensure_scratch(c, 1, cw * 8, err)) return st;
    if (int st = ensure_scratch(c, 2, 64, err)) return st;
    uint32_t total = 0;
    for (size_t w0 = 0; w0 < need; w0 += cw) {
        const size_t m = need - w0 < cw ? need - w0 : cw;
        const size_t bases = (w0 + m == need) ? n_bases - w0 * 32 : m * 32;
        HIPCHK(hipMemcpyAsync(c->scratch[0], a + w0, m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->scratch[1], b + w0, m * 8, hipMemcpyHostToDevice, c->stream));
        bitnuc_err e;
        int st = bitnuc_hdist_dev(c, reinterpret_cast<const uint64_t *>(c->scratch[0]), m,
                                  reinterpret_cast<const uint64_t *>(c->scratch[1]), m, bases,
                                  reinterpret_cast<uint32_t *>(c->scratch[2]), &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
        uint32_t part = 0;
        HIPCHK(hipMemcpyAsync(&part, c->scratch[2], 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        total += part; // u32 wrap-around like the reference's accumulator (multi.rs:130)
    }
    *out = total;
    return BITNUC_OK;
}

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
    const int est_mode = c->owner_est < 3 ? c->owner_est : (total_words >= 16 * (unsigned long long)count ? 2 : 0);
    if (est_mode == 0) block_owner_kernel<0><<<og, kBlock, 0, c->stream>>>(po, pw, count, total_words, ntiles, ratio64, o);
    else if (est_mode == 1) block_owner_kernel<1><<<og, kBlock, 0, c->stream>>>(po, pw, count, total_words, ntiles, ratio64, o);
    else block_owner_kernel<2><<<og, kBlock, 0, c->stream>>>(po, pw, count, total_words, ntiles, ratio64, o);
    HIPCHK(hipGetLastError());
    *recs = o;
    return BITNUC_OK;
}

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
    word_offsets_finish<false><<<(unsigned)nblocks, kBlock, 0, c->stream>>>(off, count, sums, wo, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_word_offsets + count, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (total_words) *total_words = (size_t)total;
    return BITNUC_OK;
}

int bitnuc_encode_batch_dev(bitnuc_ctx *c, const uint8_t *d_seq, const uint64_t *d_offsets, const uint64_t *d_word_offsets, size_t count, size_t total_words, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || total_words == 0) return BITNUC_OK;
    if (!d_seq || !d_offsets || !d_word_offsets || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned long long *po = reinterpret_cast<const unsigned long long *>(d_offsets), *pw = reinterpret_cast<const unsigned long long *>(d_word_offsets);
    unsigned long long *o = reinterpret_cast<unsigned long long *>(d_out);
    const TileRec *recs;
    if (int st = batch_owners(c, d_offsets, d_word_offsets, count, total_words, &recs, err)) return st;
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    const size_t per_block = (size_t)kBatchTile * kBatchWaves;
    const unsigned grid = grid_for(c, (total_words + per_block - 1) / per_block);
    switch (c->batch_abl) {
#ifdef BITNUC_SWEEP_VARIANTS // timing-only ablations (tools/ab_batch_ablate.py, evidence build only): anything but 0 produces wrong words
#define ABL_CASE(A) case A: encode_batch2_kernel<A><<<grid, kBlock, 0, c->stream>>>(d_seq, po, pw, count, total_words, recs, o, slot); break;
    ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(8) ABL_CASE(9) ABL_CASE(11)
#undef ABL_CASE
#endif
    default: encode_batch2_kernel<0><<<grid, kBlock, 0, c->stream>>>(d_seq, po, pw, count, total_words, recs, o, slot);
    }
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_decode_batch_dev(bitnuc_ctx *c, const uint64_t *d_words, const uint64_t *d_word_offsets, const uint64_t *d_offsets, size_t count, size_t total_words, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || total_words == 0) return BITNUC_OK;
    if (!d_words || !d_offsets || !d_word_offsets || !d_out || (reinterpret_cast<uintptr_t>(d_words) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned long long *po = reinterpret_cast<const unsigned long long *>(d_offsets), *pw = reinterpret_cast<const unsigned long long *>(d_word_offsets);
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(d_words);
    const TileRec *recs;
    if (int st = batch_owners(c, d_offsets, d_word_offsets, count, total_words, &recs, err)) return st;
    const size_t per_block = (size_t)kBatchTile * kBatchWaves;
    const unsigned grid = grid_for(c, (total_words + per_block - 1) / per_block);
    switch (c->batch_abl) {
#ifdef BITNUC_SWEEP_VARIANTS
#define ABL_CASE(A) case A: decode_batch2_kernel<A><<<grid, kBlock, 0, c->stream>>>(w, pw, po, count, total_words, recs, d_out); break;
    ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(9) ABL_CASE(11) ABL_CASE(15)
#undef ABL_CASE
#endif
    default: decode_batch2_kernel<0><<<grid, kBlock, 0, c->stream>>>(w, pw, po, count, total_words, recs, d_out);
    }
    HIPCHK(hipGetLastError());
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
    if (c->batch_host_plan) { // the layout plan: word offsets + tile bases + pad bytes in one go (the context keeps one for its host calls)
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
        if (c->batch_host_plan) { if (int st = bitnuc_encode_batch_plan_dev(c, c->host_plan, d_seq, reinterpret_cast<uint64_t *>(c->scratch[1]), err)) return st; }
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
    if (c->batch_host_plan) { // the caller's word_offsets were checked against the offsets above; the plan rebuilds them on the device
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
    size_t cap_wo = 0, cap_base = 0, cap_P = 0;
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
    // sums -> scan of the sums -> (host learns the total and sizes the plan) -> offsets + pad bytes + tile bases in one pass
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
    HIPCHK(hipMemsetAsync(p->d_P, 0, total + 2 + kBatchTile, c->stream));
    if (count) {
        word_offsets_finish<true><<<(unsigned)nblocks, kBlock, 0, c->stream>>>(off, count, sums, p->d_wo, p->d_P, p->d_base);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream)); // the build is synchronous: the plan's tables may be read on any stream afterwards
    p->seq_begin = ends[1];
    p->seq_end = ends[2];
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
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    const int threads = c->plan_enc_block; // 64, 128 or 256 threads: a wave owns a tile, so any number of waves per workgroup works
    const size_t per_block = (size_t)kBatchTile * (size_t)(threads / 64);
    const unsigned long long blocks = (p->total_words + per_block - 1) / per_block;
    unsigned long long *o = reinterpret_cast<unsigned long long *>(d_out);
    const unsigned long long U = (unsigned long long)c->plan_enc_tiles;
    const unsigned grid = grid_for(c, (blocks + U - 1) / U, threads);
#define PLAN_ENC(UU) encode_batch_plan_kernel<UU><<<grid, threads, 0, c->stream>>>(d_seq, p->d_base, p->d_P, p->total_words, p->seq_begin, p->seq_end, o, slot)
    if constexpr (kEvidenceBuild) {
#define PLAN_ENC_ABL(A) encode_batch_plan_kernel<1, A><<<grid, threads, 0, c->stream>>>(d_seq, p->d_base, p->d_P, p->total_words, p->seq_begin, p->seq_end, o, slot)
        if (U == 2) PLAN_ENC(2);
        else if (U == 4) PLAN_ENC(4);
        else switch (c->plan_enc_abl) { // timing-only ablations, right only for 32-base reads (tools/ab_plan_enc_ablate.py)
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

int bitnuc_decode_batch_plan_dev(bitnuc_ctx *c, const bitnuc_batch_plan *p, const uint64_t *d_words, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!p || !p->built || p->device != c->device) return fail(err, BITNUC_UNSUPPORTED);
    if (p->total_words == 0) return BITNUC_OK;
    if (!d_words || !d_out || (reinterpret_cast<uintptr_t>(d_words) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(d_words);
    const int tiles_per_wave = c->plan_tiles;
    const size_t per_block2 = (size_t)kBatchTile * kBatchWaves * (size_t)tiles_per_wave;
    const unsigned grid2 = grid_for(c, (p->total_words + per_block2 - 1) / per_block2);
#define PLAN_DEC(POL, U) decode_batch_plan_kernel<POL, U><<<grid2, kBlock, 0, c->stream>>>(w, p->d_base, p->d_P, p->total_words, d_out)
#define PLAN_DEC_U(POL) do { if (tiles_per_wave == 1) PLAN_DEC(POL, 1); else if (tiles_per_wave == 2) PLAN_DEC(POL, 2); else PLAN_DEC(POL, 4); } while (0)
    if constexpr (kEvidenceBuild) {
        if (c->plan_store == 0) PLAN_DEC_U(0);
        else if (c->plan_store == 1) PLAN_DEC_U(1);
        else PLAN_DEC_U(2);
    } else {
        PLAN_DEC(2, 1);
    }
#undef PLAN_DEC_U
#undef PLAN_DEC
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
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
    if (stride == read_len) encode_fixed_kernel<false><<<grid, kBlock, 0, c->stream>>>(d_seq, (unsigned)read_len, stride, wpr, magic, magic64, total, seq_end, c->fixed_stream, o, slot);
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
    else if (c->fixed_dec_strip == 2 || !kEvidenceBuild) decode_fixed_tile_kernel<2><<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, wpr, magic, magic64, total, d_out);
    else if constexpr (kEvidenceBuild) {
        if (c->fixed_dec_strip) decode_fixed_strip_kernel<<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, wpr, magic, magic64, total, d_out);
        else decode_fixed_kernel<true><<<grid, kBlock, 0, c->stream>>>(w, (unsigned)read_len, stride, wpr, magic, magic64, total, d_out);
    }
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_encode_fixed(bitnuc_ctx *c, const uint8_t *seq, size_t read_len, size_t stride, size_t count, uint64_t *out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0 || read_len == 0) return BITNUC_OK;
    if (stride < read_len || !seq || !out) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
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
        c->pending[c->n_pending - 1].base = (unsigned long long)r0 * stride; // report the index in the caller's buffer
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
    const unsigned grid = grid_for(c, (count / 4 + kBlock - 1) / kBlock + 1);
    const unsigned long long *a = reinterpret_cast<const unsigned long long *>(d_a), *b = reinterpret_cast<const unsigned long long *>(d_b);
    if (query_mode) hdist_words_kernel<true><<<grid, kBlock, 0, c->stream>>>(a, nullptr, query, count, (unsigned)len, d_dist);
    else hdist_words_kernel<false><<<grid, kBlock, 0, c->stream>>>(a, b, 0, count, (unsigned)len, d_dist);
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

// ---- multi-GPU: RCCL all-gather of the packed words -------------------------------------------
// RCCL is bound at run time (dlopen) so that single-GPU users do not need librccl.so.
extern "C++" {
namespace {
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    void *CommInitRank = nullptr; // takes ncclUniqueId by value: called through a typed pointer below
    bool ok = false;
};
struct UniqueIdBytes { char internal[BITNUC_UNIQUE_ID_BYTES]; }; // == ncclUniqueId
constexpr int kNcclUint64 = 5;                                     // ncclUint64 (rccl.h)

RcclApi &rccl() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (api.handle) {
            api.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(api.handle, "ncclGetUniqueId"));
            api.CommInitAll = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(api.handle, "ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(api.handle, "ncclCommDestroy"));
            api.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(api.handle, "ncclAllGather"));
            api.GroupStart = reinterpret_cast<int (*)()>(dlsym(api.handle, "ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<int (*)()>(dlsym(api.handle, "ncclGroupEnd"));
            api.CommInitRank = dlsym(api.handle, "ncclCommInitRank");
            api.ok = api.GetUniqueId && api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd && api.CommInitRank;
        }
    }
    return api;
}
int fail_rccl(bitnuc_err *e, int rc) { // ncclResult_t in backend_code, offset so it cannot be mistaken for a hipError_t
    if (e) { memset(e, 0, sizeof *e); e->status = BITNUC_BACKEND_ERROR; e->backend_code = 10000 + rc; }
    return BITNUC_BACKEND_ERROR;
}
} // namespace
} // extern "C++"

struct bitnuc_comm {
    void *nccl = nullptr; // ncclComm_t
    int nranks = 0, rank = 0, device = 0;
};

int bitnuc_comm_get_unique_id(uint8_t id[BITNUC_UNIQUE_ID_BYTES], bitnuc_err *err) {
    clear_err(err);
    if (!id) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    UniqueIdBytes u;
    if (int rc = r.GetUniqueId(&u)) return fail_rccl(err, rc);
    memcpy(id, u.internal, BITNUC_UNIQUE_ID_BYTES);
    return BITNUC_OK;
}

int bitnuc_comm_init_rank(bitnuc_ctx *c, int nranks, int rank, const uint8_t id[BITNUC_UNIQUE_ID_BYTES], bitnuc_comm **out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(err, BITNUC_UNSUPPORTED);
    *out = nullptr;
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    DeviceGuard g(c->device);
    UniqueIdBytes u;
    memcpy(u.internal, id, BITNUC_UNIQUE_ID_BYTES);
    void *comm = nullptr;
    auto init = reinterpret_cast<int (*)(void **, int, UniqueIdBytes, int)>(r.CommInitRank);
    if (int rc = init(&comm, nranks, u, rank)) return fail_rccl(err, rc);
    bitnuc_comm *bc = new bitnuc_comm();
    bc->nccl = comm; bc->nranks = nranks; bc->rank = rank; bc->device = c->device;
    *out = bc;
    return BITNUC_OK;
}

int bitnuc_comm_init_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, bitnuc_err *err) {
    clear_err(err);
    if (n_gpus < 1 || n_gpus > 64 || !ctxs || !comms) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    for (int i = 0; i < n_gpus; ++i) { ctxs[i] = nullptr; comms[i] = nullptr; }
    for (int i = 0; i < n_gpus; ++i)
        if (int st = bitnuc_ctx_create(i, &ctxs[i], err)) {
            for (int j = 0; j < i; ++j) { bitnuc_ctx_destroy(ctxs[j]); ctxs[j] = nullptr; }
            return st;
        }
    void *raw[64];
    int devs[64];
    for (int i = 0; i < n_gpus; ++i) devs[i] = i;
    if (int rc = r.CommInitAll(raw, n_gpus, devs)) {
        for (int j = 0; j < n_gpus; ++j) { bitnuc_ctx_destroy(ctxs[j]); ctxs[j] = nullptr; }
        return fail_rccl(err, rc);
    }
    for (int i = 0; i < n_gpus; ++i) {
        bitnuc_comm *bc = new bitnuc_comm();
        bc->nccl = raw[i]; bc->nranks = n_gpus; bc->rank = i; bc->device = i;
        comms[i] = bc;
    }
    return BITNUC_OK;
}

void bitnuc_comm_destroy(bitnuc_comm *comm) {
    if (!comm) return;
    RcclApi &r = rccl();
    if (r.ok && comm->nccl) {
        DeviceGuard g(comm->device);
        (void)r.CommDestroy(comm->nccl);
    }
    delete comm;
}

int bitnuc_comm_nranks(const bitnuc_comm *comm) { return comm ? comm->nranks : 0; }
int bitnuc_comm_rank(const bitnuc_comm *comm) { return comm ? comm->rank : -1; }

int bitnuc_allgather_words_dev(bitnuc_ctx *c, bitnuc_comm *comm, const uint64_t *d_local, size_t count, uint64_t *d_all, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!comm || comm->device != c->device) return fail(err, BITNUC_UNSUPPORTED);
    if (count == 0) return BITNUC_OK;
    if (!d_local || !d_all) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    DeviceGuard g(c->device);
    if (int rc = r.AllGather(d_local, d_all, count, kNcclUint64, comm->nccl, c->stream)) return fail_rccl(err, rc);
    return BITNUC_OK;
}

int bitnuc_encode_sharded_allgather_dev(bitnuc_ctx *c, bitnuc_comm *comm, const uint8_t *d_seq_shard, size_t shard_len, uint64_t *d_all, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!comm || comm->device != c->device) return fail(err, BITNUC_UNSUPPORTED);
    // every rank contributes the same number of whole words: shard_len must be a multiple of 32
    // (ragged tails belong in the last rank of a bitnuc_amd.dist.shard_range-style split + padding)
    if (shard_len % 32 != 0) return fail(err, BITNUC_INVALID_LENGTH, shard_len);
    if (shard_len == 0) return BITNUC_OK;
    const size_t count = shard_len / 32;
    uint64_t *mine = d_all + (size_t)comm->rank * count;
    if (int st = bitnuc_encode_dev(c, d_seq_shard, shard_len, mine, err)) return st;
    return bitnuc_allgather_words_dev(c, comm, mine, count, d_all, err);
}

int bitnuc_encode_sharded_allgather_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const uint8_t *const *d_seq_shards, size_t shard_len, uint64_t *const *d_alls, bitnuc_err *err) {
    clear_err(err);
    if (n_gpus < 1 || !ctxs || !comms || !d_seq_shards || !d_alls) return fail(err, BITNUC_UNSUPPORTED);
    if (shard_len % 32 != 0) return fail(err, BITNUC_INVALID_LENGTH, shard_len);
    if (shard_len == 0) return BITNUC_OK;
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    const size_t count = shard_len / 32;
    for (int i = 0; i < n_gpus; ++i) // encode phase: independent, no communication
        if (int st = bitnuc_encode_dev(ctxs[i], d_seq_shards[i], shard_len, d_alls[i] + (size_t)i * count, err)) return st;
    if (int rc = r.GroupStart()) return fail_rccl(err, rc);
    for (int i = 0; i < n_gpus; ++i) {
        DeviceGuard g(ctxs[i]->device);
        if (int rc = r.AllGather(d_alls[i] + (size_t)i * count, d_alls[i], count, kNcclUint64, comms[i]->nccl, ctxs[i]->stream)) {
            (void)r.GroupEnd();
            return fail_rccl(err, rc);
        }
    }
    if (int rc = r.GroupEnd()) return fail_rccl(err, rc);
    for (int i = 0; i < n_gpus; ++i) {
        bitnuc_err e;
        if (int st = bitnuc_ctx_sync(ctxs[i], &e)) { if (err) *err = e; return st; }
    }
    return BITNUC_OK;
}

// ---- single-word API: host code; batches of one on the device when forced ----------------
int bitnuc_as_2bit(bitnuc_ctx *c, const uint8_t *seq, size_t len, uint64_t *out, bitnuc_err *err) {
    clear_err(err);
    if (len > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, len); // packing/naive.rs:5-7: before any base is looked at
    if (!out || (len && !seq)) return fail(err, BITNUC_UNSUPPORTED);
    if (c && c->force_gpu) return bitnuc_as_2bit_batch(c, seq, len, len ? len : 1, 1, out, err);
    uint64_t w = 0;
    const int bad = bitnuc_host::pack_word(seq, len, &w);
    if (bad >= 0) {
        if (err) { memset(err, 0, sizeof *err); err->status = BITNUC_INVALID_BASE; err->byte = seq[bad]; err->index = (uint64_t)bad; }
        return BITNUC_INVALID_BASE;
    }
    *out = w;
    return BITNUC_OK;
}

int bitnuc_from_2bit(bitnuc_ctx *c, uint64_t packed, size_t n, uint8_t *out, bitnuc_err *err) {
    clear_err(err);
    if (n > 32) return fail(err, BITNUC_INVALID_LENGTH, n); // unpacking/naive.rs:8-10
    if (n == 0) return BITNUC_OK;
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    if (c && c->force_gpu) return bitnuc_decode(c, &packed, 1, n, out, err);
    bitnuc_host::unpack_word(packed, n, out);
    return BITNUC_OK;
}

int bitnuc_hdist_scalar(bitnuc_ctx *c, uint64_t u, uint64_t v, size_t len, uint32_t *out, bitnuc_err *err) {
    clear_err(err);
    if (len > 32) return fail(err, BITNUC_INVALID_LENGTH, len); // hamming/scalar.rs:13-15
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    if (c && c->force_gpu) return bitnuc_hdist(c, &u, 1, &v, 1, len, out, err);
    *out = bitnuc_host::hdist_word(u, v, len);
    return BITNUC_OK;
}

// Diagnostic (tools/host_path.py): GB/s of the staging pool's parallel memcpy of `bytes` with `threads` threads;
// mode 0: pageable -> pageable, 1: pageable -> pinned (hipHostMalloc), 2: pinned -> pageable.  < 0 on failure.
double bitnuc_selftime_host_copy(size_t bytes, int threads, int mode) {
    if (bytes < 4096 || threads < 1 || threads > 64 || mode < 0 || mode > 2) return -1.0;
    uint8_t *page = static_cast<uint8_t *>(malloc(bytes)), *other = nullptr;
    if (!page) return -1.0;
    memset(page, 65, bytes);
    if (mode == 0) { other = static_cast<uint8_t *>(malloc(bytes)); if (other) memset(other, 1, bytes); }
    else if (hipHostMalloc(reinterpret_cast<void **>(&other), bytes, hipHostMallocDefault) != hipSuccess) other = nullptr;
    if (!other) { free(page); return -1.0; }
    if (mode != 0) memset(other, 1, bytes);
    double best = 0.0;
    {
        CopyPool pool(threads);
        for (int rep = 0; rep < 4; ++rep) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            if (mode == 2) pool.copy(page, other, bytes); else pool.copy(other, page, bytes);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            if (rep > 0 && (double)bytes / sec / 1e9 > best) best = (double)bytes / sec / 1e9;
        }
    }
    if (mode == 0) free(other); else (void)hipHostFree(other);
    free(page);
    return best;
}

// Diagnostic (bench.py's small_call_latency block): mean ns per call of the HOST path over `iters` calls on the
// reference's bench input (cyclic "ACGT", benches/simd_comparison.rs:4-7), timed here so that no binding overhead is in it.
// op: 0 as_2bit, 1 from_2bit, 2 encode, 3 decode, 4 hdist_scalar.  Returns < 0 on a bad argument.
double bitnuc_selftime_small(int op, size_t n, size_t iters) {
    if (iters == 0 || n == 0 || n > ((size_t)1 << 20) || ((op == 0 || op == 1 || op == 4) && n > 32)) return -1.0;
    std::vector<uint8_t> seq(n), back(n);
    for (size_t i = 0; i < n; ++i) seq[i] = "ACGT"[i & 3];
    std::vector<uint64_t> words(words_for(n) + 1);
    size_t nw = 0;
    bitnuc_err e;
    if (bitnuc_encode(nullptr, seq.data(), n, words.data(), &nw, &e) != BITNUC_OK && n < kDefaultHostCutoff) return -1.0;
    volatile uint64_t sink = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t it = 0; it < iters; ++it) {
        uint64_t w = 0;
        uint32_t d = 0;
        seq[0] = "AC"[it & 1]; // the input changes between calls: the compiler cannot hoist the work
        switch (op) {
        case 0: (void)bitnuc_as_2bit(nullptr, seq.data(), n, &w, &e); sink += w; break;
        case 1: (void)bitnuc_from_2bit(nullptr, words[0] ^ it, n, back.data(), &e); sink += back[0]; break;
        case 2: (void)bitnuc_encode(nullptr, seq.data(), n, words.data(), &nw, &e); sink += words[0]; break;
        case 3: words[0] ^= it & 3; (void)bitnuc_decode(nullptr, words.data(), nw, n, back.data(), &e); sink += back[0]; break;
        case 4: (void)bitnuc_hdist_scalar(nullptr, words[0] ^ it, words[0], n, &d, &e); sink += d; break;
        default: return -1.0;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    (void)sink;
    return ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / (double)iters;
}

} // extern "C"
