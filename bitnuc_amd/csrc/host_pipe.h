// host_pipe.h -- the pipelined host-pointer path shared by the units that offer host-pointer bulk calls (codec.hip: encode / decode;
// kmer.hip: k-mer batches, windows, scan).  Internal (needs HIP); the staging pool it drives is host_pool.h (no HIP, sanitized).
#pragma once
#include "runtime.h"
#include "host_pool.h"

#include <stdlib.h>

#include <atomic>

// ---- pipelined host-pointer path --------------------------------------------------------------------
// Two engines run the same jobs (pipe_run dispatches on bitnuc_ctx::pipe_impl): the DIRECT engine ships (pipe_run_direct below: the
// calling thread and one mover thread issue pageable copies, no pinned buffers), the STAGED engine described next is kept behind
// BITNUC_PIPE_IMPL=staged for platforms whose runtime copies pageable memory through bounce buffers of its own.
//
// The staged engine:
// A caller's buffers are pageable.  Handing them to hipMemcpyAsync makes the runtime stage them through its own
// pinned bounce buffers on the calling thread, serialising copy-in, kernel and copy-out.  Here the library owns the
// staging: a worker pool (host_pool.h) copies chunk c+1 from the caller's memory into one of three pinned input buffers
// while the DMA engines move the neighbouring chunks (H2D on one stream, D2H on another) and the kernel runs on the
// context's stream; a second, asynchronous pool hands chunk c-2 back to the caller meanwhile.  Events order the three
// streams and guard buffer reuse; the host waits only when a pinned buffer is about to be overwritten.
//
// How many threads copy is decided per direction when the pipe is created.  The HEAVY side (stage-in for encode: 1 B per base;
// hand-back for decode: 1 B per base) gets 8 threads, the light side (0.25 B per base) 4 (2 below a 12-CPU budget), both capped by the CPUs this process
// may use (affinity mask AND cgroup quota: cores_usable() -- a 16-core quota on a 256-thread host shows 256 CPUs in its mask).
// Measured on the GPU box (tools/host_path_r03.py, profiles/r03_host_path.txt; 10^9 bases, pinned engines 56 GB/s each way and
// full duplex): 6 / 8 / 11 heavy threads give encode 51.7 / 52.0 / 52.0 and decode 49.3 / 49.5 / 49.9 Gbases/s; round 2's
// hand-back pool of threads/2+1 = 4 workers gave decode 39-45.  A creation-time calibration (smallest thread count whose memcpy
// of a 32 MiB probe buffer beats the DMA rate by 30 %) was tried and REMOVED: the probe buffer stays in the host's L3, 2-3
// threads look sufficient (84-88 GB/s), and the real pipeline, whose sources stream from DRAM, then ran decode at 39 Gbases/s.
constexpr int kPipeDepth = 3; // buffer sets in flight: the host hands chunk c-2 to the caller while chunk c-1 is on the DMA engines and chunk c is staged
constexpr size_t kPipeChunkDefault = (size_t)32 << 20; // bases per chunk; BITNUC_PIPE_CHUNK_MB overrides
constexpr size_t kPipeMin = (size_t)8 << 20;           // inputs below this stay on the simple path (latency, not bandwidth, matters there)

// Three kinds of buffer sets, kPipeDepth of each, pinned + device: A = chunk + 64 bytes (ASCII-sized), B = chunk / 4 + 64 bytes
// (word-sized), C = a second A-sized set that only the scan needs (input AND output are a byte per base); C is allocated on first use.
enum { kBufA = 0, kBufB = 1, kBufC = 2 };

struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[kPipeDepth] = {}, ev_k[kPipeDepth] = {}, ev_out[kPipeDepth] = {};
    uint8_t *pin[3][kPipeDepth] = {}, *dev[3][kPipeDepth] = {};
    bitnuc_host::CopyPool *pool = nullptr;     // stage-in: blocking, the calling thread takes a slice
    bitnuc_host::CopyPool *pool_out = nullptr; // hand-back to the caller: asynchronous, overlaps the next chunk's stage-in
    bitnuc_host::TaskThread *mover = nullptr;  // the direct engine's second mover (created on first use)
    size_t chunk = kPipeChunkDefault; // bases per chunk (a multiple of 32)
    int enc_in = 1, enc_out = 1, dec_in = 1, dec_out = 1; // copy threads per direction and side
    int cores_visible = 1, cores_quota = 0, cores_usable = 1, heavy_cap = 1;
    cpu_set_t local_cpus;
    int pool_threads = 1;
    int numa_node = -1, bound_cpus = 0; // the GPU's NUMA node (-1 = unknown) and how many of its CPUs the workers are bound to (0 = not bound)
    bool ok = false;
    size_t buf_bytes(int kind) const { return kind == kBufB ? chunk / 4 + 64 : chunk + 64; }
};

namespace bitnuc_rt {

inline void pipe_free(HostPipe *p) {
    if (!p) return;
    delete p->mover;    // runs what is still queued, then joins
    delete p->pool_out; // joins its workers (waits for an outstanding hand-back) before the pinned buffers go
    delete p->pool;
    for (int kind = 0; kind < 3; ++kind)
        for (int i = 0; i < kPipeDepth; ++i) {
            if (p->pin[kind][i]) (void)hipHostFree(p->pin[kind][i]);
            if (p->dev[kind][i]) (void)hipFree(p->dev[kind][i]);
        }
    for (int i = 0; i < kPipeDepth; ++i) {
        if (p->ev_in[i]) (void)hipEventDestroy(p->ev_in[i]);
        if (p->ev_k[i]) (void)hipEventDestroy(p->ev_k[i]);
        if (p->ev_out[i]) (void)hipEventDestroy(p->ev_out[i]);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
}

// device buffers of a kind; `pinned`: also the staged engine's pinned host buffers (the direct engine has none)
inline hipError_t pipe_alloc_kind(HostPipe *p, int kind, bool pinned) {
    hipError_t rc = hipSuccess;
    for (int i = 0; i < kPipeDepth && rc == hipSuccess; ++i) {
        if (pinned && !p->pin[kind][i]) rc = hipHostMalloc(reinterpret_cast<void **>(&p->pin[kind][i]), p->buf_bytes(kind), hipHostMallocDefault);
        if (rc == hipSuccess && !p->dev[kind][i]) rc = hipMalloc(&p->dev[kind][i], p->buf_bytes(kind));
    }
    return rc;
}

// kinds: bit mask of the buffer sets the caller needs (1 << kBufA | ...); A and B are allocated with the pipe, C on first use
inline int pipe_get(bitnuc_ctx *c, HostPipe **out, bitnuc_err *err, unsigned kinds = (1u << kBufA) | (1u << kBufB)) {
    if (c->pipe && c->pipe->ok) {
        if ((kinds & (1u << kBufC)) && !c->pipe->dev[kBufC][kPipeDepth - 1]) {
            const hipError_t rcC = pipe_alloc_kind(c->pipe, kBufC, false);
            if (rcC != hipSuccess) return fail_hip(err, rcC);
        }
        *out = c->pipe;
        return BITNUC_OK;
    }
    if (c->pipe) { pipe_free(c->pipe); c->pipe = nullptr; } // a pipe that an aborted call left in an unknown state: rebuild
    HostPipe *p = new HostPipe();
    hipError_t rc = hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking);
    if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking);
    if (const char *e = getenv("BITNUC_PIPE_CHUNK_MB")) {
        const long v = atol(e);
        if (v >= 1 && v <= 1024) p->chunk = (size_t)v << 20;
    }
    for (int i = 0; i < kPipeDepth && rc == hipSuccess; ++i) {
        rc = hipEventCreateWithFlags(&p->ev_in[i], hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&p->ev_k[i], hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&p->ev_out[i], hipEventDisableTiming);
    }
    if (rc == hipSuccess) rc = pipe_alloc_kind(p, kBufA, false);
    if (rc == hipSuccess) rc = pipe_alloc_kind(p, kBufB, false);
    if (rc == hipSuccess && (kinds & (1u << kBufC))) rc = pipe_alloc_kind(p, kBufC, false);
    if (rc != hipSuccess) { pipe_free(p); return fail_hip(err, rc); }
    // CPU budget: what the affinity mask AND the cgroup quota allow, minus one for the HIP runtime's own threads
    p->cores_visible = bitnuc_host::cores_visible();
    p->cores_quota = bitnuc_host::cores_quota();
    p->cores_usable = bitnuc_host::cores_usable();
    const int budget = p->cores_usable > 2 ? p->cores_usable - 1 : 2;
    int light = budget >= 12 ? 4 : (budget >= 8 ? 2 : 1); // the 0.25 B-per-base side sits on the caller's critical path: it must never become the slow one
    if (const int v = bitnuc_host::env_threads("BITNUC_HOST_THREADS_LIGHT")) light = v;
    int cap = budget - light;
    if (cap > bitnuc_host::kPoolMaxThreads - 1) cap = bitnuc_host::kPoolMaxThreads - 1;
    if (cap < 1) cap = 1;
    const int forced = bitnuc_host::env_threads("BITNUC_HOST_THREADS"); // the heavy side's thread count, as given
    if (forced) cap = forced < bitnuc_host::kPoolMaxThreads ? forced : bitnuc_host::kPoolMaxThreads - 1;
    p->heavy_cap = cap;
    const int heavy = forced ? cap : (cap < 8 ? cap : 8);
    const int most = heavy > light ? heavy : light;
    // NUMA: the runtime places pinned host memory on the node the GPU hangs off, and a copy thread is fast when it sits on the node
    // of its SOURCE (local reads, posted remote writes).  The hand-back pool reads pinned memory, so its workers are bound to the
    // GPU's local CPUs; the stage-in pool reads the CALLER's memory, whose node the library cannot know -- its workers stay free
    // and the scheduler keeps them near the calling thread that wakes them, which is where the caller's pages usually are.
    // tools/host_topology.py on a two-socket box, fresh contexts, 10^9 bases: free hand-back workers gave decode 35.9-51.2 Gbases/s
    // (median 44.7), bound ones 46.2-50.7 (median 49.8); bound STAGE-IN workers cost encode 10 % (44-51 against a steady 51-52).
    // BITNUC_PIPE_NUMA=0 leaves every worker free.
    cpu_set_t local;
    int n_local = 0;
    const char *numa_env = getenv("BITNUC_PIPE_NUMA");
    if (!(numa_env && atoi(numa_env) == 0)) {
        char bdf[32] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, c->device) == hipSuccess) {
            for (char *q = bdf; *q; ++q) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a'); // sysfs names are lower case
            n_local = bitnuc_host::pci_local_cpus(bdf, &local, &p->numa_node);
        } else (void)hipGetLastError();
    }
    p->bound_cpus = n_local >= most ? n_local : 0; // only when the node offers at least as many CPUs as the pool has threads
    if (p->bound_cpus) p->local_cpus = local;
    p->pool_threads = most; // the staged engine creates its pools on first use (pipe_staged_prepare)
    p->enc_in = p->dec_out = heavy;
    p->enc_out = p->dec_in = light < most ? light : most;
    p->ok = true;
    c->pipe = p;
    *out = p;
    return BITNUC_OK;
}

// A call that leaves the pipelined loop early (a HIP error mid-loop) must not leave copies, kernels or error slots behind:
// the next call assumes an idle pipe.  Unless dismissed, the guard waits for the hand-back pool, the three streams, and
// empties the slot ring; the pipe is rebuilt by the next call.
struct PipeAbort {
    bitnuc_ctx *c;
    HostPipe *p;
    bool dismissed = false;
    ~PipeAbort() {
        if (p->pool_out) p->pool_out->wait(); // no return leaves workers writing into the caller's buffer
        if (p->mover) p->mover->drain();
        if (dismissed) return;
        (void)hipStreamSynchronize(p->s_in);
        (void)hipStreamSynchronize(p->s_out);
        bitnuc_err e;
        (void)drain(c, &e);
        (void)hipGetLastError();
        p->ok = false;
    }
};

// The engine.  A job is cut into chunks; chunk ci is staged from the caller's memory into pinned buffer [in_kind][ci % depth]
// (blocking parallel memcpy, `in_threads` threads incl. the caller), copied to the device on s_in, worked on by `launch` on the
// context's stream, copied back on s_out and handed to the caller by the asynchronous pool two iterations later (`out_threads`
// workers), while chunk ci + 1 is being staged.  Job:
//   size_t nchunks; int in_kind, out_kind (kBufA / kBufB / kBufC), in_threads, out_threads;
//   const void *in_src(size_t ci); size_t in_bytes(size_t ci);  void *out_dst(size_t ci); size_t out_bytes(size_t ci);
//   int launch(size_t ci, const uint8_t *d_in, uint8_t *d_out, bitnuc_err *err);   // enqueue on c->stream (takes its own error slot)
// Returns after the last hand-back; the kernels' error slots are the caller's to drain (one drain: launch order = sequence order).
// The DIRECT engine (pipe_impl 1).  On this platform the runtime pins a pageable buffer in place and a pageable hipMemcpyAsync
// runs at the pinned rate (56 GB/s either way, profiles/r03_pageable_copy_rates.txt) -- but it blocks its caller until the copy is
// done, so one thread gets one direction at a time.  Two threads then do what the staged engine needs 8 + 4 copy threads and
// three sets of pinned buffers for: the CALLING thread copies chunk c straight from the caller's memory to device buffer
// c % depth and launches on it, the MOVER thread (host_pool.h: TaskThread) waits for kernel c's event and copies its output straight
// into the caller's memory.  A device buffer pair is reused once the mover has finished the chunk that used it (a host-side
// ticket: by then the kernel has read its input and the copy has left its output), so no other guard is needed.
template <class Job>
int pipe_run_direct(bitnuc_ctx *c, HostPipe *p, Job &job, bitnuc_err *err) {
    constexpr int D = kPipeDepth;
    if (!p->mover) {
        p->mover = new bitnuc_host::TaskThread();
        const int dev = c->device;
        p->mover->post([dev] { (void)hipSetDevice(dev); });
    }
    bitnuc_host::TaskThread &w = *p->mover;
    std::atomic<int> mover_rc{0}; // (declared before the guard: the tasks it waits for write here)
    struct Drain { // however the loop ends, nothing is still being written into the caller's memory when this call returns
        bitnuc_host::TaskThread &w;
        ~Drain() { w.drain(); }
    } drain_guard{w};
    const uint64_t base = w.tickets(); // all finished: the previous call drained
    for (size_t ci = 0; ci < job.nchunks; ++ci) {
        const int b = (int)(ci % D);
        if (ci >= (size_t)D) w.wait_done(base + ci - D + 1); // chunk ci-D is with the caller: device buffers b are free
        if (const int rc = mover_rc.load()) return fail_hip(err, (hipError_t)rc);
        const size_t nin = job.in_bytes(ci), nout = job.out_bytes(ci);
        uint8_t *dev_in = p->dev[job.in_kind][b], *dev_out = p->dev[job.out_kind][b];
        HIPCHK(hipMemcpyAsync(dev_in, job.in_src(ci), nin, hipMemcpyHostToDevice, p->s_in)); // pageable source: back when the copy is done
        HIPCHK(hipEventRecord(p->ev_in[b], p->s_in));
        HIPCHK(hipStreamWaitEvent(c->stream, p->ev_in[b], 0)); // (a caller that passes pinned memory gets a truly asynchronous copy)
        if (int st = job.launch(ci, dev_in, dev_out, err)) return st;
        HIPCHK(hipEventRecord(p->ev_k[b], c->stream));
        void *dst = job.out_dst(ci);
        hipStream_t s_out = p->s_out;
        hipEvent_t ev_k = p->ev_k[b];
        w.post([=, &mover_rc] {
            hipError_t e = hipStreamWaitEvent(s_out, ev_k, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(dst, dev_out, nout, hipMemcpyDeviceToHost, s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(s_out);
            if (e != hipSuccess) mover_rc.store((int)e);
        });
    }
    w.drain();
    if (const int rc = mover_rc.load()) return fail_hip(err, (hipError_t)rc);
    return BITNUC_OK;
}

// what only the staged engine needs: pinned buffers of the two kinds a job uses and the two copy pools
inline int pipe_staged_prepare(HostPipe *p, int in_kind, int out_kind, bitnuc_err *err) {
    for (int kind : {in_kind, out_kind}) {
        if (p->pin[kind][kPipeDepth - 1]) continue;
        const hipError_t rc = pipe_alloc_kind(p, kind, true);
        if (rc != hipSuccess) return fail_hip(err, rc);
    }
    if (!p->pool) p->pool = new bitnuc_host::CopyPool(p->pool_threads); // the caller + pool_threads - 1 workers, free (see pipe_get)
    if (!p->pool_out) p->pool_out = new bitnuc_host::CopyPool(p->pool_threads + 1, p->bound_cpus ? &p->local_cpus : nullptr); // pool_threads workers (the caller's slice index is unused in asynchronous jobs)
    return BITNUC_OK;
}

template <class Job>
int pipe_run(bitnuc_ctx *c, HostPipe *p, Job &job, bitnuc_err *err) {
    if (c->pipe_impl == 1) return pipe_run_direct(c, p, job, err);
    if (int st = pipe_staged_prepare(p, job.in_kind, job.out_kind, err)) return st;
    constexpr int D = kPipeDepth, LAG = kPipeDepth - 1;
    for (size_t ci = 0; ci < job.nchunks + LAG; ++ci) {
        const int b = (int)(ci % D);
        if (ci < job.nchunks) {
            const size_t nin = job.in_bytes(ci), nout = job.out_bytes(ci);
            uint8_t *pin_in = p->pin[job.in_kind][b], *dev_in = p->dev[job.in_kind][b], *pin_out = p->pin[job.out_kind][b], *dev_out = p->dev[job.out_kind][b];
            if (ci >= (size_t)D) HIPCHK(hipEventSynchronize(p->ev_in[b])); // pinned input b: its previous H2D has left
            p->pool->copy(pin_in, job.in_src(ci), nin, job.in_threads);
            if (ci >= (size_t)D) HIPCHK(hipStreamWaitEvent(p->s_in, p->ev_k[b], 0)); // device input b: the kernel of chunk ci-D has read it
            HIPCHK(hipMemcpyAsync(dev_in, pin_in, nin, hipMemcpyHostToDevice, p->s_in));
            HIPCHK(hipEventRecord(p->ev_in[b], p->s_in));
            HIPCHK(hipStreamWaitEvent(c->stream, p->ev_in[b], 0));
            if (ci >= (size_t)D) HIPCHK(hipStreamWaitEvent(c->stream, p->ev_out[b], 0)); // device output b: its D2H of chunk ci-D is done
            if (int st = job.launch(ci, dev_in, dev_out, err)) return st;
            HIPCHK(hipEventRecord(p->ev_k[b], c->stream));
            HIPCHK(hipStreamWaitEvent(p->s_out, p->ev_k[b], 0));
            p->pool_out->wait(); // pinned output b is being handed to the caller since the previous iteration
            HIPCHK(hipMemcpyAsync(pin_out, dev_out, nout, hipMemcpyDeviceToHost, p->s_out));
            HIPCHK(hipEventRecord(p->ev_out[b], p->s_out));
        }
        if (ci >= (size_t)LAG) { // hand chunk ci-LAG to the caller: its D2H finished long ago, the DMA queues stay full meanwhile
            const size_t j = ci - LAG;
            HIPCHK(hipEventSynchronize(p->ev_out[(int)(j % D)]));
            p->pool_out->start(job.out_dst(j), p->pin[job.out_kind][(int)(j % D)], job.out_bytes(j), job.out_threads); // overlaps the next chunk's stage-in
        }
    }
    p->pool_out->wait();
    return BITNUC_OK;
}

} // namespace bitnuc_rt
