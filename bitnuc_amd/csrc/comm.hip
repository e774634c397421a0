// comm.hip -- multi-GPU: shard + concatenate (BASELINE config 4) and the xGMI link probe.
// u64 words never share state (the reference's loop has no carry, src/utils/packing/avx.rs:138-145), so shards encode
// independently; the only exchange of the path is the final concatenation of the packed words over xGMI:
//   * one shot: ncclAllGather in place on the context's stream;
//   * chunked overlap (SURVEY 8e iii): the shard is encoded piece by piece on the context's stream while a second stream
//     moves the finished pieces, IN PLACE, with grouped point-to-point ncclSend / ncclRecv -- on MI355X's full mesh of
//     point-to-point xGMI links that uses all P-1 links of a GPU at once (a ring would be per-link bound).
// Two ways to hold the ranks: one thread or process per rank (bitnuc_comm_init_rank + the per-rank entry points), or ONE thread
// for all of them (bitnuc_comm_init_all[_devices] + the _all entry points, which put every rank's part of an exchange into one
// ncclGroupStart / ncclGroupEnd); the per-rank entry points refuse the second kind of communicator instead of blocking in it.
// RCCL is bound at run time (dlopen) so that single-GPU users do not need librccl.so.  No kernels here.
#include "runtime.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <time.h>

#include <mutex>
#include <vector>

using namespace bitnuc_rt;

namespace {

struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    void *CommInitRank = nullptr; // takes ncclUniqueId by value: called through a typed pointer below
    bool ok = false;
};
struct UniqueIdBytes { char internal[BITNUC_UNIQUE_ID_BYTES]; }; // == ncclUniqueId
constexpr int kNcclUint64 = 5;                                     // ncclUint64 (rccl.h)

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // "librccl.so.1" first: it is the SONAME, so a process that already holds an RCCL (PyTorch bundles its own copy under
        // that soname) gets THAT instance back instead of a second RCCL runtime beside it; otherwise the system's is loaded
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        api.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitAll = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(api.handle, "ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(api.handle, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(api.handle, "ncclAllGather"));
        api.Send = reinterpret_cast<int (*)(const void *, size_t, int, int, void *, hipStream_t)>(dlsym(api.handle, "ncclSend"));
        api.Recv = reinterpret_cast<int (*)(void *, size_t, int, int, void *, hipStream_t)>(dlsym(api.handle, "ncclRecv"));
        api.Broadcast = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(api.handle, "ncclBroadcast"));
        api.GroupStart = reinterpret_cast<int (*)()>(dlsym(api.handle, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<int (*)()>(dlsym(api.handle, "ncclGroupEnd"));
        api.CommInitRank = dlsym(api.handle, "ncclCommInitRank");
        api.ok = api.GetUniqueId && api.CommInitAll && api.CommDestroy && api.AllGather && api.Send && api.Recv && api.Broadcast &&
                 api.GroupStart && api.GroupEnd && api.CommInitRank;
    });
    return api;
}
int fail_rccl(bitnuc_err *e, int rc) { // ncclResult_t in backend_code, offset so it cannot be mistaken for a hipError_t
    if (e) { memset(e, 0, sizeof *e); e->status = BITNUC_BACKEND_ERROR; e->backend_code = 10000 + rc; }
    return BITNUC_BACKEND_ERROR;
}

double now_s() {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

} // namespace

struct bitnuc_comm {
    void *nccl = nullptr; // ncclComm_t
    int nranks = 0, rank = 0, device = 0;
    // created by bitnuc_comm_init_all[_devices]: ONE host thread holds every rank of the communicator, so an exchange must be issued
    // for all ranks inside one ncclGroupStart / ncclGroupEnd (the _all entry points).  A per-rank call from that thread would block in
    // its group's end waiting for peers that the same thread has not yet been able to issue: the per-rank entry points refuse it.
    bool single_process = false;
    bool threaded = false; // bitnuc_comm_set_threaded: the host runs one thread per rank of an init_all communicator after all
    hipStream_t xfer = nullptr;        // the second stream of the chunked overlap (created on first use)
    std::vector<hipEvent_t> piece_done; // piece c encoded (recorded on the context's stream)
    hipEvent_t all_moved = nullptr;     // every piece exchanged (recorded on xfer)
};

namespace {

int comm_overlap_resources(bitnuc_comm *comm, int n_chunks, bitnuc_err *err) {
    if (!comm->xfer) HIPCHK(hipStreamCreateWithFlags(&comm->xfer, hipStreamNonBlocking));
    if (!comm->all_moved) HIPCHK(hipEventCreateWithFlags(&comm->all_moved, hipEventDisableTiming));
    while ((int)comm->piece_done.size() < n_chunks) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        comm->piece_done.push_back(e);
    }
    return BITNUC_OK;
}

// see bitnuc_comm::single_process (a communicator of one rank has no peer to wait for)
bool per_rank_call_would_block(const bitnuc_comm *comm) { return comm->single_process && !comm->threaded && comm->nranks > 1; }

bool gather_by_broadcast() {
    static const bool bcast = [] { const char *e = getenv("BITNUC_GATHER_MODE"); return e && !strcmp(e, "bcast"); }();
    return bcast;
}

// the arguments every _all entry point shares: n_gpus contexts + the communicators bitnuc_comm_init_all made for them, in rank order
int check_all_args(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const uint8_t *const *d_seq_shards, size_t shard_len, uint64_t *const *d_alls, bitnuc_err *err) {
    if (n_gpus < 1 || n_gpus > 64 || !ctxs || !comms || !d_seq_shards || !d_alls) return fail(err, BITNUC_UNSUPPORTED);
    for (int i = 0; i < n_gpus; ++i) {
        if (!ctxs[i] || !comms[i]) return fail(err, BITNUC_UNSUPPORTED);
        if (!comms[i]->single_process || comms[i]->threaded || comms[i]->nranks != n_gpus || comms[i]->rank != i || comms[i]->device != ctxs[i]->device) return fail(err, BITNUC_UNSUPPORTED);
    }
    if (shard_len % 32 != 0) return fail(err, BITNUC_INVALID_LENGTH, shard_len);
    if (shard_len)
        for (int i = 0; i < n_gpus; ++i)
            if (!d_seq_shards[i] || !d_alls[i] || (reinterpret_cast<uintptr_t>(d_alls[i]) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    return BITNUC_OK;
}

// Wait for every rank's stream; the first data error in rank order is the call's (err.value = that rank, err.index relative to its shard).
int sync_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_err *err) {
    int st_first = BITNUC_OK;
    for (int i = 0; i < n_gpus; ++i) {
        bitnuc_err e;
        const int st = bitnuc_ctx_sync(ctxs[i], &e);
        if (st != BITNUC_OK && st_first == BITNUC_OK) {
            st_first = st;
            if (st == BITNUC_INVALID_BASE) e.value = (uint64_t)i;
            if (err) *err = e;
        }
    }
    return st_first;
}

} // namespace

extern "C" {

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

int bitnuc_comm_init_all_devices(int n_gpus, const int *devices, bitnuc_ctx **ctxs, bitnuc_comm **comms, bitnuc_err *err) {
    clear_err(err);
    if (n_gpus < 1 || n_gpus > 64 || !ctxs || !comms) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    int devs[64];
    for (int i = 0; i < n_gpus; ++i) { ctxs[i] = nullptr; comms[i] = nullptr; devs[i] = devices ? devices[i] : i; }
    for (int i = 0; i < n_gpus; ++i)
        if (int st = bitnuc_ctx_create(devs[i], &ctxs[i], err)) {
            for (int j = 0; j < i; ++j) { bitnuc_ctx_destroy(ctxs[j]); ctxs[j] = nullptr; }
            return st;
        }
    void *raw[64];
    if (int rc = r.CommInitAll(raw, n_gpus, devs)) {
        for (int j = 0; j < n_gpus; ++j) { bitnuc_ctx_destroy(ctxs[j]); ctxs[j] = nullptr; }
        return fail_rccl(err, rc);
    }
    for (int i = 0; i < n_gpus; ++i) {
        bitnuc_comm *bc = new bitnuc_comm();
        bc->nccl = raw[i]; bc->nranks = n_gpus; bc->rank = i; bc->device = devs[i]; bc->single_process = true;
        comms[i] = bc;
    }
    return BITNUC_OK;
}

int bitnuc_comm_init_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, bitnuc_err *err) {
    return bitnuc_comm_init_all_devices(n_gpus, nullptr, ctxs, comms, err);
}

void bitnuc_comm_destroy(bitnuc_comm *comm) {
    if (!comm) return;
    DeviceGuard g(comm->device);
    if (comm->xfer) (void)hipStreamSynchronize(comm->xfer);
    RcclApi &r = rccl();
    if (r.ok && comm->nccl) (void)r.CommDestroy(comm->nccl);
    for (hipEvent_t e : comm->piece_done) (void)hipEventDestroy(e);
    if (comm->all_moved) (void)hipEventDestroy(comm->all_moved);
    if (comm->xfer) (void)hipStreamDestroy(comm->xfer);
    delete comm;
}

int bitnuc_comm_nranks(const bitnuc_comm *comm) { return comm ? comm->nranks : 0; }
int bitnuc_comm_rank(const bitnuc_comm *comm) { return comm ? comm->rank : -1; }
int bitnuc_comm_single_process(const bitnuc_comm *comm) { return comm ? (int)comm->single_process : -1; }
int bitnuc_comm_set_threaded(bitnuc_comm *comm, int threaded) {
    if (!comm) return -1;
    const int prev = (int)comm->threaded;
    comm->threaded = threaded != 0;
    return prev;
}

int bitnuc_allgather_words_dev(bitnuc_ctx *c, bitnuc_comm *comm, const uint64_t *d_local, size_t count, uint64_t *d_all, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!comm || comm->device != c->device || per_rank_call_would_block(comm)) return fail(err, BITNUC_UNSUPPORTED);
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
    if (!comm || comm->device != c->device || per_rank_call_would_block(comm)) return fail(err, BITNUC_UNSUPPORTED);
    // every rank contributes the same number of whole words: shard_len must be a multiple of 32
    // (ragged tails belong in the last rank of a bitnuc_amd.dist.shard_range-style split + padding)
    if (shard_len % 32 != 0) return fail(err, BITNUC_INVALID_LENGTH, shard_len);
    if (shard_len == 0) return BITNUC_OK;
    const size_t count = shard_len / 32;
    uint64_t *mine = d_all + (size_t)comm->rank * count;
    if (int st = bitnuc_encode_dev(c, d_seq_shard, shard_len, mine, err)) return st;
    return bitnuc_allgather_words_dev(c, comm, mine, count, d_all, err);
}

// Chunked overlap, in place.  Piece p = words [count p / n, count (p+1) / n) of every rank's shard.  The context's stream
// encodes the pieces back to back; after each one an event lets the transfer stream exchange that piece: this rank SENDS
// its piece to every peer and RECEIVES every peer's piece straight into d_all + peer * count + w0 -- no staging buffer, no
// strided copy.  The context's stream finally waits for the transfer stream, so bitnuc_ctx_sync covers the whole call.
// BITNUC_GATHER_MODE=bcast moves a piece with P grouped ncclBroadcast calls (root = owner, in place) instead.
int bitnuc_encode_sharded_allgather_overlapped_dev(bitnuc_ctx *c, bitnuc_comm *comm, const uint8_t *d_seq_shard, size_t shard_len, int n_chunks, uint64_t *d_all, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!comm || comm->device != c->device || per_rank_call_would_block(comm) || n_chunks < 1 || n_chunks > 4096) return fail(err, BITNUC_UNSUPPORTED);
    if (shard_len % 32 != 0) return fail(err, BITNUC_INVALID_LENGTH, shard_len);
    if (shard_len == 0) return BITNUC_OK;
    if (!d_seq_shard || !d_all || (reinterpret_cast<uintptr_t>(d_all) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    DeviceGuard g(c->device);
    if (int st = comm_overlap_resources(comm, n_chunks, err)) return st;
    const bool bcast = gather_by_broadcast();
    const size_t count = shard_len / 32;
    const int P = comm->nranks, me = comm->rank;
    uint64_t *mine = d_all + (size_t)me * count;
    // the transfer stream must not run ahead of what the caller already queued on the context's stream (d_all's previous readers).
    // (Implied by the first piece_done wait below, which is recorded later on the same stream -- tools/multirank_mutation_check.py,
    // defect C, shows no scenario needs it; kept so that the rule does not depend on the loop's first iteration.)
    HIPCHK(hipEventRecord(comm->all_moved, c->stream));
    HIPCHK(hipStreamWaitEvent(comm->xfer, comm->all_moved, 0));
    // however this call ends -- also on an error half way -- the context's stream waits for what the transfer stream was given, so
    // that bitnuc_ctx_sync() covers it and the caller may reuse or free d_all after that sync
    struct Join {
        bitnuc_ctx *c;
        bitnuc_comm *comm;
        ~Join() {
            if (hipEventRecord(comm->all_moved, comm->xfer) == hipSuccess) (void)hipStreamWaitEvent(c->stream, comm->all_moved, 0);
            else (void)hipGetLastError();
        }
    } join{c, comm};
    for (int p = 0; p < n_chunks; ++p) {
        const size_t w0 = count * (size_t)p / (size_t)n_chunks, w1 = count * (size_t)(p + 1) / (size_t)n_chunks;
        if (w1 == w0) continue;
        if (int st = encode_dev_at(c, d_seq_shard + 32 * w0, 32 * (w1 - w0), mine + w0, 32ull * w0, err)) return st;
        HIPCHK(hipEventRecord(comm->piece_done[(size_t)p], c->stream));
        HIPCHK(hipStreamWaitEvent(comm->xfer, comm->piece_done[(size_t)p], 0));
        if (P == 1) continue;
        if (int rc = r.GroupStart()) return fail_rccl(err, rc);
        int rc = 0;
        for (int s = 0; s < P && rc == 0; ++s) {
            uint64_t *theirs = d_all + (size_t)s * count + w0;
            if (bcast) rc = r.Broadcast(theirs, theirs, w1 - w0, kNcclUint64, s, comm->nccl, comm->xfer);
            else if (s != me) {
                rc = r.Send(mine + w0, w1 - w0, kNcclUint64, s, comm->nccl, comm->xfer);
                if (rc == 0) rc = r.Recv(theirs, w1 - w0, kNcclUint64, s, comm->nccl, comm->xfer);
            }
        }
        const int rc_end = r.GroupEnd();
        if (rc) return fail_rccl(err, rc);
        if (rc_end) return fail_rccl(err, rc_end);
    }
    return BITNUC_OK; // (~Join: the context's stream now waits for the transfer stream)
}

int bitnuc_encode_sharded_allgather_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const uint8_t *const *d_seq_shards, size_t shard_len, uint64_t *const *d_alls, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_all_args(n_gpus, ctxs, comms, d_seq_shards, shard_len, d_alls, err)) return st;
    if (shard_len == 0) return BITNUC_OK;
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    const size_t count = shard_len / 32;
    for (int i = 0; i < n_gpus; ++i) // encode phase: independent, no communication
        if (int st = bitnuc_encode_dev(ctxs[i], d_seq_shards[i], shard_len, d_alls[i] + (size_t)i * count, err)) { (void)sync_all(n_gpus, ctxs, nullptr); return st; }
    if (int rc = r.GroupStart()) { (void)sync_all(n_gpus, ctxs, nullptr); return fail_rccl(err, rc); } // the encodes are queued: the caller gets its buffers back synchronised
    int rc = 0;
    for (int i = 0; i < n_gpus && rc == 0; ++i) {
        DeviceGuard g(ctxs[i]->device);
        rc = r.AllGather(d_alls[i] + (size_t)i * count, d_alls[i], count, kNcclUint64, comms[i]->nccl, ctxs[i]->stream);
    }
    const int rc_end = r.GroupEnd();
    const int st = sync_all(n_gpus, ctxs, err);
    if (rc) return fail_rccl(err, rc);
    if (rc_end) return fail_rccl(err, rc_end);
    return st;
}

// The chunked in-place exchange of bitnuc_encode_sharded_allgather_overlapped_dev, driven for ALL ranks by the calling thread: piece
// by piece, every device's stream gets its encode, then ONE ncclGroupStart ... every rank's sends and receives (on that rank's
// transfer stream) ... ncclGroupEnd -- the form RCCL requires of a thread that holds several ranks.  Synchronises every stream
// before it returns, like the one-shot _all form.
int bitnuc_encode_sharded_allgather_overlapped_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const uint8_t *const *d_seq_shards, size_t shard_len, int n_chunks, uint64_t *const *d_alls, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_all_args(n_gpus, ctxs, comms, d_seq_shards, shard_len, d_alls, err)) return st;
    if (n_chunks < 1 || n_chunks > 4096) return fail(err, BITNUC_UNSUPPORTED);
    if (shard_len == 0) return BITNUC_OK;
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    const bool bcast = gather_by_broadcast();
    const size_t count = shard_len / 32;
    const int P = n_gpus;
    // however this call ends -- also in the set-up loop below, after earlier ranks' transfer streams were chained -- every context's
    // stream is made to wait for its transfer stream (where it exists) and is then synchronised, so the caller owns all buffers again
    auto finish = [&](int st_call, bitnuc_err *e_call) {
        for (int i = 0; i < P; ++i) {
            if (!comms[i]->xfer || !comms[i]->all_moved) continue;
            DeviceGuard g(ctxs[i]->device);
            if (hipEventRecord(comms[i]->all_moved, comms[i]->xfer) == hipSuccess) (void)hipStreamWaitEvent(ctxs[i]->stream, comms[i]->all_moved, 0);
            else (void)hipGetLastError();
        }
        bitnuc_err e_sync;
        const int st_sync = sync_all(P, ctxs, &e_sync);
        if (st_call != BITNUC_OK) { if (err && e_call) *err = *e_call; return st_call; }
        if (st_sync != BITNUC_OK && err) *err = e_sync;
        return st_sync;
    };
    bitnuc_err e;
    memset(&e, 0, sizeof e);
    for (int i = 0; i < P; ++i) {
        DeviceGuard g(ctxs[i]->device);
        if (int st = comm_overlap_resources(comms[i], n_chunks, &e)) return finish(st, &e);
        // rank i's transfer stream starts behind what its context's stream already holds (d_alls[i]'s previous readers)
        hipError_t h = hipEventRecord(comms[i]->all_moved, ctxs[i]->stream);
        if (h == hipSuccess) h = hipStreamWaitEvent(comms[i]->xfer, comms[i]->all_moved, 0);
        if (h != hipSuccess) { fail_hip(&e, h); return finish(BITNUC_BACKEND_ERROR, &e); }
    }
    for (int p = 0; p < n_chunks; ++p) {
        const size_t w0 = count * (size_t)p / (size_t)n_chunks, w1 = count * (size_t)(p + 1) / (size_t)n_chunks;
        if (w1 == w0) continue;
        for (int i = 0; i < P; ++i) { // piece p of every shard: encode on the rank's own stream, then release it to the rank's transfer stream
            DeviceGuard g(ctxs[i]->device);
            uint64_t *mine = d_alls[i] + (size_t)i * count;
            if (int st = encode_dev_at(ctxs[i], d_seq_shards[i] + 32 * w0, 32 * (w1 - w0), mine + w0, 32ull * w0, &e)) return finish(st, &e);
            hipError_t h = hipEventRecord(comms[i]->piece_done[(size_t)p], ctxs[i]->stream);
            if (h == hipSuccess) h = hipStreamWaitEvent(comms[i]->xfer, comms[i]->piece_done[(size_t)p], 0);
            if (h != hipSuccess) { fail_hip(&e, h); return finish(BITNUC_BACKEND_ERROR, &e); }
        }
        if (P == 1) continue;
        if (int rc = r.GroupStart()) { fail_rccl(&e, rc); return finish(BITNUC_BACKEND_ERROR, &e); }
        int rc = 0;
        for (int i = 0; i < P && rc == 0; ++i) {
            DeviceGuard g(ctxs[i]->device);
            uint64_t *mine = d_alls[i] + (size_t)i * count;
            for (int s = 0; s < P && rc == 0; ++s) {
                uint64_t *theirs = d_alls[i] + (size_t)s * count + w0;
                if (bcast) rc = r.Broadcast(theirs, theirs, w1 - w0, kNcclUint64, s, comms[i]->nccl, comms[i]->xfer);
                else if (s != i) {
                    rc = r.Send(mine + w0, w1 - w0, kNcclUint64, s, comms[i]->nccl, comms[i]->xfer);
                    if (rc == 0) rc = r.Recv(theirs, w1 - w0, kNcclUint64, s, comms[i]->nccl, comms[i]->xfer);
                }
            }
        }
        const int rc_end = r.GroupEnd();
        if (rc || rc_end) { fail_rccl(&e, rc ? rc : rc_end); return finish(BITNUC_BACKEND_ERROR, &e); }
    }
    return finish(BITNUC_OK, nullptr);
}

// ---- a ragged batch across ranks (include/bitnuc_hip.h) --------------------------------------------------------------
int bitnuc_batch_shard_ranges(const uint64_t *offsets, size_t count, int nranks, size_t *seq_first, uint64_t *word_first, bitnuc_err *err) {
    clear_err(err);
    if (nranks < 1 || nranks > 4096 || !seq_first || !word_first || (count && !offsets)) return fail(err, BITNUC_UNSUPPORTED);
    for (size_t i = 0; i < count; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(err, BITNUC_INVALID_RANGE, i);
    // W[i] = words before sequence i (every sequence pads its own last word: packing/avx.rs:147-148); two passes, no allocation
    uint64_t total = 0;
    for (size_t i = 0; i < count; ++i) total += words_for((size_t)(offsets[i + 1] - offsets[i]));
    size_t i = 0;
    uint64_t w = 0; // = W[i]
    for (int r = 0; r < nranks; ++r) {
        const uint64_t target = (uint64_t)(((unsigned __int128)total * (unsigned)r) / (unsigned)nranks);
        while (i < count && w < target) { w += words_for((size_t)(offsets[i + 1] - offsets[i])); ++i; }
        seq_first[r] = i;
        word_first[r] = w;
    }
    seq_first[nranks] = count;
    word_first[nranks] = total;
    return BITNUC_OK;
}

namespace {
// rank `me`'s part of the in-place exchange of unequal counts, inside the caller's group
int allgatherv_issue(RcclApi &r, bitnuc_comm *comm, hipStream_t stream, const size_t *counts, const size_t *first, uint64_t *d_all, bool bcast) {
    const int P = comm->nranks, me = comm->rank;
    int rc = 0;
    for (int s = 0; s < P && rc == 0; ++s) {
        if (bcast) { if (counts[s]) rc = r.Broadcast(d_all + first[s], d_all + first[s], counts[s], kNcclUint64, s, comm->nccl, stream); }
        else if (s != me) {
            if (counts[me]) rc = r.Send(d_all + first[me], counts[me], kNcclUint64, s, comm->nccl, stream);
            if (rc == 0 && counts[s]) rc = r.Recv(d_all + first[s], counts[s], kNcclUint64, s, comm->nccl, stream);
        }
    }
    return rc;
}
} // namespace

int bitnuc_allgatherv_words_dev(bitnuc_ctx *c, bitnuc_comm *comm, const size_t *counts, uint64_t *d_all, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!comm || comm->device != c->device || per_rank_call_would_block(comm) || !counts || comm->nranks > 4096) return fail(err, BITNUC_UNSUPPORTED);
    std::vector<size_t> first((size_t)comm->nranks + 1, 0);
    for (int s = 0; s < comm->nranks; ++s) first[(size_t)s + 1] = first[(size_t)s] + counts[s];
    if (first[(size_t)comm->nranks] == 0 || comm->nranks == 1) return BITNUC_OK;
    if (!d_all || (reinterpret_cast<uintptr_t>(d_all) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    DeviceGuard g(c->device);
    if (int rc = r.GroupStart()) return fail_rccl(err, rc);
    const int rc = allgatherv_issue(r, comm, c->stream, counts, first.data(), d_all, gather_by_broadcast());
    const int rc_end = r.GroupEnd();
    if (rc) return fail_rccl(err, rc);
    if (rc_end) return fail_rccl(err, rc_end);
    return BITNUC_OK;
}

int bitnuc_allgatherv_words_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const size_t *counts, uint64_t *const *d_alls, bitnuc_err *err) {
    clear_err(err);
    if (n_gpus < 1 || n_gpus > 64 || !ctxs || !comms || !counts || !d_alls) return fail(err, BITNUC_UNSUPPORTED);
    for (int i = 0; i < n_gpus; ++i) {
        if (!ctxs[i] || !comms[i]) return fail(err, BITNUC_UNSUPPORTED);
        if (!comms[i]->single_process || comms[i]->threaded || comms[i]->nranks != n_gpus || comms[i]->rank != i || comms[i]->device != ctxs[i]->device) return fail(err, BITNUC_UNSUPPORTED);
    }
    std::vector<size_t> first((size_t)n_gpus + 1, 0);
    for (int s = 0; s < n_gpus; ++s) first[(size_t)s + 1] = first[(size_t)s] + counts[s];
    if (first[(size_t)n_gpus])
        for (int i = 0; i < n_gpus; ++i)
            if (!d_alls[i] || (reinterpret_cast<uintptr_t>(d_alls[i]) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    RcclApi &r = rccl();
    if (!r.ok) return fail_rccl(err, -1);
    int rc = 0, rc_end = 0;
    if (first[(size_t)n_gpus] && n_gpus > 1) {
        rc = r.GroupStart();
        if (rc == 0) {
            const bool bcast = gather_by_broadcast();
            for (int i = 0; i < n_gpus && rc == 0; ++i) {
                DeviceGuard g(ctxs[i]->device);
                rc = allgatherv_issue(r, comms[i], ctxs[i]->stream, counts, first.data(), d_alls[i], bcast);
            }
            rc_end = r.GroupEnd();
        }
    }
    const int st = sync_all(n_gpus, ctxs, err); // every path: the streams are synchronised before the call returns
    if (rc) return fail_rccl(err, rc);
    if (rc_end) return fail_rccl(err, rc_end);
    return st;
}

// ---- xGMI link probe (measure the per-link rate on the box before quoting a fabric roofline) -----------
// hipMemcpyPeerAsync of `bytes` from src_device to each of dst_devices[0..n): one link at a time (gb_s_each[i], best of
// `reps`), then all n at once on n streams (*gb_s_all = aggregate outbound rate of src_device).  Host wall clock around
// stream synchronisation; buffers are allocated and freed here.  Needs every device visible in THIS process.
int bitnuc_peer_link_probe(int src_device, const int *dst_devices, int n, size_t bytes, int reps, double *gb_s_each, double *gb_s_all, bitnuc_err *err) {
    clear_err(err);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (n < 1 || n > 63 || !dst_devices || bytes < 4096 || reps < 1 || src_device < 0 || src_device >= ndev) return fail(err, BITNUC_UNSUPPORTED, (uint64_t)ndev);
    for (int i = 0; i < n; ++i)
        if (dst_devices[i] < 0 || dst_devices[i] >= ndev || dst_devices[i] == src_device) return fail(err, BITNUC_UNSUPPORTED, (uint64_t)ndev);
    int prev = 0;
    (void)hipGetDevice(&prev);
    std::vector<void *> dst((size_t)n, nullptr);
    std::vector<hipStream_t> streams((size_t)n, nullptr);
    void *src = nullptr;
    hipError_t rc = hipSuccess;
    for (int i = 0; i < n && rc == hipSuccess; ++i) {
        int can = 0;
        rc = hipDeviceCanAccessPeer(&can, src_device, dst_devices[i]);
        if (rc == hipSuccess && !can) rc = hipErrorPeerAccessUnsupported;
        if (rc != hipSuccess) break;
        rc = hipSetDevice(dst_devices[i]);
        if (rc == hipSuccess) { const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = e; else (void)hipGetLastError(); }
        if (rc == hipSuccess) rc = hipMalloc(&dst[(size_t)i], bytes);
        if (rc == hipSuccess) rc = hipSetDevice(src_device);
        if (rc == hipSuccess) { const hipError_t e = hipDeviceEnablePeerAccess(dst_devices[i], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = e; else (void)hipGetLastError(); }
        if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking);
    }
    if (rc == hipSuccess) rc = hipSetDevice(src_device);
    if (rc == hipSuccess) rc = hipMalloc(&src, bytes);
    if (rc == hipSuccess) rc = hipMemset(src, 0x5A, bytes);
    if (rc == hipSuccess) rc = hipDeviceSynchronize();
    double all = 0.0;
    if (rc == hipSuccess) {
        for (int i = 0; i < n && rc == hipSuccess; ++i) { // one link at a time
            double best = 0.0;
            for (int rep = 0; rep <= reps && rc == hipSuccess; ++rep) { // rep 0 = warm-up
                const double t0 = now_s();
                rc = hipMemcpyPeerAsync(dst[(size_t)i], dst_devices[i], src, src_device, bytes, streams[(size_t)i]);
                if (rc == hipSuccess) rc = hipStreamSynchronize(streams[(size_t)i]);
                const double gbs = (double)bytes / (now_s() - t0) / 1e9;
                if (rep > 0 && gbs > best) best = gbs;
            }
            if (gb_s_each) gb_s_each[i] = best;
        }
        for (int rep = 0; rep <= reps && rc == hipSuccess; ++rep) { // all links at once
            const double t0 = now_s();
            for (int i = 0; i < n && rc == hipSuccess; ++i) rc = hipMemcpyPeerAsync(dst[(size_t)i], dst_devices[i], src, src_device, bytes, streams[(size_t)i]);
            for (int i = 0; i < n && rc == hipSuccess; ++i) rc = hipStreamSynchronize(streams[(size_t)i]);
            const double gbs = (double)bytes * n / (now_s() - t0) / 1e9;
            if (rep > 0 && gbs > all) all = gbs;
        }
    }
    if (gb_s_all) *gb_s_all = all;
    (void)hipSetDevice(src_device);
    for (int i = 0; i < n; ++i) if (streams[(size_t)i]) (void)hipStreamDestroy(streams[(size_t)i]);
    if (src) (void)hipFree(src);
    for (int i = 0; i < n; ++i)
        if (dst[(size_t)i]) { (void)hipSetDevice(dst_devices[i]); (void)hipFree(dst[(size_t)i]); }
    (void)hipSetDevice(prev);
    if (rc != hipSuccess) return fail_hip(err, rc);
    return BITNUC_OK;
}

} // extern "C"
