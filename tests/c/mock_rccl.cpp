// mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that lets P ranks (threads of one process, each with its
// own bitnuc_ctx and communicator) run the multi-rank code of bitnuc_amd/csrc/comm.hip on ONE GPU.
//
// Nothing here is product code and nothing in the product knows about it: tests/test_gpu_multirank_mock.py builds it as
// `librccl.so.1` into a scratch directory that it puts first on LD_LIBRARY_PATH of a child process, so comm.hip's dlopen of the
// soname finds this file instead of RCCL.  What it checks is everything of config 4 that is OURS -- piece boundaries, slot
// addresses, which peer gets which piece, stream / event ordering between the encode stream and the transfer stream, in-place
// reuse of the gathered buffer, error reporting per rank -- for P = 2..8.  What it cannot check is RCCL itself and the fabric.
//
// Semantics kept from NCCL's documentation (the only parts comm.hip relies on):
//   * operations are enqueued on the caller's stream and take effect in stream order;
//   * point-to-point operations between a pair of ranks match in FIFO order; a send and its receive must agree on the byte count
//     (a mismatch is an error here: stricter than NCCL, which is what a test wants);
//   * a send buffer may be reused in stream order after the call: the sender's stream waits until the receiver has copied;
//   * operations inside ncclGroupStart / ncclGroupEnd are issued together (sends never block the receives of the same group);
//   * ncclCommInitRank returns when every rank of the id has joined.
// A receive whose send does not show up within 60 s (MOCK_RCCL_PATIENCE_MS) returns ncclInternalError instead of hanging the box.
// MOCK_RCCL_DELAY_US=n makes the "fabric" slow: every receive's copy is preceded, in the receiver's stream order, by a host
// function that sleeps n microseconds.  Transfers then finish long after the encode that feeds them (as on xGMI, where the
// gather costs about 8x the encode), so a missing wait between the two streams shows up as stale words instead of passing by luck.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int kOk = 0, kUnhandledHip = 1, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5;
// how long a receive waits for its send (and a send for its receive): MOCK_RCCL_PATIENCE_MS, default 60 s
const auto kPatience = std::chrono::milliseconds([] { const char *e = getenv("MOCK_RCCL_PATIENCE_MS"); return e && atoi(e) > 0 ? atoi(e) : 60000; }());

struct Msg {
    const void *src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;  // recorded on the sender's stream when the message was posted
    hipEvent_t copied = nullptr; // recorded on the receiver's stream after its copy
    bool acked = false;
};
struct World {
    int nranks = 0, joined = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box; // (from, to) -> FIFO
};
struct Comm {
    std::shared_ptr<World> w;
    int rank = 0, device = 0;
};
std::mutex g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;
int g_ids = 0;
std::atomic<uint64_t> g_sends{0}, g_recvs{0};

enum Kind { P_SEND, P_RECV, P_LOCAL };
struct Op {
    Kind kind;
    const void *send;
    void *recv;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

int delay_us() {
    static const int v = [] { const char *e = getenv("MOCK_RCCL_DELAY_US"); return e ? atoi(e) : 0; }();
    return v;
}
void sleeper(void *) { usleep((useconds_t)delay_us()); }

size_t dtype_bytes(int t) { // ncclDataType_t
    switch (t) {
    case 0: case 1: return 1;          // int8 / uint8
    case 2: case 3: case 7: return 4;  // int32 / uint32 / float
    case 4: case 5: case 8: return 8;  // int64 / uint64 / double
    case 6: case 9: return 2;          // half / bfloat16
    default: return 0;
    }
}

int flush(std::vector<Op> &ops) {
    struct Mine { std::shared_ptr<Msg> m; Comm *comm; hipStream_t stream; };
    std::vector<Mine> posted;
    int rc = kOk;
    for (const Op &op : ops) { // 1. every send of the group is posted before anything can block
        if (op.kind != P_SEND) continue;
        if (hipSetDevice(op.comm->device) != hipSuccess) return kUnhandledHip;
        auto m = std::make_shared<Msg>();
        m->src = op.send;
        m->bytes = op.bytes;
        if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(m->ready, op.stream) != hipSuccess) return kUnhandledHip;
        {
            std::lock_guard<std::mutex> g(op.comm->w->mu);
            op.comm->w->box[{op.comm->rank, op.peer}].push_back(m);
            ++g_sends;
        }
        op.comm->w->cv.notify_all();
        posted.push_back({m, op.comm, op.stream});
    }
    for (const Op &op : ops) { // 2. local copies (out-of-place collectives)
        if (op.kind != P_LOCAL) continue;
        if (hipSetDevice(op.comm->device) != hipSuccess) return kUnhandledHip;
        if (hipMemcpyAsync(op.recv, op.send, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return kUnhandledHip;
    }
    for (const Op &op : ops) { // 3. receives: wait for the matching send, copy in the receiver's stream order
        if (op.kind != P_RECV) continue;
        if (hipSetDevice(op.comm->device) != hipSuccess) return kUnhandledHip;
        std::shared_ptr<Msg> m;
        {
            std::unique_lock<std::mutex> g(op.comm->w->mu);
            auto &q = op.comm->w->box[{op.peer, op.comm->rank}];
            if (!op.comm->w->cv.wait_for(g, kPatience, [&] { return !q.empty(); })) { rc = kInternal; continue; }
            m = q.front();
            q.pop_front();
            ++g_recvs;
        }
        hipError_t e = hipSuccess;
        if (m->bytes != op.bytes) rc = kInvalidArgument;
        else {
            e = hipStreamWaitEvent(op.stream, m->ready, 0);
            if (e == hipSuccess && delay_us() > 0) e = hipLaunchHostFunc(op.stream, sleeper, nullptr);
            if (e == hipSuccess) e = hipMemcpyAsync(op.recv, m->src, op.bytes, hipMemcpyDefault, op.stream);
        }
        hipEvent_t done = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(done, op.stream);
        if (e != hipSuccess) rc = kUnhandledHip;
        {
            std::lock_guard<std::mutex> g(op.comm->w->mu);
            m->copied = done;
            m->acked = true;
        }
        op.comm->w->cv.notify_all();
    }
    for (Mine &p : posted) { // 4. the send buffer is the sender's again once the receiver has copied: in the sender's stream order
        std::unique_lock<std::mutex> g(p.comm->w->mu);
        if (!p.comm->w->cv.wait_for(g, kPatience, [&] { return p.m->acked; })) { rc = kInternal; continue; }
        g.unlock();
        (void)hipSetDevice(p.comm->device);
        if (p.m->copied) {
            if (hipStreamWaitEvent(p.stream, p.m->copied, 0) != hipSuccess) rc = kUnhandledHip;
            (void)hipEventDestroy(p.m->copied); // released when the pending wait has been served
        }
        (void)hipEventDestroy(p.m->ready);
    }
    return rc;
}

int enqueue(std::vector<Op> &&prims) {
    for (Op &o : prims) t_ops.push_back(o);
    if (t_depth > 0) return kOk;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return flush(ops);
}

} // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> g(g_mu);
    snprintf(id->internal, sizeof id->internal, "mock-rccl:%d:%d", (int)getpid(), ++g_ids);
    return kOk;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto &slot = g_worlds[std::string(id.internal, sizeof id.internal)];
        if (!slot) { slot = std::make_shared<World>(); slot->nranks = nranks; }
        w = slot;
    }
    if (w->nranks != nranks) return kInvalidArgument;
    Comm *c = new Comm();
    c->w = w;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return kUnhandledHip; }
    {
        std::unique_lock<std::mutex> g(w->mu);
        ++w->joined;
        w->cv.notify_all();
        if (!w->cv.wait_for(g, kPatience, [&] { return w->joined >= w->nranks; })) { delete c; return kInternal; }
    }
    *comm = c;
    return kOk;
}

int ncclCommInitAll(void **comms, int n, const int *devs) {
    if (!comms || n < 1) return kInvalidArgument;
    auto w = std::make_shared<World>();
    w->nranks = w->joined = n;
    for (int i = 0; i < n; ++i) {
        Comm *c = new Comm();
        c->w = w;
        c->rank = i;
        c->device = devs ? devs[i] : i;
        comms[i] = c;
    }
    return kOk;
}

int ncclCommDestroy(void *comm) {
    delete static_cast<Comm *>(comm);
    return kOk;
}

int ncclGroupStart() { ++t_depth; return kOk; }
int ncclGroupEnd() {
    if (t_depth <= 0) return kInvalidUsage;
    if (--t_depth > 0) return kOk;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return flush(ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t b = dtype_bytes(dtype);
    if (!c || !b || peer < 0 || peer >= c->w->nranks || peer == c->rank) return kInvalidArgument;
    return enqueue({Op{P_SEND, buf, nullptr, count * b, peer, c, stream}});
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t b = dtype_bytes(dtype);
    if (!c || !b || peer < 0 || peer >= c->w->nranks || peer == c->rank) return kInvalidArgument;
    return enqueue({Op{P_RECV, nullptr, buf, count * b, peer, c, stream}});
}
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t b = dtype_bytes(dtype) * count;
    if (!c || !b) return kInvalidArgument;
    std::vector<Op> prims;
    char *out = static_cast<char *>(recv);
    for (int p = 0; p < c->w->nranks; ++p) {
        if (p == c->rank) { if (send != out + (size_t)p * b) prims.push_back(Op{P_LOCAL, send, out + (size_t)p * b, b, p, c, stream}); continue; }
        prims.push_back(Op{P_SEND, send, nullptr, b, p, c, stream});
        prims.push_back(Op{P_RECV, nullptr, out + (size_t)p * b, b, p, c, stream});
    }
    return enqueue(std::move(prims));
}
int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t b = dtype_bytes(dtype) * count;
    if (!c || !b || root < 0 || root >= c->w->nranks) return kInvalidArgument;
    std::vector<Op> prims;
    if (c->rank == root) {
        for (int p = 0; p < c->w->nranks; ++p) if (p != root) prims.push_back(Op{P_SEND, send, nullptr, b, p, c, stream});
        if (send != recv) prims.push_back(Op{P_LOCAL, send, recv, b, root, c, stream});
    } else prims.push_back(Op{P_RECV, nullptr, recv, b, root, c, stream});
    return enqueue(std::move(prims));
}

// how many point-to-point messages have been posted / matched in this process so far (the driver checks the schedule's size,
// and that it really ran against this mock)
void mock_rccl_totals(uint64_t *sends, uint64_t *recvs) {
    if (sends) *sends = g_sends.load();
    if (recvs) *recvs = g_recvs.load();
}

} // extern "C"
