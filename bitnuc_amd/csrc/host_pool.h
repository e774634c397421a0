// host_pool.h -- the host-side staging pool of the pipelined host-pointer path (codec.hip: encode_pipelined /
// decode_pipelined) and the CPU budget it is sized from.  Plain C++17 + pthreads, NO HIP: tests/c/host_sanitize.cpp
// compiles this header (and host_word.h) with -fsanitize=address,undefined and -fsanitize=thread on the CPU build.
//
// The reference is single-threaded (src/utils/unpacking/avx.rs:37 holds its only static); these threads exist only to
// move the caller's pageable bytes to and from pinned memory at PCIe speed, they never touch codec arithmetic.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace bitnuc_host {

// CPUs this process may run on (affinity mask) and the CPU-time quota of its cgroup, in cores (0 = no quota).
// A container with a 16-core quota on a 256-thread host reports 256 in its affinity mask: sizing thread pools
// from the mask alone oversubscribes the quota, sizing them from a fixed guess undersubscribes it.
inline int cores_visible() {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n >= 1) return n;
    }
    return 1;
}
inline int cores_quota() {
    // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: two files
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long period = 0;
        const int got = fscanf(f, "%31s %lld", q, &period);
        fclose(f);
        if (got == 2 && period > 0 && strcmp(q, "max") != 0) {
            const long long quota = atoll(q);
            if (quota > 0) return (int)((quota + period - 1) / period);
        }
        return 0;
    }
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
    if (quota > 0 && period > 0) return (int)((quota + period - 1) / period);
    return 0;
}
inline int cores_usable() {
    const int vis = cores_visible(), quota = cores_quota();
    return quota > 0 && quota < vis ? quota : vis;
}
// "0-63,128-191" (sysfs cpulist format) -> cpu_set_t; returns the number of CPUs set
inline int parse_cpulist(const char *text, cpu_set_t *out) {
    CPU_ZERO(out);
    const char *p = text;
    while (*p) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1) break; p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0) CPU_SET((int)c, out);
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    return CPU_COUNT(out);
}
// the CPUs local to a PCI device ("0000:23:00.0"), intersected with what this process may run on; 0 = unknown
inline int pci_local_cpus(const char *bdf, cpu_set_t *out, int *numa_node) {
    char path[256], buf[4096] = {0};
    CPU_ZERO(out);
    if (numa_node) *numa_node = -1;
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    cpu_set_t local, mine;
    if (parse_cpulist(buf, &local) == 0 || sched_getaffinity(0, sizeof mine, &mine) != 0) return 0;
    CPU_AND(out, &local, &mine);
    if (numa_node) {
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
        if (FILE *g = fopen(path, "r")) { if (fscanf(g, "%d", numa_node) != 1) *numa_node = -1; fclose(g); }
    }
    return CPU_COUNT(out);
}
inline int env_threads(const char *name) { // 0 = not set / out of range
    if (const char *e = getenv(name)) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) return v;
    }
    return 0;
}

constexpr int kPoolMaxThreads = 12; // copy threads a pool is created with (the per-call count is chosen below this)

// A pool of copy threads.  Two ways to use it, never mixed while a job is outstanding:
//   copy(d, s, n, use)   blocking: `use` threads (the caller is one of them) copy disjoint 4 KiB-aligned slices;
//   start(d, s, n, use)  asynchronous: `use` WORKERS copy, the caller goes on and calls wait() before it touches
//                        either buffer again or starts the next job.
// One caller thread per pool (the context's single-thread contract, include/bitnuc_hip.h).  Every shared field is
// written under `mu` before the generation counter moves and read by a worker only after it has seen the new
// generation under `mu`; a new job cannot be posted before `pending` of the previous one has reached zero.
struct CopyPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    uint8_t *dst = nullptr;
    const uint8_t *src = nullptr;
    size_t bytes = 0, slice = 0;
    unsigned generation = 0, pending = 0;
    bool stop = false;
    int skip = 0;      // 1 while an asynchronous job runs: worker i then owns slice i - 1 (the caller takes none)
    bool busy = false; // caller-side only: an asynchronous job has been started and not yet waited for
    int n = 1;         // workers + the calling thread

    // `cpus` (optional): the workers are bound to this CPU set -- the pipelined host path passes the CPUs of the NUMA node the
    // GPU hangs off, where the runtime also puts pinned host memory (the calling thread is the caller's and is left alone)
    explicit CopyPool(int nthreads, const cpu_set_t *cpus = nullptr) : n(nthreads < 1 ? 1 : nthreads) {
        if (cpus) { bind = *cpus; bound = CPU_COUNT(&bind) > 0; }
        for (int i = 1; i < n; ++i) threads.emplace_back([this, i] { run(i); });
    }
    ~CopyPool() {
        wait();
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;

    void copy_slice(int i) const {
        const size_t a = slice * (size_t)(i - skip);
        if (a >= bytes) return;
        const size_t m = bytes - a < slice ? bytes - a : slice;
        memcpy(dst + a, src + a, m);
    }
    cpu_set_t bind;
    bool bound = false;
    void run(int i) {
        if (bound) (void)pthread_setaffinity_np(pthread_self(), sizeof bind, &bind); // best effort
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
    static size_t slice_for(size_t nbytes, int parts) { return ((nbytes + (size_t)parts - 1) / (size_t)parts + 4095) & ~(size_t)4095; }
    int clamp_use(int use, int most) const { return use < 1 ? 1 : (use > most ? most : use); }

    void start(void *d, const void *s_, size_t nbytes, int use = kPoolMaxThreads) {
        wait();
        if (n == 1 || nbytes == 0) { if (nbytes) memcpy(d, s_, nbytes); return; }
        {
            std::lock_guard<std::mutex> g(mu);
            dst = static_cast<uint8_t *>(d);
            src = static_cast<const uint8_t *>(s_);
            bytes = nbytes;
            slice = slice_for(nbytes, clamp_use(use, n - 1)); // workers only
            skip = 1;
            pending = (unsigned)(n - 1);
            ++generation;
        }
        busy = true;
        cv_work.notify_all();
    }
    void wait() {
        if (!busy) return;
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return pending == 0; });
        busy = false;
    }
    void copy(void *d, const void *s_, size_t nbytes, int use = kPoolMaxThreads) { // blocking parallel memcpy
        wait();
        if (n == 1 || use <= 1 || nbytes < ((size_t)1 << 20)) { if (nbytes) memcpy(d, s_, nbytes); return; }
        {
            std::lock_guard<std::mutex> g(mu);
            dst = static_cast<uint8_t *>(d);
            src = static_cast<const uint8_t *>(s_);
            bytes = nbytes;
            slice = slice_for(nbytes, clamp_use(use, n));
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

// One helper thread that runs posted tasks in order (the "direct" host pipeline's second mover: while the calling thread copies
// chunk c+1 towards the device, this thread copies chunk c-1 back into the caller's memory).  post() returns a ticket = the
// number of tasks posted so far; wait_done(t) returns when t tasks have finished; drain() when all posted ones have.  One poster
// thread (the context's single-thread contract).  Tasks still queued when the object is destroyed are run first.
struct TaskThread {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::function<void()>> q;
    uint64_t posted = 0, done = 0;
    bool stop = false;
    std::thread t;

    TaskThread() : t([this] { run(); }) {}
    ~TaskThread() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv_work.notify_all();
        t.join();
    }
    TaskThread(const TaskThread &) = delete;
    TaskThread &operator=(const TaskThread &) = delete;

    void run() {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> g(mu);
                cv_work.wait(g, [&] { return stop || !q.empty(); });
                if (q.empty()) return; // stop, and nothing left to run
                fn = std::move(q.front());
                q.pop_front();
            }
            fn();
            {
                std::lock_guard<std::mutex> g(mu);
                ++done;
            }
            cv_done.notify_all();
        }
    }
    uint64_t post(std::function<void()> fn) {
        uint64_t ticket;
        {
            std::lock_guard<std::mutex> g(mu);
            q.push_back(std::move(fn));
            ticket = ++posted;
        }
        cv_work.notify_one();
        return ticket;
    }
    uint64_t tickets() {
        std::lock_guard<std::mutex> g(mu);
        return posted;
    }
    void wait_done(uint64_t ticket) {
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return done >= ticket; });
    }
    void drain() {
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return done >= posted; });
    }
};

} // namespace bitnuc_host
