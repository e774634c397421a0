// multirank_driver.cpp -- TEST: config 4 through the C ABI with P ranks on ONE GPU (threads of this process, one bitnuc_ctx and
// one communicator each), against tests/c/mock_rccl.cpp standing in for librccl.so.1 (see that file for what this can and cannot
// show).  For every scenario the gathered buffer of EVERY rank must equal a single-GPU bitnuc_encode_dev of the concatenated
// input ("bit-identical to a single-GPU encode of the concatenated input"); that single-GPU encode is what the
// parity tests pin to the oracle.
//
//   multirank_driver P shard_len n_chunks mode rounds [bad_rank bad_offset]
//     mode: oneshot | overlap           (overlap + BITNUC_GATHER_MODE=bcast in the environment = the broadcast exchange)
//           oneshot_all | overlap_all   ONE thread holds all P ranks (bitnuc_comm_init_all_devices on device 0 P times -- the mock
//                                       accepts the duplicate device, RCCL would not) and drives them with the _all entry points;
//                                       also checks that the per-rank entry points REFUSE such a communicator (they would block)
//           ragged | ragged_all         a RAGGED BATCH split by whole sequences (SURVEY 8e sentence 2): shard_len = number of sequences, n_chunks = seed of
//                                       their lengths; every rank encodes its run (bitnuc_batch_shard_ranges -> bitnuc_batch_plan_build_dev ->
//                                       bitnuc_encode_batch_plan_dev) into its slot and the UNEQUAL word counts are gathered in place
//                                       (bitnuc_allgatherv_words_dev / _all); expected = ONE context's plan encode of the whole batch
//     rounds: calls back to back on the SAME buffers with new data each round (in-place reuse: the transfer stream of round r+1
//             must not run ahead of round r's readers)
//     bad_rank / bad_offset: plant an invalid byte in that rank's shard in the last round: that rank's sync must report
//             InvalidBase('N') with the shard-relative offset, every other rank's must succeed
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bitnuc_hip.h"

namespace {

struct Barrier { // C++17: no std::barrier
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> g(mu);
        const unsigned my = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return; }
        cv.wait(g, [&] { return gen != my; });
    }
};

std::atomic<int> g_fail{0};
std::mutex g_print;
void complain(int rank, const std::string &what) {
    std::lock_guard<std::mutex> g(g_print);
    fprintf(stderr, "rank %d: %s\n", rank, what.c_str());
    ++g_fail;
}
// a rank that cannot go on ends the process: the others would wait for it at the next barrier or receive
#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { complain(rank, std::string(#call) + ": " + hipGetErrorString(e_)); _exit(1); } } while (0)
#define BNOK(call) do { int st_ = (call); if (st_ != BITNUC_OK) { complain(rank, std::string(#call) + ": status " + std::to_string(st_) + " backend " + std::to_string(err.backend_code)); _exit(1); } } while (0)

constexpr uint64_t kSeed = 0xB17C0DE;

// One thread, all ranks: the _all forms over a bitnuc_comm_init_all_devices communicator.
void run_single_process(int P, size_t shard_len, int n_chunks, bool overlap, int rounds, int bad_rank, size_t bad_offset, const std::vector<std::vector<uint64_t>> &expect) {
    const int rank = -1; // for the macros' messages
    const size_t count = shard_len / 32, total_words = count * (size_t)P;
    bitnuc_err err;
    memset(&err, 0, sizeof err);
    std::vector<int> devs((size_t)P, 0);
    std::vector<bitnuc_ctx *> ctxs((size_t)P, nullptr);
    std::vector<bitnuc_comm *> comms((size_t)P, nullptr);
    BNOK(bitnuc_comm_init_all_devices(P, devs.data(), ctxs.data(), comms.data(), &err));
    std::vector<uint8_t *> d_seq((size_t)P, nullptr);
    std::vector<uint64_t *> d_all((size_t)P, nullptr);
    for (int i = 0; i < P; ++i) {
        if (bitnuc_comm_nranks(comms[(size_t)i]) != P || bitnuc_comm_rank(comms[(size_t)i]) != i || bitnuc_comm_single_process(comms[(size_t)i]) != 1) complain(i, "communicator reports the wrong rank / size / kind");
        HIPOK(hipMalloc(&d_seq[(size_t)i], shard_len));
        HIPOK(hipMalloc(&d_all[(size_t)i], total_words * 8 + 64));
        HIPOK(hipMemset(d_all[(size_t)i], 0xEE, total_words * 8 + 64));
    }
    HIPOK(hipDeviceSynchronize());
    if (P > 1) { // a per-rank call from the thread that holds all ranks would wait for peers it cannot reach: refused, nothing enqueued
        uint64_t sends0 = 0, recvs0 = 0, sends1 = 0, recvs1 = 0;
        auto totals = reinterpret_cast<void (*)(uint64_t *, uint64_t *)>(dlsym(RTLD_DEFAULT, "mock_rccl_totals"));
        totals(&sends0, &recvs0);
        const int a = bitnuc_encode_sharded_allgather_overlapped_dev(ctxs[0], comms[0], d_seq[0], shard_len, n_chunks, d_all[0], &err);
        const int b = bitnuc_encode_sharded_allgather_dev(ctxs[0], comms[0], d_seq[0], shard_len, d_all[0], &err);
        const int c = bitnuc_allgather_words_dev(ctxs[0], comms[0], d_all[0], count, d_all[0], &err);
        totals(&sends1, &recvs1);
        if (a != BITNUC_UNSUPPORTED || b != BITNUC_UNSUPPORTED || c != BITNUC_UNSUPPORTED || sends0 != sends1 || recvs0 != recvs1)
            complain(0, "per-rank entry points on a single-process communicator: statuses " + std::to_string(a) + " " + std::to_string(b) + " " + std::to_string(c) + " (expected 6 6 6, nothing sent)");
        // and the _all forms refuse communicators that are not the whole single-process set in rank order
        std::vector<bitnuc_comm *> swapped(comms);
        std::swap(swapped[0], swapped[1]);
        if (bitnuc_encode_sharded_allgather_overlapped_all(P, ctxs.data(), swapped.data(), d_seq.data(), shard_len, n_chunks, d_all.data(), &err) != BITNUC_UNSUPPORTED) complain(0, "_overlapped_all accepted communicators out of rank order");
        if (bitnuc_encode_sharded_allgather_all(P - 1, ctxs.data(), comms.data(), d_seq.data(), shard_len, d_all.data(), &err) != BITNUC_UNSUPPORTED) complain(0, "_all accepted a subset of the ranks");
    }
    std::vector<uint64_t> got(total_words + 8);
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < P; ++i) BNOK(bitnuc_nucgen_dev(ctxs[(size_t)i], d_seq[(size_t)i], shard_len, kSeed + (uint64_t)r, (uint64_t)i * shard_len, 0, &err));
        const bool plant = r == rounds - 1 && bad_rank >= 0;
        if (plant) { BNOK(bitnuc_ctx_sync(ctxs[(size_t)bad_rank], &err)); HIPOK(hipMemset(d_seq[(size_t)bad_rank] + bad_offset, 'N', 1)); HIPOK(hipDeviceSynchronize()); }
        const int st = overlap ? bitnuc_encode_sharded_allgather_overlapped_all(P, ctxs.data(), comms.data(), d_seq.data(), shard_len, n_chunks, d_all.data(), &err)
                               : bitnuc_encode_sharded_allgather_all(P, ctxs.data(), comms.data(), d_seq.data(), shard_len, d_all.data(), &err);
        if (plant) {
            if (st != BITNUC_INVALID_BASE || err.byte != 'N' || err.index != bad_offset || err.value != (uint64_t)bad_rank)
                complain(bad_rank, "planted byte: status " + std::to_string(st) + " byte " + std::to_string(err.byte) + " index " + std::to_string(err.index) + " rank " + std::to_string(err.value));
        } else if (st != BITNUC_OK) complain(-1, "_all call: status " + std::to_string(st) + " backend " + std::to_string(err.backend_code));
        // the _all forms synchronise every stream themselves: the buffers are the caller's now (plain blocking copies below)
        for (int i = 0; i < P; ++i) {
            HIPOK(hipMemcpy(got.data(), d_all[(size_t)i], total_words * 8 + 64, hipMemcpyDeviceToHost));
            for (size_t s = 0; s < (size_t)P; ++s) {
                if ((int)s == bad_rank && r == rounds - 1) continue;
                if (memcmp(got.data() + s * count, expect[(size_t)r].data() + s * count, count * 8) != 0) {
                    size_t w = 0;
                    while (got[s * count + w] == expect[(size_t)r][s * count + w]) ++w;
                    complain(i, "round " + std::to_string(r) + ": slot of rank " + std::to_string(s) + " differs from the single-GPU encode at word " + std::to_string(w) + " of " + std::to_string(count));
                    break;
                }
            }
            for (size_t k = 0; k < 8; ++k)
                if (got[total_words + k] != 0xEEEEEEEEEEEEEEEEull) { complain(i, "wrote past the gathered buffer"); break; }
        }
    }
    for (int i = 0; i < P; ++i) {
        bitnuc_comm_destroy(comms[(size_t)i]);
        (void)hipFree(d_seq[(size_t)i]);
        (void)hipFree(d_all[(size_t)i]);
        bitnuc_ctx_destroy(ctxs[(size_t)i]);
    }
}


// ---- ragged batch: whole sequences per rank, in-place gather of unequal word counts ------------------------------------------------
std::vector<uint64_t> ragged_offsets(size_t nseq, uint64_t seed) {
    std::vector<uint64_t> off(nseq + 1, 0);
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = 0; i < nseq; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        uint64_t len = (x >> 11) % 400;                 // reads of 0..399 bases: empty sequences occur
        if (x % 7 == 0) len = 0;
        if (nseq > 8 && i == nseq / 3) len = 200000 + (x % 1000); // one sequence longer than a fair share of any P here
        off[i + 1] = off[i] + len;
    }
    return off;
}

// threaded_all: the communicators come from bitnuc_comm_init_all_devices but every rank gets its own thread (ordinary NCCL usage): after
// bitnuc_comm_set_threaded(comm, 1) the per-rank entry points accept them and the _all forms refuse them
void run_ragged(int P, size_t nseq, uint64_t seed, bool single_process, int rounds, bool threaded_all = false) {
    int rank = -1;
    bitnuc_err err;
    memset(&err, 0, sizeof err);
    const std::vector<uint64_t> off = ragged_offsets(nseq, seed);
    const size_t total_bases = (size_t)off[nseq];
    std::vector<size_t> seq_first((size_t)P + 1);
    std::vector<uint64_t> word_first((size_t)P + 1);
    BNOK(bitnuc_batch_shard_ranges(off.data(), nseq, P, seq_first.data(), word_first.data(), &err));
    const size_t total_words = (size_t)word_first[(size_t)P];
    std::vector<size_t> counts((size_t)P);
    int nonempty = 0;
    for (int r = 0; r < P; ++r) { counts[(size_t)r] = (size_t)(word_first[(size_t)r + 1] - word_first[(size_t)r]); nonempty += counts[(size_t)r] != 0; }
    // expected per round: one context, the whole batch
    std::vector<std::vector<uint64_t>> expect((size_t)rounds, std::vector<uint64_t>(total_words + 1));
    {
        bitnuc_ctx *c = nullptr;
        BNOK(bitnuc_ctx_create(0, &c, &err));
        uint8_t *d_seq = nullptr; uint64_t *d_off = nullptr, *d_words = nullptr;
        HIPOK(hipMalloc(&d_seq, total_bases + 64)); HIPOK(hipMalloc(&d_off, (nseq + 1) * 8)); HIPOK(hipMalloc(&d_words, total_words * 8 + 64));
        HIPOK(hipMemcpy(d_off, off.data(), (nseq + 1) * 8, hipMemcpyHostToDevice));
        bitnuc_batch_plan *plan = nullptr;
        size_t tw = 0;
        BNOK(bitnuc_batch_plan_create(c, &plan, &err));
        BNOK(bitnuc_batch_plan_build_dev(c, plan, d_off, nseq, &tw, &err));
        if (tw != total_words) complain(-1, "bitnuc_batch_shard_ranges' total " + std::to_string(total_words) + " != the plan's " + std::to_string(tw));
        for (int r = 0; r < rounds; ++r) {
            if (total_bases) BNOK(bitnuc_nucgen_dev(c, d_seq, total_bases, kSeed + (uint64_t)r, 0, 2 /* lower-case mix */, &err));
            BNOK(bitnuc_encode_batch_plan_dev(c, plan, d_seq, d_words, &err));
            BNOK(bitnuc_ctx_sync(c, &err));
            HIPOK(hipMemcpy(expect[(size_t)r].data(), d_words, total_words * 8, hipMemcpyDeviceToHost));
        }
        bitnuc_batch_plan_destroy(plan);
        (void)hipFree(d_seq); (void)hipFree(d_off); (void)hipFree(d_words);
        bitnuc_ctx_destroy(c);
    }
    struct RankState { bitnuc_ctx *c = nullptr; bitnuc_comm *comm = nullptr; bitnuc_batch_plan *plan = nullptr; uint8_t *d_seq = nullptr; uint64_t *d_off = nullptr, *d_all = nullptr; size_t nbases = 0, base0 = 0, nloc = 0; };
    auto setup = [&](RankState &st, int r) {
        const int rank = r;
        bitnuc_err err; // (the lambdas run on the ranks' threads: each call reports through its own)
        memset(&err, 0, sizeof err);
        const size_t s0 = seq_first[(size_t)r], s1 = seq_first[(size_t)r + 1];
        st.nloc = s1 - s0; st.base0 = (size_t)off[s0]; st.nbases = (size_t)(off[s1] - off[s0]);
        std::vector<uint64_t> local(st.nloc + 1);
        for (size_t i = 0; i <= st.nloc; ++i) local[i] = off[s0 + i] - off[s0]; // the run's offsets rebased to 0
        HIPOK(hipMalloc(&st.d_seq, st.nbases + 64)); HIPOK(hipMalloc(&st.d_off, (st.nloc + 1) * 8)); HIPOK(hipMalloc(&st.d_all, total_words * 8 + 64));
        HIPOK(hipMemcpy(st.d_off, local.data(), (st.nloc + 1) * 8, hipMemcpyHostToDevice));
        HIPOK(hipMemset(st.d_all, 0xEE, total_words * 8 + 64));
        size_t tw = 0;
        BNOK(bitnuc_batch_plan_create(st.c, &st.plan, &err));
        BNOK(bitnuc_batch_plan_build_dev(st.c, st.plan, st.d_off, st.nloc, &tw, &err));
        if (tw != counts[(size_t)r]) complain(r, "the run's plan holds " + std::to_string(tw) + " words, the split says " + std::to_string(counts[(size_t)r]));
    };
    auto encode = [&](RankState &st, int r, int round) {
        const int rank = r;
        bitnuc_err err;
        memset(&err, 0, sizeof err);
        if (st.nbases) BNOK(bitnuc_nucgen_dev(st.c, st.d_seq, st.nbases, kSeed + (uint64_t)round, (uint64_t)st.base0, 2, &err)); // exactly this rank's bases of the stream
        BNOK(bitnuc_encode_batch_plan_dev(st.c, st.plan, st.d_seq, st.d_all + word_first[(size_t)r], &err));
    };
    auto check = [&](RankState &st, int r, int round) {
        const int rank = r;
        std::vector<uint64_t> got(total_words + 8);
        HIPOK(hipMemcpy(got.data(), st.d_all, total_words * 8 + 64, hipMemcpyDeviceToHost));
        if (memcmp(got.data(), expect[(size_t)round].data(), total_words * 8) != 0) {
            size_t w = 0;
            while (got[w] == expect[(size_t)round][w]) ++w;
            int owner = 0;
            while (w >= word_first[(size_t)owner + 1]) ++owner;
            complain(r, "round " + std::to_string(round) + ": word " + std::to_string(w) + " (slot of rank " + std::to_string(owner) + ") differs from the single-context batch encode");
        }
        for (size_t k = 0; k < 8; ++k)
            if (got[total_words + k] != 0xEEEEEEEEEEEEEEEEull) { complain(r, "wrote past the gathered buffer"); break; }
    };
    auto teardown = [&](RankState &st) {
        bitnuc_batch_plan_destroy(st.plan);
        (void)hipFree(st.d_seq); (void)hipFree(st.d_off); (void)hipFree(st.d_all);
    };
    if (single_process) {
        std::vector<int> devs((size_t)P, 0);
        std::vector<bitnuc_ctx *> ctxs((size_t)P, nullptr);
        std::vector<bitnuc_comm *> comms((size_t)P, nullptr);
        BNOK(bitnuc_comm_init_all_devices(P, devs.data(), ctxs.data(), comms.data(), &err));
        std::vector<RankState> st((size_t)P);
        std::vector<uint64_t *> alls((size_t)P);
        for (int r = 0; r < P; ++r) { st[(size_t)r].c = ctxs[(size_t)r]; st[(size_t)r].comm = comms[(size_t)r]; setup(st[(size_t)r], r); alls[(size_t)r] = st[(size_t)r].d_all; }
        if (P > 1 && bitnuc_allgatherv_words_dev(ctxs[0], comms[0], counts.data(), alls[0], &err) != BITNUC_UNSUPPORTED) complain(0, "bitnuc_allgatherv_words_dev accepted a single-process communicator");
        for (int round = 0; round < rounds; ++round) {
            for (int r = 0; r < P; ++r) encode(st[(size_t)r], r, round);
            BNOK(bitnuc_allgatherv_words_all(P, ctxs.data(), comms.data(), counts.data(), alls.data(), &err)); // synchronises every stream
            for (int r = 0; r < P; ++r) check(st[(size_t)r], r, round);
        }
        for (int r = 0; r < P; ++r) { teardown(st[(size_t)r]); bitnuc_comm_destroy(comms[(size_t)r]); bitnuc_ctx_destroy(ctxs[(size_t)r]); }
    } else {
        uint8_t id[BITNUC_UNIQUE_ID_BYTES];
        BNOK(bitnuc_comm_get_unique_id(id, &err));
        std::vector<bitnuc_ctx *> all_ctxs((size_t)P, nullptr);
        std::vector<bitnuc_comm *> all_comms((size_t)P, nullptr);
        if (threaded_all) {
            std::vector<int> devs((size_t)P, 0);
            BNOK(bitnuc_comm_init_all_devices(P, devs.data(), all_ctxs.data(), all_comms.data(), &err));
            for (int r = 0; r < P; ++r)
                if (bitnuc_comm_set_threaded(all_comms[(size_t)r], 1) != 0 || bitnuc_comm_set_threaded(all_comms[(size_t)r], 1) != 1) complain(r, "bitnuc_comm_set_threaded does not return the previous setting");
            std::vector<uint64_t *> none((size_t)P, nullptr);
            if (P > 1 && bitnuc_allgatherv_words_all(P, all_ctxs.data(), all_comms.data(), counts.data(), none.data(), &err) != BITNUC_UNSUPPORTED) complain(-1, "_all form accepted communicators that were declared thread-per-rank");
        }
        Barrier bar(P);
        std::vector<std::thread> threads;
        for (int r = 0; r < P; ++r)
            threads.emplace_back([&, r] {
                const int rank = r;
                bitnuc_err err;
                memset(&err, 0, sizeof err);
                RankState st;
                if (threaded_all) { st.c = all_ctxs[(size_t)r]; st.comm = all_comms[(size_t)r]; }
                else {
                    BNOK(bitnuc_ctx_create(0, &st.c, &err));
                    BNOK(bitnuc_comm_init_rank(st.c, P, r, id, &st.comm, &err));
                }
                setup(st, r);
                for (int round = 0; round < rounds; ++round) {
                    encode(st, r, round);
                    BNOK(bitnuc_allgatherv_words_dev(st.c, st.comm, counts.data(), st.d_all, &err));
                    if (round != rounds - 1 && rounds > 2) continue; // no host wait between rounds except where the result is checked
                    BNOK(bitnuc_ctx_sync(st.c, &err));
                    check(st, r, round);
                    bar.wait(); // peers read this rank's slot out of d_all: nobody starts the next round's encode into it before all have checked
                }
                bar.wait();
                teardown(st);
                bitnuc_comm_destroy(st.comm);
                bitnuc_ctx_destroy(st.c);
            });
        for (auto &t : threads) t.join();
    }
    uint64_t sends = 0, recvs = 0;
    auto totals = reinterpret_cast<void (*)(uint64_t *, uint64_t *)>(dlsym(RTLD_DEFAULT, "mock_rccl_totals"));
    totals(&sends, &recvs);
    const bool bcast = getenv("BITNUC_GATHER_MODE") && !strcmp(getenv("BITNUC_GATHER_MODE"), "bcast");
    const uint64_t want = (uint64_t)rounds * (uint64_t)nonempty * (uint64_t)(P - 1); // every non-empty rank sends its slot to every peer, nobody sends an empty one
    if (!bcast && (sends != want || recvs != want)) { fprintf(stderr, "schedule: %llu sends / %llu receives, expected %llu each\n", (unsigned long long)sends, (unsigned long long)recvs, (unsigned long long)want); ++g_fail; }
    if (g_fail.load()) { fprintf(stderr, "FAILED: %d complaint(s)\n", g_fail.load()); _exit(1); }
    printf("ok ragged P=%d sequences=%zu bases=%zu words=%zu nonempty_ranks=%d mode=%s rounds=%d messages=%llu\n", P, nseq, total_bases, total_words, nonempty,
           single_process ? "ragged_all" : threaded_all ? "ragged_threaded" : "ragged", rounds, (unsigned long long)sends);
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s P shard_len n_chunks oneshot|overlap|oneshot_all|overlap_all rounds [bad_rank bad_offset]\n", argv[0]); return 2; }
    const int P = atoi(argv[1]);
    const size_t shard_len = strtoull(argv[2], nullptr, 10);
    const int n_chunks = atoi(argv[3]);
    const bool overlap = !strncmp(argv[4], "overlap", 7);
    const bool single_process = strstr(argv[4], "_all") != nullptr;
    const int rounds = atoi(argv[5]);
    const int bad_rank = argc > 7 ? atoi(argv[6]) : -1;
    const size_t bad_offset = argc > 7 ? strtoull(argv[7], nullptr, 10) : 0;
    if (!strncmp(argv[4], "ragged", 6)) {
        if (P < 1 || P > 16 || rounds < 1) return 2;
        if (!dlsym(RTLD_DEFAULT, "mock_rccl_totals")) {
            bitnuc_err e0;
            uint8_t id0[BITNUC_UNIQUE_ID_BYTES];
            (void)bitnuc_comm_get_unique_id(id0, &e0); // binds the RCCL stand-in
        }
        if (!dlsym(RTLD_DEFAULT, "mock_rccl_totals")) { fprintf(stderr, "this driver must run against tests/c/mock_rccl.cpp (LD_LIBRARY_PATH), not a real RCCL\n"); return 3; }
        run_ragged(P, shard_len, (uint64_t)n_chunks, single_process, rounds, !strcmp(argv[4], "ragged_threaded"));
        return 0;
    }
    if (P < 1 || P > 16 || shard_len % 32 || rounds < 1) return 2;
    const size_t count = shard_len / 32, total_words = count * (size_t)P;

    uint8_t id[BITNUC_UNIQUE_ID_BYTES];
    bitnuc_err err;
    if (bitnuc_comm_get_unique_id(id, &err) != BITNUC_OK) { fprintf(stderr, "no unique id (backend %d)\n", err.backend_code); return 3; }
    auto totals = reinterpret_cast<void (*)(uint64_t *, uint64_t *)>(dlsym(RTLD_DEFAULT, "mock_rccl_totals"));
    if (!totals) { fprintf(stderr, "this driver must run against tests/c/mock_rccl.cpp (LD_LIBRARY_PATH), not a real RCCL\n"); return 3; }

    // expected words per round: one context encodes the concatenation of all shards
    std::vector<std::vector<uint64_t>> expect((size_t)rounds, std::vector<uint64_t>(total_words));
    {
        bitnuc_ctx *c = nullptr;
        if (bitnuc_ctx_create(0, &c, &err) != BITNUC_OK) return 3;
        uint8_t *d_seq = nullptr;
        uint64_t *d_words = nullptr;
        if (hipMalloc(&d_seq, shard_len * P) != hipSuccess || hipMalloc(&d_words, total_words * 8) != hipSuccess) return 3;
        for (int r = 0; r < rounds; ++r) {
            if (bitnuc_nucgen_dev(c, d_seq, shard_len * P, kSeed + (uint64_t)r, 0, 0, &err) != BITNUC_OK) return 3;
            if (bitnuc_encode_dev(c, d_seq, shard_len * P, d_words, &err) != BITNUC_OK || bitnuc_ctx_sync(c, &err) != BITNUC_OK) return 3;
            if (hipMemcpy(expect[(size_t)r].data(), d_words, total_words * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        }
        (void)hipFree(d_seq);
        (void)hipFree(d_words);
        bitnuc_ctx_destroy(c);
    }

    Barrier bar(P);
    std::vector<std::thread> threads;
    if (single_process) run_single_process(P, shard_len, n_chunks, overlap, rounds, bad_rank, bad_offset, expect);
    for (int rank = 0; rank < P && !single_process; ++rank)
        threads.emplace_back([&, rank] {
            bitnuc_err err;
            memset(&err, 0, sizeof err);
            bitnuc_ctx *c = nullptr;
            bitnuc_comm *comm = nullptr;
            BNOK(bitnuc_ctx_create(0, &c, &err));
            BNOK(bitnuc_comm_init_rank(c, P, rank, id, &comm, &err));
            if (bitnuc_comm_nranks(comm) != P || bitnuc_comm_rank(comm) != rank) complain(rank, "communicator reports the wrong rank / size");
            uint8_t *d_seq = nullptr;
            uint64_t *d_all = nullptr;
            HIPOK(hipMalloc(&d_seq, shard_len));
            HIPOK(hipMalloc(&d_all, total_words * 8 + 64));
            HIPOK(hipMemset(d_all, 0xEE, total_words * 8 + 64));
            std::vector<uint64_t> got(total_words + 8);
            for (int r = 0; r < rounds; ++r) {
                // rank-disjoint slice of the round's stream, generated in place (asynchronous on the context's stream)
                BNOK(bitnuc_nucgen_dev(c, d_seq, shard_len, kSeed + (uint64_t)r, (uint64_t)rank * shard_len, 0, &err));
                const bool plant = r == rounds - 1 && rank == bad_rank;
                if (plant) { BNOK(bitnuc_ctx_sync(c, &err)); HIPOK(hipMemset(d_seq + bad_offset, 'N', 1)); HIPOK(hipDeviceSynchronize()); }
                if (overlap) BNOK(bitnuc_encode_sharded_allgather_overlapped_dev(c, comm, d_seq, shard_len, n_chunks, d_all, &err));
                else BNOK(bitnuc_encode_sharded_allgather_dev(c, comm, d_seq, shard_len, d_all, &err));
                // no host wait between rounds except where the result is checked: rounds 0..rounds-2 are checked only when rounds <= 2
                const bool check = r == rounds - 1 || rounds <= 2;
                if (!check) continue;
                const int st = bitnuc_ctx_sync(c, &err);
                if (plant) {
                    if (st != BITNUC_INVALID_BASE || err.byte != 'N' || err.index != bad_offset)
                        complain(rank, "planted byte: status " + std::to_string(st) + " byte " + std::to_string(err.byte) + " index " + std::to_string(err.index));
                } else if (st != BITNUC_OK) complain(rank, "sync: status " + std::to_string(st) + " backend " + std::to_string(err.backend_code));
                HIPOK(hipMemcpy(got.data(), d_all, total_words * 8 + 64, hipMemcpyDeviceToHost));
                for (size_t s = 0; s < (size_t)P; ++s) {
                    if ((int)s == bad_rank && r == rounds - 1) continue; // the failing shard's words are unspecified
                    if (memcmp(got.data() + s * count, expect[(size_t)r].data() + s * count, count * 8) != 0) {
                        size_t w = 0;
                        while (got[s * count + w] == expect[(size_t)r][s * count + w]) ++w;
                        complain(rank, "round " + std::to_string(r) + ": slot of rank " + std::to_string(s) + " differs from the single-GPU encode at word " + std::to_string(w) + " of " + std::to_string(count));
                        break;
                    }
                }
                for (size_t i = 0; i < 8; ++i)
                    if (got[total_words + i] != 0xEEEEEEEEEEEEEEEEull) { complain(rank, "wrote past the gathered buffer"); break; }
                bar.wait(); // peers read this rank's slot straight out of d_all: nobody starts the next round's encode into it, or frees it, before all have checked
            }
            bar.wait();
            bitnuc_comm_destroy(comm);
            (void)hipFree(d_seq);
            (void)hipFree(d_all);
            bitnuc_ctx_destroy(c);
        });
    for (auto &t : threads) t.join();
    uint64_t sends = 0, recvs = 0;
    totals(&sends, &recvs);
    // one message per ordered pair of ranks per non-empty piece (or per call for the one-shot form), each matched exactly once
    size_t pieces = 1;
    if (overlap) {
        pieces = 0;
        for (int p = 0; p < n_chunks; ++p) pieces += count * (size_t)(p + 1) / (size_t)n_chunks > count * (size_t)p / (size_t)n_chunks;
    }
    const uint64_t want = (uint64_t)rounds * pieces * (uint64_t)P * (uint64_t)(P - 1);
    if (sends != want || recvs != want) { fprintf(stderr, "schedule: %llu sends / %llu receives, expected %llu each\n", (unsigned long long)sends, (unsigned long long)recvs, (unsigned long long)want); ++g_fail; }
    if (g_fail.load()) { fprintf(stderr, "FAILED: %d complaint(s)\n", g_fail.load()); return 1; }
    printf("ok P=%d shard_len=%zu chunks=%d mode=%s rounds=%d messages=%llu\n", P, shard_len, n_chunks, argv[4], rounds, (unsigned long long)sends);
    return 0;
}
