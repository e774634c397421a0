"""Config 4 through the C ABI with 2..8 ranks on ONE GPU (-m gpu).

No box this project can reach has more than one GPU, so the multi-rank code of bitnuc_amd/csrc/comm.hip never met a second rank
on real hardware.  Here it does, against a stand-in for RCCL: tests/c/mock_rccl.cpp is built as `librccl.so.1` into a scratch
directory that goes first on the child's LD_LIBRARY_PATH (a pytest temporary directory) (comm.hip binds RCCL by dlopen of that soname), and
tests/c/multirank_driver.cpp runs P ranks as threads, one bitnuc_ctx + communicator each, all on device 0.  Checked for every
scenario: every rank's gathered buffer == a single-GPU encode of the concatenated input, nothing written past it, the number
of point-to-point messages == rounds x pieces x P x (P - 1), errors reported by the rank that owns the invalid byte only.
What this cannot show is RCCL itself and the xGMI fabric: tests/test_gpu_round3.py holds those tests (skipped below 2 GPUs).
"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from bitnuc_amd import build
    lib = build.ensure_built()
    BUILD = str(tmp_path_factory.mktemp("mock_rccl"))  # outside the repository: no file named like RCCL ever sits in the tree
    mock, exe = os.path.join(BUILD, "librccl.so.1"), os.path.join(BUILD, "multirank_driver")
    subprocess.run(["hipcc", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wl,-soname,librccl.so.1", "-o", mock,
                    os.path.join(ROOT, "tests", "c", "mock_rccl.cpp")], check=True, capture_output=True, timeout=600)
    subprocess.run(["hipcc", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "c", "multirank_driver.cpp"), "-L" + os.path.dirname(lib), "-lbitnuc_hip",
                    "-Wl,-rpath," + os.path.dirname(lib), "-ldl", "-lpthread"], check=True, capture_output=True, timeout=600)

    def run(*args, **env_extra):
        env = dict(os.environ, LD_LIBRARY_PATH=BUILD + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), **env_extra)
        r = subprocess.run([exe, *[str(a) for a in args]], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and r.stdout.startswith("ok "), (args, env_extra, r.stdout[-2000:], r.stderr[-4000:])
        return r.stdout
    return run


ODD = 32 * 100_003  # 100 003 words per shard: no piece boundary falls on a round number


@pytest.mark.parametrize("P", [2, 3, 4, 8])
def test_one_shot_gather(driver, P):
    driver(P, ODD, 1, "oneshot", 2)


@pytest.mark.parametrize("P,chunks", [(2, 1), (2, 3), (2, 8), (3, 5), (4, 8), (5, 7), (8, 8), (8, 13)])
def test_chunked_in_place_gather(driver, P, chunks):
    out = driver(P, ODD, chunks, "overlap", 2)
    assert f"messages={2 * chunks * P * (P - 1)}" in out


@pytest.mark.parametrize("P,chunks", [(2, 4), (4, 8), (8, 8)])
def test_chunked_gather_broadcast_exchange(driver, P, chunks):
    driver(P, ODD, chunks, "overlap", 2, BITNUC_GATHER_MODE="bcast")


@pytest.mark.parametrize("P,shard,chunks,mode,rounds", [(8, ODD, 8, "overlap", 2), (2, 32 * 4_000_003, 4, "overlap", 2), (4, ODD, 6, "overlap", 5),
                                                        (4, ODD, 1, "oneshot", 3), (4, ODD, 4, "bcast", 2)])
def test_slow_fabric(driver, P, shard, chunks, mode, rounds):
    """The mock's transfers take 2 ms each (host function in the receiving stream): they finish long after the encode that feeds
    them, as on xGMI where the gather costs ~8x the encode.  A missing wait between the context's stream and the transfer stream
    (bitnuc_ctx_sync returning before the exchange is done, a transfer starting before its piece is encoded) then shows as stale
    words -- tools/multirank_mutation_check.py removes each wait in turn and shows these scenarios fail
    (profiles/r03_multirank_mock_mutations.txt)."""
    extra = {"BITNUC_GATHER_MODE": "bcast"} if mode == "bcast" else {}
    driver(P, shard, chunks, "overlap" if mode == "bcast" else mode, rounds, MOCK_RCCL_DELAY_US="2000", **extra)


@pytest.mark.parametrize("mode", ["oneshot", "overlap"])
def test_back_to_back_rounds_reuse_the_buffer_in_stream_order(driver, mode):
    """Five calls on the same buffers with new data each time and NO host wait in between: the transfer stream of call r+1 must
    not overwrite what a peer is still reading from call r (the all_moved handshake), and the last result must be call 4's."""
    driver(4, ODD, 6, mode, 5)
    driver(8, 32 * 20_001, 4, mode, 5)


def test_fewer_words_than_pieces(driver):
    """3 words per shard in 8 pieces: five pieces are empty and must be skipped identically by every rank."""
    out = driver(4, 96, 8, "overlap", 2)
    assert "messages=%d" % (2 * 3 * 4 * 3) in out
    driver(2, 32, 4, "overlap", 1)


@pytest.mark.parametrize("mode", ["oneshot", "overlap"])
def test_invalid_byte_is_reported_by_its_rank_only(driver, mode):
    """InvalidBase with the shard-relative offset on the rank that owns the byte (piece 5 of 8); every other rank succeeds and
    holds the other shards' words."""
    driver(4, ODD, 8, mode, 2, 2, 5 * (ODD // 8) + 12345)
    driver(8, ODD, 8, mode, 1, 7, ODD - 1)


# ---- one thread holds all ranks: bitnuc_comm_init_all[_devices] + the _all entry points -----------------------------------------
@pytest.mark.parametrize("P", [2, 4, 8])
def test_single_process_one_shot_all(driver, P):
    driver(P, ODD, 1, "oneshot_all", 2)


@pytest.mark.parametrize("P,chunks", [(2, 1), (2, 3), (3, 5), (4, 8), (8, 8), (8, 13)])
def test_single_process_chunked_in_place_gather_all(driver, P, chunks):
    """bitnuc_encode_sharded_allgather_overlapped_all: per piece ONE group holds every rank's sends and receives, issued by the one
    thread that owns all ranks.  The same driver run checks that the per-rank entry points refuse the communicator (status 6, nothing
    sent) -- against this mock, as against RCCL, a per-rank group issued rank after rank from one thread waits for peers for ever
    (here: the mock's 60 s patience, then ncclInternalError)."""
    out = driver(P, ODD, chunks, "overlap_all", 2)
    assert f"messages={2 * chunks * P * (P - 1)}" in out


@pytest.mark.parametrize("P,chunks,mode", [(8, 8, "sendrecv"), (4, 6, "sendrecv"), (2, 4, "sendrecv"), (4, 4, "bcast")])
def test_single_process_all_slow_fabric(driver, P, chunks, mode):
    extra = {"BITNUC_GATHER_MODE": "bcast"} if mode == "bcast" else {}
    driver(P, ODD if P > 2 else 32 * 4_000_003, chunks, "overlap_all", 3, MOCK_RCCL_DELAY_US="2000", **extra)
    driver(P, ODD, 1, "oneshot_all", 2, MOCK_RCCL_DELAY_US="2000")


@pytest.mark.parametrize("mode", ["oneshot_all", "overlap_all"])
def test_single_process_all_reports_the_rank_of_an_invalid_byte(driver, mode):
    """err.value = the rank whose shard holds the byte, err.index relative to that shard; every other rank's slot is still exchanged."""
    driver(4, ODD, 8, mode, 2, 2, 5 * (ODD // 8) + 12345)
    driver(8, ODD, 8, mode, 1, 7, ODD - 1)


def test_single_process_all_fewer_words_than_pieces(driver):
    out = driver(4, 96, 8, "overlap_all", 2)
    assert "messages=%d" % (2 * 3 * 4 * 3) in out


# ---- a ragged batch split by whole sequences + in-place gather of UNEQUAL word counts (SURVEY 8e sentence 2) ------------------------------
@pytest.mark.parametrize("P", [2, 4, 8])
@pytest.mark.parametrize("mode", ["ragged", "ragged_all"])
def test_ragged_batch_whole_sequences_allgatherv(driver, P, mode):
    """bitnuc_batch_shard_ranges -> per-rank bitnuc_batch_plan + bitnuc_encode_batch_plan_dev into the rank's slot ->
    bitnuc_allgatherv_words_dev (threads, one per rank) / bitnuc_allgatherv_words_all (one thread): every rank's buffer == ONE context's
    plan encode of the whole batch (which tests/test_gpu_parity.py pins to the oracle's per-sequence loop).  3000 reads of 0..399 bases
    with empty sequences and one 200 000-base sequence (longer than a fair share: some rank is left with no sequence at P = 8); a rank
    with an empty slot sends nothing: messages == rounds x non-empty ranks x (P - 1)."""
    out = driver(P, 3000, 11, mode, 2)
    assert "nonempty_ranks=" in out
    driver(P, 5, 3, mode, 1)  # fewer sequences than ranks at P = 8


@pytest.mark.parametrize("mode", ["ragged", "ragged_all"])
def test_ragged_batch_slow_fabric_and_broadcast_exchange(driver, mode):
    driver(4, 3000, 5, mode, 3, MOCK_RCCL_DELAY_US="2000")
    driver(4, 3000, 5, mode, 2, BITNUC_GATHER_MODE="bcast")


def test_init_all_communicators_driven_by_one_thread_per_rank(driver):
    """bitnuc_comm_set_threaded: communicators made by bitnuc_comm_init_all_devices whose ranks each get their own host thread (ordinary
    NCCL usage) -- the per-rank entry points accept them after the declaration, the _all forms refuse them."""
    driver(4, 3000, 7, "ragged_threaded", 2)
    driver(8, 300, 2, "ragged_threaded", 1)


def test_ragged_batch_back_to_back_rounds(driver):
    """Five rounds on the same buffers, new data each round, no host wait in between (per-rank form): round r+1's encode into a rank's slot
    is ordered behind round r's sends from it by the context's stream."""
    driver(4, 2000, 9, "ragged", 5)
