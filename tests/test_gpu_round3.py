"""GPU tests added in round 3 (all through the C ABI):

* the lifetime of data-error slots: launches recorded into a hipGraph keep a persistent slot that every sync examines and
  re-arms (capture -> sync -> replay on an invalid byte -> sync raises; later launches and replays are clean), and the slot
  ring grows instead of synchronising inside an asynchronous call (packing/avx.rs:86-91 is the error rule kept);
* config 4 through the C ABI on 2 / 4 / 8 GPUs (skipped where fewer devices are visible): single-process init_all,
  one process per GPU through init_rank, one-shot and chunked-overlap (in-place) all-gather, both exchange modes; the
  chunked form on a 1-rank communicator against the one-shot form on the one GPU every box has;
* the pipelined host-pointer path with >= 9 chunks, so that every buffer-reuse guard of the three-stream pipeline runs
  (BITNUC_PIPE_CHUNK_MB=1), with invalid bytes planted in late chunks -- encode / decode and, through the same engine, the
  host-pointer k-mer batch (dense, every window, strided with gaps) and scan;
* the pipe's thread budget (bitnuc_host_pipe_info) and the xGMI link probe's argument rule;
* table-driven ragged batches through the asynchronous plan emission against the explicit plan and the oracle loop.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xB17C0DE


# ---- error-slot lifetime ----------------------------------------------------------------------------------------------
def test_captured_launch_keeps_its_error_slot_across_syncs(oracle):
    """after the first sync a replayed launch used to latch into a slot nobody looked at (error lost)
    and the stale value was later reported against an unrelated launch."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = 1_000_003
    nw = (n + 31) // 32
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c = bn.Context(0, stream=s.cuda_stream)
        seq = torch.empty(n, dtype=torch.uint8, device=dev)
        other = torch.empty(n, dtype=torch.uint8, device=dev)
        words = torch.empty(nw, dtype=torch.int64, device=dev)
        words2 = torch.empty(nw, dtype=torch.int64, device=dev)
        c.nucgen_dev(seq, n, 1)
        c.nucgen_dev(other, n, 9)
        c.encode_dev(seq, n, words)  # warm-up outside the capture
        c.sync()
        assert c.get("captured_slots") == 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            c.encode_dev(seq, n, words)
        assert c.get("captured_slots") == 1
        c.sync()  # the sync that used to orphan the captured launch's slot
        # replay on an invalid byte AFTER that sync: the error must surface at the next sync, with its byte and index
        seq[777_001] = ord("N")
        g.replay()
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", ord("N"), 777_001)
        # an ordinary valid launch afterwards is clean: nothing stale is reported against it
        c.encode_dev(other, n, words2)
        c.sync()
        assert np.array_equal(words2[:2000].cpu().numpy().view(np.uint64), oracle.encode(other[:64000].cpu().numpy()))
        # the slot was re-armed: a replay on valid data is clean, a second invalid replay is reported again
        seq[777_001] = ord("A")
        g.replay()
        c.sync()
        seq[5] = 0xFF
        g.replay()
        c.encode_dev(other, n, words2)  # an ordinary launch queued behind the replay does not hide it
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        assert (ei.value.byte, ei.value.index) == (0xFF, 5)
        # ordinary launches are reported before captured ones (include/bitnuc_hip.h)
        other[123] = ord("x")
        g.replay()
        c.encode_dev(other, n, words2)
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        assert (ei.value.byte, ei.value.index) == (ord("x"), 123)
        c.sync()  # both slots were re-armed by the sync that reported
        # host-pointer calls report their own error, not a captured launch's
        g.replay()  # seq[5] is still invalid
        h = oracle.nucgen(4096, 3)
        c.set_variant("force_gpu", 1)
        assert np.array_equal(c.encode_array(h), oracle.encode(h))
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        assert (ei.value.byte, ei.value.index) == (0xFF, 5)
        c.close()


def test_slot_ring_grows_without_synchronising(oracle):
    """More asynchronous launches than the ring's first block between two syncs: the ring grows (round 2 drained the stream
    inside the 4097th call), every launch keeps its own slot and the first error in launch order is the one reported."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    c = bn.Context(0)
    n = 4096
    launches = 4096 + 4096 + 900  # crosses two block boundaries
    seq = torch.from_numpy(oracle.nucgen(n, SEED)).to(dev)
    bad_a = seq.clone()
    bad_a[100] = ord("N")
    bad_b = seq.clone()
    bad_b[7] = ord("Z")
    words = torch.empty(n // 32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for i in range(launches):
        src = bad_a if i == 6000 else (bad_b if i == 8500 else seq)
        c.encode_dev(src, n, words)
    with pytest.raises(bn.NucleotideError) as ei:
        c.sync()
    assert (ei.value.byte, ei.value.index) == (ord("N"), 100)
    c.sync()  # everything was re-armed / released
    c.encode_dev(seq, n, words)
    c.sync()
    assert np.array_equal(words.cpu().numpy().view(np.uint64), oracle.encode(seq.cpu().numpy()))
    # the other order: the later launch's error must not win
    for i in range(5000):
        c.encode_dev(bad_b if i == 4500 else (bad_a if i == 4999 else seq), n, words)
    with pytest.raises(bn.NucleotideError) as ei:
        c.sync()
    assert (ei.value.byte, ei.value.index) == (ord("Z"), 7)
    c.close()


# ---- config 4 through the C ABI ---------------------------------------------------------------------------------------
def _gpus():
    import torch
    return torch.cuda.device_count()


WORLDS = [2, 4, 8]


@pytest.mark.parametrize("chunks", [1, 5], ids=["one_piece", "five_pieces"])
def test_overlapped_allgather_single_rank_equals_one_shot(oracle, chunks):
    """bitnuc_encode_sharded_allgather_overlapped_dev on a 1-rank RCCL communicator (every box has one GPU): the pieces,
    their events and the second stream run; the result equals the one-shot form and the oracle; an invalid byte in a late
    piece is reported with its shard-relative index."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    c = bn.Context(0)
    comm = bn.Comm(c, 1, 0, bn.Comm.unique_id())
    n = 32 * 300_007
    seq = torch.from_numpy(oracle.nucgen(n, SEED)).to(dev)
    one = torch.zeros(n // 32, dtype=torch.int64, device=dev)
    two = torch.zeros(n // 32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    comm.encode_sharded_allgather_dev(seq, n, one)
    comm.encode_sharded_allgather_overlapped_dev(seq, n, chunks, two)
    c.sync()
    assert torch.equal(one, two)
    assert np.array_equal(two.cpu().numpy().view(np.uint64), oracle.encode(seq.cpu().numpy()))
    bad = n - 1000
    seq[bad] = ord("N")
    comm.encode_sharded_allgather_overlapped_dev(seq, n, chunks, two)
    with pytest.raises(bn.NucleotideError) as ei:
        c.sync()
    assert (ei.value.byte, ei.value.index) == (ord("N"), bad)
    with pytest.raises(bn.NucleotideError) as ei:  # shard_len must be whole words
        comm.encode_sharded_allgather_overlapped_dev(seq, n - 5, chunks, two)
    assert ei.value.kind == "InvalidLength"
    comm.close()
    c.close()


@pytest.mark.parametrize("world", WORLDS)
def test_multi_gpu_sharded_allgather_single_process(oracle, world):
    """bitnuc_comm_init_all + bitnuc_encode_sharded_allgather_all / _overlapped_all on `world` GPUs == single-GPU encode of the concatenation."""
    if _gpus() < world:
        pytest.skip(f"needs >= {world} GPUs")
    import torch
    from bitnuc_amd import _lib as L
    import bitnuc_amd as bn
    lib = L.load()
    n = 32 * 1_000_003  # per shard
    ctxs, comms = (C.c_void_p * world)(), (C.c_void_p * world)()
    err = L.BitnucErr()
    assert lib.bitnuc_comm_init_all(world, ctxs, comms, C.byref(err)) == 0, err.backend_code
    shards, alls = [], []
    for r in range(world):
        d = torch.device("cuda", r)
        shards.append(torch.from_numpy(oracle.nucgen(n, SEED, r * n)).to(d))
        alls.append(torch.zeros(world * n // 32, dtype=torch.int64, device=d))
    for r in range(world):
        torch.cuda.synchronize(r)
    sp = (C.c_void_p * world)(*[t.data_ptr() for t in shards])
    ap = (C.c_void_p * world)(*[t.data_ptr() for t in alls])
    assert lib.bitnuc_encode_sharded_allgather_all(world, ctxs, comms, sp, n, ap, C.byref(err)) == 0, err.backend_code
    c0 = bn.Context(0)  # single-GPU encode of the concatenation
    whole = torch.cat([s.to("cuda:0") for s in shards])
    ref = torch.empty(world * n // 32, dtype=torch.int64, device="cuda:0")
    c0.encode_dev(whole, world * n, ref)
    c0.sync()
    for r in range(world):
        assert torch.equal(alls[r].to("cuda:0"), ref), r
    assert np.array_equal(ref[:4096].cpu().numpy().view(np.uint64), oracle.encode(oracle.nucgen(32 * 4096, SEED)))
    # the chunked in-place exchange driven by this one thread: per piece one RCCL group holds every rank's sends and receives
    for chunks in (1, 4, 7):
        for r in range(world):
            alls[r].zero_()
            torch.cuda.synchronize(r)
        assert lib.bitnuc_encode_sharded_allgather_overlapped_all(world, ctxs, comms, sp, n, chunks, ap, C.byref(err)) == 0, (chunks, err.backend_code)
        for r in range(world):
            assert torch.equal(alls[r].to("cuda:0"), ref), (chunks, r)
    # the per-rank entry points refuse a communicator whose ranks all live in this thread (they would wait for each other)
    assert lib.bitnuc_comm_single_process(comms[0]) == 1
    assert lib.bitnuc_encode_sharded_allgather_overlapped_dev(ctxs[0], comms[0], shards[0].data_ptr(), n, 4, alls[0].data_ptr(), C.byref(err)) == L.UNSUPPORTED
    assert lib.bitnuc_encode_sharded_allgather_dev(ctxs[0], comms[0], shards[0].data_ptr(), n, alls[0].data_ptr(), C.byref(err)) == L.UNSUPPORTED
    c0.close()
    for r in range(world):
        lib.bitnuc_comm_destroy(comms[r])
        lib.bitnuc_ctx_destroy(ctxs[r])
    # the same through the Python mirror (bn.CommGroup), on the devices in reverse order (rank r on device world - 1 - r)
    g = bn.CommGroup(world, devices=list(range(world - 1, -1, -1)))
    shards_r = [shards[r].to(torch.device("cuda", world - 1 - r)) for r in range(world)]
    alls_r = [torch.zeros(world * n // 32, dtype=torch.int64, device=torch.device("cuda", world - 1 - r)) for r in range(world)]
    for r in range(world):
        torch.cuda.synchronize(r)
    for chunks in (0, 5):
        g.encode_sharded_allgather(shards_r, n, alls_r, n_chunks=chunks)
        for r in range(world):
            assert torch.equal(alls_r[r].to("cuda:0"), ref), (chunks, r)
            alls_r[r].zero_()
            torch.cuda.synchronize(world - 1 - r)
    g.close()


def _rank_worker(rank, world, uid_path, n, mode, q):
    import time
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if mode == "bcast":
        os.environ["BITNUC_GATHER_MODE"] = "bcast"
    import bitnuc_amd as bn
    import oracle_py
    torch.cuda.set_device(rank)
    c = bn.Context(rank)
    if rank == 0:
        uid = bn.Comm.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    comm = bn.Comm(c, world, rank, uid)
    dev = torch.device("cuda", rank)
    shard = torch.from_numpy(oracle_py.nucgen(n, SEED, rank * n)).to(dev)
    expect = oracle_py.encode(oracle_py.nucgen(world * n, SEED))
    ok = True
    for chunks in (0, 1, 4):  # 0 = the one-shot ncclAllGather form
        allw = torch.zeros(world * n // 32, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        if chunks == 0:
            comm.encode_sharded_allgather_dev(shard, n, allw)
        else:
            comm.encode_sharded_allgather_overlapped_dev(shard, n, chunks, allw)
        c.sync()
        ok = ok and bool(np.array_equal(allw.cpu().numpy().view(np.uint64), expect))
    q.put((rank, ok))
    comm.close()
    c.close()


@pytest.mark.parametrize("mode", ["sendrecv", "bcast"])
@pytest.mark.parametrize("world", WORLDS)
def test_multi_gpu_sharded_allgather_one_process_per_gpu(oracle, tmp_path, world, mode):
    """bitnuc_comm_init_rank in `world` processes (one per GPU): after the one-shot and after the chunked in-place exchange
    every rank holds the packed words of the whole sequence."""
    if _gpus() < world:
        pytest.skip(f"needs >= {world} GPUs")
    import torch.multiprocessing as mp
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    uid_path = str(tmp_path / "uid.bin")
    n = 32 * 250_001
    procs = [mctx.Process(target=_rank_worker, args=(r, world, uid_path, n, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(world)) == {r: True for r in range(world)}


def test_peer_link_probe_arguments():
    """The xGMI probe needs two devices in one process; on a one-GPU box it says so (Unsupported, value = device count)
    instead of inventing a number; with two or more it returns a positive rate per link and for all links at once."""
    import bitnuc_amd as bn
    from bitnuc_amd import api
    n = _gpus()
    with pytest.raises(bn.NucleotideError) as ei:
        api.peer_link_probe(0, [0])  # src == dst
    assert ei.value.kind == "Unsupported"
    with pytest.raises(bn.NucleotideError) as ei:
        api.peer_link_probe(0, [n])  # no such device
    assert ei.value.kind == "Unsupported"
    if n >= 2:
        r = api.peer_link_probe(0, list(range(1, n)), nbytes=64 << 20, reps=2)
        assert len(r["gb_s_each"]) == n - 1 and all(x > 1.0 for x in r["gb_s_each"]) and r["gb_s_all"] > 1.0


# ---- pipelined host-pointer path ---------------------------------------------------------------------------------------
_PIPE_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import bitnuc_amd as bn
import oracle_py as oracle
SEED = 0xB17C0DE
chunk = 1 << 20
c = bn.Context(0)
info = c.host_pipe_info()
assert info["chunk_bases"] == chunk and info["depth"] == 3, info
for n in (9 * chunk + 17, 12 * chunk, 11 * chunk + chunk // 2 + 5):
    s = oracle.nucgen(n, SEED + n, 0, 2)
    expect = oracle.encode(s)
    w = np.zeros(len(expect), dtype=np.uint64)
    c.encode_into(s, w)
    assert np.array_equal(w, expect), n
    d = np.zeros(n, dtype=np.uint8)
    c.decode_into(w, n, d)
    assert np.array_equal(d, s & 0xDF), n
    # a second pass over the same buffers with different data: a stale chunk of pass 1 would show
    s2 = oracle.nucgen(n, SEED ^ n, 5, 0)
    c.encode_into(s2, w)
    assert np.array_equal(w, oracle.encode(s2)), n
    c.decode_into(w, n, d)
    assert np.array_equal(d, s2), n
n = 10 * chunk + 1000
s = oracle.nucgen(n, SEED, 0, 0)
expect = oracle.encode(s)
for bad in (3 * chunk + 5, 4 * chunk - 1, 7 * chunk, 9 * chunk + 33, 10 * chunk + 999):
    t = s.copy()
    t[bad] = ord("N")
    if bad + 2 * chunk < n:
        t[bad + 2 * chunk] = ord("X")  # an invalid byte in a later chunk must not win
    try:
        c.encode_array(t)
        raise SystemExit("no error for " + str(bad))
    except bn.NucleotideError as e:
        assert (e.kind, e.byte, e.index) == ("InvalidBase", ord("N"), bad), (bad, e.kind, e.byte, e.index)
        assert np.array_equal(e.words, expect[: bad // 32]), bad
    assert np.array_equal(c.encode_array(s), expect)  # the pipe is idle and clean after an error
# the k-mer host calls ride the same engine (round 3): dense 31-mers, every window, strided with gaps, the scan (input AND output a byte
# per base: the second A-sized buffer set) -- many 1 Mi chunks each, against the oracle, then the first invalid byte from a late chunk
k = 31
cnt = 400_003
km = oracle.nucgen(cnt * k, SEED + 1, 0, 2)
assert np.array_equal(c.as_2bit_batch(km, k, k, cnt), oracle.as_2bit_batch(km, k, k, cnt))
nwin_src = oracle.nucgen(10 * chunk + 777, SEED + 2, 0, 2)
for kk in (31, 32, 5):
    cw = len(nwin_src) - kk + 1
    assert np.array_equal(c.as_2bit_batch(nwin_src, kk, 1, cw), oracle.as_2bit_batch(nwin_src, kk, 1, cw)), kk
gap = oracle.nucgen(9 * chunk, SEED + 3, 0, 0)
cg = (len(gap) - 21) // 40 + 1
assert np.array_equal(c.as_2bit_batch(gap, 21, 40, cg), oracle.as_2bit_batch(gap, 21, 40, cg))
ref = oracle.nucgen(9 * chunk + 17, SEED + 4, 0, 2)
for kk in (31, 32, 7):
    q = oracle.as_2bit(ref[12345:12345 + kk])
    assert np.array_equal(c.kmer_hdist_scan(ref, kk, q), oracle.kmer_hdist_scan(ref, kk, q)), kk
for bad in (5 * chunk + 7, 8 * chunk + 31, len(ref) - 1):
    t = ref.copy()
    t[bad] = ord("N")
    if bad + chunk < len(t):
        t[bad + chunk] = ord("X")
    for call in (lambda: c.kmer_hdist_scan(t, 31, 0), lambda: c.as_2bit_batch(t, 31, 1, len(t) - 30)):
        try:
            call()
            raise SystemExit("no error for " + str(bad))
        except bn.NucleotideError as e:
            assert (e.kind, e.byte, e.index) == ("InvalidBase", ord("N"), bad), (bad, e.kind, e.byte, e.index)
assert np.array_equal(c.kmer_hdist_scan(ref, 31, 0), oracle.kmer_hdist_scan(ref, 31, 0))  # idle and clean after the errors
# fixed-length reads from host memory: back to back (encode and decode pipelined) and newline-separated (encode pipelined)
Lr, cr = 150, 70_001
for stride in (Lr, Lr + 1):
    fr = oracle.nucgen(cr * stride, SEED + 5, 0, 2)
    if stride != Lr:
        fr[Lr::stride] = ord("\n")  # separators are never examined
    expw = np.concatenate([oracle.encode(fr[i * stride:i * stride + Lr]) for i in range(cr)])
    got = c.encode_fixed(fr, Lr, stride, cr)  # (count, words per read)
    assert np.array_equal(got.reshape(-1), expw), stride
    backr = c.decode_fixed(got, Lr, stride, out=fr.copy() if stride != Lr else None)
    ref_up = fr & 0xDF if stride == Lr else np.where(fr == ord("\n"), fr, fr & 0xDF)
    assert np.array_equal(backr[:cr * stride], ref_up), stride
    bad = 61_234 * stride + 77
    t = fr.copy()
    t[bad] = ord("N")
    t[bad + 5 * stride] = ord("X")
    try:
        c.encode_fixed(t, Lr, stride, cr)
        raise SystemExit("no error")
    except bn.NucleotideError as e:
        assert (e.kind, e.byte, e.index) == ("InvalidBase", ord("N"), bad), (stride, e.byte, e.index)
c.close()
print("pipe child ok", info)
"""


@pytest.mark.parametrize("engine", ["staged", "direct"])
def test_pipelined_host_path_reuses_every_buffer(oracle, engine):
    """with 32 Mi-base chunks no test input reached the `ci >= depth` guards of the three-stream
    pipeline.  A fresh process with BITNUC_PIPE_CHUNK_MB=1 runs 9-12 chunks per call: every pinned / device buffer is reused
    3-4 times, encode and decode are checked against the oracle, two passes with different data over the same caller
    buffers, and invalid bytes sit in chunks >= 3 with a later invalid byte that must not win."""
    env = dict(os.environ, BITNUC_PIPE_CHUNK_MB="1", BITNUC_HOST_CUTOFF="0", BITNUC_PIPE_IMPL=engine)  # both engines of csrc/host_pipe.h
    r = subprocess.run([sys.executable, "-c", _PIPE_CHILD, ROOT], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "pipe child ok" in r.stdout


def test_host_pipe_budget_respects_the_cpu_quota():
    """The staging pools are sized from the CPUs this process may use (affinity AND cgroup quota), not from a constant."""
    import bitnuc_amd as bn
    c = bn.Context(0)
    info = c.host_pipe_info()
    assert 1 <= info["cores_usable"] <= info["cores_visible"]
    if info["cores_quota"]:
        assert info["cores_usable"] <= info["cores_quota"]
    for side in ("encode", "decode"):
        total = info[f"{side}_stage_in_threads"] + info[f"{side}_hand_back_threads"]
        assert 2 <= total <= max(3, info["cores_usable"]), info
    assert info["encode_stage_in_threads"] >= info["encode_hand_back_threads"]  # 1 B per base in, 0.25 B out
    assert info["decode_hand_back_threads"] >= info["decode_stage_in_threads"]  # 0.25 B per base in, 1 B out
    c.close()


# ---- table-driven ragged batches: asynchronous plan emission ---------------------------------------------------------------
def _oracle_batch(oracle, seq, off):
    words, wo = [], [0]
    for i in range(len(off) - 1):
        s = seq[int(off[i]):int(off[i + 1])]
        w = oracle.encode(s) if len(s) else np.zeros(0, np.uint64)  # the reference's idiom: one encode() per sequence
        words.append(w)
        wo.append(wo[-1] + len(w))
    return (np.concatenate(words) if words else np.zeros(0, np.uint64)), np.array(wo, dtype=np.int64)


def test_table_driven_batch_emits_its_plan_asynchronously(ctx, oracle):
    """bitnuc_encode_batch_dev / bitnuc_decode_batch_dev (offset tables only): one pass emits pad bytes + tile bases into
    context scratch (no memset, no host sync), then the plan kernels run.  Against the oracle's per-sequence loop
    (packing/avx.rs:147-148: every sequence pads its own last word) for length mixes that reach every path of the emit
    kernel: threads whose eight sequences span <= 256 / > 256 pad bytes (the wave-cooperative zeroing), empties, counts around
    the 8-per-thread and 2048-per-workgroup granularity, tables at 8-byte-aligned addresses, a batch that starts anywhere in
    its buffer; back-to-back calls with different batches reuse the scratch plan in stream order; first invalid byte."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31337)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    shapes = [
        [150] * 3001, [32] * 4099, list(rng.integers(1, 5, size=5003)), [0, 0, 5, 0, 37, 0, 0, 0, 64, 0] * 203,
        [1000003], [7, 300001, 13, 2049, 2048, 2047, 5], [9000] * 37 + [3] * 5 + [20000, 1, 0, 0, 8193] * 9,
        list(rng.integers(1, 400, size=2047)) + [100000, 31, 32, 33, 64, 1], [1, 0, 1, 1, 0, 0, 1] * 700 + [31, 1, 33, 0, 1] * 50,
        [0] * 1000 + [40] + [0] * 2000 + [7, 0, 0, 33] + [0] * 500, [1100] * 16385, [5], [0, 0, 0], [8192 * 32] * 3 + [1] * 8,
    ]
    queued = []
    for k, lengths in enumerate(shapes):
        count = len(lengths)
        off = np.zeros(count + 1, dtype=np.int64)
        off[1:] = np.cumsum(lengths)
        pre = int(rng.integers(0, 40))
        off += pre
        body = alpha[rng.integers(0, 8, size=int(off[-1]) - pre)]
        buf = np.concatenate([np.full(pre, ord("N"), np.uint8), body, np.full(19, ord("N"), np.uint8)])
        ew, ewo = _oracle_batch(oracle, buf, off)
        hold = torch.zeros(count + 2, dtype=torch.int64, device=dev)
        hold[1:] = torch.from_numpy(off).to(dev)
        d_off = hold[1:]  # 8 bytes into its allocation
        wo_hold = torch.zeros(count + 2, dtype=torch.int64, device=dev)
        d_wo = wo_hold[1:]
        d_seq = torch.from_numpy(buf).to(dev)
        torch.cuda.synchronize()
        total = ctx.batch_word_offsets_dev(d_off, count, d_wo)
        assert total == len(ew) and np.array_equal(d_wo.cpu().numpy(), ewo), k
        words = torch.full((total + 2,), -1, dtype=torch.int64, device=dev)
        back = torch.zeros(len(buf), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.encode_batch_dev(d_seq, d_off, d_wo, count, total, words)
        ctx.decode_batch_dev(words, d_wo, d_off, count, total, back)
        queued.append((k, pre, buf, ew, words, back, total))  # no sync between batches: the scratch plan is reused in stream order
    ctx.sync()
    for k, pre, buf, ew, words, back, total in queued:
        assert np.array_equal(words[:total].cpu().numpy().view(np.uint64), ew), k
        assert (words[total:] == -1).all(), k
        h = back.cpu().numpy()
        n = len(buf) - pre - 19
        assert bytes(h[pre:pre + n]) == bytes(buf[pre:pre + n]).upper() and not h[:pre].any() and not h[pre + n:].any(), k
    # the first invalid byte in buffer order, through the table-driven entry point
    lengths = list(rng.integers(1, 300, size=1500))
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    off[1:] = np.cumsum(lengths)
    buf = alpha[rng.integers(0, 4, size=int(off[-1]))].copy()
    p1, p2 = int(off[700]) + 3, int(off[900])
    buf[p1], buf[p2] = ord("N"), ord("X")
    d_off, d_seq = torch.from_numpy(off).to(dev), torch.from_numpy(buf).to(dev)
    d_wo = torch.zeros(len(off), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(d_off, len(lengths), d_wo)
    words = torch.zeros(total, dtype=torch.int64, device=dev)
    ctx.encode_batch_dev(d_seq, d_off, d_wo, len(lengths), total, words)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", ord("N"), p1)


def test_bytes_before_a_batch_never_reach_its_first_word(ctx, oracle):
    """The batch kernels load aligned 16-byte chunks; what precedes the batch's first base inside its first chunk is not the
    batch's (here: bytes that are not bases at all).  Found in round 3: such a byte in the same DWORD as the first bases used
    to spill into their codes through enc4's multiply-add.  Every lead 1..15 x {plan, tables, fixed-length back-to-back},
    device pointers, against the oracle loop."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(99)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, count = 75, 300
    body = alpha[rng.integers(0, 4, size=L * count)]
    exp = np.concatenate([oracle.encode(body[i * L:(i + 1) * L]) for i in range(count)])
    wpr = (L + 31) // 32
    for lead in range(0, 16):
        for junk in (ord("N"), 0xFF, ord("\n"), 0x00):
            buf = np.concatenate([np.full(lead, junk, np.uint8), body, np.full(7, junk, np.uint8)])
            hold = torch.zeros(len(buf) + 16, dtype=torch.uint8, device=dev)
            assert hold.data_ptr() % 16 == 0
            hold[:len(buf)] = torch.from_numpy(buf).to(dev)
            off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L + lead
            wo = torch.zeros(count + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            total = ctx.batch_word_offsets_dev(off, count, wo)
            assert total == count * wpr
            w_tab = torch.zeros(total, dtype=torch.int64, device=dev)
            w_plan = torch.zeros(total, dtype=torch.int64, device=dev)
            w_fix = torch.zeros(total, dtype=torch.int64, device=dev)
            plan = bn.BatchPlan(ctx, off, count)
            ctx.encode_batch_dev(hold, off, wo, count, total, w_tab)
            plan.encode_dev(hold, w_plan)
            ctx.encode_fixed_dev(hold.data_ptr() + lead, L, L, count, w_fix)
            ctx.sync()
            for name, w in (("tables", w_tab), ("plan", w_plan), ("fixed", w_fix)):
                assert np.array_equal(w.cpu().numpy().view(np.uint64), exp), (name, lead, junk)
            plan.close()


# ---- every window of a sequence: line-aligned rounds, windows computed where they are stored -------------------------------
@pytest.mark.parametrize("rounds_per_trip", [1, 2, 4])
def test_windows_line_aligned_rounds_vs_oracle(ctx, sweep_ctx, oracle, rounds_per_trip):
    """kmer_slide2_kernel (`for w in seq.windows(k) { as_2bit(w) }`, src/lib.rs:170-173): rounds of 1024 windows whose 30-base
    halo comes from the next round's registers or one extra load; sizes around the 1024 / 1056-byte round and trip
    boundaries, every k class (<= 16, 17..31, 32), first invalid byte incl. the halo positions."""
    import bitnuc_amd as bn
    ctx = ctx if rounds_per_trip == 4 else sweep_ctx  # the product ships 4 rounds per trip; 1 and 2 live in the evidence build
    prev = ctx.set_variant("slide2_rounds", rounds_per_trip)
    assert ctx.get("slide_impl") == 1 and ctx.get("slide2_rounds") == rounds_per_trip
    rng = np.random.default_rng(4242 + rounds_per_trip)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    try:
        for k in (1, 2, 15, 16, 17, 21, 31, 32):
            for n in (1055, 1056, 1057, 1056 + k - 1, 2047, 2048, 2079, 2080, 2081, 4 * 1024 + 31, 4 * 1024 + 32, 4 * 1024 + 33,
                      5 * 1024 + 40, 8 * 1024 + 32, 9 * 1024 + 500, 200003):
                if n < k:
                    continue
                s = alpha[rng.integers(0, 8, size=n)]
                count = n - k + 1
                assert np.array_equal(ctx.as_2bit_batch(s, k, 1, count), oracle.as_2bit_batch(s, k, 1, count)), (k, n)
        k, n = 31, 50000
        s = alpha[rng.integers(0, 4, size=n)].copy()
        for pos in (0, 15, 16, 1023, 1024, 1025, 1039, 1040, 1055, 1056, 4095, 4096, 4 * 1024 + 31, 20000, n - 1):
            t = s.copy()
            t[pos] = ord("N")
            if pos + 7 < n:
                t[pos + 7] = ord("X")  # a later invalid byte never wins
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.as_2bit_batch(t, k, 1, n - k + 1)
            assert (ei.value.byte, ei.value.index) == (ord("N"), pos), pos
        t = np.concatenate([s, np.frombuffer(b"N", dtype=np.uint8)])  # a byte past the last window is never examined
        assert np.array_equal(ctx.as_2bit_batch(t, k, 1, n - k + 1), oracle.as_2bit_batch(s, k, 1, n - k + 1))
    finally:
        ctx.set_variant("slide2_rounds", prev)


def test_windows_both_formulations_agree_at_scale(sweep_ctx, oracle):
    """10^8 bases, k = 31: the strip kernel of rounds 1-2 (rounds of 992 windows) and the line-aligned kernel write the same
    10^8 - 30 words; spot blocks against the oracle."""
    import torch
    ctx = sweep_ctx
    dev = torch.device("cuda:0")
    n, k = 10**8 + 13, 31
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, SEED)
    count = n - k + 1
    a = torch.zeros(count, dtype=torch.int64, device=dev)
    b = torch.zeros(count, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    prev = ctx.set_variant("slide_impl", 0)
    ctx.as_2bit_batch_dev(seq, k, 1, count, a)
    ctx.set_variant("slide_impl", 1)
    ctx.as_2bit_batch_dev(seq, k, 1, count, b)
    ctx.sync()
    ctx.set_variant("slide_impl", prev)
    assert torch.equal(a, b)
    h = seq.cpu().numpy()
    for start in (0, 1024 * 777 - 40, count - 5000):
        exp = oracle.as_2bit_batch(h[start:start + 5000 + k - 1], k, 1, 5000)
        assert np.array_equal(b[start:start + 5000].cpu().numpy().view(np.uint64), exp), start


def test_encode_quad_variants_vs_oracle(sweep_ctx, oracle):
    """encode variants 47..62 (evidence build): 16-byte stores by a register quad transpose (encode_quad_kernel) -- same
    words as the oracle at tailed sizes, first invalid byte with its index."""
    import bitnuc_amd as bn
    ctx = sweep_ctx
    enc0 = ctx.get("encode")
    rng = np.random.default_rng(5150)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    try:
        for v in range(47, 63):
            assert ctx.set_variant("encode", v) != -2
            for n in (1, 31, 4095, 4096, 4097, 8192 * 2 + 5, 16384 * 4 + 5, 1000003, (1 << 22) + 17):
                s = alpha[rng.integers(0, 8, size=n)]
                assert np.array_equal(ctx.encode_array(s), oracle.encode(s)), (v, n)
            s = alpha[rng.integers(0, 4, size=300000)].copy()
            s[123457] = ord("N")
            s[200000] = ord("X")
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.encode_array(s)
            assert (ei.value.byte, ei.value.index) == (ord("N"), 123457), v
    finally:
        ctx.set_variant("encode", enc0)
    assert ctx.set_variant("encode", 63) == -2


def test_bench_force_dist_prints_the_multi_gpu_blocks():
    """`bench.py --force-dist` on the one GPU every box has: torch.distributed + RCCL at world size 1, so the N>1 line's
    blocks are all exercised -- allgather_packed with its xGMI roofline entry (null + reason at one rank), the in-place
    chunked end-to-end form, and the C-ABI block (1-rank RCCL communicator: one-shot == chunked overlap); rc 0, one line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "5", "--warmup", "2", "--bases", str(10**8),
                        "--no-extras", "--no-traffic", "--cpu-sample", str(10**7), "--cpu-reps", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["rccl_ok"] is True
    ag = line["allgather_packed"]
    assert ag["own_slot_ok"] is True and ag["roofline"]["value"] is None and "one rank" in ag["roofline"]["reason"]
    e2e = line["encode_allgather_end_to_end"]
    assert e2e["one_shot_ok"] is True and e2e["overlap8_ok"] is True
    cab = line["c_abi_allgather"]
    assert cab.get("overlap_equals_one_shot") is True and cab["one_shot_ms"] > 0 and cab["overlap8_ms"] > 0, cab
    assert "cpu_baseline" in line and line["cpu_baseline"].get("value", 0) > 0


def test_hdist_words_coalesced_kernel_vs_oracle(ctx, sweep_ctx, oracle):
    """Many-pair / one-query hdist_scalar (hamming/scalar.rs:11-48): the coalesced-load kernel (whole 256-word wave tiles, the
    stored bytes gathered from neighbouring lanes) and the four-contiguous-words kernel give the oracle's distances for
    counts around the tile size, every len class, 16- and 8-byte aligned inputs."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(808)
    product = ctx
    for impl in (1, 0):
        ctx = product if impl == 1 else sweep_ctx  # the four-contiguous-words kernel lost its A/B: evidence build only
        prev = ctx.set_variant("hdist_words_impl", impl)
        try:
            for count in (1, 255, 256, 257, 511, 512, 1000, 256 * 37 + 3, 100003):
                for length in (0, 1, 16, 31, 32):
                    a = rng.integers(0, 1 << 63, size=count + 1, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=count + 1, dtype=np.uint64)
                    b = a ^ (rng.integers(0, 1 << 63, size=count + 1, dtype=np.uint64) & rng.integers(0, 1 << 63, size=count + 1, dtype=np.uint64))
                    for shift in (0, 1):  # 16-byte aligned tables, then 8 bytes into them
                        da = torch.from_numpy(a.view(np.int64)).to(dev)[shift:]
                        db = torch.from_numpy(b.view(np.int64)).to(dev)[shift:]
                        n = count + 1 - shift
                        out = torch.full((n + 8,), 0xEE, dtype=torch.uint8, device=dev)
                        torch.cuda.synchronize()
                        ctx.hdist_pairs_dev(da, db, n, length, out)
                        ctx.sync()
                        exp = oracle.hdist_pairs(a[shift:], b[shift:], length)
                        h = out.cpu().numpy()
                        assert np.array_equal(h[:n], exp) and (h[n:] == 0xEE).all(), (impl, count, length, shift)
                        q = int(b[0])
                        ctx.hdist_query_dev(q, da, n, length, out)
                        ctx.sync()
                        assert np.array_equal(out.cpu().numpy()[:n], oracle.hdist_pairs(a[shift:], np.full(n, q, dtype=np.uint64), length)), (impl, count, length, shift)
        finally:
            ctx.set_variant("hdist_words_impl", prev)


def test_every_async_entry_point_can_be_captured_and_replayed(oracle):
    """Launch-bound pipelines capture their inner loop once and replay it: after one warm-up call (scratch growth is an allocation)
    every asynchronous entry point -- bulk codec, table-driven and planned batches, fixed-length reads, dense k-mers, windows,
    scan, fused count, bulk hdist, base counts, one-query hdist, split -- is recorded into ONE hipGraph (8 error slots become
    persistent), replayed on new data three times and compared with direct calls; then a replay on an invalid byte is reported
    by the next sync with its byte and index, and the replay after it is clean."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    L, count = 150, 20011
    n = L * count
    k = 31
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c = bn.Context(0, stream=s.cuda_stream)
        ref_c = bn.Context(0, stream=s.cuda_stream)  # direct calls for comparison
        seq = torch.empty(n, dtype=torch.uint8, device=dev)
        off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
        wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
        c.nucgen_dev(seq, n, 1)
        torch.cuda.synchronize()
        total = c.batch_word_offsets_dev(off, count, wo)
        plan = bn.BatchPlan(c, off, count)
        nw = (n + 31) // 32
        nk = n // k

        def buffers():
            return dict(words=torch.zeros(nw, dtype=torch.int64, device=dev), back=torch.zeros(n, dtype=torch.uint8, device=dev),
                        bw=torch.zeros(total, dtype=torch.int64, device=dev), bback=torch.zeros(n, dtype=torch.uint8, device=dev),
                        pw=torch.zeros(total, dtype=torch.int64, device=dev), fw=torch.zeros(total, dtype=torch.int64, device=dev),
                        fback=torch.zeros(n, dtype=torch.uint8, device=dev), km=torch.zeros(nk, dtype=torch.int64, device=dev),
                        win=torch.zeros(n - k + 1, dtype=torch.int64, device=dev), dist=torch.zeros(n - k + 1, dtype=torch.uint8, device=dev),
                        cnt=torch.zeros(1, dtype=torch.int64, device=dev), hd=torch.zeros(1, dtype=torch.int32, device=dev),
                        bc=torch.zeros(4, dtype=torch.int64, device=dev), hq=torch.zeros(nw, dtype=torch.uint8, device=dev),
                        sl=torch.zeros(nw, dtype=torch.int64, device=dev), sr=torch.zeros(nw, dtype=torch.int64, device=dev))

        def step(cx, b, pl):
            cx.encode_dev(seq, n, b["words"])
            cx.decode_dev(b["words"], nw, n, b["back"])
            cx.encode_batch_dev(seq, off, wo, count, total, b["bw"])
            cx.decode_batch_dev(b["bw"], wo, off, count, total, b["bback"])
            pl.encode_dev(seq, b["pw"])
            cx.encode_fixed_dev(seq, L, L, count, b["fw"])
            cx.decode_fixed_dev(b["fw"], L, L, count, b["fback"])
            cx.as_2bit_batch_dev(seq, k, k, nk, b["km"])
            cx.as_2bit_batch_dev(seq, k, 1, n - k + 1, b["win"])
            cx.kmer_hdist_scan_dev(seq, n, k, 0x0123456789ABCDEF & ((1 << 62) - 1), b["dist"])
            cx.kmer_hdist_count_dev(seq, n, k, 0x0123456789ABCDEF & ((1 << 62) - 1), 20, b["cnt"])
            cx.hdist_dev(b["words"], nw, b["pw"], nw, min(n, 32 * min(nw, total)), b["hd"])
            cx.base_counts_dev(b["words"], nw, n, b["bc"])
            cx.hdist_query_dev(0x1111222233334444, b["words"], nw, 32, b["hq"])
            cx.split_packed_dev(b["words"], nw, n, n // 2 + 5, b["sl"], b["sr"], canonical=True)

        got, exp = buffers(), buffers()
        ref_plan = bn.BatchPlan(ref_c, off, count)
        step(c, got, plan)  # warm-up: scratch of the table-driven path grows here, outside the capture
        c.sync()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step(c, got, plan)
        assert c.get("captured_slots") == 8  # encode, tables encode, plan encode, fixed encode, dense k-mers, windows (their tail kernels share the call's slot), scan, count
        for seed in (2, 3, 4):
            c.nucgen_dev(seq, n, seed)
            for t in got.values():
                t.zero_()
            g.replay()
            c.sync()
            step(ref_c, exp, ref_plan)
            ref_c.sync()
            for name in got:
                assert torch.equal(got[name], exp[name]), (seed, name)
            assert torch.equal(got["back"], seq) and torch.equal(got["bback"], seq) and torch.equal(got["fback"], seq)
            assert np.array_equal(got["words"][:1000].cpu().numpy().view(np.uint64), oracle.encode(seq[:32000].cpu().numpy()))
        seq[n - 77] = ord("N")
        g.replay()
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        assert (ei.value.byte, ei.value.index) == (ord("N"), n - 77)
        seq[n - 77] = ord("C")
        g.replay()
        c.sync()
        plan.close()
        ref_plan.close()
        ref_c.close()
        c.close()
