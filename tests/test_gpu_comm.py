"""GPU tests of the multi-GPU layer (bitnuc_amd/csrc/comm.hip, BASELINE config 4): the chunked in-place gather on a one-rank
communicator against the one-shot form, the >= 2-GPU cases (skipped where fewer devices are visible: single-process init_all, one process
per GPU through init_rank, both exchange modes), the link probe's argument rule, the Python mirror of the _all forms, and bench.py's N > 1
path on one shared GPU.  The shard rule: packing/avx.rs:138-145 (no carry between words).  P ranks on one GPU against a mock RCCL:
tests/test_gpu_multirank_mock.py.  (Filed by component in round 5; tests from test_gpu_round2/3/4.py unchanged.)"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xB17C0DE


# ---- config 4 through the C ABI ---------------------------------------------------------------------------------------
def _gpus():
    import torch
    return torch.cuda.device_count()


WORLDS = [2, 4, 8]


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


def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: fresh rank processes, one JSON line, rc 0; the N>1 line carries
    config 4's side measurements (here over gloo, both ranks sharing the one GPU: a rehearsal of the control flow)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                        "--steps", "5", "--warmup", "2", "--bases", str(10**8)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["rccl_ok"] is None
    assert line["allgather_packed"]["own_slot_ok"] is True and "encode_allgather_end_to_end" in line
    # round 5: the ragged batch split by whole sequences -- each rank plan-encodes its run on the GPU, unequal word counts gathered (over gloo: on host copies),
    # every word compared with one rank's encode of the WHOLE batch
    rb = line["ragged_batch_sharded"]
    assert rb["all_slots_ok"] is True and sum(rb["words_per_rank"]) == rb["total_words"] and rb["gather_ms"] > 0, rb
    # ... and every slot of the gathered buffer against the closed form of the seeded stream (2^20 words of each rank's shard here)
    assert line["allgather_packed"]["all_slots_ok"] is True and line["allgather_packed"]["first_bad_slot"] is None
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                          "--bases", str(10**7), "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=dict(env, BITNUC_BENCH_PLANT_BAD_SLOT="1"))
    assert bad.returncode != 0, (bad.returncode, bad.stderr[-2000:])  # (which rank the launcher names as the root cause depends on who is torn down first: the line is the evidence)
    bl = json.loads([l for l in bad.stdout.splitlines() if l.strip()][0])
    assert bl["allgather_packed"]["all_slots_ok"] is False and bl["allgather_packed"]["first_bad_slot"] == 1  # a wrong word in the PEER's slot is caught
    # round 3: the fabric roofline entry (one shared GPU: null + the reason), the C-ABI block's skip reason, the CPU baseline on an N > 1 line
    assert line["allgather_packed"]["roofline"]["value"] is None and "device" in line["allgather_packed"]["roofline"]["reason"]
    assert "skipped" in line["c_abi_allgather"] and line["cpu_baseline"]["value"] > 0
    assert line["roofline"]["kernel"] in ("encode_kernel", "decode_kernel") and "roofline_step" in line


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


@pytest.mark.parametrize("world", [1] + WORLDS)  # world 1: the same code on the one GPU every box has (the gather is then a no-op)
def test_multi_gpu_ragged_batch_by_whole_sequences_allgatherv(oracle, world):
    """north_star's split on real GPUs (skipped below `world` devices): a ragged batch -- reads of 0..300 bases, empty ones, one sequence longer
    than a fair share -- partitioned by whole sequences (bitnuc_batch_shard_ranges), every rank's run plan-encoded on ITS device straight into
    its slot, the unequal word counts gathered in place by bitnuc_allgatherv_words_all (one thread holds all ranks); every rank's buffer ==
    one GPU's plan encode of the whole batch == the oracle's per-sequence loop (src/utils/mod.rs:22-25; padding rule packing/avx.rs:147-148)."""
    if _gpus() < world:
        pytest.skip(f"needs >= {world} GPUs")
    import torch
    import bitnuc_amd as bn
    rng = np.random.default_rng(1234 + world)
    count = 20000 * world
    lens = rng.integers(0, 301, size=count).astype(np.uint64)
    lens[::97] = 0
    lens[count // 3] = 3_000_001
    off = np.zeros(count + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    seq = oracle.nucgen(int(off[-1]), SEED + 5, flags=2)
    seq_first, word_first = bn.batch_shard_ranges(off, world)
    total = int(word_first[-1])
    counts = [int(word_first[r + 1] - word_first[r]) for r in range(world)]
    # (a rank may be left without a sequence: allowed, it sends nothing)
    g = bn.CommGroup(world)
    alls, keep = [], []
    for r in range(world):
        d = torch.device("cuda", r)
        s0, s1 = int(seq_first[r]), int(seq_first[r + 1])
        local = (off[s0:s1 + 1] - off[s0]).astype(np.int64)
        mine = torch.from_numpy(seq[int(off[s0]):int(off[s1])].copy()).to(d)
        out = torch.full((total + 8,), -1, dtype=torch.int64, device=d)
        c = g.context(r)
        plan = bn.BatchPlan(c, torch.from_numpy(local).to(d), s1 - s0)
        assert plan.total_words == counts[r]
        torch.cuda.synchronize(r)
        if counts[r]:
            plan.encode_dev(mine, out[int(word_first[r]):])
        alls.append(out)
        keep.append((plan, mine))
    g.allgatherv_words(counts, alls)  # synchronises every stream
    c0 = bn.Context(0)
    plan_all = bn.BatchPlan(c0, torch.from_numpy(off.astype(np.int64)).to("cuda:0"), count)
    ref = torch.empty(total, dtype=torch.int64, device="cuda:0")
    plan_all.encode_dev(torch.from_numpy(seq).to("cuda:0"), ref)
    c0.sync()
    for r in range(world):
        got = alls[r].to("cuda:0")
        assert torch.equal(got[:total], ref) and bool((got[total:] == -1).all()), r
    # ... and the oracle's per-sequence loop on a prefix of the batch
    h = ref.cpu().numpy().view(np.uint64)
    W = np.concatenate([[0], np.cumsum((lens + 31) // 32)]).astype(np.int64)
    for i in list(range(0, 200)) + [count // 3, count - 1]:
        s = seq[int(off[i]):int(off[i + 1])]
        assert np.array_equal(h[W[i]:W[i + 1]], oracle.encode(s) if len(s) else np.zeros(0, np.uint64)), i
    for plan, _ in keep:
        plan.close()
    plan_all.close()
    c0.close()
    g.close()


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


def test_comm_group_python_mirror(oracle):
    """bn.CommGroup (bitnuc_comm_init_all_devices + the _all entry points) on the one GPU every box has: a one-rank group through the
    real RCCL, one-shot and chunked, equals the oracle; an invalid byte comes back with its rank; wider groups are exercised against
    the stand-in RCCL (tests/test_gpu_multirank_mock.py) and, where two or more GPUs exist, in tests/test_gpu_round3.py."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    g = bn.CommGroup(1, devices=[0])
    n = 32 * 70_001
    seq = torch.from_numpy(oracle.nucgen(n, 3)).to(dev)
    expect = oracle.encode(seq.cpu().numpy())
    for chunks in (0, 1, 6):
        out = torch.zeros(n // 32, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        g.encode_sharded_allgather([seq], n, [out], n_chunks=chunks)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), expect), chunks
    seq[12345] = ord("N")
    torch.cuda.synchronize()
    with pytest.raises(bn.NucleotideError) as ei:
        g.encode_sharded_allgather([seq], n, [out], n_chunks=4)
    assert (ei.value.kind, ei.value.byte, ei.value.index, ei.value.rank) == ("InvalidBase", ord("N"), 12345, 0)
    with pytest.raises(bn.NucleotideError) as ei:
        g.encode_sharded_allgather([seq], n - 1, [out])
    assert ei.value.kind == "InvalidLength"
    g.close()
    with pytest.raises(bn.BackendError):
        bn.CommGroup(2, devices=[0, 99])  # no such device: nothing leaks, a backend error comes back
