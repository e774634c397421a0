"""CPU tests of the N>1 path: shard arithmetic and the packed all-gather, world_size 2
over gloo.  The per-rank encoder is injected; here it is the oracle (tests may use it),
on the GPU it is Context.encode_dev."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from bitnuc_amd.dist import shard_range, shard_word_counts
    for n in [0, 1, 31, 32, 33, 1000, 10**9, 10**9 + 17, 8 * 10**9]:
        for world in [1, 2, 3, 4, 8]:
            prev = 0
            for r in range(world):
                a, b = shard_range(n, r, world)
                assert a == prev and a % 32 == 0 or a == n
                assert b >= a
                prev = b
            assert prev == n
            assert sum(shard_word_counts(n, world)) == (n + 31) // 32
    # BASELINE config 4: 8 x 10^9 bases -> 8 equal shards of 31 250 000 words
    assert shard_word_counts(8 * 10**9, 8) == [31_250_000] * 8


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_py
        from bitnuc_amd.dist import encode_sharded, shard_range
        a, b = shard_range(n, rank, world)
        shard = oracle_py.nucgen(b - a, 0xB17C0DE, first=a)  # every rank regenerates its own slice

        def enc(s):
            return torch.from_numpy(oracle_py.encode(s).view(np.int64).copy()) if len(s) else torch.zeros(0, dtype=torch.int64)
        full = encode_sharded(enc, shard, n)
        expect = oracle_py.encode(oracle_py.nucgen(n, 0xB17C0DE))
        q.put((rank, bool(np.array_equal(full.numpy().view(np.uint64), expect))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [64 * 1000, 64 * 1000 + 17, 33])
def test_two_rank_sharded_encode_equals_single(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


def _worker_overlap(rank, world, port, shard_words, n_chunks, in_place, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_py
        from bitnuc_amd.dist import allgather_packed, encode_allgather_overlapped
        n = 32 * shard_words
        shard = oracle_py.nucgen(n, 0xB17C0DE, first=rank * n)
        words = torch.zeros(shard_words, dtype=torch.int64)
        slots = []

        def enc2(w0, w1):  # legacy form: returns the words (one copy per piece into the output)
            words[w0:w1] = torch.from_numpy(oracle_py.encode(shard[32 * w0:32 * w1]).view(np.int64).copy())
            return words[w0:w1]

        def enc3(w0, w1, dst):  # in-place form: the per-rank encoder writes straight into its slot of the output
            dst.copy_(torch.from_numpy(oracle_py.encode(shard[32 * w0:32 * w1]).view(np.int64).copy()))
            words[w0:w1] = dst
            slots.append(dst.data_ptr())
        out = torch.full((world * shard_words,), -1, dtype=torch.int64) if in_place else None
        full = encode_allgather_overlapped(enc3 if in_place else enc2, shard_words, n_chunks, words, out=out)
        expect = oracle_py.encode(oracle_py.nucgen(world * n, 0xB17C0DE))
        ok = bool(np.array_equal(full.numpy().view(np.uint64), expect)) and torch.equal(full, allgather_packed(words))
        if in_place:  # the result IS the caller's buffer and every piece was encoded inside this rank's slot of it
            lo = out.data_ptr() + 8 * rank * shard_words
            ok = ok and full.data_ptr() == out.data_ptr() and all(lo <= p < lo + 8 * shard_words for p in slots)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shard_words,n_chunks,in_place", [(2, 1000, 8, True), (2, 7, 8, False), (2, 64, 1, True), (2, 1001, 3, False),
                                                                 (3, 1001, 5, True), (4, 500, 8, True)])
def test_overlapped_allgather_in_place_equals_single(world, shard_words, n_chunks, in_place):
    """The chunked, in-place exchange (batched point-to-point: what RCCL runs as grouped ncclSend / ncclRecv) over gloo with 2, 3
    and 4 ranks == the plain all-gather == the oracle's encode of the whole sequence."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + shard_words + 7 * world) % 2000
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, shard_words, n_chunks, in_place, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {r: True for r in range(world)}


def test_scan_shards_with_halo_concatenate_to_the_single_result():
    """the sliding scan shards like the codec, each shard reading a (k-1)-base halo; no exchange.  Every rank's
    windows (computed here by the oracle on exactly the bytes scan_shard_range hands it) concatenate to the unsharded scan."""
    import oracle_py
    from bitnuc_amd.dist import scan_shard_range
    for n, k in [(1000, 31), (31, 31), (30, 31), (100003, 31), (4096, 1), (5000, 32), (64, 17)]:
        seq = oracle_py.nucgen(n, 0xB17C0DE)
        q = oracle_py.as_2bit(seq[:k]) if n >= k else 0
        whole = oracle_py.kmer_hdist_scan(seq, k, q) if n >= k else np.zeros(0, np.uint8)
        for world in (1, 2, 3, 8):
            parts, covered = [], 0
            for r in range(world):
                first, count, nread = scan_shard_range(n, k, r, world)
                assert first == covered and (first % 32 == 0 or count == 0)
                covered += count
                if count:
                    assert first + nread <= n
                    parts.append(oracle_py.kmer_hdist_scan(seq[first:first + nread], k, q))
                    assert len(parts[-1]) == count
            assert covered == len(whole)
            got = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
            assert np.array_equal(got, whole), (n, k, world)
