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


def _worker_overlap(rank, world, port, shard_words, n_chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_py
        from bitnuc_amd.dist import allgather_packed, encode_allgather_overlapped
        n = 32 * shard_words
        shard = oracle_py.nucgen(n, 0xB17C0DE, first=rank * n)
        words = torch.zeros(shard_words, dtype=torch.int64)

        def enc(w0, w1):  # the per-rank encoder is injected: the oracle here, Context.encode_dev on a GPU
            words[w0:w1] = torch.from_numpy(oracle_py.encode(shard[32 * w0:32 * w1]).view(np.int64).copy())
            return words[w0:w1]
        full = encode_allgather_overlapped(enc, shard_words, n_chunks, words)
        expect = oracle_py.encode(oracle_py.nucgen(world * n, 0xB17C0DE))
        same_as_plain = torch.equal(full, allgather_packed(words))
        q.put((rank, bool(np.array_equal(full.numpy().view(np.uint64), expect)) and same_as_plain))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shard_words,n_chunks", [(1000, 8), (7, 8), (64, 1), (1001, 3)])
def test_two_rank_overlapped_allgather_equals_single(shard_words, n_chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + shard_words) % 2000
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, shard_words, n_chunks, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}
