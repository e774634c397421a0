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


# ---- a ragged batch sharded by WHOLE sequences (SURVEY 8e sentence 2; north_star: "batches of independent sequences shard trivially") ----
RAGGED_CASES = {
    "reads": lambda rng: rng.integers(1, 400, size=300),
    "empties": lambda rng: np.array([0, 0, 5, 0, 64, 0, 0, 33, 32, 0, 1, 0, 0], dtype=np.int64),
    "one_long": lambda rng: np.concatenate([rng.integers(1, 50, size=5), [40000], rng.integers(1, 50, size=6)]),  # longer than a fair share of 3 ranks
    "long_first": lambda rng: np.array([9000, 3, 3, 3], dtype=np.int64),                                          # ranks 1.. of 3 start after it: one is left with no sequence
    "single": lambda rng: np.array([77], dtype=np.int64),
    "nothing": lambda rng: np.zeros(0, dtype=np.int64),
    "all_empty": lambda rng: np.zeros(6, dtype=np.int64),
}


def _ragged_offsets(name):
    rng = np.random.default_rng(sum(map(ord, name)))
    lens = np.asarray(RAGGED_CASES[name](rng), dtype=np.uint64)
    off = np.zeros(lens.size + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    return off


def test_batch_shard_ranges_partition_by_whole_sequences():
    """dist.batch_shard_ranges: contiguous runs of whole sequences that cover the batch, word_first = the word prefix at each run's first
    sequence, every run but a covered one within one sequence's words of the fair share; bitnuc_batch_shard_ranges (C ABI, host
    arithmetic: runs without a GPU) gives the same arrays; decreasing offsets are INVALID_RANGE."""
    import bitnuc_amd as bn
    from bitnuc_amd.dist import batch_shard_ranges, batch_word_prefix, batch_shard
    for name in RAGGED_CASES:
        off = _ragged_offsets(name)
        W = batch_word_prefix(off)
        count, total = off.size - 1, int(W[-1])
        assert np.array_equal(W[1:] - W[:-1], (off[1:] - off[:-1] + np.uint64(31)) // np.uint64(32))  # every sequence pads its own last word
        for world in (1, 2, 3, 4, 8):
            seq_first, word_first = batch_shard_ranges(off, world)
            assert seq_first[0] == 0 and seq_first[-1] == count and bool((seq_first[1:] >= seq_first[:-1]).all())
            assert np.array_equal(word_first, W[seq_first.astype(np.int64)]) and word_first[-1] == total
            longest = int((W[1:] - W[:-1]).max()) if count else 0
            for r in range(world):
                got = int(word_first[r + 1] - word_first[r])
                assert got <= total // world + 1 + longest, (name, world, r)  # balanced up to one sequence
                s0, s1, b0, b1, local, w0, nw = batch_shard(off, r, world)
                assert (s0, s1, w0, nw) == (int(seq_first[r]), int(seq_first[r + 1]), int(word_first[r]), got) and local[0] == 0 and int(local[-1]) == b1 - b0
            c_seq, c_word = bn.batch_shard_ranges(off, world)
            assert np.array_equal(c_seq, seq_first) and np.array_equal(c_word, word_first), (name, world)
    with pytest.raises(bn.NucleotideError):
        bn.batch_shard_ranges(np.array([0, 10, 5], dtype=np.uint64), 2)


def _worker_ragged(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_py
        from bitnuc_amd.dist import encode_batch_sharded, batch_word_prefix
        off = _ragged_offsets(name)
        total_bases = int(off[-1])
        touched = []

        def enc(b0, b1, local, dst):  # the reference's caller loop (src/utils/mod.rs:22-25) over THIS rank's sequences only
            seq = oracle_py.nucgen(b1 - b0, 0xB17C0DE, first=b0)  # the rank regenerates exactly its own bytes
            touched.append((b0, b1))
            wo = [0]
            for i in range(len(local) - 1):
                s = seq[int(local[i]):int(local[i + 1])]
                w = oracle_py.encode(s) if len(s) else np.zeros(0, np.uint64)
                dst[wo[-1]:wo[-1] + len(w)].copy_(torch.from_numpy(w.view(np.int64).copy()))
                wo.append(wo[-1] + len(w))
            assert wo[-1] == dst.numel()
            return np.array(wo, dtype=np.uint64)
        words, table = encode_batch_sharded(enc, off, torch.zeros(0, dtype=torch.int64))
        # expectation: the same loop over the WHOLE batch on one rank
        whole = oracle_py.nucgen(total_bases, 0xB17C0DE)
        parts = [oracle_py.encode(whole[int(off[i]):int(off[i + 1])]) for i in range(off.size - 1) if off[i + 1] > off[i]]
        expect = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
        ok = bool(np.array_equal(words.numpy().view(np.uint64), expect)) and bool(np.array_equal(table.numpy().view(np.uint64), batch_word_prefix(off)))
        q.put((rank, ok, touched[0][1] - touched[0][0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "reads"), (3, "reads"), (2, "empties"), (3, "empties"), (3, "one_long"), (3, "long_first"), (2, "single"), (2, "nothing"), (3, "all_empty")])
def test_ragged_batch_sharded_by_whole_sequences_equals_single(world, name):
    """gloo world 2 / 3: every rank encodes its run of WHOLE sequences (the oracle's per-sequence loop as the injected encoder) into its
    slot, unequal word counts are gathered in place, the global word_offsets table is rank prefix + local tables -- and both equal one
    per-sequence loop over the whole batch.  Cases: empty sequences, a rank left with no sequence, one sequence longer than a fair share."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + 13 * world + sum(map(ord, name))) % 2000
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [q.get(timeout=10) for _ in range(world)]
    assert all(ok for _, ok, _ in res), res
    assert sum(nb for _, _, nb in res) == int(_ragged_offsets(name)[-1])  # the ranks' byte ranges partition the batch: nobody read a peer's bases
    if name == "long_first" and world == 3:
        assert sorted(nb for _, _, nb in res)[0] == 0  # a rank was left with no sequence
