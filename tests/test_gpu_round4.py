"""GPU tests added in round 4 (all through the C ABI):

* context scratch that a recorded hipGraph holds: a table-driven ragged batch captured after warm-up, then a LARGER ordinary batch on
  the same context (the scratch plan grows), then the replay -- the replay still writes through the old buffer, which therefore
  must stay alive; growth inside a capture is refused cleanly and the capture stays usable;
* the order of data errors across an implicit drain: the deferred error is reported first, the later one by the following sync;
* bitnuc_decode (host pointers, pipelined) keeps an InvalidBase latched by earlier asynchronous launches for the next sync;
* every output element of BASELINE configs 3 and 5 at full size (10^8 dense 31-mers, 10^9 - 30 windows) against an independent
  closed form / the oracle run over the whole input on the host cores.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xB17C0DE


def _reads(oracle, count, L, seed):
    import torch
    dev = torch.device("cuda:0")
    seq = torch.from_numpy(oracle.nucgen(count * L, seed)).to(dev)
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    return seq, off


def _oracle_batch(oracle, seq, count, L):
    h = seq.cpu().numpy()
    return np.concatenate([oracle.encode(h[i * L:(i + 1) * L]) for i in range(count)])


def test_scratch_held_by_a_graph_outlives_a_larger_batch(oracle):
    """plan_emit_kernel and the plan kernels of the table-driven batch calls take their layout plan from context
    scratch; a graph recorded after warm-up holds those addresses.  A later, larger table-driven call must not free them."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    L, small, large = 150, 4001, 60013
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c = bn.Context(0, stream=s.cuda_stream)
        seq_s, off_s = _reads(oracle, small, L, 11)
        seq_l, off_l = _reads(oracle, large, L, 12)
        wo_s = torch.empty(small + 1, dtype=torch.int64, device=dev)
        wo_l = torch.empty(large + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        tot_s = c.batch_word_offsets_dev(off_s, small, wo_s)
        tot_l = c.batch_word_offsets_dev(off_l, large, wo_l)
        out_s = torch.zeros(tot_s, dtype=torch.int64, device=dev)
        back_s = torch.zeros(small * L, dtype=torch.uint8, device=dev)
        out_l = torch.zeros(tot_l, dtype=torch.int64, device=dev)
        back_l = torch.zeros(large * L, dtype=torch.uint8, device=dev)
        c.encode_batch_dev(seq_s, off_s, wo_s, small, tot_s, out_s)  # warm-up: scratch sized for the small batch
        c.decode_batch_dev(out_s, wo_s, off_s, small, tot_s, back_s)
        c.sync()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            c.encode_batch_dev(seq_s, off_s, wo_s, small, tot_s, out_s)
            c.decode_batch_dev(out_s, wo_s, off_s, small, tot_s, back_s)
            # a batch that needs MORE scratch than the context holds cannot be recorded: refused before anything is touched ...
            with pytest.raises(bn.NucleotideError) as ei:
                c.encode_batch_dev(seq_l, off_l, wo_l, large, tot_l, out_l)
            assert ei.value.kind == "Unsupported"
        # ... and the capture survived the refusal
        expect_s = _oracle_batch(oracle, seq_s, small, L)
        out_s.zero_(); back_s.zero_()
        g.replay()
        c.sync()
        assert np.array_equal(out_s.cpu().numpy().view(np.uint64), expect_s) and torch.equal(back_s, seq_s)
        # the larger ordinary batch: the scratch plan grows.  Fill memory churn in between so that a freed buffer would be reused.
        c.encode_batch_dev(seq_l, off_l, wo_l, large, tot_l, out_l)
        c.decode_batch_dev(out_l, wo_l, off_l, large, tot_l, back_l)
        c.sync()
        assert np.array_equal(out_l.cpu().numpy().view(np.uint64), _oracle_batch(oracle, seq_l, large, L)) and torch.equal(back_l, seq_l)
        churn = [torch.full((1 << 20,), 0x5A, dtype=torch.uint8, device=dev) for _ in range(64)]
        # replay on NEW data of the recorded shape: it runs plan_emit_kernel into the buffer the graph holds
        seq_s.copy_(torch.from_numpy(oracle.nucgen(small * L, 13)).to(dev))
        expect_s = _oracle_batch(oracle, seq_s, small, L)
        for _ in range(3):
            out_s.zero_(); back_s.zero_()
            g.replay()
            c.encode_batch_dev(seq_l, off_l, wo_l, large, tot_l, out_l)  # ordinary calls interleaved: they use the new buffer
            c.sync()
            assert np.array_equal(out_s.cpu().numpy().view(np.uint64), expect_s) and torch.equal(back_s, seq_s)
        assert all(bool((t == 0x5A).all()) for t in churn), "a replay wrote into memory that had been given back"
        c.close()


def test_two_data_errors_across_an_implicit_drain_are_both_reported(oracle):
    """An InvalidBase latched by an asynchronous launch, then a host-pointer call (which starts from an empty ring and so finds it,
    defers it), then another asynchronous launch with its own invalid byte: the first sync reports the first error, the second sync
    the second one (the later one used to be dropped)."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    c = bn.Context(0)
    n = 3_000_000
    a = torch.from_numpy(oracle.nucgen(n, 1)).to(dev)
    b = a.clone()
    a[1234] = ord("N")
    b[2_999_999] = ord("x")
    words = torch.empty((n + 31) // 32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    c.encode_dev(a, n, words)
    host = oracle.nucgen(2_500_000, 3)
    w = c.encode_array(host)  # host-pointer call: its own result is clean, the latched error is kept
    assert np.array_equal(w, oracle.encode(host))
    back = c.decode_array(w, host.size)  # decode as well (above the host cutoff: the GPU path)
    assert np.array_equal(back, host)
    c.encode_dev(b, n, words)
    with pytest.raises(bn.NucleotideError) as e1:
        c.sync()
    assert (e1.value.byte, e1.value.index) == (ord("N"), 1234)
    with pytest.raises(bn.NucleotideError) as e2:
        c.sync()
    assert (e2.value.byte, e2.value.index) == (ord("x"), 2_999_999)
    c.sync()
    c.close()


def test_pipelined_host_decode_keeps_a_pending_async_error(oracle):
    """bitnuc_decode above the pipeline threshold (>= 8 Mi bases) after an asynchronous encode that latched an InvalidBase: the
    decode's result is its own, the error surfaces at the next sync."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    c = bn.Context(0)
    n = 1_000_000
    a = torch.from_numpy(oracle.nucgen(n, 1)).to(dev)
    a[99] = 0
    words = torch.empty((n + 31) // 32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    c.encode_dev(a, n, words)
    m = 20_000_003
    host = oracle.nucgen(m, 7)
    packed = oracle.encode(host)
    assert np.array_equal(c.decode_array(packed, m), host)
    with pytest.raises(bn.NucleotideError) as ei:
        c.sync()
    assert (ei.value.byte, ei.value.index) == (0, 99)
    c.close()


# ---- BASELINE configs 3 and 5, every output element ---------------------------------------------------------------------------
def _lsr(x, s):  # logical shift right on int64 tensors (s: int or tensor, 1 <= s <= 63)
    return (x >> s) & ~(torch_min_i64() >> (s - 1))


def torch_min_i64():
    import torch
    return torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda:0")


def _generator_words(first, count, seed):
    """Words first .. first+count of the seeded stream's 2-bit encoding: base i of the stream IS field i % 32 of
    splitmix64(seed + (i / 32 + 1) * 0x9E3779B97F4A7C15) (include/bitnuc_hip.h, bitnuc_nucgen_dev), so the stream's packed form is
    that word sequence -- computed here with torch integer arithmetic, independently of every kernel of the library."""
    import torch
    idx = torch.arange(first + 1, first + count + 1, dtype=torch.int64, device="cuda:0")
    z = idx * (-7046029254386353131) + seed        # 0x9E3779B97F4A7C15 as i64, wraps mod 2^64
    z = (z ^ _lsr(z, 30)) * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    z = (z ^ _lsr(z, 27)) * (-7723592293110705685)  # 0x94D049BB133111EB
    return z ^ _lsr(z, 31)


def test_config3_every_dense_31mer_against_the_closed_form(ctx, oracle):
    """BASELINE config 3 at full size: 10^8 back-to-back 31-mers (naive.rs:3-20 per k-mer).  The input is the seeded stream, whose
    2-bit encoding is the generator's own word sequence; k-mer j is bits [62 j, 62 j + 62) of that bit stream.  ALL 10^8 output
    words are compared with that closed form (three blocks of 5 000 were compared before), in chunks of 2^24."""
    import torch
    dev = torch.device("cuda:0")
    count, k = 10**8, 31
    seq = torch.empty(count * k, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, count * k, SEED)
    out = torch.empty(count, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.as_2bit_batch_dev(seq, k, k, count, out)
    ctx.sync()
    # pin the closed form itself to the oracle on one block (generator words == oracle encode of the oracle's stream)
    assert np.array_equal(_generator_words(5_000_000, 4096, SEED).cpu().numpy().view(np.uint64), oracle.encode(oracle.nucgen(32 * 4096, SEED, 32 * 5_000_000)))
    CH = 1 << 24
    compared = 0
    for j0 in range(0, count, CH):
        m = min(CH, count - j0)
        w_first = (62 * j0) >> 6
        w_count = ((62 * (j0 + m) + 63) >> 6) - w_first + 1  # one word beyond: the last k-mer's high part may be empty
        W = _generator_words(w_first, w_count, SEED)
        bit = torch.arange(j0, j0 + m, dtype=torch.int64, device=dev) * 62
        wi = (bit >> 6) - w_first
        sh = bit & 63
        lo = torch.where(sh == 0, W[wi], _lsr(W[wi], torch.clamp(sh, min=1)))
        hi = torch.where(sh <= 2, torch.zeros_like(lo), W[wi + 1] << ((64 - sh) & 63))
        expect = (lo | hi) & ((1 << 62) - 1)
        if not torch.equal(out[j0:j0 + m], expect):
            bad = int((out[j0:j0 + m] != expect).nonzero()[0]) + j0
            h = seq[bad * k:(bad + 1) * k].cpu().numpy()
            raise AssertionError(f"k-mer {bad}: kernel {int(out[bad]) & (2**64 - 1):#x}, closed form {int(expect[bad - j0]) & (2**64 - 1):#x}, oracle {oracle.as_2bit(h):#x}")
        compared += m
    assert compared == count
    # and the closed form agrees with the oracle's per-k-mer loop where the old test looked
    for j0 in (0, 12_345_678, count - 5000):
        h = seq[j0 * k:(j0 + 5000) * k].cpu().numpy()
        assert np.array_equal(out[j0:j0 + 5000].cpu().numpy().view(np.uint64), oracle.as_2bit_batch(h, k, k, 5000))


def test_config5_every_window_of_the_scan_against_the_oracle(ctx, oracle):
    """BASELINE config 5 at full size: 10^9 bases, k = 31, one query: ALL 10^9 - 30 distances against the oracle's loop
    (naive.rs:3-20 per window, then hamming/scalar.rs:11-48) run over the whole input on the host cores -- 32-window-aligned slices
    from bitnuc_amd.dist.scan_shard_range, one thread each (the oracle is the checker here, nothing of it is timed or shipped).
    Compared by a 64-bit sum per 1 MiB block, with a full compare of a block that differs."""
    import ctypes as C
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from bitnuc_amd.dist import scan_shard_range
    dev = torch.device("cuda:0")
    n, k = 10**9, 31
    nwin = n - k + 1
    ref = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(ref, n, SEED)
    ctx.sync()
    qpos = 777_777_777
    q = oracle.as_2bit(ref[qpos:qpos + k].cpu().numpy())
    dist = torch.empty(nwin, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.kmer_hdist_scan_dev(ref, n, k, q, dist)
    ctx.sync()
    assert int(dist[qpos]) == 0 and int(dist.max()) <= k
    h_ref = ref.cpu().numpy()
    assert np.array_equal(h_ref[:1 << 20], oracle.nucgen(1 << 20, SEED))  # the input is the oracle's stream too
    h_got = dist.cpu().numpy()
    del ref, dist
    h_exp = np.zeros(nwin, dtype=np.uint8)
    h_exp[::4096] = 1  # touch the pages before the threads do
    lib = oracle.lib()
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    parts = 8 * threads

    def run(r):
        first, cnt, nread = scan_shard_range(n, k, r, parts)
        if cnt == 0:
            return 0
        e = oracle.OrcErr()
        st = lib.orc_kmer_hdist_scan(C.c_void_p(h_ref.ctypes.data + first), nread, k, C.c_uint64(q), C.c_void_p(h_exp.ctypes.data + first), C.byref(e))
        assert st == 0, (r, st)
        return cnt
    with ThreadPoolExecutor(threads) as ex:
        done = sum(ex.map(run, range(parts)))
    assert done == nwin
    # 64-bit sum per 1 MiB block (as u64 lanes of 8 distances: carries cannot cancel a difference within a lane pair by accident the
    # way a byte sum could), full compare where a block differs
    BLK = 1 << 20
    whole = nwin // BLK * BLK
    sums_got = h_got[:whole].view(np.uint64).reshape(-1, BLK // 8).sum(axis=1, dtype=np.uint64)
    sums_exp = h_exp[:whole].view(np.uint64).reshape(-1, BLK // 8).sum(axis=1, dtype=np.uint64)
    badblocks = np.nonzero(sums_got != sums_exp)[0]
    for b in badblocks[:1]:
        i = int(np.nonzero(h_got[b * BLK:(b + 1) * BLK] != h_exp[b * BLK:(b + 1) * BLK])[0][0]) + int(b) * BLK
        raise AssertionError(f"window {i}: kernel {h_got[i]}, oracle {h_exp[i]} ({len(badblocks)} of {whole // BLK} blocks differ)")
    assert np.array_equal(h_got[whole:], h_exp[whole:])
    assert np.array_equal(h_got, h_exp)  # a 1 GB memcmp is cheap: the block sums above only localise a failure


@pytest.mark.parametrize("impl", [2, 3, 5], ids=["chunks12", "chunks20", "chunks32"])
def test_scan3_first_invalid_byte_and_later_bytes(sweep_ctx, oracle, impl):
    """kmer_scan3_kernel (a wave owns 12 / 20 / 32 consecutive rounds and carries the halo planes): the first invalid byte wins at round,
    trip and chunk boundaries and inside the halo positions; a byte after the last window is never examined; the shipped form gives
    the same answers (hamming/scalar.rs:11-48 over naive.rs:3-20 per window)."""
    import bitnuc_amd as bn
    ctx = sweep_ctx
    rng = np.random.default_rng(77 + impl)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    C = {2: 12, 3: 20, 4: 16, 5: 32}[impl]
    n = (2 * C + 5) * 1024 + 77
    s = alpha[rng.integers(0, 8, size=n)].copy()
    k, q = 31, 0x0123456789ABCDEF & ((1 << 62) - 1)
    prev = ctx.set_variant("scan_impl", impl)
    try:
        assert ctx.get("scan_impl") == impl
        assert np.array_equal(ctx.kmer_hdist_scan(s, k, q), oracle.kmer_hdist_scan(s, k, q))
        for pos in (0, 15, 16, 1023, 1024, 1025, 1039, 1040, 1055, 1056, 4095, 4096, 4 * 1024 + 31, C * 1024 - 1, C * 1024, C * 1024 + 17, C * 1024 + 31, C * 1024 + 32,
                    2 * C * 1024 - 1, 2 * C * 1024 + 1, (2 * C + 4) * 1024 + 5, n - k - 1, n - 1):
            t = s.copy()
            t[pos] = ord("N")
            if pos + 9 < n:
                t[pos + 9] = ord("X")  # a later invalid byte never wins
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.kmer_hdist_scan(t, k, q)
            assert (ei.value.byte, ei.value.index) == (ord("N"), pos), pos
        for kk in (1, 2, 16, 17, 32):
            for m in (kk, 1056, 1057, C * 1024 + 31, C * 1024 + 32, C * 1024 + 33, n):
                if m < kk:
                    continue
                assert np.array_equal(ctx.kmer_hdist_scan(s[:m], kk, q), oracle.kmer_hdist_scan(s[:m], kk, q)), (kk, m)
    finally:
        ctx.set_variant("scan_impl", prev)


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
