"""GPU tests of the asynchronous contract of the C ABI (include/bitnuc_hip.h "Asynchronous errors", "Capture rules"): per-launch error
slots and their lifetime, the slot ring growing instead of synchronising, hipGraph capture and replay of every _dev entry point, context
scratch held by a recorded launch, the order of data errors across an implicit drain (packing/avx.rs:86-91 is the error rule kept).
(Filed by component in round 5; the tests came from test_gpu_round3.py / test_gpu_round4.py unchanged.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xB17C0DE


# ---- table-driven ragged batches: asynchronous plan emission ---------------------------------------------------------------
def _oracle_batch(oracle, seq, off):
    words, wo = [], [0]
    for i in range(len(off) - 1):
        s = seq[int(off[i]):int(off[i + 1])]
        w = oracle.encode(s) if len(s) else np.zeros(0, np.uint64)  # the reference's idiom: one encode() per sequence
        words.append(w)
        wo.append(wo[-1] + len(w))
    return (np.concatenate(words) if words else np.zeros(0, np.uint64)), np.array(wo, dtype=np.int64)


def _reads(oracle, count, L, seed):
    import torch
    dev = torch.device("cuda:0")
    seq = torch.from_numpy(oracle.nucgen(count * L, seed)).to(dev)
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    return seq, off


def _oracle_fixed_batch(oracle, seq, count, L):
    h = seq.cpu().numpy()
    return np.concatenate([oracle.encode(h[i * L:(i + 1) * L]) for i in range(count)])


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
        expect_s = _oracle_fixed_batch(oracle, seq_s, small, L)
        out_s.zero_(); back_s.zero_()
        g.replay()
        c.sync()
        assert np.array_equal(out_s.cpu().numpy().view(np.uint64), expect_s) and torch.equal(back_s, seq_s)
        # the larger ordinary batch: the scratch plan grows.  Fill memory churn in between so that a freed buffer would be reused.
        c.encode_batch_dev(seq_l, off_l, wo_l, large, tot_l, out_l)
        c.decode_batch_dev(out_l, wo_l, off_l, large, tot_l, back_l)
        c.sync()
        assert np.array_equal(out_l.cpu().numpy().view(np.uint64), _oracle_fixed_batch(oracle, seq_l, large, L)) and torch.equal(back_l, seq_l)
        churn = [torch.full((1 << 20,), 0x5A, dtype=torch.uint8, device=dev) for _ in range(64)]
        # replay on NEW data of the recorded shape: it runs plan_emit_kernel into the buffer the graph holds
        seq_s.copy_(torch.from_numpy(oracle.nucgen(small * L, 13)).to(dev))
        expect_s = _oracle_fixed_batch(oracle, seq_s, small, L)
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
    # A, host-pointer call (defers A), B, ANOTHER host-pointer call (its drain finds B while A is still deferred), then a third
    # error C that only the sync's own drain sees: three syncs report A, B, C in that order -- the deferred errors are a FIFO
    c.encode_dev(a, n, words)
    assert np.array_equal(c.encode_array(host), w)
    c.encode_dev(b, n, words)
    assert np.array_equal(c.decode_array(w, host.size), host)
    b2 = a.clone()
    b2[1234] = ord("A")
    b2[77] = ord("!")
    torch.cuda.synchronize()
    c.encode_dev(b2, n, words)
    seen = []
    for _ in range(3):
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        seen.append((ei.value.byte, ei.value.index))
    assert seen == [(ord("N"), 1234), (ord("x"), 2_999_999), (ord("!"), 77)], seen
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
