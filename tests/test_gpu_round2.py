"""GPU parity tests added in round 2 (all through the C ABI, kernels forced by the `ctx` fixture):

* SURVEY 8(d)'s parity variants at BASELINE size (10^9 + 17 bases): lower-case mix p = 0.25, the benches' cyclic
  `bases[i % 4]` pattern (benches/simd_comparison.rs:4-7), and 'N' / 0x00 / 0xFF planted at {0, 15, 16, 31, 32, L-1}
  with (byte, index) and the words before the failing chunk checked (packing/mod.rs:181-196, packing/avx.rs:86-91,142-143);
* the pipelined host-pointer path (pinned double buffers, three streams) against the oracle, errors in every chunk;
* host code vs kernels on both sides of the size-dispatch cutoff;
* (config 4 through the C ABI on 2 / 4 / 8 GPUs now lives in tests/test_gpu_round3.py)
* bench.py --gpus 2 started with no launcher on one shared GPU.
"""
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
RNG = np.random.default_rng(777)


def rand_seq(n, lower=0.25):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[RNG.integers(0, 4, size=n)]
    return np.where(RNG.random(n) < lower, s | 0x20, s).astype(np.uint8)


def gen_words(torch, n, seed, dev):
    """Closed form of the seeded stream's packed words (the generator's words ARE the packed words)."""
    nw = (n + 31) // 32
    idx = torch.arange(1, nw + 1, dtype=torch.int64, device=dev)
    z = idx * (-7046029254386353131) + seed  # 0x9E3779B97F4A7C15 as i64, wraps mod 2^64

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)   # 0x94D049BB133111EB
    z = z ^ lsr(z, 31)
    if n % 32:
        z[-1] &= (1 << (2 * (n % 32))) - 1
    return z


def test_nucgen_lowercase_and_cyclic_flags(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    for n, first, flags in [(1000, 0, 2), (100003, 32 * 77, 2), (4097, 5, 2), (333, 1 << 40, 2), (1000, 3, 3), (70, 17, 3), (65, 31, 2)]:
        t = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.nucgen_dev(t, n, SEED, first, flags)
        ctx.sync()
        h = t.cpu().numpy()
        exp = oracle.nucgen(n, SEED, first, flags)
        assert np.array_equal(h[:n], exp), (n, first, flags)
        assert not h[n:].any()
    big = oracle.nucgen(1 << 20, SEED, 0, 2)
    frac = float((big & 0x20).astype(bool).mean())
    assert 0.24 < frac < 0.26  # p = 0.25
    assert set(np.unique(big)) == set(b"ACGTacgt")


def test_full_size_lowercase_mix_and_cyclic(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    n = 10**9 + 17
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    # (1) lower-case mix p = 0.25: same words as the upper-case stream (as_2bit("acgt") == as_2bit("ACGT"), packing/mod.rs:181),
    #     decode gives the upper-cased input
    ctx.nucgen_dev(seq, n, SEED, 0, 2)
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    assert torch.equal(words, gen_words(torch, n, SEED, dev))
    assert torch.equal(back, seq & 0xDF)
    lower = int(((seq & 0x20) != 0).sum().item())
    assert 0.2499 < lower / n < 0.2501
    for off in (0, 32 * 12_345_678, (n // 32 - 4096) * 32):  # spot blocks against the CPU oracle, bytes as generated
        m = min(32 * 4096, n - off)
        h = seq[off:off + m].cpu().numpy()
        assert np.array_equal(h, oracle.nucgen(m, SEED, off, 2))
        assert np.array_equal(words[off // 32: off // 32 + (m + 31) // 32].cpu().numpy().view(np.uint64), oracle.encode(h))
    # (2) the benches' cyclic pattern: every full word is 0xE4E4..E4, the 17-base tail word its low 34 bits
    ctx.nucgen_dev(seq, n, SEED, 0, 1)
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    e4 = int(np.uint64(0xE4E4E4E4E4E4E4E4).view(np.int64))
    assert bool((words[:-1] == e4).all().item())
    assert int(words[-1].item()) == 0xE4E4E4E4E4E4E4E4 & ((1 << 34) - 1)
    assert torch.equal(back, seq)
    assert bytes(seq[:8].cpu().numpy()) == b"ACGTACGT" and bytes(seq[-5:].cpu().numpy()) == bytes(b"ACGT"[(n - 5 + i) % 4] for i in range(5))


def test_full_size_planted_invalid_bytes(ctx, oracle):
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = 10**9 + 17
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    ctx.nucgen_dev(seq, n, SEED)
    ctx.sync()
    expect = gen_words(torch, n, SEED, dev)
    for pos in (0, 15, 16, 31, 32, n - 1):
        orig = int(seq[pos].item())
        for byte in (ord("N"), 0x00, 0xFF):
            seq[pos] = byte
            later = pos + 1000 if pos + 1000 < n else None
            if later is not None:
                keep = int(seq[later].item())
                seq[later] = ord("X")  # a later invalid byte must not be the one reported
            words.fill_(-1)
            torch.cuda.synchronize()
            ctx.encode_dev(seq, n, words)
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.sync()
            assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", byte, pos), (pos, byte)
            # the reference's Vec holds the words of the chunks before the failing one (packing/avx.rs:142-143)
            k = pos // 32
            assert torch.equal(words[:k], expect[:k]), (pos, byte)
            if later is not None:
                seq[later] = keep
        seq[pos] = orig
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.sync()
    assert torch.equal(words, expect)


# ---- host-pointer path ------------------------------------------------------------------------------------
@pytest.fixture()
def host_ctx():
    """A context with the library's default size dispatch (not forced to the GPU)."""
    import bitnuc_amd
    c = bitnuc_amd.Context(0)
    yield c
    c.close()


def test_host_code_and_kernels_agree_around_the_cutoff(host_ctx, ctx, oracle):
    cutoff, cutoff_d = host_ctx.get("host_cutoff"), host_ctx.get("host_cutoff_decode")
    assert (cutoff, cutoff_d) == (1 << 20, 1 << 19) and host_ctx.get("force_gpu") == 0 and ctx.get("force_gpu") == 1  # the measured crossovers
    for n in (1, 31, 32, 33, 1000, cutoff_d - 1, cutoff_d, cutoff_d + 1, cutoff - 1, cutoff, cutoff + 1, 3 * cutoff + 5):
        s = rand_seq(n)
        wh, wg = host_ctx.encode_array(s), ctx.encode_array(s)
        assert np.array_equal(wh, wg) and np.array_equal(wh, oracle.encode(s)), n
        assert np.array_equal(host_ctx.decode_array(wh, n), ctx.decode_array(wg, n)), n
        t = rand_seq(n)
        wt = ctx.encode_array(t)
        assert host_ctx.hdist(wh, wt, n) == ctx.hdist(wg, wt, n) == oracle.hdist(wh, wt, n), n
    # a lowered cutoff moves the boundary; force_gpu removes it
    host_ctx.set_variant("host_cutoff", 100)
    s = rand_seq(99)
    assert np.array_equal(host_ctx.encode_array(s), oracle.encode(s))
    s = rand_seq(100)
    assert np.array_equal(host_ctx.encode_array(s), oracle.encode(s))
    # same error, same truncated Vec on both sides of the dispatch
    import bitnuc_amd as bn
    for c in (host_ctx, ctx):
        s = rand_seq(50).copy()
        s[40] = ord("N")
        with pytest.raises(bn.NucleotideError) as ei:
            c.encode_array(s)
        assert (ei.value.byte, ei.value.index, len(ei.value.words)) == (ord("N"), 40, 1)
    for c in (host_ctx, ctx):  # single words: host code vs a batch of one on the device
        assert c.as_2bit(b"ACTGGAAAATTTTAAGG") == 0x283FC02B4  # packing/mod.rs:173
        assert c.from_2bit_alloc(71620941647064936, 28) == b"AGGCTTGAGGCCCATTCTCTGATCGTTT"  # unpacking/mod.rs:206-214
        assert c.hdist_scalar(c.as_2bit(b"ACTGACTG"), c.as_2bit(b"TGCATGCA"), 8) == 8  # hamming/scalar.rs:93-100


@pytest.mark.parametrize("pipeline", [1, 0], ids=["pipelined", "simple"])
def test_host_pointer_bulk_path_vs_oracle(host_ctx, oracle, pipeline):
    import bitnuc_amd as bn
    host_ctx.set_variant("host_pipeline", pipeline)
    chunk = 32 << 20
    for n in (8 << 20, chunk + 17, 2 * chunk + chunk // 2 + 5):
        s = oracle.nucgen(n, SEED + n, 0, 2)
        w = host_ctx.encode_array(s)
        assert np.array_equal(w, oracle.encode(s)), n
        d = host_ctx.decode_array(w, n)
        assert np.array_equal(d, s & 0xDF), n
    # errors: first invalid byte in sequence order, whichever chunk holds it, and the words before it
    n = 2 * chunk + 1000
    s = oracle.nucgen(n, SEED, 0, 0)
    expect = oracle.encode(s)
    for bad in (0, chunk - 1, chunk, chunk + 33, 2 * chunk + 999):
        t = s.copy()
        t[bad] = ord("N")
        if bad + chunk < n:
            t[bad + chunk] = ord("X")  # an invalid byte in a later chunk must not win
        with pytest.raises(bn.NucleotideError) as ei:
            host_ctx.encode_array(t)
        assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", ord("N"), bad), bad
        assert np.array_equal(ei.value.words, expect[: bad // 32]), bad
    # the context stays usable after an error
    assert np.array_equal(host_ctx.encode_array(s), expect)


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
    # round 3: the fabric roofline entry (one shared GPU: null + the reason), the C-ABI block's skip reason, the CPU baseline on an N > 1 line
    assert line["allgather_packed"]["roofline"]["value"] is None and "device" in line["allgather_packed"]["roofline"]["reason"]
    assert "skipped" in line["c_abi_allgather"] and line["cpu_baseline"]["value"] > 0
    assert line["roofline"]["kernel"] in ("encode_kernel", "decode_kernel") and "roofline_step" in line
