"""GPU tests of the host-pointer entry points (the drop-in forms for callers whose data lives in host memory; PCIe-bound, never the reported
value): host code vs kernels on both sides of the size-dispatch cutoff, the pipelined path against the oracle with errors in
every chunk, every buffer-reuse guard of the pipeline (>= 9 chunks, both engines, in a child process), the pipe's thread budget.
(Filed by component in round 5; tests from test_gpu_round2.py / test_gpu_round3.py unchanged.)"""
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


# ---- host-pointer path ------------------------------------------------------------------------------------
@pytest.fixture()
def host_ctx():
    """A context with the library's default size dispatch (not forced to the GPU)."""
    import bitnuc_amd
    c = bitnuc_amd.Context(0)
    yield c
    c.close()


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
