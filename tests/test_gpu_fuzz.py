"""Differential fuzzing of the device entry points against the CPU oracle (-m gpu).

Seeded, so a failure reproduces; every case draws its own length / alignment / content:
  * lengths cluster around the kernels' tile edges (32, 64, 992, 1024, 4096, 16384, 65536 +- a few) besides uniform draws;
  * device pointers start at arbitrary byte offsets inside a larger allocation (u64 operands at arbitrary multiples of 8);
  * every output sits between two canary regions that must come back untouched (no out-of-bounds stores, no stores to
    separator / padding bytes the entry point does not own);
  * inputs mix lower case in, and a share of the cases plants invalid bytes: the first one in buffer order has to come back
    as InvalidBase(byte) with its offset, exactly as the oracle reports it (src/utils/packing/naive.rs:10-16); invalid bytes
    OUTSIDE what an entry point reads (between strided k-mers, around a batch) must not be reported.
All cases of one entry point run in one test (one process, one context), failures are collected and reported together.
"""
import os

import numpy as np
import pytest

import bitnuc_amd as bn

def _rng(seed):
    """The drawn cases are fixed (the seeds below) so that a failure reproduces; BITNUC_FUZZ_SEED=n XORs n into every seed for a one-off
    run over other cases (profiles/README.md, round 4: three extra seeds, all green)."""
    return np.random.default_rng(seed ^ int(os.environ.get("BITNUC_FUZZ_SEED", "0"), 0))


pytestmark = pytest.mark.gpu

CANARY = 0xA5
EDGES = (1, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 992, 1023, 1024, 1025, 2048, 4095, 4096, 4097, 8192, 16383, 16384, 16385,
         32768, 65535, 65536, 65537, 262144, 1 << 20)


def _torch():
    import torch
    return torch


def draw_len(rng, most, least=0):
    r = rng.random()
    if r < 0.45:
        e = int(EDGES[rng.integers(len(EDGES))]) * int(rng.integers(1, 4)) + int(rng.integers(-3, 4))
    elif r < 0.75:
        e = int(rng.integers(least, 5000))
    else:
        e = int(rng.integers(least, most + 1))
    return max(least, min(most, e))


def draw_seq(rng, n, lower=0.25):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    if lower > 0 and n:
        m = rng.random(n) < lower
        s = np.where(m, s | 0x20, s).astype(np.uint8)
    return s


BAD_BYTES = (ord("N"), 0, 0xFF, ord("n"), ord("U"), ord("@"), ord("B"), 0x80 | ord("A"), ord(" "), ord("\n"))


def plant(rng, s, lo=0, hi=None, most=3):
    """Overwrite 1..most positions of s[lo:hi] with invalid bytes; -> sorted positions."""
    hi = s.size if hi is None else hi
    if hi <= lo:
        return []
    pos = sorted({int(rng.integers(lo, hi)) for _ in range(int(rng.integers(1, most + 1)))})
    if rng.random() < 0.3:  # tile edges are where a per-tile first-bad reduction goes wrong
        pos = sorted(set(pos) | {min(hi - 1, max(lo, lo + int(EDGES[rng.integers(len(EDGES))]) - int(rng.integers(0, 2))))})
    for p in pos:
        s[p] = BAD_BYTES[rng.integers(len(BAD_BYTES))]
    return pos


class Arena:
    """A device allocation holding one operand at an arbitrary byte offset, canaries on both sides."""

    def __init__(self, nbytes, offset, fill=None, pad=256):
        torch = _torch()
        self.pad, self.n, self.off = pad, int(nbytes), int(offset)
        self.host = np.full(self.off + pad + self.n + pad + 64, CANARY, dtype=np.uint8)
        if fill is not None:
            self.host[self.lo:self.lo + self.n] = np.frombuffer(np.ascontiguousarray(fill).tobytes(), dtype=np.uint8)
        self.t = torch.from_numpy(self.host.copy()).cuda()
        assert self.t.data_ptr() % 256 == 0

    @property
    def lo(self):
        return self.off + self.pad

    @property
    def ptr(self):
        return self.t.data_ptr() + self.lo

    def back(self):
        """-> (payload bytes, canaries intact?)"""
        h = self.t.cpu().numpy()
        ok = bool((h[:self.lo] == CANARY).all() and (h[self.lo + self.n:] == CANARY).all())
        return h[self.lo:self.lo + self.n].copy(), ok


def expect_error(ctx, oracle_call):
    """Run the oracle; -> None if it succeeds else (kind, byte, index)."""
    import oracle_py
    try:
        return None, oracle_call()
    except oracle_py.OracleError as e:
        return (e.kind, e.byte, e.index), None


def sync_error(ctx):
    try:
        ctx.sync()
        return None
    except bn.NucleotideError as e:
        return (e.kind, getattr(e, "byte", None), getattr(e, "index", None))


# ------------------------------------------------------------------------------------------------------------------
def test_fuzz_encode_decode_dev(ctx, oracle):
    rng = _rng(0xE1C0DE)
    fails = []
    for case in range(400):
        n = draw_len(rng, 3_000_000, 1)
        s = draw_seq(rng, n, lower=0.25 if case % 3 else 0.0)
        bad = plant(rng, s) if rng.random() < 0.3 else []
        nw = (n + 31) // 32
        src = Arena(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
        dst = Arena(8 * nw, 8 * int(rng.integers(0, 8)))
        ctx.encode_dev(src.ptr, n, dst.ptr)
        got_err = sync_error(ctx)
        want_err, want = expect_error(ctx, lambda: oracle.encode(s))
        words, intact = dst.back()
        tag = f"encode case {case}: n={n} src+{src.off} dst+{dst.off} bad={bad[:3]}"
        if not intact:
            fails.append(tag + ": canary overwritten")
        if want_err is not None:
            if got_err != (want_err[0], want_err[1], bad[0]):
                fails.append(tag + f": error {got_err}, oracle {want_err}")
            continue
        if got_err is not None:
            fails.append(tag + f": unexpected error {got_err}")
            continue
        words = words.view(np.uint64)
        if not np.array_equal(words, want):
            fails.append(tag + f": first differing word {int(np.flatnonzero(words != want)[0])}")
            continue
        # decode what was encoded: n_bases anywhere in the last word's range, sometimes more words than needed
        nb = n if rng.random() < 0.6 else int(rng.integers(max(1, 32 * (nw - 1) + 1), 32 * nw + 1))
        extra = int(rng.integers(0, 3))
        wsrc = Arena(8 * (nw + extra), 8 * int(rng.integers(0, 8)), np.concatenate([want, np.full(extra, 0xFFFFFFFFFFFFFFFF, np.uint64)]))
        out = Arena(nb, int(rng.integers(0, 64)) if case % 5 else 0)
        ctx.decode_dev(wsrc.ptr, nw + extra, nb, out.ptr)
        got_err = sync_error(ctx)
        text, intact = out.back()
        tag = f"decode case {case}: n_bases={nb} words={nw}+{extra} src+{wsrc.off} dst+{out.off}"
        if got_err is not None:
            fails.append(tag + f": unexpected error {got_err}")
        elif not intact:
            fails.append(tag + ": canary overwritten")
        elif not np.array_equal(text, oracle.decode(want, nb)):
            fails.append(tag + f": first differing byte {int(np.flatnonzero(text != oracle.decode(want, nb))[0])}")
    assert not fails, "\n".join(fails[:20]) + f"\n({len(fails)} failing cases)"


def test_fuzz_decode_short_buffer(ctx):
    """n_words < ceil(n_bases / 32) is InvalidLength(n_bases) before anything is launched (src/utils/unpacking/mod.rs:42-45)."""
    rng = _rng(0x5407)
    for _ in range(20):
        nb = draw_len(rng, 100_000, 33)
        need = (nb + 31) // 32
        have = int(rng.integers(0, need))
        w = Arena(8 * max(have, 1), 0, np.zeros(max(have, 1), np.uint64))
        out = Arena(nb, 0)
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.decode_dev(w.ptr, have, nb, out.ptr)
            ctx.sync()
        assert (ei.value.kind, ei.value.len) == ("InvalidLength", nb)
        assert out.back()[1] and not (out.back()[0] != CANARY).any()


def test_fuzz_kmer_batch_dev(ctx, oracle):
    rng = _rng(0xBA7C4)
    fails = []
    for case in range(600):
        k = int(rng.integers(1, 33)) if case % 3 else (31, 32, 21, 16)[case % 4]
        r = rng.random()
        if r < 0.35:
            stride = k  # dense
        elif r < 0.6:
            stride = (1, 2, 4, 8, 16)[rng.integers(5)]  # overlapping windows (the slide kernels for the power-of-two strides)
        elif r < 0.8:
            stride = k + int(rng.integers(1, 40))  # records with separators
        else:
            stride = int(rng.integers(1, 70))
        count = draw_len(rng, 200_000, 1)
        n = (count - 1) * stride + k
        s = draw_seq(rng, n + 8)[:n]
        gaps_bad = False
        if stride > k and rng.random() < 0.5:  # separators are not bases: '\n', '>' ... must never be looked at
            m = (np.arange(n) % stride) >= k
            s[m] = np.frombuffer(b"\n>N\x00", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            gaps_bad = True
        bad = []
        if rng.random() < 0.3:
            cand = plant(rng, s.copy())  # positions only
            bad = [p for p in cand if stride <= k or (p % stride) < k]
            for p in bad:
                s[p] = BAD_BYTES[rng.integers(len(BAD_BYTES))]
        src = Arena(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
        dst = Arena(8 * count, 8 * int(rng.integers(0, 8)))
        ctx.as_2bit_batch_dev(src.ptr, k, stride, count, dst.ptr)
        got_err = sync_error(ctx)
        want_err, want = expect_error(ctx, lambda: oracle.as_2bit_batch(s, k, stride, count))
        words, intact = dst.back()
        tag = f"batch case {case}: k={k} stride={stride} count={count} src+{src.off} dst+{dst.off} bad={bad[:3]} gaps_bad={gaps_bad}"
        if not intact:
            fails.append(tag + ": canary overwritten")
        if want_err is not None:
            if got_err is None or got_err[:2] != want_err[:2] or got_err[2] != want_err[2]:
                fails.append(tag + f": error {got_err}, oracle {want_err}")
            continue
        if got_err is not None:
            fails.append(tag + f": unexpected error {got_err}")
        elif not np.array_equal(words.view(np.uint64), want):
            fails.append(tag + f": first differing k-mer {int(np.flatnonzero(words.view(np.uint64) != want)[0])}")
    assert not fails, "\n".join(fails[:20]) + f"\n({len(fails)} failing cases)"


def test_fuzz_scan_dev(ctx, oracle):
    torch = _torch()
    rng = _rng(0x5CA4)
    fails = []
    for case in range(400):
        k = int(rng.integers(1, 33)) if case % 3 else 31
        n = draw_len(rng, 2_000_000, k)
        s = draw_seq(rng, n)
        bad = plant(rng, s) if rng.random() < 0.3 else []
        query = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)  # bits above 2k are ignored (scalar.rs:26-31)
        nwin = n - k + 1
        src = Arena(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
        dst = Arena(nwin, int(rng.integers(0, 64)) if case % 5 else 0)
        ctx.kmer_hdist_scan_dev(src.ptr, n, k, query, dst.ptr)
        got_err = sync_error(ctx)
        want_err, want = expect_error(ctx, lambda: oracle.kmer_hdist_scan(s, k, query))
        dist, intact = dst.back()
        tag = f"scan case {case}: n={n} k={k} src+{src.off} dst+{dst.off} bad={bad[:3]}"
        if not intact:
            fails.append(tag + ": canary overwritten")
        if want_err is not None:
            if got_err != (want_err[0], want_err[1], bad[0]):
                fails.append(tag + f": error {got_err}, oracle {want_err}")
            continue
        if got_err is not None:
            fails.append(tag + f": unexpected error {got_err}")
            continue
        if not np.array_equal(dist, want):
            fails.append(tag + f": first differing window {int(np.flatnonzero(dist != want)[0])}")
            continue
        tau = int(rng.integers(0, k + 1))
        cnt = torch.full((3,), -1, dtype=torch.int64, device="cuda")
        ctx.kmer_hdist_count_dev(src.ptr, n, k, query, tau, cnt[1:].data_ptr())
        ctx.sync()
        c = cnt.cpu().numpy()
        if (int(c[0]), int(c[1]), int(c[2])) != (-1, int((want <= tau).sum()), -1):
            fails.append(tag + f": count(d <= {tau}) = {c.tolist()}, oracle {int((want <= tau).sum())}")
    assert not fails, "\n".join(fails[:20]) + f"\n({len(fails)} failing cases)"


def test_fuzz_packed_word_kernels(ctx, oracle):
    """hdist (bulk), base_counts, hdist_pairs / hdist_query, split_packed on random packed buffers."""
    torch = _torch()
    rng = _rng(0x9ACCED)
    fails = []
    for case in range(300):
        nb = draw_len(rng, 4_000_000, 1)
        nw = (nb + 31) // 32
        ea, eb = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        a = rng.integers(0, 1 << 63, nw + ea, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, nw + ea, dtype=np.uint64)
        b = a.copy()[:nw]
        flips = rng.integers(0, nw, max(1, nw // int(rng.integers(1, 50))))
        b[flips] ^= rng.integers(0, 1 << 63, flips.size, dtype=np.uint64)
        b = np.concatenate([b, rng.integers(0, 1 << 63, eb, dtype=np.uint64)])
        A = Arena(8 * a.size, 8 * int(rng.integers(0, 8)), a)
        B = Arena(8 * b.size, 8 * int(rng.integers(0, 8)), b)
        tag = f"packed case {case}: n_bases={nb} words={nw}+{ea}/{eb} a+{A.off} b+{B.off}"
        res = torch.full((3,), 0xDEAD, dtype=torch.int32, device="cuda")
        ctx.hdist_dev(A.ptr, a.size, B.ptr, b.size, nb, res[1:].data_ptr())
        ctx.sync()
        r = res.cpu().numpy()
        if (int(r[0]), int(r[1]), int(r[2])) != (0xDEAD, oracle.hdist(a, b, nb), 0xDEAD):
            fails.append(tag + f": hdist {r.tolist()}, oracle {oracle.hdist(a, b, nb)}")
        cnt = Arena(32, 0)
        ctx.base_counts_dev(A.ptr, a.size, nb, cnt.ptr)
        ctx.sync()
        c, intact = cnt.back()
        if not intact or [int(x) for x in c.view(np.uint64)] != oracle.base_counts(a, nb):
            fails.append(tag + f": base_counts {c.view(np.uint64).tolist()}, oracle {oracle.base_counts(a, nb)} canaries {intact}")
        # many pairs / one query, word length 1..32
        length = int(rng.integers(1, 33))
        m = min(a.size, b.size)
        d = Arena(m, int(rng.integers(0, 64)) if case % 3 else 0)
        ctx.hdist_pairs_dev(A.ptr, B.ptr, m, length, d.ptr)
        ctx.sync()
        got, intact = d.back()
        if not intact or not np.array_equal(got, oracle.hdist_pairs(a[:m], b[:m], length)):
            fails.append(tag + f": hdist_pairs len={length} dst+{d.off} differs (canaries {intact})")
        q = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
        d = Arena(m, int(rng.integers(0, 64)) if case % 3 else 0)
        ctx.hdist_query_dev(q, A.ptr, m, length, d.ptr)
        ctx.sync()
        got, intact = d.back()
        if not intact or not np.array_equal(got, oracle.hdist_pairs(a[:m], np.full(m, q, np.uint64), length)):
            fails.append(tag + f": hdist_query len={length} dst+{d.off} differs (canaries {intact})")
        # split_packed at an arbitrary base index, as the reference leaves its two buffers (functions/split.rs:63-99)
        idx = int(rng.integers(0, nb + 1)) if case % 4 else (0, nb, min(nb, 32), max(0, nb - 32))[(case // 4) % 4]
        try:
            wl, wr = oracle.split_packed(a[:nw], nb, idx)
        except Exception as e:  # the oracle refuses what the reference refuses
            wl = wr = None
            want_kind = getattr(e, "kind", type(e).__name__)
        if wl is not None:
            nl, nr = ctx.split_packed_sizes(nw, nb, idx)
            if (nl, nr) != (wl.size, wr.size):
                fails.append(tag + f": split sizes {(nl, nr)} oracle {(wl.size, wr.size)} idx={idx}")
                continue
            Lb, Rb = Arena(8 * nl, 8 * int(rng.integers(0, 8))), Arena(8 * nr, 8 * int(rng.integers(0, 8)))
            ctx.split_packed_dev(A.ptr, nw, nb, idx, Lb.ptr, Rb.ptr)
            ctx.sync()
            (gl, il), (gr, ir) = Lb.back(), Rb.back()
            if not (il and ir):
                fails.append(tag + f": split idx={idx} canary overwritten")
            elif not (np.array_equal(gl.view(np.uint64), wl) and np.array_equal(gr.view(np.uint64), wr)):
                fails.append(tag + f": split idx={idx} differs")
        else:
            try:
                ctx.split_packed_sizes(nw, nb, idx)
                fails.append(tag + f": split idx={idx} accepted, oracle {want_kind}")
            except bn.NucleotideError:
                pass
    assert not fails, "\n".join(fails[:20]) + f"\n({len(fails)} failing cases)"


def _ragged_case(rng, case):
    style = case % 5
    if style == 0:
        count = int(rng.integers(1, 40_000))
        lens = rng.integers(0, 200, count)
    elif style == 1:
        count = int(rng.integers(1, 3000))
        lens = rng.integers(100, 400, count)
        lens[rng.integers(0, count, max(1, count // 20))] = 0  # empty reads keep a zero-word slot
    elif style == 2:
        count = int(rng.integers(1, 200))
        lens = rng.integers(0, 60_000, count)  # reads longer than a tile
    elif style == 3:
        count = int(rng.integers(1, 20_000))
        lens = np.full(count, (150, 32, 31, 33, 64, 1)[rng.integers(6)])
    else:
        count = int(rng.integers(1, 2000))
        lens = np.where(rng.random(count) < 0.02, rng.integers(10_000, 100_000, count), rng.integers(0, 300, count))
    off = np.zeros(count + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens.astype(np.uint64))
    return count, off


def test_fuzz_ragged_batch_dev(ctx, oracle):
    """encode_batch_dev / decode_batch_dev (tables) and the same batch through a layout plan."""
    torch = _torch()
    rng = _rng(0x4A66ED)
    fails = []
    for case in range(200):
        count, off = _ragged_case(rng, case)
        pre = int(rng.integers(0, 100)) if case % 3 else 0
        total = int(off[-1])
        s = draw_seq(rng, total)
        buf = np.concatenate([np.full(pre, ord("N"), np.uint8), s, np.full(7, ord("N"), np.uint8)])  # neighbours are not bases
        off2 = off + np.uint64(pre)
        bad = []
        if rng.random() < 0.3 and total:
            bad = [p + pre for p in plant(rng, s)]
            buf[pre:pre + total] = s
        want_wo = np.zeros(count + 1, dtype=np.uint64)
        want_wo[1:] = np.cumsum((np.diff(off) + np.uint64(31)) // np.uint64(32))
        tw = int(want_wo[-1])
        want_err = None
        try:
            want = np.concatenate([oracle.encode(s[int(off[i]):int(off[i + 1])]) if off[i + 1] > off[i] else np.zeros(0, np.uint64) for i in range(count)]) if count < 5000 else None
        except Exception as e:
            want_err, want = ("InvalidBase", e.byte, None), None
        if want is None and want_err is None and not bad:  # large counts: the vectorised restatement of the same definition
            codes = ((s >> 1) & 3).astype(np.uint64)  # A/a 0, C/c 1, T/t 2, G/g 3 -> fix the G/T order below
            codes = np.where(codes == 2, np.uint64(3), np.where(codes == 3, np.uint64(2), codes))
            want = np.zeros(tw, dtype=np.uint64)
            read = np.repeat(np.arange(count), np.diff(off).astype(np.int64))
            inpos = np.arange(total, dtype=np.int64) - off[read].astype(np.int64)
            np.bitwise_or.at(want, want_wo[read].astype(np.int64) + inpos // 32, codes << (np.uint64(2) * (inpos % 32).astype(np.uint64)))
        src_off = int(rng.integers(0, 64)) if case % 4 else 0
        S = Arena(buf.size, src_off, buf)
        d_off = torch.from_numpy(off2.view(np.int64)).cuda()
        d_wo = torch.full((count + 3,), -7, dtype=torch.int64, device="cuda")
        tag = f"ragged case {case}: count={count} bases={total} pre={pre} src+{src_off} bad={bad[:3]}"
        got_tw = ctx.batch_word_offsets_dev(d_off, count, d_wo[1:].data_ptr())
        wo_back = d_wo.cpu().numpy()
        if got_tw != tw or not np.array_equal(wo_back[1:count + 2].view(np.uint64), want_wo) or wo_back[0] != -7 or wo_back[-1] != -7:
            fails.append(tag + f": word offsets differ (total {got_tw} vs {tw})")
            continue
        plan = bn.BatchPlan(ctx, d_off, count)
        if plan.total_words != tw:
            fails.append(tag + f": plan total {plan.total_words} vs {tw}")
        for via in ("tables", "plan"):
            W = Arena(8 * tw, 8 * int(rng.integers(0, 8)))
            if via == "tables":
                ctx.encode_batch_dev(S.ptr, d_off, d_wo[1:].data_ptr(), count, tw, W.ptr)
            else:
                plan.encode_dev(S.ptr, W.ptr)
            got_err = sync_error(ctx)
            words, intact = W.back()
            if not intact:
                fails.append(tag + f" [{via}]: encode canary overwritten")
            if bad:
                if got_err is None or got_err[0] != "InvalidBase" or got_err[2] != bad[0] or got_err[1] != int(buf[bad[0]]):
                    fails.append(tag + f" [{via}]: error {got_err}, expected InvalidBase({int(buf[bad[0]])}) at {bad[0]}")
                continue
            if got_err is not None:
                fails.append(tag + f" [{via}]: unexpected error {got_err}")
                continue
            if want is not None and not np.array_equal(words.view(np.uint64), want):
                fails.append(tag + f" [{via}]: first differing word {int(np.flatnonzero(words.view(np.uint64) != want)[0])}")
                continue
            # decode back into a buffer whose neighbours and (none here) gaps must stay as they were
            Wsrc = Arena(8 * tw, 8 * int(rng.integers(0, 8)), words.view(np.uint64))
            out_off = int(rng.integers(0, 64)) if case % 2 else 0
            # the decode writes [off2[0], off2[-1]) relative to its base pointer: give it the whole buffer extent
            D = Arena(buf.size, out_off)
            if via == "tables":
                ctx.decode_batch_dev(Wsrc.ptr, d_wo[1:].data_ptr(), d_off, count, tw, D.ptr)
            else:
                plan.decode_dev(Wsrc.ptr, D.ptr)
            got_err = sync_error(ctx)
            text, intact = D.back()
            if got_err is not None or not intact:
                fails.append(tag + f" [{via}]: decode error {got_err} canaries {intact}")
            elif not (np.array_equal(text[pre:pre + total], s & 0xDF) and (text[:pre] == CANARY).all() and (text[pre + total:] == CANARY).all()):
                fails.append(tag + f" [{via}]: decode differs or wrote outside [offsets[0], offsets[-1])")
        plan.close()
    assert not fails, "\n".join(fails[:20]) + f"\n({len(fails)} failing cases)"


def test_fuzz_fixed_reads_dev(ctx, oracle):
    rng = _rng(0xF17ED)
    fails = []
    for case in range(300):
        L = int(rng.integers(1, 700)) if case % 3 else (150, 32, 31, 33, 64, 100, 250, 151)[(case // 3) % 8]
        stride = L if rng.random() < 0.5 else L + int(rng.integers(1, 9))
        count = draw_len(rng, max(1, 3_000_000 // max(L, 8)), 1)
        n = (count - 1) * stride + L
        s = draw_seq(rng, n)
        if stride > L:
            s[(np.arange(n) % stride) >= L] = ord("\n")  # separators: never read as bases, never overwritten by the decode
        bad = []
        if rng.random() < 0.3:
            cand = plant(rng, s.copy())
            bad = [p for p in cand if (p % stride) < L]
            for p in bad:
                s[p] = BAD_BYTES[rng.integers(len(BAD_BYTES))]
        wpr = (L + 31) // 32
        S = Arena(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
        W = Arena(8 * wpr * count, 8 * int(rng.integers(0, 8)))
        ctx.encode_fixed_dev(S.ptr, L, stride, count, W.ptr)
        got_err = sync_error(ctx)
        words, intact = W.back()
        tag = f"fixed case {case}: L={L} stride={stride} count={count} src+{S.off} bad={bad[:3]}"
        if not intact:
            fails.append(tag + ": encode canary overwritten")
        if bad:
            if got_err is None or got_err[0] != "InvalidBase" or got_err[2] != bad[0] or got_err[1] != int(s[bad[0]]):
                fails.append(tag + f": error {got_err}, expected InvalidBase({int(s[bad[0]])}) at {bad[0]}")
            continue
        if got_err is not None:
            fails.append(tag + f": unexpected error {got_err}")
            continue
        rows = words.view(np.uint64).reshape(count, wpr)
        if L <= 32:
            want = oracle.as_2bit_batch(s, L, stride, count).reshape(count, 1)
        else:
            pick = sorted({0, count - 1, *[int(x) for x in rng.integers(0, count, 40)]})
            want = None
            for r in pick:
                if not np.array_equal(rows[r], oracle.encode(s[r * stride:r * stride + L])):
                    fails.append(tag + f": read {r} differs")
                    break
        if want is not None and not np.array_equal(rows, want):
            fails.append(tag + f": first differing read {int(np.flatnonzero((rows != want).any(axis=1))[0])}")
            continue
        Wsrc = Arena(8 * wpr * count, 8 * int(rng.integers(0, 8)), words.view(np.uint64))
        D = Arena(n, int(rng.integers(0, 64)) if case % 2 else 0, np.full(n, ord("#"), np.uint8))
        ctx.decode_fixed_dev(Wsrc.ptr, L, stride, count, D.ptr)
        got_err = sync_error(ctx)
        text, intact = D.back()
        expect = np.where((np.arange(n) % stride) < L, s & 0xDF, ord("#")).astype(np.uint8)
        if got_err is not None or not intact:
            fails.append(tag + f": decode error {got_err} canaries {intact}")
        elif not np.array_equal(text, expect):
            fails.append(tag + f": decode differs at byte {int(np.flatnonzero(text != expect)[0])} (separators must stay)")
    assert not fails, "\n".join(fails[:20]) + f"\n({len(fails)} failing cases)"


# ------------------------------------------------------------------------------------------------------------------
# host-pointer entry points: the three size regimes (host code below the cutoff, one staged launch, the chunked three-stream
# pipeline) with caller buffers at arbitrary addresses between canaries.  A fresh process, so that BITNUC_PIPE_CHUNK_MB=1
# makes an 8..14 Mi-base call run 8..14 chunks (every pinned / device buffer reused several times per call).
_HOST_FUZZ_CHILD = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import bitnuc_amd as bn
from bitnuc_amd import _lib as L
import oracle_py as oracle
from test_gpu_fuzz import draw_seq, plant, CANARY, _rng

ctx = bn.Context(0)
lib = ctx._lib
rng = _rng(0x4057F)
CUT = (512 << 10, 1 << 20, 8 << 20)
fails = []


def draw_n(least=1):
    r = rng.random()
    if r < 0.3:
        return max(least, int(rng.integers(least, 3000)))
    if r < 0.75:
        return max(least, int(CUT[rng.integers(3)]) + int(rng.integers(-70, 70)))
    if r < 0.9:
        return int(rng.integers(least, 2 << 20))
    return int(rng.integers(8 << 20, 14 << 20))


class HostBuf:
    def __init__(self, nbytes, offset, fill=None):
        self.raw = np.full(nbytes + offset + 256 + 320, CANARY, dtype=np.uint8)
        base = (-self.raw.ctypes.data) % 64  # offsets are relative to a 64-byte line
        self.lo, self.n = base + 128 + offset, nbytes
        if fill is not None:
            self.raw[self.lo:self.lo + nbytes] = np.frombuffer(np.ascontiguousarray(fill).tobytes(), dtype=np.uint8)
    @property
    def ptr(self):
        return C.c_void_p(self.raw.ctypes.data + self.lo)
    def back(self):
        ok = bool((self.raw[:self.lo] == CANARY).all() and (self.raw[self.lo + self.n:] == CANARY).all())
        return self.raw[self.lo:self.lo + self.n], ok


def status(st, err):
    if st == L.OK:
        return None
    return (st, int(err.byte), int(err.index), int(err.value))


for case in range(150):
    n = draw_n()
    s = draw_seq(rng, n, lower=0.25 if case % 3 else 0.0)
    bad = plant(rng, s) if rng.random() < 0.3 else []
    nw = (n + 31) // 32
    src = HostBuf(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
    dst = HostBuf(8 * nw, 8 * int(rng.integers(0, 8)))
    got_nw, err = C.c_size_t(0), L.BitnucErr()
    st = lib.bitnuc_encode(ctx._h, src.ptr, n, dst.ptr, C.byref(got_nw), C.byref(err))
    words, intact = dst.back()
    tag = f"host encode case {case}: n={n} src+{src.lo % 64} dst+{dst.lo % 64} bad={bad[:3]}"
    if not intact:
        fails.append(tag + ": canary overwritten")
    try:
        want = oracle.encode(s)
    except oracle.OracleError as e:
        # the reference's Vec holds the words of the chunks before the failing one (packing/avx.rs:139-141)
        if st != L.INVALID_BASE or (int(err.byte), int(err.index)) != (e.byte, bad[0]) or got_nw.value != e.words.size \
                or not np.array_equal(words.view(np.uint64)[:got_nw.value], e.words):
            fails.append(tag + f": status {status(st, err)} words {got_nw.value}, oracle byte {e.byte} words {e.words.size}")
        continue
    if st != L.OK or got_nw.value != nw or not np.array_equal(words.view(np.uint64), want):
        fails.append(tag + f": status {status(st, err)} words {got_nw.value}/{nw}")
        continue
    nb = n if rng.random() < 0.6 else int(rng.integers(max(1, 32 * (nw - 1) + 1), 32 * nw + 1))
    extra = int(rng.integers(0, 3))
    wsrc = HostBuf(8 * (nw + extra), 8 * int(rng.integers(0, 8)), np.concatenate([want, np.full(extra, 0xFFFFFFFFFFFFFFFF, np.uint64)]))
    out = HostBuf(nb, int(rng.integers(0, 64)) if case % 5 else 0)
    st = lib.bitnuc_decode(ctx._h, wsrc.ptr, nw + extra, nb, out.ptr, C.byref(err))
    text, intact = out.back()
    if st != L.OK or not intact or not np.array_equal(text, oracle.decode(want, nb)):
        fails.append(f"host decode case {case}: n_bases={nb} words={nw}+{extra} dst+{out.lo % 64}: status {status(st, err)} canaries {intact}")

for case in range(120):
    k = int(rng.integers(1, 33)) if case % 3 else (31, 32, 21, 16)[case % 4]
    r = rng.random()
    stride = k if r < 0.35 else ((1, 2, 4, 8, 16)[rng.integers(5)] if r < 0.6 else (k + int(rng.integers(1, 40)) if r < 0.8 else int(rng.integers(1, 70))))
    n = draw_n(k)
    count = (n - k) // stride + 1
    n = (count - 1) * stride + k
    s = draw_seq(rng, n)
    if stride > k and rng.random() < 0.5:
        m = (np.arange(n) % stride) >= k
        s[m] = np.frombuffer(b"\n>N\x00", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
    bad = []
    if rng.random() < 0.3:
        bad = [p for p in plant(rng, s.copy()) if stride <= k or (p % stride) < k]
        for p in bad:
            s[p] = ord("N")
    src = HostBuf(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
    dst = HostBuf(8 * count, 8 * int(rng.integers(0, 8)))
    err = L.BitnucErr()
    st = lib.bitnuc_as_2bit_batch(ctx._h, src.ptr, k, stride, count, dst.ptr, C.byref(err))
    words, intact = dst.back()
    tag = f"host batch case {case}: k={k} stride={stride} count={count} src+{src.lo % 64} bad={bad[:3]}"
    if not intact:
        fails.append(tag + ": canary overwritten")
    try:
        want = oracle.as_2bit_batch(s, k, stride, count)
    except oracle.OracleError as e:
        if st != L.INVALID_BASE or (int(err.byte), int(err.index)) != (e.byte, e.index):
            fails.append(tag + f": status {status(st, err)}, oracle byte {e.byte} index {e.index}")
        continue
    if st != L.OK or not np.array_equal(words.view(np.uint64), want):
        fails.append(tag + f": status {status(st, err)} or words differ")

for case in range(80):
    k = int(rng.integers(1, 33)) if case % 3 else 31
    n = draw_n(k)
    s = draw_seq(rng, n)
    bad = plant(rng, s) if rng.random() < 0.3 else []
    query = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
    src = HostBuf(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
    dst = HostBuf(n - k + 1, int(rng.integers(0, 64)) if case % 5 else 0)
    err = L.BitnucErr()
    st = lib.bitnuc_kmer_hdist_scan(ctx._h, src.ptr, n, k, C.c_uint64(query), dst.ptr, C.byref(err))
    dist, intact = dst.back()
    tag = f"host scan case {case}: n={n} k={k} src+{src.lo % 64} dst+{dst.lo % 64} bad={bad[:3]}"
    if not intact:
        fails.append(tag + ": canary overwritten")
    try:
        want = oracle.kmer_hdist_scan(s, k, query)
    except oracle.OracleError as e:
        if st != L.INVALID_BASE or (int(err.byte), int(err.index)) != (e.byte, bad[0]):
            fails.append(tag + f": status {status(st, err)}, oracle byte {e.byte}")
        continue
    if st != L.OK or not np.array_equal(dist, want):
        fails.append(tag + f": status {status(st, err)} or distances differ")

for case in range(60):
    Lr = int(rng.integers(1, 700)) if case % 3 else (150, 32, 31, 33, 64, 100, 250, 151)[(case // 3) % 8]
    stride = Lr if rng.random() < 0.5 else Lr + int(rng.integers(1, 9))
    n = draw_n(Lr)
    count = (n - Lr) // stride + 1
    n = (count - 1) * stride + Lr
    s = draw_seq(rng, n)
    if stride > Lr:
        s[(np.arange(n) % stride) >= Lr] = ord("\n")
    wpr = (Lr + 31) // 32
    src = HostBuf(n, int(rng.integers(0, 64)) if case % 4 else 0, s)
    W = HostBuf(8 * wpr * count, 8 * int(rng.integers(0, 8)))
    err = L.BitnucErr()
    st = lib.bitnuc_encode_fixed(ctx._h, src.ptr, Lr, stride, count, W.ptr, C.byref(err))
    words, intact = W.back()
    tag = f"host fixed case {case}: L={Lr} stride={stride} count={count} src+{src.lo % 64}"
    rows = words.view(np.uint64).reshape(count, wpr)
    pick = sorted({0, count - 1, *[int(x) for x in rng.integers(0, count, 60)]})
    if st != L.OK or not intact or any(not np.array_equal(rows[r], oracle.encode(s[r * stride:r * stride + Lr])) for r in pick):
        fails.append(tag + f": encode status {status(st, err)} canaries {intact} or reads differ")
        continue
    D = HostBuf(n, int(rng.integers(0, 64)) if case % 2 else 0, np.full(n, ord("#"), np.uint8))
    Wc = HostBuf(8 * wpr * count, 8 * int(rng.integers(0, 8)), words.view(np.uint64))
    st = lib.bitnuc_decode_fixed(ctx._h, Wc.ptr, Lr, stride, count, D.ptr, C.byref(err))
    text, intact = D.back()
    expect = np.where((np.arange(n) % stride) < Lr, s & 0xDF, ord("#")).astype(np.uint8)
    if st != L.OK or not intact or not np.array_equal(text, expect):
        fails.append(tag + f": decode status {status(st, err)} canaries {intact} or bytes differ (separators must stay)")

for case in range(60):
    nb = draw_n()
    nw = (nb + 31) // 32
    a = rng.integers(0, 1 << 63, nw, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, nw, dtype=np.uint64)
    b = a.copy()
    flips = rng.integers(0, nw, max(1, nw // int(rng.integers(1, 50))))
    b[flips] ^= rng.integers(0, 1 << 63, flips.size, dtype=np.uint64)
    if ctx.hdist(a, b, nb) != oracle.hdist(a, b, nb):
        fails.append(f"host hdist case {case}: n_bases={nb}")
    if ctx.base_counts(a, nb) != oracle.base_counts(a, nb):
        fails.append(f"host base_counts case {case}: n_bases={nb}")
    length = int(rng.integers(1, 33))
    if not np.array_equal(ctx.hdist_pairs(a, b, length), oracle.hdist_pairs(a, b, length)):
        fails.append(f"host hdist_pairs case {case}: words={nw} len={length}")

print("\n".join(fails[:30]))
print(f"host fuzz: {len(fails)} failing cases")
sys.exit(1 if fails else 0)
"""


@pytest.mark.parametrize("engine", ["staged", "direct"])
def test_fuzz_host_pointer_calls(engine):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BITNUC_PIPE_CHUNK_MB="1", BITNUC_PIPE_IMPL=engine)  # both engines of csrc/host_pipe.h
    env.pop("BITNUC_FORCE_GPU", None)
    env.pop("BITNUC_HOST_CUTOFF", None)
    r = subprocess.run([sys.executable, "-c", _HOST_FUZZ_CHILD, root], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and "host fuzz: 0 failing cases" in r.stdout, (r.stdout[-6000:], r.stderr[-3000:])


def test_plan_accessors_and_host_copy_diagnostic(ctx, oracle):
    """The small entry points nothing else calls: a plan's count / total words / device word-offset table, and the staged engine's
    copy-rate diagnostic (argument rule + a positive rate for each of its three modes)."""
    torch = _torch()
    lens = np.array([0, 1, 31, 32, 33, 150, 0, 64, 1000], dtype=np.uint64)
    off = np.zeros(lens.size + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    plan = bn.BatchPlan(ctx, d_off, lens.size)
    lib = ctx._lib
    want_wo = np.zeros(lens.size + 1, dtype=np.uint64)
    want_wo[1:] = np.cumsum((lens + np.uint64(31)) // np.uint64(32))
    assert lib.bitnuc_batch_plan_count(plan._h) == lens.size
    assert lib.bitnuc_batch_plan_total_words(plan._h) == plan.total_words == int(want_wo[-1])
    # the device table of the plan, used as the caller-provided word offsets of the table-driven encode: right words <=> right table
    ptr = plan.word_offsets_ptr
    assert ptr and ptr % 8 == 0
    rng = _rng(5)
    s_host = draw_seq(rng, int(off[-1]))
    d_seq = torch.from_numpy(s_host).cuda()
    out = torch.full((int(want_wo[-1]) + 1,), -1, dtype=torch.int64, device="cuda")
    ctx.encode_batch_dev(d_seq, d_off, ptr, lens.size, int(want_wo[-1]), out)
    ctx.sync()
    want = np.concatenate([oracle.encode(s_host[int(off[i]):int(off[i + 1])]) for i in range(lens.size) if lens[i]])
    got = out.cpu().numpy()
    assert np.array_equal(got[:-1].view(np.uint64), want) and got[-1] == -1
    plan.close()
    assert lib.bitnuc_selftime_host_copy(100, 2, 0) < 0 and lib.bitnuc_selftime_host_copy(1 << 22, 0, 0) < 0 and lib.bitnuc_selftime_host_copy(1 << 22, 2, 3) < 0
    for mode in (0, 1, 2):
        assert lib.bitnuc_selftime_host_copy(8 << 20, 3, mode) > 0.1


def test_host_pipeline_with_pinned_caller_memory(oracle):
    """A caller that hands PINNED buffers to the host-pointer calls gets truly asynchronous copies from the runtime (a pageable copy
    blocks its issuer, a pinned one does not): the direct engine's ordering must then rest on its events and tickets alone.
    150 M bases = 5 chunks of 32 Mi (every device buffer reused), encode and decode, both engines, checked against the oracle on
    slices and as a round trip."""
    import torch
    import bitnuc_amd
    n = 150_000_001
    nw = (n + 31) // 32
    seq_t = torch.empty(n, dtype=torch.uint8).pin_memory()
    words_t = torch.empty(nw, dtype=torch.int64).pin_memory()
    back_t = torch.empty(n, dtype=torch.uint8).pin_memory()
    seq, words, back = seq_t.numpy(), words_t.numpy().view(np.uint64), back_t.numpy()
    rng = _rng(77)
    seq[:] = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, n)]
    c = bitnuc_amd.Context(0)
    for engine in (1, 0):
        c.set_variant("pipe_impl", engine)
        words[:] = 0
        back[:] = 0
        assert c.encode_into(seq, words) == nw
        for lo in (0, 32 * 1_048_576 - 64, 32 * 3_000_000, n - 1000 - (n - 1000) % 32):
            assert np.array_equal(words[lo // 32:(lo + 992) // 32], oracle.encode(seq[lo:lo + 992])), (engine, lo)
        assert np.array_equal(words[-1:], oracle.encode(seq[32 * (nw - 1):]))
        c.decode_into(words, n, back)
        assert np.array_equal(back, seq & 0xDF), engine
        # an invalid byte in the fourth chunk, a later one that must not win
        seq[3 * 33_554_432 + 17] = ord("N")
        seq[4 * 33_554_432 + 5] = ord("X")
        with pytest.raises(bitnuc_amd.NucleotideError) as ei:
            c.encode_into(seq, words)
        assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", ord("N"), 3 * 33_554_432 + 17)
        seq[3 * 33_554_432 + 17] = ord("A")
        seq[4 * 33_554_432 + 5] = ord("C")
    c.close()
