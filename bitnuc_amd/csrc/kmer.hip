// kmer.hip -- k-mer compositions of the hot path behind the C ABI (include/bitnuc_hip.h): batched as_2bit over many
// <= 32-mers (BASELINE config 3, README.md:52-56), every window of a sequence (src/lib.rs:170-173), the sliding pack +
// Hamming scan (config 5: packing/mod.rs:80-110 o hamming/scalar.rs:11-48) and bulk hdist (hamming/multi.rs:121-160).
// Kernels: kmer_device.h; config 5 on the matrix cores: scan_mfma_device.h.
#include "runtime.h"
#include "kmer_device.h"
#include "scan_mfma_device.h"
#include "host_word.h"
#include "host_pipe.h"

using namespace bitnuc_dev;
using namespace bitnuc_rt;

namespace {
hipError_t launch_batch(bitnuc_ctx *c, const uint8_t *kmers, size_t k, size_t stride, size_t count, uint64_t *out,
                        unsigned long long *slot) {
    unsigned long long *o = reinterpret_cast<unsigned long long *>(out);
    size_t done = 0;
    if (stride == k && count >= 64 && knobs(c).batch_dense) {
        // dense layout: whole waves of 64 k-mers go through the bulk-encode-shaped kernel
        const unsigned long long items = count / 64;
        const int un = knobs(c).dense_unroll, kb = knobs(c).kmer_block;
        const unsigned grid = grid_for(c, (items + (kb / 64) * un - 1) / ((kb / 64) * un), kb);
#define DENSE_LAUNCH(AL, NL, NS, U) kmer_dense_kernel<AL, NL, NS, U><<<grid, kb, 0, c->stream>>>(kmers, (unsigned)k, items, o, slot)
#define DENSE_POLICY(U)                                              \
    switch (knobs(c).dense_policy) { /* bit0: nt loads, bit1: nt stores */ \
    case 0: DENSE_LAUNCH(true, false, false, U); break;              \
    case 1: DENSE_LAUNCH(true, true, false, U); break;               \
    case 2: DENSE_LAUNCH(true, false, true, U); break;               \
    default: DENSE_LAUNCH(true, true, true, U); break;               \
    }
        if (!aligned16(kmers)) { DENSE_LAUNCH(false, false, false, 1); }
        else if constexpr (!kEvidenceBuild) { DENSE_LAUNCH(true, true, true, 1); } // the shipped form: dense_policy 3, dense_unroll 1
        else if (un == 1) { DENSE_POLICY(1) }
        else if (un == 2) { DENSE_POLICY(2) }
        else { DENSE_POLICY(4) }
#undef DENSE_POLICY
#undef DENSE_LAUNCH
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = items * 64;
        if (done == count) return hipSuccess;
    }
    if (stride == 1 && done == 0 && knobs(c).batch_slide && knobs(c).slide_impl == 1 && aligned16(kmers) && aligned16(out) && count - 1 + k >= 1056) {
        // every window of a sequence (src/lib.rs:170-173): line-aligned rounds of 1024 windows, computed where they are stored
        const unsigned long long rounds = (count - 1 + k - 32) >> 10; // round r reads bytes [1024 r, 1024 r + 1056)
        const int U = kEvidenceBuild ? knobs(c).slide2_rounds : 4; // the shipped form: 4 rounds per trip (profiles/r03_ab_windows.txt)
        const unsigned grid = grid_for(c, (rounds + (unsigned long long)U * (kBlock / 64) - 1) / ((unsigned long long)U * (kBlock / 64)));
        const bool nts = (knobs(c).dense_policy & 2) != 0;
#define SLIDE2(NT, UU) kmer_slide2_kernel<NT, UU><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot)
        if constexpr (!kEvidenceBuild) { SLIDE2(true, 4); }
        else if (U == 1) { if (nts) SLIDE2(true, 1); else SLIDE2(false, 1); }
        else if (U == 2) { if (nts) SLIDE2(true, 2); else SLIDE2(false, 2); }
        else { if (nts) SLIDE2(true, 4); else SLIDE2(false, 4); }
#undef SLIDE2
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = (size_t)(rounds << 10); // < count: the windows of the last partial KiB follow below
    }
    // (k >= stride: every byte of the span belongs to some k-mer, so validating whole 16-byte groups examines no byte the
    // reference's loop would not; with gaps between k-mers the general kernel looks at each k-mer's own bytes only)
    if ((stride == 1 || stride == 2 || stride == 4 || stride == 8 || stride == 16) && k >= stride && done == 0 && knobs(c).batch_slide &&
        aligned16(kmers) && aligned16(out) && (count - 1) * stride + k >= 1024) {
        // windows at a small power-of-two stride (1 = every window of a sequence): whole 1 KiB wave rounds through the
        // sliding kernel, 992 / stride windows each; the round that would read past the batch's last byte is left over
        const unsigned long long rounds = ((count - 1) * stride + k - 1024) / kScanWaveWindows + 1;
        const unsigned long long per_wave = (unsigned long long)knobs(c).slide_rounds;
        const unsigned grid = grid_for(c, (rounds + per_wave * (kBlock / 64) - 1) / (per_wave * (kBlock / 64)));
        const bool nts = (knobs(c).dense_policy & 2) != 0;
#define SLIDE_U(S, NT) do { if constexpr (kEvidenceBuild) { \
                              if (per_wave == 2) { kmer_slide_kernel<S, NT, 2><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); break; } \
                              if (per_wave == 4) { kmer_slide_kernel<S, NT, 4><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); break; } \
                              if (per_wave == 8) { kmer_slide_kernel<S, NT, 8><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); break; } } \
                            kmer_slide_kernel<S, NT, 1><<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, rounds, o, slot); } while (0)
#define SLIDE(S) do { if (nts) SLIDE_U(S, true); else SLIDE_U(S, false); } while (0)
        switch (stride) {
        case 1: SLIDE(1); break;
        case 2: SLIDE(2); break;
        case 4: SLIDE(4); break;
        case 8: SLIDE(8); break;
        default: SLIDE(16); break;
        }
#undef SLIDE
#undef SLIDE_U
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = (size_t)(rounds * (kScanWaveWindows / stride));
        if (done >= count) return hipSuccess;
    }
    if (stride >= 3 && stride < 32 && k >= stride && done == 0 && knobs(c).batch_slide && aligned16(kmers) &&
        (count - 1) * stride + k >= 1024) {
        // any other small stride with overlapping k-mers: the sliding round with per-lane window selection
        const unsigned long long rounds = ((count - 1) * stride + k - 1024) / kScanWaveWindows + 1;
        const unsigned grid = grid_for(c, (rounds + kBlock / 64 - 1) / (kBlock / 64));
        const unsigned magic = (unsigned)((0x100000000ull + stride - 1) / stride); // exact floor(t / stride) for t < 2^16
        const unsigned long long magic64 = ~0ull / stride + 1; // stride >= 3: no overflow
        kmer_slide_any_kernel<<<grid, kBlock, 0, c->stream>>>(kmers, (unsigned)k, (unsigned)stride, magic, magic64, rounds, o, slot);
        hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        done = (size_t)((rounds * kScanWaveWindows + stride - 1) / stride); // k-mers that start before the last round's end
        if (done >= count) return hipSuccess;
    }
    // general strides, and the < 64 k-mers a dense batch leaves over.  The error slot holds
    // byte offsets relative to `kmers`, so the leftover launch passes the offset it starts at.
    const size_t rest = count - done;
    const unsigned grid = grid_for(c, (rest + kBlock - 1) / kBlock);
    if (stride <= (size_t)kStagedMaxStride)
        kmer_batch_kernel<true><<<grid, kBlock, 0, c->stream>>>(kmers + done * stride, (unsigned)k, stride, rest, o + done, slot, done * stride);
    else
        kmer_batch_kernel<false><<<grid, kBlock, 0, c->stream>>>(kmers + done * stride, (unsigned)k, stride, rest, o + done, slot, done * stride);
    return hipGetLastError();
}

// de-interleave a packed query into its two bit-planes (bit i = low / high code bit of base i)
void query_planes(uint64_t query, size_t k, uint32_t *ql, uint32_t *qh) {
    *ql = *qh = 0;
    for (unsigned i = 0; i < k; ++i) {
        *ql |= (uint32_t)((query >> (2 * i)) & 1) << i;
        *qh |= (uint32_t)((query >> (2 * i + 1)) & 1) << i;
    }
}

#ifdef BITNUC_SWEEP_VARIANTS
// The query's operand of the natural-layout matrix-core scan (evidence/scan_mfma_evidence.h: ScanMfmaTable): per window shift rho and K-step, the nibbles that
// are 1.0 where a channel differs from the query's base (hamming/scalar.rs:33-47 counts the differing 2-bit fields).
// match = true: the nibbles are -1.0 (0b1010) where a channel EQUALS the query's base and the accumulators start at 2^23 + k 2^(8 (r & 3)) (r & 3 = 3: 2^23 + k): the
// product counts the matches down from k -- the same distance with a third of the non-zero entries (one channel of four instead of three).
void scan_mfma_table(uint64_t query, size_t k, ScanMfmaTable *t, bool match = false) {
    uint8_t lo[80], hi[80]; // [16 + i]: channels (A, C) and (G, T) of query position i; zero outside [0, k)
    memset(lo, 0, sizeof lo);
    memset(hi, 0, sizeof hi);
    for (size_t i = 0; i < k; ++i) {
        const unsigned q = (unsigned)((query >> (2 * i)) & 3);
        if (match) {
            lo[16 + i] = (uint8_t)((q == 0 ? 0x0A : 0) | (q == 1 ? 0xA0 : 0));
            hi[16 + i] = (uint8_t)((q == 2 ? 0x0A : 0) | (q == 3 ? 0xA0 : 0));
        } else {
            lo[16 + i] = (uint8_t)((q != 0 ? 0x02 : 0) | (q != 1 ? 0x20 : 0));
            hi[16 + i] = (uint8_t)((q != 2 ? 0x02 : 0) | (q != 3 ? 0x20 : 0));
        }
    }
    for (int j = 0; j < 4; ++j) t->c[j] = 8388608.f + (match ? (float)((unsigned)k << (j == 3 ? 0 : 8 * j)) : 0.f);
    memset(t->w[16], 0, sizeof t->w[16]);
    for (int rho = 0; rho < 16; ++rho)
        for (int s = 0; s < 6; ++s)
            for (int i = 0; i < 4; ++i) {
                const int p0 = 16 * (s >> 1) + 8 * (s & 1) + 4 * (i >> 1); // position of byte 0 of this dword
                const uint8_t *src = (i & 1) ? hi : lo;
                uint32_t w = 0;
                for (int b = 0; b < 4; ++b) w |= (uint32_t)src[16 + p0 + b - rho] << (8 * b);
                t->w[rho][4 * s + i] = w;
            }
}
#endif

// The query's operand of the segment tiling with four channels per base (scan_mfma_device.h: CountMfmaTable; the shipped scan, the four-channel count of the
// evidence build): row m of K-block h only depends on delta = m - 8 h.
// thresholded (kmer_count_mfma_kernel's EMIT 1, 2): result register r (rows with m & 3 = r & 3 = j) must end at 2^23 + (32 + tau - d) 2^(6 j) for j < 3 and at
// 2 d - 2 tau - 1 for j = 3 (scan_mfma_device.h).  match = false: the entries mark the channels that DIFFER from the query's base (-1.0 for j < 3, +1.0 for j = 3)
// and the accumulators start at 2^23 + (32 + tau) 2^(6 j) / -(2 tau + 1).  match = true: they mark the channel that EQUALS it (+1.0 / -1.0: a third of the non-zero
// entries), d = k - matches, and the accumulators start at 2^23 + (32 + tau - k) 2^(6 j) / 2 k - 2 tau - 1.  A threshold no window can miss (tau >= k) gets the
// all-zero table and the start values of tau = k: every field reads 32, every j = 3 result -1.
void count_mfma_table(uint64_t query, size_t k, CountMfmaTable *t, bool thresholded = false, unsigned tau = 0, bool match = false) {
    uint8_t lo[128], hi[128]; // [32 + i]
    memset(lo, 0, sizeof lo);
    memset(hi, 0, sizeof hi);
    const bool all = thresholded && tau >= k;
    for (size_t i = 0; i < k && !all; ++i) {
        const unsigned q = (unsigned)((query >> (2 * i)) & 3);
        if (thresholded && match) {
            lo[32 + i] = (uint8_t)((q == 0 ? 0x02 : 0) | (q == 1 ? 0x20 : 0));
            hi[32 + i] = (uint8_t)((q == 2 ? 0x02 : 0) | (q == 3 ? 0x20 : 0));
        } else {
            lo[32 + i] = (uint8_t)((q != 0 ? 0x02 : 0) | (q != 1 ? 0x20 : 0));
            hi[32 + i] = (uint8_t)((q != 2 ? 0x02 : 0) | (q != 3 ? 0x20 : 0));
        }
    }
    const unsigned te = all ? (unsigned)k : tau; // tau < k <= 32 otherwise
    for (int j = 0; j < 4; ++j) {
        if (!thresholded) t->c[j] = 0.f;
        else if (j < 3) t->c[j] = 8388608.f + (float)((match ? 32u + te - (unsigned)k : 32u + te) << (6 * j));
        else t->c[j] = match ? (float)(2 * (int)k - 2 * (int)te - 1) : -(float)(2 * te + 1);
    }
    if (all) for (int j = 0; j < 3; ++j) t->c[j] = 8388608.f + (float)(32u << (6 * j)), t->c[3] = -1.f;
    for (int d = -8; d < 32; ++d)
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) {
                const int p0 = 16 * j + 4 * (i >> 1); // position of byte 0 of this dword, relative to 32 n + 8 h
                const uint8_t *src = (i & 1) ? hi : lo;
                uint32_t w = 0;
                for (int b = 0; b < 4; ++b) w |= (uint32_t)src[32 + p0 + b - d] << (8 * b);
                // the sign bit of every non-zero nibble (0b0010 -> 0b1010) where the row counts DOWN: j < 3 with differing channels, j = 3 with equal ones
                if (thresholded && ((((d + 8) & 3) != 3) != match)) w |= w << 2;
                t->w[d + 8][4 * j + i] = w;
            }
}

// ... and of the three-channel count (scan_mfma_device.h: Count3MfmaTable): per lane (row m = lane & 31, K-block h = lane >> 5) and K-step, the 32 nibbles that meet
// the lane's operand -- K-steps 0 / 1: the (A, C) bytes of positions 32 s + 16 h + b; K-step 2: the G nibbles of positions 32 h .. + 31, byte b holding positions 8 (b >> 2) + (b & 3) and that + 4.
// d = #(q_i != T) + sum over the window of v(q_i, channel) x[channel], v = -1 on channel q for q in {A, C, G}, +1 on all three for q = T; rows with m & 3 < 3 carry
// -v and start at 2^23 + (32 + tau - #(q_i != T)) 2^(6 j) (they end at 32 + tau - d), rows with m & 3 = 3 carry v at scale 2 and start at 2 #(q_i != T) - 2 tau - 1.
// distance = true (evidence build's three-channel scan): every row carries v and starts at 2^23 + #(q_i != T) 2^(8 j) (j = 3: 2^23 + #): the product is d itself.
void count3_mfma_table(uint64_t query, size_t k, unsigned tau, Count3MfmaTable *t, bool distance = false) {
    const bool all = !distance && tau >= k;
    unsigned non_t = 0;
    for (size_t i = 0; i < k; ++i) non_t += ((query >> (2 * i)) & 3) != 3;
    auto nibble = [&](int m, int p, unsigned ch) -> uint32_t {
        const int i = p - m;
        if (all || i < 0 || i >= (int)k) return 0u;
        const unsigned q = (unsigned)((query >> (2 * i)) & 3);
        const int v = q == 3 ? 1 : (ch == q ? -1 : 0);
        const int e = distance || (m & 3) == 3 ? v : -v;
        return e == 0 ? 0u : e > 0 ? 0x2u : 0xAu;
    };
    for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 31, h = lane >> 5;
        for (int s = 0; s < 3; ++s)
            for (int i = 0; i < 4; ++i) {
                uint32_t w = 0;
                for (int bb = 0; bb < 4; ++bb) {
                    const int b = 4 * i + bb;
                    const int gp = 32 * h + 8 * (b >> 2) + (b & 3); // K-step 2, byte b of the lane's 16: bases gp (low nibble) and gp + 4 (high nibble) of positions 32 h .. + 31
                    const uint32_t lo = s < 2 ? nibble(m, 32 * s + 16 * h + b, 0) : nibble(m, gp, 2);
                    const uint32_t hi = s < 2 ? nibble(m, 32 * s + 16 * h + b, 1) : nibble(m, gp + 4, 2);
                    w |= (lo | hi << 4) << (8 * bb);
                }
                t->w[lane][4 * s + i] = w;
            }
    }
    if (distance) {
        for (int j = 0; j < 4; ++j) t->c[j] = 8388608.f + (float)(non_t << (j == 3 ? 0 : 8 * j));
        return;
    }
    for (int j = 0; j < 3; ++j) t->c[j] = 8388608.f + (float)((all ? 32u : 32u + tau - non_t) << (6 * j));
    t->c[3] = all ? -1.f : (float)(2 * (int)non_t - 2 * (int)tau - 1);
}

// grid of the matrix-core scan: resident waves that walk the rounds (each wave builds its constant operand once)
unsigned scan_mfma_grid(const bitnuc_ctx *c, unsigned long long rounds, int U, bool persist) {
    const unsigned long long want = rounds / ((kBlock / 64) * (unsigned long long)U) + 1; // one trip per wave (+ 1: the tail loop needs a workgroup even without a whole round)
    const unsigned long long cap = persist ? (unsigned long long)c->num_cu * (unsigned)knobs(c).scan_mfma_grid : 0x7FFFFFFFull;
    return (unsigned)(want < cap ? want : cap);
}

// ... and of the fused count in its own tiling: a bounded number of workgroups (each arrives once at the accumulator and the ticket)
unsigned count_mfma_grid(const bitnuc_ctx *c, unsigned long long rounds, int U) {
    const unsigned long long want = rounds / ((kBlock / 64) * (unsigned long long)U) + 1, cap = (unsigned long long)c->num_cu * (unsigned)knobs(c).scan_mfma_count_grid;
    return (unsigned)(want < cap ? want : cap);
}

hipError_t launch_scan(bitnuc_ctx *c, const uint8_t *ref, size_t n, size_t k, uint64_t query, uint8_t *dist,
                       unsigned long long *slot) {
    uint32_t ql, qh;
    query_planes(query, k, &ql, &qh);
    const bool al = aligned16(ref) && aligned16(dist);
    if (knobs(c).scan_impl == 8 && al) {
        // The shipped form: the contraction on the matrix cores in the tiling the fused count introduced -- a column is a segment of 32 windows, a row one of its 32
        // shifts: four MFMAs per 1024 windows -- one trip of 4 rounds per wave (the dispatcher walks the trips), one-hot operands through a wave-private LDS strip,
        // 2^23 bias + row scales for the byte pack, two v_permlane32_swap put the packed dwords in store order, nt loads and stores (scan_mfma_device.h:
        // kmer_scan_seg_mfma_kernel; profiles/r05_ab_scan_seg.txt against the natural-layout tiling's six MFMAs, which shipped first)
        CountMfmaTable ct;
        count_mfma_table(query, k, &ct);
        for (int j = 0; j < 4; ++j) ct.c[j] = 8388608.f;
        const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
        if constexpr (!kEvidenceBuild) {
            // workgroups of ONE wave (the strips are wave-private, nothing is shared inside a workgroup): 19 waves fit a CU's LDS instead of 16, and the dispatcher
            // refills a CU wave by wave (profiles/r05_ab_scan_block.txt: 1 % faster from idle than workgroups of four waves)
            kmer_scan_seg_mfma_kernel<3, 4, 64><<<(unsigned)(rounds / 4 + 1), 64, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, ct);
            return hipGetLastError();
        }
#ifdef BITNUC_SWEEP_VARIANTS
        const int U = knobs(c).scan_mfma_unroll;
        const unsigned grid = scan_mfma_grid(c, rounds, U, false);
        if (knobs(c).scan_mfma_ch3) { // three channels per base: three MFMAs per 1024 windows
            Count3MfmaTable c3;
            count3_mfma_table(query, k, 0u, &c3, true);
            if (U == 2) kmer_scan_seg3_mfma_kernel<3, 2><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, c3);
            else if (U == 3) kmer_scan_seg3_mfma_kernel<3, 3><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, c3);
            else kmer_scan_seg3_mfma_kernel<3, 4><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, c3);
            return hipGetLastError();
        }
        if (knobs(c).scan_mfma_block != kBlock && U == 4 && !knobs(c).scan_mfma_ch3) { // workgroups of one wave (ships) or two
            const int kb = knobs(c).scan_mfma_block;
            const unsigned g2 = (unsigned)(rounds / ((kb / 64) * 4ull) + 1);
            if (kb == 128) kmer_scan_seg_mfma_kernel<3, 4, 128><<<g2, 128, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, ct);
            else kmer_scan_seg_mfma_kernel<3, 4, 64><<<g2, 64, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, ct);
            return hipGetLastError();
        }
        if (U == 2) kmer_scan_seg_mfma_kernel<3, 2><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, ct);
        else if (U == 3) kmer_scan_seg_mfma_kernel<3, 3><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, ct);
        else kmer_scan_seg_mfma_kernel<3, 4><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, dist, slot, ct);
        return hipGetLastError();
#endif
    }
#ifdef BITNUC_SWEEP_VARIANTS
    if (knobs(c).scan_impl == 7 && al) { // the natural-layout tiling (six MFMAs per 1024 windows, results already in store order): round 5's first matrix-core form
        ScanMfmaTable tab;
        scan_mfma_table(query, k, &tab, knobs(c).scan_mfma_match != 0 && knobs(c).scan_mfma_pack == 1);
        const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
        const int U = knobs(c).scan_mfma_unroll, pack = knobs(c).scan_mfma_pack, shift = knobs(c).scan_mfma_shift;
        const bool persist = knobs(c).scan_mfma_persist != 0, ntld = (knobs(c).scan_mfma_policy & 1) != 0;
        const unsigned grid = scan_mfma_grid(c, rounds, U, persist);
#define SCANM(P, UU, PK, SH, PS) kmer_scan_mfma_kernel<P, UU, false, PK, SH, PS><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, 0u, dist, nullptr, nullptr, nullptr, slot, tab)
#define SCANM_PS(P, UU, PK, SH) do { if (persist) SCANM(P, UU, PK, SH, true); else SCANM(P, UU, PK, SH, false); } while (0)
#define SCANM_NT(UU, PK, SH) do { if (ntld) SCANM_PS(3, UU, PK, SH); else SCANM_PS(2, UU, PK, SH); } while (0)
#define SCANM_U(PK, SH) do { if (U == 2) SCANM_NT(2, PK, SH); else if (U == 3 && SH == 4 && PK == 1) SCANM_NT(3, 1, 4); else SCANM_NT(4, PK, SH); } while (0)
        if (shift == 0) { if (pack == 0) SCANM_PS(3, 2, 0, 0); else SCANM_PS(3, 2, 1, 0); }
        else if (shift == 1) { if (pack == 0) SCANM_U(0, 1); else if (pack == 1) SCANM_U(1, 1); else SCANM_U(2, 1); }
        else if (shift == 2) { if (pack == 0) SCANM_U(0, 2); else if (pack == 1) SCANM_U(1, 2); else SCANM_U(2, 2); }
        else if (shift == 3) { if (pack == 0) SCANM_U(0, 3); else if (pack == 1) SCANM_U(1, 3); else SCANM_U(2, 3); }
        else if (shift == 4) { if (pack == 0) SCANM_U(0, 4); else if (pack == 1) SCANM_U(1, 4); else SCANM_U(2, 4); }
        else if (shift == 6) SCANM_U(1, 6);
        else { if (pack == 0) SCANM_U(0, 5); else SCANM_U(1, 5); }
#undef SCANM_U
#undef SCANM_NT
#undef SCANM_PS
#undef SCANM
        return hipGetLastError();
    }
#endif
    const int unroll = knobs(c).scan_unroll, kb = knobs(c).kmer_block;
#ifdef BITNUC_SWEEP_VARIANTS
    if (knobs(c).scan_impl >= 2 && knobs(c).scan_impl <= 5 && al) { // line-aligned rounds, a wave owns consecutive rounds and carries the halo planes (kmer_scan3_kernel)
        const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
        const int impl = knobs(c).scan_impl;
        const int C = impl == 2 ? 12 : impl == 3 ? 20 : impl == 4 ? 16 : 32;
        const unsigned long long waves = (rounds + C - 1) / C;
        const unsigned long long blocks = waves / (kBlock / 64) + 1; // (+ 1: the tail loop needs a workgroup even when there is no whole round)
        const unsigned grid = (unsigned)(blocks < 0x7FFFFFFFull ? blocks : 0x7FFFFFFFull);
#define SCAN3(CC) kmer_scan3_kernel<true, true, 4, CC><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, ql, qh, dist, slot)
        if (C == 12) SCAN3(12); else if (C == 20) SCAN3(20); else if (C == 16) SCAN3(16); else SCAN3(32);
#undef SCAN3
        return hipGetLastError();
    }
#endif
#ifdef BITNUC_SWEEP_VARIANTS
    // rounds 2-4's bit-plane scan (v_alignbit + v_bcnt per window: VALU-issue bound, profiles/r05_ab_scan_mfma*.txt): evidence build
    if (al && knobs(c).scan_impl == 1 && unroll == 4 && knobs(c).scan_policy == 3 && kb == kBlock) { // GEN 1 (two-LUT planes + scalar halo), what round 4 shipped
        const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
        const unsigned grid = grid_for(c, rounds / ((kBlock / 64) * 4) + 1, kBlock);
        kmer_scan2_kernel<true, true, true, 4, false, 1><<<grid, kBlock, 0, c->stream>>>(ref, n, (unsigned)k, query, ql, qh, 0u, dist, nullptr, nullptr, nullptr, slot);
        return hipGetLastError();
    }
#endif
    if ((knobs(c).scan_impl == 1 || knobs(c).scan_impl == 6) && al) { // line-aligned rounds of 1024 windows, round 2-3's plane build (GEN 0): evidence build (6 = that form at the shipped policy)
        const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
        const unsigned grid = grid_for(c, rounds / ((kb / 64) * unroll) + 1, kb);
#define SCAN2(NL, NS, U) kmer_scan2_kernel<true, NL, NS, U, false><<<grid, kb, 0, c->stream>>>(ref, n, (unsigned)k, query, ql, qh, 0u, dist, nullptr, nullptr, nullptr, slot)
#define SCAN2_POLICY(U)                                              \
    switch (knobs(c).scan_policy) { /* bit0: nt loads, bit1: nt stores */ \
    case 0: SCAN2(false, false, U); break;                           \
    case 1: SCAN2(true, false, U); break;                            \
    case 2: SCAN2(false, true, U); break;                            \
    default: SCAN2(true, true, U); break;                            \
    }
        if constexpr (!kEvidenceBuild) { (void)0; } // (the product returned above)
        else if (unroll == 1) { SCAN2_POLICY(1) } else if (unroll == 2) { SCAN2_POLICY(2) } else { SCAN2_POLICY(4) }
#undef SCAN2_POLICY
#undef SCAN2
        return hipGetLastError();
    }
    const unsigned long long rounds = n >= 1024 ? (n - 1024) / kScanWaveWindows + 1 : 0;
    const unsigned grid = grid_for(c, rounds / ((kb / 64) * unroll) + 1, kb);
#define SCAN_LAUNCH(AL, NL, NS, U) \
    kmer_scan_kernel<AL, NL, NS, U><<<grid, kb, 0, c->stream>>>(ref, n, (unsigned)k, query, ql, qh, dist, slot)
#define SCAN_POLICY(U)                                             \
    switch (knobs(c).scan_policy) { /* bit0: nt loads, bit1: nt stores */ \
    case 0: SCAN_LAUNCH(true, false, false, U); break;             \
    case 1: SCAN_LAUNCH(true, true, false, U); break;              \
    case 2: SCAN_LAUNCH(true, false, true, U); break;              \
    default: SCAN_LAUNCH(true, true, true, U); break;              \
    }
    if (!al || !kEvidenceBuild) { // the product reaches this only for unaligned pointers
        SCAN_LAUNCH(false, false, false, 1);
    } else if constexpr (kEvidenceBuild) {
        if (unroll == 1) { SCAN_POLICY(1) } else if (unroll == 2) { SCAN_POLICY(2) } else { SCAN_POLICY(4) }
    }
#undef SCAN_POLICY
#undef SCAN_LAUNCH
    return hipGetLastError();
}
// ---- host-pointer k-mer calls through the pipelined staging of host_pipe.h ---------------------------------------------------
// The drop-in forms of configs 3 and 5 for a caller whose data lives in host memory (README.md:52-56's host loop over as_2bit;
// the window idiom of src/lib.rs:170-173): PCIe-bound, so the point is to keep both DMA engines busy -- round 2's simple path
// (one pageable hipMemcpyAsync in, kernel, one out, host wait, per 128 MiB chunk) left each idle two thirds of the time.
// A chunk is as many k-mers as fit BOTH pinned buffers; the side that moves more bytes per k-mer (8 out against `stride` in) gets
// the large A buffers (chunk + 64), the other the B buffers (chunk / 4 + 64).  Errors: one slot per chunk with the chunk's byte offset as index base, one drain at the end (launch order =
// sequence order), so the first invalid examined byte of the whole call is reported; `out` past it is unspecified.
int batch_pipelined(bitnuc_ctx *c, const uint8_t *kmers, size_t k, size_t stride, size_t count, uint64_t *out, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    PipeAbort guard{c, p};
    // the side that moves more bytes per k-mer gets the large (A) buffers: 8 bytes out against `stride` bytes in
    const bool in_heavy = stride >= 8;
    const size_t in_cap = in_heavy ? p->chunk : p->chunk / 4, out_cap = in_heavy ? p->chunk / 4 : p->chunk;
    size_t per = out_cap / 8;                                             // words that fit the output buffer
    const size_t by_bytes = in_cap > k ? (in_cap - k) / stride + 1 : 1;   // k-mers whose bytes fit the input buffer
    if (by_bytes < per) per = by_bytes;
    if (per >= 1024) per &= ~(size_t)1023; // whole dense items / window rounds per chunk where the chunk is large enough to care
    if (per == 0) per = 1;
    struct Job {
        bitnuc_ctx *c; const uint8_t *kmers; size_t k, stride, count, per; uint64_t *out;
        size_t nchunks; int in_kind, out_kind, in_threads, out_threads;
        size_t items(size_t ci) const { return count - ci * per < per ? count - ci * per : per; }
        const void *in_src(size_t ci) const { return kmers + ci * per * stride; }
        size_t in_bytes(size_t ci) const { return (items(ci) - 1) * stride + k; }
        void *out_dst(size_t ci) const { return out + ci * per; }
        size_t out_bytes(size_t ci) const { return items(ci) * 8; }
        int launch(size_t ci, const uint8_t *d_in, uint8_t *d_out, bitnuc_err *err) const {
            unsigned long long *slot;
            if (int st = take_slot(c, (unsigned long long)(ci * per) * stride, &slot, err)) return st;
            HIPCHK(launch_batch(c, d_in, k, stride, items(ci), reinterpret_cast<uint64_t *>(d_out), slot));
            return BITNUC_OK;
        }
    } job{c, kmers, k, stride, count, per, out, (count + per - 1) / per};
    job.in_kind = in_heavy ? kBufA : kBufB;
    job.out_kind = in_heavy ? kBufB : kBufA;
    job.in_threads = in_heavy ? p->enc_in : p->dec_in;
    job.out_threads = in_heavy ? p->enc_out : p->dec_out;
    if (int st = pipe_run(c, p, job, err)) return st;
    bitnuc_err e;
    const int st = drain(c, &e);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    guard.dismissed = true;
    if (st != BITNUC_OK) { if (err) *err = e; return st; }
    return BITNUC_OK;
}

// windows [off, off + w) of a chunk need bases [off, off + w + k - 1): consecutive chunks overlap by the k - 1 halo bases
int scan_pipelined(bitnuc_ctx *c, const uint8_t *ref, size_t n, size_t k, uint64_t query, uint8_t *dist, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err, (1u << kBufA) | (1u << kBufB) | (1u << kBufC))) return st; // input AND output are a byte per base
    PipeAbort guard{c, p};
    struct Job {
        bitnuc_ctx *c; const uint8_t *ref; size_t nwin, k, per; uint64_t query; uint8_t *dist;
        size_t nchunks; int in_kind = kBufA, out_kind = kBufC, in_threads, out_threads;
        size_t items(size_t ci) const { return nwin - ci * per < per ? nwin - ci * per : per; }
        const void *in_src(size_t ci) const { return ref + ci * per; }
        size_t in_bytes(size_t ci) const { return items(ci) + k - 1; }
        void *out_dst(size_t ci) const { return dist + ci * per; }
        size_t out_bytes(size_t ci) const { return items(ci); }
        int launch(size_t ci, const uint8_t *d_in, uint8_t *d_out, bitnuc_err *err) const {
            unsigned long long *slot;
            if (int st = take_slot(c, ci * per, &slot, err)) return st;
            HIPCHK(launch_scan(c, d_in, in_bytes(ci), k, query, d_out, slot));
            return BITNUC_OK;
        }
    } job{c, ref, n - k + 1, k, p->chunk, query, dist, (n - k + 1 + p->chunk - 1) / p->chunk}; // chunk windows + 31 halo bases fit chunk + 64
    job.in_threads = p->enc_in;  // a byte in and a byte out per window: both sides are heavy
    job.out_threads = p->dec_out;
    if (int st = pipe_run(c, p, job, err)) return st;
    bitnuc_err e;
    const int st = drain(c, &e);
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    guard.dismissed = true;
    if (st != BITNUC_OK) { if (err) *err = e; return st; }
    return BITNUC_OK;
}

} // namespace

extern "C" {

int bitnuc_as_2bit_batch_dev(bitnuc_ctx *c, const uint8_t *d_kmers, size_t k, size_t stride, size_t count, uint64_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0) return BITNUC_OK;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k); // packing/naive.rs:5-7, before any base
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 7) || stride == 0) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (k == 0) { // as_2bit(b"") == Ok(0)
        HIPCHK(hipMemsetAsync(d_out, 0, sizeof(uint64_t) * count, c->stream));
        return BITNUC_OK;
    }
    if (!d_kmers) return fail(err, BITNUC_UNSUPPORTED);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    HIPCHK(launch_batch(c, d_kmers, k, stride, count, d_out, slot));
    return BITNUC_OK;
}

int bitnuc_kmer_hdist_scan_dev(bitnuc_ctx *c, const uint8_t *d_ref, size_t n, size_t k, uint64_t query, uint8_t *d_dist, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (k == 0 || n < k) return BITNUC_OK; // no windows
    if (!d_ref || !d_dist) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    HIPCHK(launch_scan(c, d_ref, n, k, query, d_dist, slot));
    return BITNUC_OK;
}

int bitnuc_kmer_hdist_count_dev(bitnuc_ctx *c, const uint8_t *d_ref, size_t n, size_t k, uint64_t query, unsigned tau, uint64_t *d_count, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (!d_count || (reinterpret_cast<uintptr_t>(d_count) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (k == 0 || n < k) { // no windows
        HIPCHK(hipMemsetAsync(d_count, 0, sizeof(uint64_t), c->stream));
        return BITNUC_OK;
    }
    if (!d_ref) return fail(err, BITNUC_UNSUPPORTED);
    unsigned long long *slot;
    if (int st = take_slot(c, 0, &slot, err)) return st;
    uint32_t ql, qh;
    query_planes(query, k, &ql, &qh);
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
    // a resident grid (the accumulator's ticket needs every workgroup to arrive; 4 trips of 4 rounds per wave keep the tail short)
    const unsigned long long want = rounds / ((kBlock / 64) * 4) + 1, cap = (unsigned long long)c->num_cu * 8;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    unsigned long long *res = reinterpret_cast<unsigned long long *>(d_count);
    if (knobs(c).scan_impl >= 7 && aligned16(d_ref)) {
        if constexpr (!kEvidenceBuild) {
            // The shipped form (scan_mfma_device.h: kmer_count3_mfma_kernel): segments of 32 windows x 32 shifts with THREE channels per base -- three MFMAs per 1024
            // windows --, the threshold inside the product (6-bit fields 32 + tau - d, three rows per register; v_or3 + v_bitop3 + v_bcnt per four windows, nothing on
            // the scalar unit), a bounded grid (one arrival per workgroup at the ticket) whose waves walk trips of four rounds and load the next trip into the registers
            // the current one has just left; 12 workgroups per CU (profiles/r05_ab_count_ch3*.txt; the four-channel form it replaced: r05_ab_count_emit*.txt).
            Count3MfmaTable c3;
            count3_mfma_table(query, k, tau, &c3);
            kmer_count3_mfma_kernel<4, true><<<count_mfma_grid(c, rounds, 4), kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, tau, res, c->d_acc + 5, c->d_tickets + 2, slot, c3);
            HIPCHK(hipGetLastError());
            return BITNUC_OK;
        }
#ifdef BITNUC_SWEEP_VARIANTS
        ScanMfmaTable tab;
        scan_mfma_table(query, k, &tab);
        if (knobs(c).scan_mfma_count_form == 2) { // three channels per base: 3 MFMAs per 1024 windows
            Count3MfmaTable c3;
            count3_mfma_table(query, k, tau, &c3);
            const int CU_ = knobs(c).scan_mfma_count_rounds;
            const unsigned g = count_mfma_grid(c, rounds, CU_);
            if (CU_ == 2) kmer_count3_mfma_kernel<2, true><<<g, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, tau, res, c->d_acc + 5, c->d_tickets + 2, slot, c3);
            else if (CU_ == 3) kmer_count3_mfma_kernel<3, true><<<g, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, tau, res, c->d_acc + 5, c->d_tickets + 2, slot, c3);
            else kmer_count3_mfma_kernel<4, true><<<g, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, tau, res, c->d_acc + 5, c->d_tickets + 2, slot, c3);
            HIPCHK(hipGetLastError());
            return BITNUC_OK;
        }
        if (knobs(c).scan_mfma_count_form == 1) { // the count's own tiling: segments of 32 windows, 4 MFMAs per 1024 windows
            CountMfmaTable ct;
            const int CU_ = knobs(c).scan_mfma_count_rounds, emit = knobs(c).scan_mfma_count_emit;
            count_mfma_table(query, k, &ct, emit != 0, tau, knobs(c).scan_mfma_match != 0);
            const unsigned g = count_mfma_grid(c, rounds, CU_);
#define COUNTOWN(UU, EM) kmer_count_mfma_kernel<UU, true, EM><<<g, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, tau, res, c->d_acc + 5, c->d_tickets + 2, slot, ct)
#define COUNTOWN_E(UU) do { if (emit == 0) COUNTOWN(UU, 0); else if (emit == 1) COUNTOWN(UU, 1); else COUNTOWN(UU, 2); } while (0)
            if (CU_ == 2) COUNTOWN_E(2); else if (CU_ == 3) COUNTOWN_E(3); else COUNTOWN_E(4);
#undef COUNTOWN_E
#undef COUNTOWN
            HIPCHK(hipGetLastError());
            return BITNUC_OK;
        }
        const int U = knobs(c).scan_mfma_unroll, shift = knobs(c).scan_mfma_shift;
        const bool persist = knobs(c).scan_mfma_count_persist != 0; // 0: one trip per wave, every workgroup arrives at the ticket (two atomics per workgroup)
        const unsigned g = scan_mfma_grid(c, rounds, U, persist);
        const bool nt = (knobs(c).scan_mfma_policy & 1) != 0;
        static_assert(kScanPartials == 1024, "runtime.hip allocates 1024 partial accumulators behind d_acc[8]");
#define COUNTM(P, UU, SH, PS) kmer_scan_mfma_kernel<P, UU, true, 0, SH, PS><<<g, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, tau, nullptr, res, PS ? c->d_acc + 5 : c->d_acc + 8, c->d_tickets + 2, slot, tab)
#define COUNTM_PS(P, UU, SH) do { if (persist) COUNTM(P, UU, SH, true); else COUNTM(P, UU, SH, false); } while (0)
#define COUNTM_NT(UU, SH) do { if (nt) COUNTM_PS(1, UU, SH); else COUNTM_PS(0, UU, SH); } while (0)
#define COUNTM_U(SH) do { if (U == 2) COUNTM_NT(2, SH); else COUNTM_NT(4, SH); } while (0)
        if (shift == 0) COUNTM(1, 2, 0, true); else if (shift == 1) COUNTM_U(1); else if (shift == 2) COUNTM_U(2); else if (shift == 3) COUNTM_U(3); else if (shift == 4) COUNTM_U(4); else COUNTM_U(5);
#undef COUNTM_PS
        if (!persist && shift != 0) scan_count_finish_kernel<<<1, kScanPartials, 0, c->stream>>>(c->d_acc + 8, res);
#undef COUNTM_U
#undef COUNTM_NT
#undef COUNTM
        HIPCHK(hipGetLastError());
        return BITNUC_OK;
#endif
    }
#ifdef BITNUC_SWEEP_VARIANTS
    if (aligned16(d_ref)) { // round 4's fused count on the bit-plane scan (0.33 ms per 10^9 windows against the matrix-core form's 0.20): evidence build
        kmer_scan2_kernel<true, true, false, 4, true, 1><<<grid, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, ql, qh, tau, nullptr, res, c->d_acc + 5, c->d_tickets + 2, slot);
        HIPCHK(hipGetLastError());
        return BITNUC_OK;
    }
#endif
    // unaligned reference pointer: the bit-plane scan with unaligned 16-byte loads
    kmer_scan2_kernel<false, false, false, 1, true><<<grid, kBlock, 0, c->stream>>>(d_ref, n, (unsigned)k, query, ql, qh, tau, nullptr, res, c->d_acc + 5, c->d_tickets + 2, slot);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_hdist_dev(bitnuc_ctx *c, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, size_t n_bases, uint32_t *d_result, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    const size_t need = words_for(n_bases);
    if (na < need || nb < need) return fail(err, BITNUC_INVALID_LENGTH, n_bases); // hamming/multi.rs:124-127
    if (!d_result) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (n_bases == 0) {
        HIPCHK(hipMemsetAsync(d_result, 0, sizeof(uint32_t), c->stream));
        return BITNUC_OK;
    }
    if (!d_a || !d_b) return fail(err, BITNUC_UNSUPPORTED);
    const unsigned long long tiles = (n_bases / 32) / (kBlock * 2) + 1;
    const unsigned grid = (unsigned)(tiles < c->hdist_blocks ? tiles : c->hdist_blocks);
    if (knobs(c).hdist_tiled) hdist_kernel<true><<<grid, kBlock, 0, c->stream>>>(reinterpret_cast<const unsigned long long *>(d_a),
                                                 reinterpret_cast<const unsigned long long *>(d_b), n_bases, d_result, reinterpret_cast<unsigned *>(c->d_acc + 4), c->d_tickets + 1);
    else hdist_kernel<false><<<grid, kBlock, 0, c->stream>>>(reinterpret_cast<const unsigned long long *>(d_a),
                                                 reinterpret_cast<const unsigned long long *>(d_b), n_bases, d_result, reinterpret_cast<unsigned *>(c->d_acc + 4), c->d_tickets + 1);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}
int bitnuc_as_2bit_batch(bitnuc_ctx *c, const uint8_t *kmers, size_t k, size_t stride, size_t count, uint64_t *out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (count == 0) return BITNUC_OK;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (!out || stride == 0) return fail(err, BITNUC_UNSUPPORTED);
    if (k == 0) { memset(out, 0, sizeof(uint64_t) * count); return BITNUC_OK; }
    if (!kmers) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    if (c->host_pipeline && (count - 1) * stride + k >= kPipeMin && stride <= ((size_t)1 << 20)) return batch_pipelined(c, kmers, k, stride, count, out, err);
    // chunk by k-mers so a staged chunk stays <= kHostChunk bytes
    size_t per = kHostChunk / stride;
    if (per == 0) per = 1;
    if (per > count) per = count;
    if (int st = ensure_scratch(c, 0, (per - 1) * stride + k + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, per * 8, err)) return st;
    for (size_t j0 = 0; j0 < count; j0 += per) {
        const size_t m = count - j0 < per ? count - j0 : per;
        const size_t bytes = (m - 1) * stride + k;
        HIPCHK(hipMemcpyAsync(c->scratch[0], kmers + j0 * stride, bytes, hipMemcpyHostToDevice, c->stream));
        unsigned long long *slot;
        if (int st = take_slot(c, (unsigned long long)j0 * stride, &slot, err)) return st;
        HIPCHK(launch_batch(c, c->scratch[0], k, stride, m, reinterpret_cast<uint64_t *>(c->scratch[1]), slot));
        HIPCHK(hipMemcpyAsync(out + j0, c->scratch[1], m * 8, hipMemcpyDeviceToHost, c->stream));
        bitnuc_err e;
        int st = drain(c, &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
    }
    return BITNUC_OK;
}

int bitnuc_kmer_hdist_scan(bitnuc_ctx *c, const uint8_t *ref, size_t n, size_t k, uint64_t query, uint8_t *dist, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (k > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, k);
    if (k == 0 || n < k) return BITNUC_OK;
    if (!ref || !dist) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    if (c->host_pipeline && n >= kPipeMin) return scan_pipelined(c, ref, n, k, query, dist, err);
    const size_t nwin = n - k + 1;
    const size_t chunk = nwin < kHostChunk ? nwin : kHostChunk; // windows per staged chunk
    if (int st = ensure_scratch(c, 0, chunk + k + 16, err)) return st;
    if (int st = ensure_scratch(c, 2, chunk + 16, err)) return st;
    for (size_t off = 0; off < nwin; off += chunk) {
        const size_t w = nwin - off < chunk ? nwin - off : chunk;
        const size_t bytes = w + k - 1;
        HIPCHK(hipMemcpyAsync(c->scratch[0], ref + off, bytes, hipMemcpyHostToDevice, c->stream));
        unsigned long long *slot;
        if (int st = take_slot(c, off, &slot, err)) return st;
        HIPCHK(launch_scan(c, c->scratch[0], bytes, k, query, c->scratch[2], slot));
        HIPCHK(hipMemcpyAsync(dist + off, c->scratch[2], w, hipMemcpyDeviceToHost, c->stream));
        bitnuc_err e;
        int st = drain(c, &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
    }
    return BITNUC_OK;
}

int bitnuc_hdist(bitnuc_ctx *c, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, size_t n_bases, uint32_t *out, bitnuc_err *err) {
    clear_err(err);
    const size_t need = words_for(n_bases);
    if (na < need || nb < need) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    if (n_bases == 0) { *out = 0; return BITNUC_OK; }
    if (!a || !b) return fail(err, BITNUC_UNSUPPORTED);
    if (on_host(c, n_bases)) { *out = bitnuc_host::hdist_small(a, b, n_bases); return BITNUC_OK; }
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    const size_t chunk_words = kHostChunk / 8;
    const size_t cw = need < chunk_words ? need : chunk_words;
    if (int st = ensure_scratch(c, 0, cw * 8, err)) return st;
    if (int st = ensure_scratch(c, 1, cw * 8, err)) return st;
    if (int st = ensure_scratch(c, 2, 64, err)) return st;
    uint32_t total = 0;
    for (size_t w0 = 0; w0 < need; w0 += cw) {
        const size_t m = need - w0 < cw ? need - w0 : cw;
        const size_t bases = (w0 + m == need) ? n_bases - w0 * 32 : m * 32;
        HIPCHK(hipMemcpyAsync(c->scratch[0], a + w0, m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->scratch[1], b + w0, m * 8, hipMemcpyHostToDevice, c->stream));
        bitnuc_err e;
        int st = bitnuc_hdist_dev(c, reinterpret_cast<const uint64_t *>(c->scratch[0]), m,
                                  reinterpret_cast<const uint64_t *>(c->scratch[1]), m, bases,
                                  reinterpret_cast<uint32_t *>(c->scratch[2]), &e);
        if (st != BITNUC_OK) { if (err) *err = e; return st; }
        uint32_t part = 0;
        HIPCHK(hipMemcpyAsync(&part, c->scratch[2], 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        total += part; // u32 wrap-around like the reference's accumulator (multi.rs:130)
    }
    *out = total;
    return BITNUC_OK;
}

} // extern "C"
