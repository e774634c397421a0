// kmer_device.h -- gfx950 device code for the k-mer compositions of the hot path:
//   * batched as_2bit over many <=32-mers at a byte stride   (BASELINE config 3)
//   * sliding-window k-mer pack + Hamming distance to a query (BASELINE config 5): the bit-plane forms of rounds 1-4 -- since round 5 the
//     shipped form runs on the matrix cores (scan_mfma_device.h); what stays in use from here is the unaligned-pointer path
//   * bulk packed-vs-packed Hamming distance (hdist)
//
// Values follow src/utils/packing/naive.rs:8-18 (pack) and
// src/utils/functions/hamming/scalar.rs:22-47 (distance of two packed words).
//
// WHAT SHIPS (the product library instantiates only these):
//   kmer_dense_kernel<nt, nt, XCD, 1>                      stride == k batches (config 3)
//   kmer_batch_kernel<STAGED>                              any other stride, leftovers, unaligned pointers
//   kmer_slide2_kernel<nt, 4>, kmer_slide_kernel<S>, kmer_slide_any_kernel     every window / small strides -> u64
//   kmer_scan_kernel<unaligned>, kmer_scan2_kernel<unaligned, COUNT>    config 5 and its fused count for UNALIGNED pointers only (aligned: scan_mfma_device.h)
//   hdist_kernel, nucgen_kernel
// EVIDENCE BUILD ONLY (-DBITNUC_SWEEP_VARIANTS; csrc/evidence/kmer_evidence.h and the other instantiations of the templates below): kmer_scan_kernel's aligned policies (rounds of 992 windows), kmer_scan2_kernel
// aligned (GEN 1: what round 4 shipped; GEN 0 and its other policies / trip lengths), kmer_scan3_kernel (round 4: a wave owns consecutive rounds), kmer_slide_kernel<1> with rounds per
// trip, the other dense unrolls / policies -- profiles/NARRATIVE_r01_r03.md 3.3-3.4, profiles/README.md.
#pragma once
#include "device_prims.h"

namespace bitnuc_dev {

// 2-bit-field mismatch count of two 32-bit halves (scalar.rs:33-47 on a half word):
// (x | x>>1) & 0x5555.. has one bit per differing base.
__device__ __forceinline__ uint32_t mismatch_bits(uint32_t x, uint32_t evenmask) { return (x | (x >> 1)) & evenmask; }

// ---------------------------------------------------------------------------------
// batched as_2bit
// ---------------------------------------------------------------------------------
// One lane per k-mer.  The lane's k bytes start at an arbitrary byte address, so it
// reads the <=9 ALIGNED dwords that cover them (an aligned dword holding one valid
// byte never crosses a page) and funnel-shifts with v_alignbyte_b32.  With STAGED the
// block first copies its contiguous byte span into LDS with coalesced 16-byte loads
// (dense strides: every HBM byte is fetched once, by a full-width access) and the
// per-lane dwords come from LDS; otherwise they come straight from global memory.
constexpr int kStagedMaxStride = 64;
constexpr int kStageBytes = (kBlock - 1) * kStagedMaxStride + 32 + 32; // span + alignment slack, multiple of 16

template <bool STAGED>
__global__ void __launch_bounds__(kBlock)
kmer_batch_kernel(const uint8_t *__restrict__ kmers, unsigned k, unsigned long long stride,
                  unsigned long long count, unsigned long long *__restrict__ out,
                  unsigned long long *__restrict__ slot, unsigned long long index_base) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGED ? kStageBytes : 16];
    const unsigned t = threadIdx.x;
    const unsigned nfull = k >> 2, rem = k & 3;
    const unsigned long long nblk = (count + kBlock - 1) / kBlock;

    for (unsigned long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const unsigned long long j0 = blk * kBlock;
        const unsigned long long j = j0 + t;
        const bool active = j < count;
        uint32_t a[10];
        unsigned sh;
        if constexpr (STAGED) {
            const unsigned long long jl = (j0 + kBlock <= count ? j0 + kBlock : count) - 1; // last k-mer of the block
            const uintptr_t lo = reinterpret_cast<uintptr_t>(kmers) + j0 * stride;
            const uintptr_t hi = reinterpret_cast<uintptr_t>(kmers) + jl * stride + k; // one past the last byte
            const uintptr_t lo16 = lo & ~(uintptr_t)15;
            const unsigned nchunk = k ? (unsigned)((hi - lo16 + 15) >> 4) : 0;
            __syncthreads(); // previous iteration's readers are done
            for (unsigned c = t; c < nchunk; c += kBlock)
                *reinterpret_cast<u32x4 *>(stage + 16 * c) = *reinterpret_cast<const u32x4 *>(lo16 + 16 * (uintptr_t)c);
            __syncthreads();
            const unsigned off = (unsigned)(lo - lo16) + (unsigned)((j - j0) * stride);
            sh = off & 3;
            const unsigned nd = (sh + k + 3) >> 2;
            const uint32_t *base = reinterpret_cast<const uint32_t *>(stage + (off & ~3u));
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = (active && (unsigned)i < nd) ? base[i] : 0u;
        } else {
            const uintptr_t p = reinterpret_cast<uintptr_t>(kmers) + j * stride;
            sh = (unsigned)(p & 3);
            const unsigned nd = (sh + k + 3) >> 2;
            const uint32_t *base = reinterpret_cast<const uint32_t *>(p & ~(uintptr_t)3);
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = (active && (unsigned)i < nd) ? base[i] : 0u;
        }
        a[9] = 0;
        if (!active) continue;

        uint32_t x[8];
        uint32_t bad = 0, lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t w = __builtin_amdgcn_alignbyte(a[i + 1], a[i], sh);
            if ((unsigned)i == nfull) { // partial dword: pad the bytes past k with 'A' (code 0, valid)
                const uint32_t keep = rem ? ((1u << (8 * rem)) - 1u) : 0u;
                w = (w & keep) | (0x41414141u & ~keep);
            } else if ((unsigned)i > nfull) {
                w = 0x41414141u;
            }
            x[i] = w;
            const uint32_t r = enc4(w, bad);
            if (i < 4) lo |= r << (8 * i); else hi |= r << (8 * (i - 4));
        }
        if (__builtin_expect(residue_is_bad(bad), 0)) {
            for (unsigned b = 0; b < k; ++b) {
                const uint32_t byte = (x[b >> 2] >> (8 * (b & 3))) & 0xFFu;
                if (!valid_base(byte)) { latch_bad(slot, index_base + j * stride + b, byte); break; }
            }
        }
        out[j] = ((unsigned long long)hi << 32) | lo;
    }
}

// ---------------------------------------------------------------------------------
// batched as_2bit, dense layout (stride == k): BASELINE config 3
// ---------------------------------------------------------------------------------
// With no gap between k-mers the batch IS a bulk encode whose 2-bit stream is cut every
// 2k bits instead of every 64: a wave takes 64 k-mers = 64k bytes = 4k whole 16-byte
// groups (so every wave's span stays 16-byte aligned), lanes load them with coalesced
// dwordx4 (1 or 2 per lane), pack each group to a u32 of codes into a wave-private LDS
// strip, and lane j then funnel-shifts its 2k bits out of strip dwords (2kj)>>5 .. +2.
// 4x less LDS traffic and ~half the VALU work of staging raw bytes (kmer_batch_kernel).
// Handles whole waves only; the host sends the < 64 leftover k-mers to kmer_batch_kernel.
template <bool ALIGNED, bool NTLD, bool NTST, int UNROLL>
__global__ void __launch_bounds__(kBlock)
kmer_dense_kernel(const uint8_t *__restrict__ kmers, unsigned k, unsigned long long nwave_items,
                  unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    __shared__ uint32_t strips[kBlock / 64][UNROLL][132];
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block(); // blockDim <= kBlock
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const unsigned ngroups = 4 * k; // 16-byte groups per 64 k-mers
    const unsigned bit = 2 * k * lane, d = bit >> 5, sh = bit & 31;
    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const u32x4 pad = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
    const bool a0 = lane < ngroups, a1 = lane + 64 < ngroups;

    // a wave owns UNROLL consecutive items (64 k-mers each) per trip; all loads are issued first
    for (unsigned long long it0 = wave * UNROLL; it0 < nwave_items; it0 += nwaves * UNROLL) {
        u32x4 v0[UNROLL], v1[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned long long it = it0 + u < nwave_items ? it0 + u : nwave_items - 1; // clamp: in bounds
            const uint8_t *base = kmers + it * 64ull * k;
            v0[u] = a0 ? load_group<NTLD, ALIGNED>(base + 16 * lane) : pad;
            v1[u] = a1 ? load_group<NTLD, ALIGNED>(base + 16 * (lane + 64)) : pad;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (it0 + u >= nwave_items) break; // wave-uniform
            const unsigned long long it = it0 + u;
            uint32_t *ws = strips[wave_in_block()][u];
            uint32_t bad = 0;
            ws[lane] = enc16(v0[u], bad);
            ws[lane + 64] = enc16(v1[u], bad);
            if (lane < 4) ws[128 + lane] = 0u; // k = 32: lane 63's funnel reads ws[128] (masked off by kmask, but no read of unwritten LDS)
            wave_lds_fence();
            const uint32_t w0 = ws[d], w1 = ws[d + 1], w2 = ws[d + 2];
            wave_lds_fence();
            const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
            const unsigned long long word = (((unsigned long long)hi << 32) | lo) & kmask;
            if constexpr (NTST) __builtin_nontemporal_store(word, out + it * 64 + lane);
            else out[it * 64 + lane] = word;
            if (__builtin_expect(residue_is_bad(bad), 0)) {
                if (a0) rescan_bytes(kmers, it * 64ull * k + 16 * lane, 16, slot);
                if (a1) rescan_bytes(kmers, it * 64ull * k + 16 * (lane + 64), 16, slot);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// sliding-window k-mer pack + Hamming distance scan: BASELINE config 5
// ---------------------------------------------------------------------------------
// dist[i] = hdist_scalar(as_2bit(ref[i..i+k]), query, k).  The distance of two packed
// words is the number of 2-bit fields that differ (hamming/scalar.rs:33-47); with the
// codes split into two bit-planes (L = low code bits, H = high code bits, one bit per
// base) that is popcount(((L>>i ^ qL) | (H>>i ^ qH)) & ones(k)) -- all 32-bit ops.
//
// Fast path: a wave reads 1024 consecutive bytes (lane l: the dwordx4 at 16l, coalesced)
// and produces the 992 windows that start in its first 62 lanes; lanes 62/63 only supply
// the 30-base halo, so there is no separate halo load and consecutive waves overlap by
// 32 bytes (3 % re-read, served by L2).  Per lane: 16 bases -> two 16-bit planes
// (v_perm LUT + validity residue as in enc4, v_dot4 with weights {1,2,4,8} gathers one bit
// per byte, chained over two dwords), planes of lanes l+1, l+2 arrive by two lane shifts, window j is a
// v_alignbit_b32 of each plane pair by j.  16 distance bytes per lane, one dwordx4 store.
// Needs bytes [wb, wb+1024) in bounds; the leftover windows go through the tail loop.
constexpr unsigned kScanWaveWindows = 992;

// lane i <- lane i+1 across the whole wave64 (gfx9 DPP wave_shl:1); lane 63 gets 0
__device__ __forceinline__ uint32_t wave_shl1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}

// 8 bases (two dwords) -> 8 low-plane bits and 2 x 8 high-plane bits: the second v_dot4 carries
// weights {16,32,64,128} and accumulates onto the first, so no shift/or is needed to join nibbles.
__device__ __forceinline__ void planes8(uint32_t x0, uint32_t x1, uint32_t &bad, uint32_t &l8, uint32_t &h8x2) {
    const uint32_t t0 = __builtin_amdgcn_perm(0x42040453u, 0x41044004u, x0 & 0x07070707u);
    const uint32_t t1 = __builtin_amdgcn_perm(0x42040453u, 0x41044004u, x1 & 0x07070707u);
    const uint32_t d0 = t0 ^ (x0 & 0xD8D8D8D8u), d1 = t1 ^ (x1 & 0xD8D8D8D8u);
    bad |= d0 | d1;
    l8 = __builtin_amdgcn_udot4(d1 & 0x01010101u, 0x80402010u, __builtin_amdgcn_udot4(d0 & 0x01010101u, 0x08040201u, 0u, false), false);
    h8x2 = __builtin_amdgcn_udot4(d1 & 0x02020202u, 0x80402010u, __builtin_amdgcn_udot4(d0 & 0x02020202u, 0x08040201u, 0u, false), false);
}

// Round 4 (the scan is VALU-issue bound and the chip lowers its clock under such a kernel, DESIGN.md 3.4: every instruction less is time): the
// plane build without the two v_and per dword -- a second 8-entry v_perm LUT yields the HIGH code bit as a byte of its own (0 / 1), the first
// LUT's byte keeps only the LOW code bit next to the validity residue, so both v_dot4 gathers take their operand as it is.
__device__ __forceinline__ void planes8_2lut(uint32_t x0, uint32_t x1, uint32_t &bad, uint32_t &l8, uint32_t &h8) {
    const uint32_t s0 = x0 & 0x07070707u, s1 = x1 & 0x07070707u; // A=1 C=3 T=4 G=7, case bit ignored
    // low code bit | the bits every base shares (0x40; T: 0x50); 0x04 for the four indices no base has
    const uint32_t tl0 = __builtin_amdgcn_perm(0x40040451u, 0x41044004u, s0), tl1 = __builtin_amdgcn_perm(0x40040451u, 0x41044004u, s1);
    // high code bit alone (G, T)
    const uint32_t h0 = __builtin_amdgcn_perm(0x01000001u, 0x00000000u, s0), h1 = __builtin_amdgcn_perm(0x01000001u, 0x00000000u, s1);
    const uint32_t d0 = tl0 ^ (x0 & 0xD8D8D8D8u), d1 = tl1 ^ (x1 & 0xD8D8D8D8u); // == the low code bit (0 / 1) iff the byte is in ACGTacgt
    bad |= d0 | d1;
    l8 = __builtin_amdgcn_udot4(d1, 0x80402010u, __builtin_amdgcn_udot4(d0, 0x08040201u, 0u, false), false);
    h8 = __builtin_amdgcn_udot4(h1, 0x80402010u, __builtin_amdgcn_udot4(h0, 0x08040201u, 0u, false), false);
}
__device__ __forceinline__ uint32_t planes16_2lut(const u32x4 &v, uint32_t &bad) { // low half: L plane, high half: H plane (16 bases)
    uint32_t la, lb, ha, hb;
    planes8_2lut(v.x, v.y, bad, la, ha);
    planes8_2lut(v.z, v.w, bad, lb, hb);
    return (la | (lb << 8)) | ((ha | (hb << 8)) << 16);
}

// The same 16-base planes word from four WAVE-UNIFORM dwords, on the scalar unit: the code of a byte is ((b >> 1) ^ (b >> 2)) & 3
// (device_prims.h code_of; no validity here: halo bytes are validated by the round that owns them), and a multiply gathers bit 0 of the
// four bytes into the top nibble (the partial products land on distinct bits, nothing carries).
__device__ __forceinline__ uint32_t planes16_uniform(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    uint32_t L = 0, H = 0;
    const uint32_t x[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = (x[i] >> 1) ^ (x[i] >> 2);
        L |= (((c & 0x01010101u) * 0x10204080u) >> 28) << (4 * i);
        H |= ((((c >> 1) & 0x01010101u) * 0x10204080u) >> 28) << (4 * i);
    }
    return L | (H << 16);
}

template <bool ALIGNED, bool NTLD, bool NTST, int UNROLL>
__global__ void __launch_bounds__(kBlock)
kmer_scan_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query,
                 uint32_t ql, uint32_t qh, uint8_t *__restrict__ dist, unsigned long long *__restrict__ slot) {
    const unsigned long long nwin = n - k + 1; // host guarantees 1 <= k <= 32, n >= k
    const unsigned long long rounds = n >= 1024 ? (n - 1024) / kScanWaveWindows + 1 : 0;
    const unsigned lane = threadIdx.x & 63;
    // scalar wave index: the per-round 64-bit address arithmetic becomes scalar ops + one constant lane offset
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block(); // blockDim <= kBlock
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    // ql / qh: the query's bit-planes (bit i = low / high code bit of base i), split on the host
    const uint32_t km = k == 32 ? ~0u : ((1u << k) - 1u);

    // a wave owns UNROLL consecutive rounds per trip; all their loads are issued first
    for (unsigned long long r0 = wave * UNROLL; r0 < rounds; r0 += nwaves * UNROLL) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned long long r = r0 + u < rounds ? r0 + u : rounds - 1; // clamp: redundant but in bounds
            v[u] = load_group<NTLD, ALIGNED>(ref + r * kScanWaveWindows + 16 * lane);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (r0 + u >= rounds) break; // wave-uniform
            const unsigned long long wb = (r0 + u) * kScanWaveWindows;
            uint32_t bad = 0, la, lb, ha, hb;
            planes8(v[u].x, v[u].y, bad, la, ha);
            planes8(v[u].z, v[u].w, bad, lb, hb);
            const uint32_t pl = (la | (lb << 8)) | ((ha | (hb << 8)) << 15); // low half: L plane, high half: H plane (16 bases)
            if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(ref, wb + 16 * lane, 16, slot);
            // planes of lanes l+1 and l+2: whole-wave DPP shifts (v_mov_b32 wave_shl:1), no LDS round trip
            const uint32_t n1 = wave_shl1(pl), n2 = wave_shl1(n1);
            const uint32_t Llo = __builtin_amdgcn_perm(n1, pl, 0x05040100u); // bases 0..31 of this lane's run
            const uint32_t Hlo = __builtin_amdgcn_perm(n1, pl, 0x07060302u);
            const uint32_t Lhi = n2 & 0xFFFFu, Hhi = n2 >> 16;               // bases 32..47
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t acc = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = 4 * q + b;
                    const uint32_t l = j ? __builtin_amdgcn_alignbit(Lhi, Llo, j) : Llo;
                    const uint32_t h = j ? __builtin_amdgcn_alignbit(Hhi, Hlo, j) : Hlo;
                    const uint32_t dcount = __builtin_popcount(((l ^ ql) | (h ^ qh)) & km);
                    acc |= dcount << (8 * b);
                }
                o[q] = acc;
            }
            if (lane < 62) {
                const u32x4 ov = {o[0], o[1], o[2], o[3]};
                store_group<NTST, ALIGNED>(dist + wb + 16 * lane, ov);
            }
        }
    }

    // tail: one window per thread, byte loads
    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const unsigned long long gt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = rounds * kScanWaveWindows + gt; i < nwin; i += nthreads) {
        unsigned long long w = 0;
        bool flagged = false;
        for (unsigned b = 0; b < k; ++b) {
            const uint32_t byte = ref[i + b];
            if (!valid_base(byte) && !flagged) { latch_bad(slot, i + b, byte); flagged = true; }
            w |= (unsigned long long)code_of(byte) << (2 * b);
        }
        const unsigned long long x = (w ^ query) & kmask;
        dist[i] = (uint8_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
}

// ---------------------------------------------------------------------------------
// scan, line-aligned rounds (round 2) + the fused `d <= tau` count
// ---------------------------------------------------------------------------------
// kmer_scan_kernel's rounds advance by 992 windows, so every wave-store of 992 distance bytes starts and ends in the
// middle of a 128-byte line that the neighbouring wave also writes (and its 1 KiB load overlaps the next one by 32 B).
// The decode kernels showed what such shared lines cost (batch_device.h, strip_drain).  Here a round is 1024 windows at a
// 1024-byte aligned offset: all 64 lanes produce 16 windows, the store is eight whole lines, loads do not overlap.  The
// 30-base halo of lanes 62/63 is the first two chunks of the NEXT round: inside a trip of UNROLL consecutive rounds they
// are already in registers (two v_readlane + two v_and_or per round), and the trip's last round gets them from one
// extra 16-byte load in lanes 0 and 1.
// COUNT: SURVEY 8(d) cfg 5's optional fused output -- only the number of windows with d <= tau leaves the chip
// (1 B read per window instead of 2 B moved): per window a compare-and-add instead of the byte pack, one u64 atomic per
// workgroup, published by the last workgroup (single launch, accumulator zero between launches, as hdist_kernel).
// GEN = 1 (round 4): the same tiling with fewer vector instructions -- the kernel is VALU-issue bound.  (a) planes16_2lut
// instead of planes8: no v_and pair per dword (a second v_perm LUT yields the high code bit as a byte of its own).  (b) The halo of the
// trip's last round is 32 bytes at a wave-uniform address: its planes are computed on the SCALAR unit (SWAR code formula + a multiply
// gather per dword) instead of by a fifth whole-wave plane build of which two lanes were used.
template <bool ALIGNED, bool NTLD, bool NTST, int UNROLL, bool COUNT, int GEN = 0>
__global__ void __launch_bounds__(kBlock)
kmer_scan2_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query,
                  uint32_t ql, uint32_t qh, unsigned tau, uint8_t *__restrict__ dist, unsigned long long *__restrict__ result,
                  unsigned long long *__restrict__ total /* zero between launches */, unsigned *__restrict__ ticket,
                  unsigned long long *__restrict__ slot) {
    const unsigned long long nwin = n - k + 1;                              // host guarantees 1 <= k <= 32, n >= k
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;       // round r reads bytes [1024 r, 1024 r + 1056)
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const uint32_t km = k == 32 ? ~0u : ((1u << k) - 1u);
    const uint32_t m63 = lane == 63 ? ~0u : 0u;
    uint32_t hits = 0;

    for (unsigned long long r0 = wave * UNROLL; r0 < rounds; r0 += nwaves * UNROLL) {
        const unsigned m = rounds - r0 < (unsigned long long)UNROLL ? (unsigned)(rounds - r0) : (unsigned)UNROLL; // valid rounds in this trip (wave-uniform)
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned long long r = r0 + u < rounds ? r0 + u : rounds - 1;
            v[u] = load_group<NTLD, ALIGNED>(ref + (r << 10) + 16 * lane);
        }
        u32x4 hv = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
        uint32_t hs0 = 0, hs1 = 0; // GEN 1: the halo's planes, wave-uniform
        if constexpr (GEN == 0) {
            if (lane < 2) hv = load_group<false, ALIGNED>(ref + ((r0 + m) << 10) + 16 * lane); // the halo of the trip's last round
        } else {
            static_assert(GEN == 0 || ALIGNED, "the scalar halo reads dwords");
            const uint32_t *hp = reinterpret_cast<const uint32_t *>(ref + ((r0 + m) << 10));
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hp[i]);
            hs0 = planes16_uniform(w[0], w[1], w[2], w[3]);
            hs1 = planes16_uniform(w[4], w[5], w[6], w[7]);
        }
        uint32_t pl[UNROLL + 1];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            uint32_t bad = 0;
            if constexpr (GEN == 0) {
                uint32_t la, lb, ha, hb;
                planes8(v[u].x, v[u].y, bad, la, ha);
                planes8(v[u].z, v[u].w, bad, lb, hb);
                pl[u] = (la | (lb << 8)) | ((ha | (hb << 8)) << 15);
            } else pl[u] = planes16_2lut(v[u], bad);
            if (__builtin_expect(residue_is_bad(bad) && (unsigned)u < m, 0)) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
        }
        if constexpr (GEN == 0) {
            uint32_t bad = 0, la, lb, ha, hb; // halo bytes are validated by the round (or tail) that owns them
            planes8(hv.x, hv.y, bad, la, ha);
            planes8(hv.z, hv.w, bad, lb, hb);
            pl[UNROLL] = (la | (lb << 8)) | ((ha | (hb << 8)) << 15);
        } else pl[UNROLL] = 0;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if ((unsigned)u >= m) break; // wave-uniform
            const bool inner = (unsigned)(u + 1) < m; // the next KiB is one of this trip's rounds (wave-uniform)
            const uint32_t nx = inner ? pl[u + 1] : pl[UNROLL];
            uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)nx, 0), h1 = (uint32_t)__builtin_amdgcn_readlane((int)nx, 1);
            if constexpr (GEN != 0) { h0 = inner ? h0 : hs0; h1 = inner ? h1 : hs1; }
            // lane 63 of a wave_shl is 0 (bound_ctrl), so the halo goes in with one v_and_or.  NOT `lane == 63 ? h : shl(x)`: hipcc turns
            // that select into a branch and runs the DPP move with lane 63 masked off, and a DPP read from a disabled lane returns 0.
            const uint32_t n1 = wave_shl1(pl[u]) | (h0 & m63);
            const uint32_t n2 = wave_shl1(n1) | (h1 & m63);
            const uint32_t Llo = __builtin_amdgcn_perm(n1, pl[u], 0x05040100u);
            const uint32_t Hlo = __builtin_amdgcn_perm(n1, pl[u], 0x07060302u);
            const uint32_t Lhi = n2 & 0xFFFFu, Hhi = n2 >> 16;
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t acc = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = 4 * q + b;
                    const uint32_t l = j ? __builtin_amdgcn_alignbit(Lhi, Llo, j) : Llo;
                    const uint32_t h = j ? __builtin_amdgcn_alignbit(Hhi, Hlo, j) : Hlo;
                    const uint32_t dcount = __builtin_popcount(((l ^ ql) | (h ^ qh)) & km);
                    if constexpr (COUNT) hits += dcount <= tau ? 1u : 0u;
                    else acc |= dcount << (8 * b);
                }
                o[q] = acc;
            }
            if constexpr (!COUNT) {
                const u32x4 ov = {o[0], o[1], o[2], o[3]};
                store_group<NTST, ALIGNED>(dist + ((r0 + u) << 10) + 16 * lane, ov);
            }
        }
    }

    // tail: one window per thread, byte loads
    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const unsigned long long gt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (rounds << 10) + gt; i < nwin; i += nthreads) {
        unsigned long long w = 0;
        bool flagged = false;
        for (unsigned b = 0; b < k; ++b) {
            const uint32_t byte = ref[i + b];
            if (!valid_base(byte) && !flagged) { latch_bad(slot, i + b, byte); flagged = true; }
            w |= (unsigned long long)code_of(byte) << (2 * b);
        }
        const unsigned long long x = (w ^ query) & kmask;
        const uint32_t d = (uint32_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
        if constexpr (COUNT) hits += d <= tau ? 1u : 0u;
        else dist[i] = (uint8_t)d;
    }
    if constexpr (COUNT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
        __shared__ uint32_t part[kBlock / 64];
        if (lane == 0) part[threadIdx.x >> 6] = hits;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long s = 0;
            for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += part[i];
            if (s) add_performed(total, s);
            if (draw_last_ticket(ticket)) *result = atomicExch(total, 0ull);
        }
    }
}

// (round 4's other scan form, kmer_scan3_kernel -- a wave owns C consecutive rounds and carries the halo planes between its trips; fewest
// instructions, 10-20 % slower, profiles/r04_ab_scan3.txt -- : csrc/evidence/kmer_evidence.h, evidence build only)

// ---------------------------------------------------------------------------------
// every window of a sequence: as_2bit over seq.windows(k)  (stride == 1, src/lib.rs:170-173)
// ---------------------------------------------------------------------------------
// out[i] = as_2bit(seq[i .. i+k]) for consecutive i: 1 B read + 8 B written per window, so the kernel is
// bound by its output.  Same tiling as the scan: a wave reads 1024 consecutive bytes (dwordx4 per lane),
// every lane packs its 16 bases into one 32-bit code word, the next two lanes' words arrive by DPP, and
// window j of a lane is the 2k bits at bit 2j of that 96-bit run (two v_alignbit + a mask).  A lane's 16
// windows are consecutive in memory (128 B), so they cross a wave-private LDS strip once (padded to 144 B
// per lane: conflict-free both ways) and leave as dwordx4 stores of 1 KiB contiguous per instruction.
// 992 windows per wave round (lanes 62/63 only supply the halo); needs bytes [wb, wb+1024) in bounds,
// the leftover windows go through kmer_batch_kernel.
//
// The same tiling serves the power-of-two strides 2, 4, 8, 16 (a lane then owns 16/S windows, every S-th bit
// position; 992 is a multiple of 16, so rounds stay aligned to the stride): S <= 4 crosses the strip, S >= 8
// stores its 16 or 8 bytes per lane directly (already contiguous across lanes).
// U = consecutive rounds per wave trip: their loads are issued before the first is packed (the strip limits a CU to 16 waves,
// so a wave with one 1 KiB load in flight leaves the read path idle while it stores its 8 KiB).
template <int S, bool NTST, int U = 1>
__global__ void __launch_bounds__(kBlock)
kmer_slide_kernel(const uint8_t *__restrict__ seq, unsigned k, unsigned long long rounds, unsigned long long *__restrict__ out,
                  unsigned long long *__restrict__ slot) {
    constexpr int W = 16 / S;                 // windows per lane and round
    constexpr int LW = W + 2;                 // u64 slots per lane in the strip (padding: conflict-free both ways)
    constexpr unsigned RW = kScanWaveWindows / S; // windows per wave round
    __shared__ __attribute__((aligned(16))) unsigned long long strips[kBlock / 64][W >= 4 ? 64 * LW : 2];
    const unsigned lane = threadIdx.x & 63;
    unsigned long long *strip = strips[wave_in_block()];
    const unsigned long long wave = (unsigned long long)blockIdx.x * (kBlock / 64) + wave_in_block();
    const unsigned long long nwaves = (unsigned long long)gridDim.x * (kBlock / 64);
    const uint32_t mlo = k >= 16 ? ~0u : (1u << (2 * k)) - 1u;
    const uint32_t mhi = k <= 16 ? 0u : (k == 32 ? ~0u : (1u << (2 * k - 32)) - 1u);
    for (unsigned long long r0 = wave * U; r0 < rounds; r0 += nwaves * U) {
      u32x4 vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) vv[u] = load_group<true, true>(seq + (r0 + u < rounds ? r0 + u : rounds - 1) * kScanWaveWindows + 16 * lane);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned long long r = r0 + u;
        if (r >= rounds) break; // wave-uniform
        const unsigned long long wb = r * kScanWaveWindows; // first base of the round
        unsigned long long *dst = out + r * RW;              // its first window
        const u32x4 v = vv[u];
        uint32_t bad = 0;
        const uint32_t c0 = enc16(v, bad);
        if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(seq, wb + 16 * lane, 16, slot);
        const uint32_t c1 = wave_shl1(c0), c2 = wave_shl1(c1);
        uint32_t lo[W], hi[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int j = i * S; // base offset of the lane's i-th window
            lo[i] = (j ? __builtin_amdgcn_alignbit(c1, c0, 2 * j) : c0) & mlo;
            hi[i] = (j ? __builtin_amdgcn_alignbit(c2, c1, 2 * j) : c1) & mhi;
        }
        if constexpr (W == 1) {
            if (lane < 62) {
                const unsigned long long w = ((unsigned long long)hi[0] << 32) | lo[0];
                if constexpr (NTST) __builtin_nontemporal_store(w, dst + lane); else dst[lane] = w;
            }
        } else if constexpr (W == 2) {
            if (lane < 62) {
                const u32x4 t = {lo[0], hi[0], lo[1], hi[1]};
                u32x4 *d = reinterpret_cast<u32x4 *>(dst) + lane;
                if constexpr (NTST) __builtin_nontemporal_store(t, d); else *d = t;
            }
        } else {
            wave_lds_fence(); // previous round's readers are done
            u32x4 *mine = reinterpret_cast<u32x4 *>(strip + LW * lane);
#pragma unroll
            for (int i = 0; i < W; i += 2) mine[i >> 1] = u32x4{lo[i], hi[i], lo[i + 1], hi[i + 1]};
            wave_lds_fence();
            // RW windows = RW/2 pairs; pair p = windows 2p, 2p+1, held by lane p / (W/2)
#pragma unroll
            for (int m = 0; m < W / 2; ++m) {
                const unsigned p = lane + 64 * m;
                if (p < RW / 2) {
                    const u32x4 t = *reinterpret_cast<const u32x4 *>(strip + LW * (p / (W / 2)) + 2 * (p % (W / 2)));
                    u32x4 *d = reinterpret_cast<u32x4 *>(dst) + p;
                    if constexpr (NTST) __builtin_nontemporal_store(t, d); else *d = t;
                }
            }
        }
      }
    }
}

// ---------------------------------------------------------------------------------
// every window of a sequence, line-aligned rounds and no LDS strip (round 3)
// ---------------------------------------------------------------------------------
// kmer_slide_kernel<1> stops at 91 % of the box's fill rate: its rounds advance by 992 windows, so every 7.75 KiB wave store
// starts and ends inside a line the neighbouring wave also writes, and its 9 KiB strip per wave limits a CU to 16 waves.
// Here a round is 1024 windows at a 1024-byte aligned input offset -- 8 KiB of output = 64 whole lines per wave round -- and the
// windows are not transposed at all: they are COMPUTED where they are stored.  Store instruction i (of 8) writes the contiguous
// KiB of windows 128 i .. 128 i + 127, lane l the two windows at 128 i + 2 l.  Those 64 bits start at bit 256 i + 4 l of the
// round's 2-bit stream, i.e. in code dword D = 8 i + (l >> 3) (dword d = the 16 bases packed by lane d), at bit 4 (l & 7): the
// lane fetches dwords D, D+1, D+2 with three ds_bpermute_b32 (the LDS crossbar, no LDS memory: nothing is allocated, occupancy
// is the register file's) and cuts both windows with v_alignbit.  24 bpermutes per round replace 8 + 8 b128 strip accesses.
// Dwords 64 and 65 (the 30-base halo) are the first two code dwords of the NEXT round: inside a trip of U consecutive rounds
// they are already in registers, the trip's last round gets them from one extra 16-byte load in lanes 0 and 1 (as kmer_scan2_kernel).
// Needs bytes [1024 r, 1024 r + 1056) in bounds; the leftover windows go through kmer_batch_kernel.
template <bool NTST, int U>
__global__ void __launch_bounds__(kBlock)
kmer_slide2_kernel(const uint8_t *__restrict__ seq, unsigned k, unsigned long long rounds, unsigned long long *__restrict__ out,
                   unsigned long long *__restrict__ slot) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (kBlock / 64) + wave_in_block();
    const unsigned long long nwaves = (unsigned long long)gridDim.x * (kBlock / 64);
    const uint32_t mlo = k >= 16 ? ~0u : (1u << (2 * k)) - 1u;
    const uint32_t mhi = k <= 16 ? 0u : (k == 32 ? ~0u : (1u << (2 * k - 32)) - 1u);
    const int src = (int)((lane >> 3) << 2);   // byte address of lane (l >> 3) for ds_bpermute; + 32 i + {0, 4, 8} as immediate offsets
    const unsigned sh = 4u * (lane & 7u);      // bit offset of the lane's first window inside dword D
    const bool g6 = (lane >> 3) == 6u, g7 = (lane >> 3) == 7u;
    for (unsigned long long r0 = wave * U; r0 < rounds; r0 += nwaves * U) {
        const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U; // valid rounds in this trip (wave-uniform)
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load_group<true, true>(seq + ((r0 + u < rounds ? r0 + u : rounds - 1) << 10) + 16 * lane);
        u32x4 hv = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
        if (lane < 2) hv = load_group<false, true>(seq + ((r0 + m) << 10) + 16 * lane); // the halo of the trip's last round
        uint32_t c[U + 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t bad = 0;
            c[u] = enc16(v[u], bad);
            if (__builtin_expect(residue_is_bad(bad) && (unsigned)u < m, 0)) rescan_bytes(seq, ((r0 + u) << 10) + 16 * lane, 16, slot);
        }
        {
            uint32_t bad = 0; // halo bytes are validated by the round (or the tail) that owns them
            c[U] = enc16(hv, bad);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((unsigned)u >= m) break; // wave-uniform
            const uint32_t nx = (unsigned)(u + 1) < m ? c[u + 1] : c[U]; // code dwords of the next KiB (wave-uniform choice)
            const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)nx, 0), h1 = (uint32_t)__builtin_amdgcn_readlane((int)nx, 1);
            unsigned long long *dst = out + ((r0 + u) << 10);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src + 32 * i, (int)c[u]);
                uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src + 32 * i + 4, (int)c[u]);
                uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src + 32 * i + 8, (int)c[u]);
                if (i == 7) { // dwords 64 / 65 wrap around in the permute: they are the halo
                    w1 = g7 ? h0 : w1;
                    w2 = g6 ? h0 : (g7 ? h1 : w2);
                }
                u32x4 t;
                t.x = __builtin_amdgcn_alignbit(w1, w0, sh) & mlo; // (a shift of 0 returns w0)
                t.y = __builtin_amdgcn_alignbit(w2, w1, sh) & mhi;
                t.z = __builtin_amdgcn_alignbit(w1, w0, sh + 2) & mlo;
                t.w = __builtin_amdgcn_alignbit(w2, w1, sh + 2) & mhi;
                u32x4 *d = reinterpret_cast<u32x4 *>(dst + 128 * i) + lane;
                if constexpr (NTST) __builtin_nontemporal_store(t, d); else *d = t;
            }
        }
    }
}

// Any other stride 3 <= S < 32 with k >= S: the same round, but a lane keeps only the windows whose position is a
// multiple of S (at most ceil(16/S) of its 16, found with one wave-uniform division for the round's first base and a
// 32-bit multiply-high step per lane), shifts them out with a per-lane v_alignbit amount, and drops them into the
// strip at their output index, so the strip holds the round's k-mers densely and in order; they leave as 8-byte
// stores, 512 contiguous bytes per instruction.
__global__ void __launch_bounds__(kBlock)
kmer_slide_any_kernel(const uint8_t *__restrict__ seq, unsigned k, unsigned S, unsigned magic /* ceil(2^32 / S) */,
                      unsigned long long magic64 /* floor(2^64 / S) + 1 */, unsigned long long rounds, unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    constexpr int kMaxOut = kScanWaveWindows / 3 + 2; // S >= 3
    __shared__ unsigned long long strips[kBlock / 64][kMaxOut + 2];
    const unsigned lane = threadIdx.x & 63;
    unsigned long long *strip = strips[wave_in_block()];
    const unsigned long long wave = (unsigned long long)blockIdx.x * (kBlock / 64) + wave_in_block();
    const unsigned long long nwaves = (unsigned long long)gridDim.x * (kBlock / 64);
    const uint32_t mlo = k >= 16 ? ~0u : (1u << (2 * k)) - 1u;
    const uint32_t mhi = k <= 16 ? 0u : (k == 32 ? ~0u : (1u << (2 * k - 32)) - 1u);
    for (unsigned long long r = wave; r < rounds; r += nwaves) {
        const unsigned long long wb = r * kScanWaveWindows; // first base of the round (wave-uniform)
        // wave-uniform divisions by S as multiply-highs (exact while x * S < 2^64)
        const unsigned long long q_w = (unsigned long long)(((unsigned __int128)wb * magic64) >> 64);
        const unsigned r_w = (unsigned)(wb - q_w * S);
        const unsigned long long first_out = q_w + (r_w ? 1 : 0);                 // first k-mer that starts in this round
        const unsigned n_out = (unsigned)((unsigned long long)(((unsigned __int128)(wb + kScanWaveWindows - 1) * magic64) >> 64) - first_out + 1); // k-mers that start in [wb, wb + 992)
        const u32x4 v = load_group<true, true>(seq + wb + 16 * lane);
        uint32_t bad = 0;
        const uint32_t c0 = enc16(v, bad);
        if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(seq, wb + 16 * lane, 16, slot);
        const uint32_t c1 = wave_shl1(c0), c2 = wave_shl1(c1);
        const unsigned t = r_w + 16 * lane;              // position of the lane's first base, relative to q_w * S
        const unsigned q_l = __umulhi(t, magic), r_l = t - q_l * S;
        unsigned j = r_l ? S - r_l : 0u;                 // offset of the lane's first kept window
        unsigned o = q_l + (r_l ? 1u : 0u) - (r_w ? 1u : 0u); // its index among the round's k-mers
        wave_lds_fence(); // previous round's readers are done
        if (lane < 62) {
#pragma unroll 1
            for (; j < 16; j += S, ++o) {
                const uint32_t lo = (j ? __builtin_amdgcn_alignbit(c1, c0, 2 * j) : c0) & mlo;
                const uint32_t hi = (j ? __builtin_amdgcn_alignbit(c2, c1, 2 * j) : c1) & mhi;
                strip[o] = ((unsigned long long)hi << 32) | lo;
            }
        }
        wave_lds_fence();
        unsigned long long *dst = out + first_out;
#pragma unroll 1
        for (unsigned i = lane; i < n_out; i += 64) __builtin_nontemporal_store(strip[i], dst + i);
    }
}

// ---------------------------------------------------------------------------------
// bulk hdist: sum over words of the per-word mismatch count (u32, wraps like Rust release)
// ---------------------------------------------------------------------------------
// TILED (round 3 experiment, tools/ab_hdist_tiled.py): the grid-stride walk at TILE granularity -- a workgroup reads 4 consecutive
// KiB-per-wave rows (16 KiB of each operand) per trip, like the read probe -- instead of 4 loads that are a whole grid apart.
template <bool TILED>
__global__ void __launch_bounds__(kBlock)
hdist_kernel(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b,
             unsigned long long n_bases, uint32_t *__restrict__ result, unsigned *__restrict__ total /* zero between launches */,
             unsigned *__restrict__ ticket) {
    // 16 B algorithmic per 32-base word pair.  Grid-stride over word PAIRS with dwordx4
    // loads when both buffers are 16-byte aligned, u64 loads otherwise; a fixed, resident
    // grid adds one partial per workgroup to a context-owned accumulator; the last to finish publishes it.
    const unsigned long long full = n_bases >> 5;
    const unsigned rem = (unsigned)(n_bases & 31);
    const unsigned long long gt = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * kBlock;
    uint32_t acc = 0;
    const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    unsigned long long done = 0;
    if (al) {
        const unsigned long long pairs = full >> 1;
        const u32x4 *a4 = reinterpret_cast<const u32x4 *>(a), *b4 = reinterpret_cast<const u32x4 *>(b);
        if constexpr (TILED) {
            constexpr unsigned long long TILE = (unsigned long long)kBlock * 4; // 16-byte pairs per workgroup trip
            const unsigned long long tiles = pairs / TILE;
            for (unsigned long long t = blockIdx.x; t < tiles; t += gridDim.x) {
                u32x4 va[4], vb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    va[u] = __builtin_nontemporal_load(a4 + t * TILE + u * kBlock + threadIdx.x);
                    vb[u] = __builtin_nontemporal_load(b4 + t * TILE + u * kBlock + threadIdx.x);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32x4 x = va[u] ^ vb[u];
                    acc += __builtin_popcount(mismatch_bits(x.x, 0x55555555u)) + __builtin_popcount(mismatch_bits(x.y, 0x55555555u)) +
                           __builtin_popcount(mismatch_bits(x.z, 0x55555555u)) + __builtin_popcount(mismatch_bits(x.w, 0x55555555u));
                }
            }
            for (unsigned long long p = tiles * TILE + gt; p < pairs; p += nthreads) { // the pairs past the last whole tile
                const u32x4 x = __builtin_nontemporal_load(a4 + p) ^ __builtin_nontemporal_load(b4 + p);
                acc += __builtin_popcount(mismatch_bits(x.x, 0x55555555u)) + __builtin_popcount(mismatch_bits(x.y, 0x55555555u)) +
                       __builtin_popcount(mismatch_bits(x.z, 0x55555555u)) + __builtin_popcount(mismatch_bits(x.w, 0x55555555u));
            }
        } else {
#pragma unroll 4
            for (unsigned long long p = gt; p < pairs; p += nthreads) {
                const u32x4 x = __builtin_nontemporal_load(a4 + p) ^ __builtin_nontemporal_load(b4 + p);
                acc += __builtin_popcount(mismatch_bits(x.x, 0x55555555u)) + __builtin_popcount(mismatch_bits(x.y, 0x55555555u)) +
                       __builtin_popcount(mismatch_bits(x.z, 0x55555555u)) + __builtin_popcount(mismatch_bits(x.w, 0x55555555u));
            }
        }
        done = pairs << 1;
    }
    for (unsigned long long w = done + gt; w < full; w += nthreads) {
        const unsigned long long x = a[w] ^ b[w];
        acc += (uint32_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) { // scalar.rs:26-33: bits above 2*rem are ignored
        const unsigned long long mask = (1ull << (2 * rem)) - 1;
        const unsigned long long x = (a[full] ^ b[full]) & mask;
        acc += (uint32_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ uint32_t part[kBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += part[i];
        if (s) add_performed(total, s); // u32 wrap-around like the reference's accumulator (multi.rs:130)
        if (draw_last_ticket(ticket)) *result = atomicExch(total, 0u);
    }
}

#ifdef BITNUC_SWEEP_VARIANTS
#include "evidence/kmer_evidence.h" // the formulations that lost their A/B: evidence build only
#endif

} // namespace bitnuc_dev
