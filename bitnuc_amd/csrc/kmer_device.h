// kmer_device.h -- gfx950 device code for the k-mer compositions of the hot path:
//   * batched as_2bit over many <=32-mers at a byte stride   (BASELINE config 3)
//   * sliding-window k-mer pack + Hamming distance to a query (BASELINE config 5)
//   * bulk packed-vs-packed Hamming distance (hdist)
//
// Values follow src/utils/packing/naive.rs:8-18 (pack) and
// src/utils/functions/hamming/scalar.rs:22-47 (distance of two packed words).
#pragma once
#include "codec_device.h"

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
                  unsigned long long *__restrict__ slot) {
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
                if (!valid_base(byte)) { atomicMin(slot, j * stride + b); break; }
            }
        }
        out[j] = ((unsigned long long)hi << 32) | lo;
    }
}

// ---------------------------------------------------------------------------------
// sliding-window k-mer pack + Hamming distance scan
// ---------------------------------------------------------------------------------
// Fast path: a wave owns 1024 consecutive windows.  Lane l loads the 16 bytes at
// window 16l with one coalesced dwordx4 and packs them to a 32-bit code stream; the
// 30-base halo it needs is the next two lanes' streams (two lane shifts) -- only lanes
// 62/63 take it from a 32-byte wave-uniform halo load.  Window j is the 64-bit field at
// bit 2j of the 96-bit stream {s2,s1,s0}: two v_alignbit_b32 with a constant shift.  Output: 16 distance bytes per lane, one coalesced dwordx4 store.
// Needs bytes [wb, wb+1056) in bounds; the (< 2080) windows left over go through the
// byte-wise tail below.
template <bool ALIGNED, bool NT>
__global__ void __launch_bounds__(kBlock)
kmer_scan_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query,
                 uint8_t *__restrict__ dist, unsigned long long *__restrict__ slot) {
    const unsigned long long nwin = n - k + 1; // host guarantees 1 <= k <= 32, n >= k
    const unsigned long long rounds = n >= 1056 ? (n - 1056) / 1024 + 1 : 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = ((unsigned long long)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * kBlock) >> 6;

    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint32_t qlo = (uint32_t)(query & kmask), qhi = (uint32_t)((query & kmask) >> 32);
    const uint32_t mlo = (uint32_t)kmask & 0x55555555u, mhi = (uint32_t)(kmask >> 32) & 0x55555555u;

    for (unsigned long long r = wave; r < rounds; r += nwaves) {
        const unsigned long long wb = r * 1024;
        const uint8_t *p = ref + wb + 16 * lane;
        const u32x4 v0 = load_group<NT, ALIGNED>(p);
        const u32x4 hv = load_group<false, ALIGNED>(ref + wb + 1024 + 16 * (lane & 1)); // 32-byte halo, two addresses per wave
        // encode once per lane; the halo arrives as the next lanes' 32-bit code
        // streams.  Every byte is validated by the lane (or tail thread) that owns it.
        uint32_t bad = 0, hbad = 0;
        const uint32_t s0 = enc16(v0, bad);
        const uint32_t hs = enc16(hv, hbad);
        const uint32_t d1 = __shfl_down(s0, 1), d2 = __shfl_down(s0, 2);
        const uint32_t h0 = __shfl(hs, 0), h1 = __shfl(hs, 1);
        const uint32_t s1 = lane < 63 ? d1 : h0;
        const uint32_t s2 = lane < 62 ? d2 : (lane == 62 ? h0 : h1);
        if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(ref, wb + 16 * lane, 16, slot);
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = 4 * q + b;
                const uint32_t lo = j ? __builtin_amdgcn_alignbit(s1, s0, 2 * j) : s0;
                const uint32_t hi = j ? __builtin_amdgcn_alignbit(s2, s1, 2 * j) : s1;
                const uint32_t d = __builtin_popcount(mismatch_bits(lo ^ qlo, mlo)) +
                                   __builtin_popcount(mismatch_bits(hi ^ qhi, mhi));
                acc |= d << (8 * b);
            }
            o[q] = acc;
        }
        const u32x4 ov = {o[0], o[1], o[2], o[3]};
        store_group<NT, ALIGNED>(dist + wb + 16 * lane, ov);
    }

    // tail: one window per thread, byte loads
    const unsigned long long gt = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long i = rounds * 1024 + gt; i < nwin; i += nthreads) {
        unsigned long long w = 0;
        bool flagged = false;
        for (unsigned b = 0; b < k; ++b) {
            const uint32_t byte = ref[i + b];
            if (!valid_base(byte) && !flagged) { atomicMin(slot, i + b); flagged = true; }
            w |= (unsigned long long)code_of(byte) << (2 * b);
        }
        const unsigned long long x = (w ^ query) & kmask;
        dist[i] = (uint8_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
}

// ---------------------------------------------------------------------------------
// bulk hdist: sum over words of the per-word mismatch count (u32, wraps like Rust release)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
hdist_kernel(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b,
             unsigned long long n_bases, uint32_t *__restrict__ result) {
    const unsigned long long full = n_bases >> 5;
    const unsigned rem = (unsigned)(n_bases & 31);
    uint32_t acc = 0;
    for (unsigned long long w = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; w < full;
         w += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long x = a[w] ^ b[w];
        acc += (uint32_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) {
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
        if (s) atomicAdd(result, s);
    }
}

} // namespace bitnuc_dev
