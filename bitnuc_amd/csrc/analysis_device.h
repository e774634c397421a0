// analysis_device.h -- reductions and per-word distances directly on packed words
// (the callers just above the codec).
//
//   base_counts : [A,C,G,T] counts of a packed sequence.  The reference decodes to ASCII
//                 and counts bytes (src/utils/analysis.rs:23-39, via PackedSequence::to_vec);
//                 here it is popcounts of the two code bit-planes at 0.25 B/base, no decode.
//   hdist_pairs : dist[i] = hdist_scalar(a[i], b[i], len)   (hamming/scalar.rs:11-48), many pairs
//   hdist_query : dist[i] = hdist_scalar(query, t[i], len), one query against many targets
#pragma once
#include "codec_device.h"

namespace bitnuc_dev {

// counts[1..3] += C,G,T of one word's bases (A is derived from the length by the host side
// of the kernel: zero padding would otherwise count as A)
__device__ __forceinline__ void count_word(unsigned long long w, uint32_t &c, uint32_t &g, uint32_t &t) {
    const unsigned long long lo = w & 0x5555555555555555ull, hi = (w >> 1) & 0x5555555555555555ull;
    t += (uint32_t)__builtin_popcountll(lo & hi);
    g += (uint32_t)__builtin_popcountll(hi & ~lo);
    c += (uint32_t)__builtin_popcountll(lo & ~hi);
}

__global__ void __launch_bounds__(kBlock)
base_counts_kernel(const unsigned long long *__restrict__ words, unsigned long long n_bases,
                   unsigned long long *__restrict__ counts /* [A,C,G,T], pre-zeroed */) {
    const unsigned long long full = n_bases >> 5;
    const unsigned rem = (unsigned)(n_bases & 31);
    const unsigned long long gt = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * kBlock;
    uint32_t c = 0, g = 0, t = 0; // < 2^32 per thread: a thread sees n_bases / nthreads bases
    unsigned long long done = 0;
    if ((reinterpret_cast<uintptr_t>(words) & 15) == 0) {
        const unsigned long long pairs = full >> 1;
        const u32x4 *w4 = reinterpret_cast<const u32x4 *>(words);
#pragma unroll 4
        for (unsigned long long p = gt; p < pairs; p += nthreads) {
            const u32x4 v = __builtin_nontemporal_load(w4 + p);
            count_word(((unsigned long long)v.y << 32) | v.x, c, g, t);
            count_word(((unsigned long long)v.w << 32) | v.z, c, g, t);
        }
        done = pairs << 1;
    }
    for (unsigned long long w = done + gt; w < full; w += nthreads) count_word(words[w], c, g, t);
    if (rem && gt == 0) // bits above 2*rem are ignored, like PackedSequence::get past `length`
        count_word(words[full] & ((1ull << (2 * rem)) - 1), c, g, t);
    unsigned long long c64 = c, g64 = g, t64 = t;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c64 += __shfl_xor(c64, off);
        g64 += __shfl_xor(g64, off);
        t64 += __shfl_xor(t64, off);
    }
    __shared__ unsigned long long part[kBlock / 64][3];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = c64;
        part[threadIdx.x >> 6][1] = g64;
        part[threadIdx.x >> 6][2] = t64;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += part[i][threadIdx.x];
        if (s) atomicAdd(counts + 1 + threadIdx.x, s);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counts, n_bases); // A = n - C - G - T, fixed up below
}

__global__ void base_counts_finish(unsigned long long *__restrict__ counts) {
    counts[0] -= counts[1] + counts[2] + counts[3];
}

// one lane -> 4 consecutive words (two dwordx4 loads per operand) -> 4 distance bytes (one dword store)
template <bool QUERY>
__global__ void __launch_bounds__(kBlock)
hdist_words_kernel(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b,
                   unsigned long long query, unsigned long long count, unsigned len, uint8_t *__restrict__ dist) {
    const unsigned long long mask = len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1); // scalar.rs:26-30
    const unsigned long long quads = count >> 2;
    const bool al = ((reinterpret_cast<uintptr_t>(a) | (QUERY ? 0 : reinterpret_cast<uintptr_t>(b))) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(dist) & 3) == 0;
    unsigned long long done = 0;
    if (al) {
        for (unsigned long long q = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; q < quads;
             q += (unsigned long long)gridDim.x * kBlock) {
            const u32x4 *pa = reinterpret_cast<const u32x4 *>(a + 4 * q);
            const u32x4 a0 = __builtin_nontemporal_load(pa), a1 = __builtin_nontemporal_load(pa + 1);
            u32x4 b0, b1;
            if constexpr (QUERY) {
                b0 = u32x4{(uint32_t)query, (uint32_t)(query >> 32), (uint32_t)query, (uint32_t)(query >> 32)};
                b1 = b0;
            } else {
                const u32x4 *pb = reinterpret_cast<const u32x4 *>(b + 4 * q);
                b0 = __builtin_nontemporal_load(pb);
                b1 = __builtin_nontemporal_load(pb + 1);
            }
            const u32x4 x0 = a0 ^ b0, x1 = a1 ^ b1;
            const uint32_t mlo = (uint32_t)mask & 0x55555555u, mhi = (uint32_t)(mask >> 32) & 0x55555555u;
            const uint32_t d0 = __builtin_popcount((x0.x | (x0.x >> 1)) & mlo) + __builtin_popcount((x0.y | (x0.y >> 1)) & mhi);
            const uint32_t d1 = __builtin_popcount((x0.z | (x0.z >> 1)) & mlo) + __builtin_popcount((x0.w | (x0.w >> 1)) & mhi);
            const uint32_t d2 = __builtin_popcount((x1.x | (x1.x >> 1)) & mlo) + __builtin_popcount((x1.y | (x1.y >> 1)) & mhi);
            const uint32_t d3 = __builtin_popcount((x1.z | (x1.z >> 1)) & mlo) + __builtin_popcount((x1.w | (x1.w >> 1)) & mhi);
            __builtin_nontemporal_store(d0 | (d1 << 8) | (d2 << 16) | (d3 << 24), reinterpret_cast<uint32_t *>(dist + 4 * q));
        }
        done = quads << 2;
    }
    for (unsigned long long i = done + (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long x = (a[i] ^ (QUERY ? query : b[i])) & mask;
        dist[i] = (uint8_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
}

} // namespace bitnuc_dev
