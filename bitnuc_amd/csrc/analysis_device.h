// analysis_device.h -- reductions and per-word distances directly on packed words
// (the callers just above the codec).
//
//   base_counts : [A,C,G,T] counts of a packed sequence.  The reference decodes to ASCII
//                 and counts bytes (src/utils/analysis.rs:23-39, via PackedSequence::to_vec);
//                 here it is popcounts of the two code bit-planes at 0.25 B/base, no decode.
//   hdist_pairs : dist[i] = hdist_scalar(a[i], b[i], len)   (hamming/scalar.rs:11-48), many pairs
//   hdist_query : dist[i] = hdist_scalar(query, t[i], len), one query against many targets
#pragma once
#include "device_prims.h"

#ifndef BITNUC_COUNTS_UNROLL
#define BITNUC_COUNTS_UNROLL 4 // 16-byte loads in flight per thread of base_counts_kernel (tools/ab_counts_unroll.py)
#endif

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
                   unsigned long long *__restrict__ counts /* [A,C,G,T] */, unsigned long long *__restrict__ acc /* C,G,T: zero between launches */,
                   unsigned *__restrict__ ticket) {
    const unsigned long long full = n_bases >> 5;
    const unsigned rem = (unsigned)(n_bases & 31);
    const unsigned long long gt = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * kBlock;
    uint32_t c = 0, g = 0, t = 0; // < 2^32 per thread: a thread sees n_bases / nthreads bases
    unsigned long long done = 0;
    if ((reinterpret_cast<uintptr_t>(words) & 15) == 0) {
        const unsigned long long pairs = full >> 1;
        const u32x4 *w4 = reinterpret_cast<const u32x4 *>(words);
        // U loads are ISSUED before the first is counted (explicit registers: with `#pragma unroll 8` on the plain loop the
        // compiler kept ONE load in flight -- load, s_waitcnt vmcnt(0), count, load ... -- and the kernel ran at half speed);
        // a lane past the end re-reads the last pair and counts zeros
        constexpr int U = BITNUC_COUNTS_UNROLL;
        for (unsigned long long p0 = gt; p0 < pairs; p0 += nthreads * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long p = p0 + (unsigned long long)u * nthreads;
                v[u] = __builtin_nontemporal_load(w4 + (p < pairs ? p : pairs - 1));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = p0 + (unsigned long long)u * nthreads < pairs;
                const unsigned long long lo = in ? ((unsigned long long)v[u].y << 32) | v[u].x : 0ull, hi = in ? ((unsigned long long)v[u].w << 32) | v[u].z : 0ull;
                count_word(lo, c, g, t);
                count_word(hi, c, g, t);
            }
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
#ifdef BITNUC_COUNTS_NO_REDUCE // timing-only experiment (tools/ab_counts_unroll.py): what the kernel costs without its arrival atomics
    if (threadIdx.x == 0 && part[0][0] == 0x123456789ull) counts[0] = 1;
    return;
#endif
    if (threadIdx.x < 3) {
        unsigned long long s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += part[i][threadIdx.x];
        if (s) add_performed(acc + threadIdx.x, s);
    }
    __syncthreads();
    if (threadIdx.x == 0 && draw_last_ticket(ticket)) {
        const unsigned long long cs = atomicExch(acc, 0ull), gs = atomicExch(acc + 1, 0ull), ts = atomicExch(acc + 2, 0ull);
        counts[0] = n_bases - (cs + gs + ts); // A: the zero padding of the last word is not a base
        counts[1] = cs;
        counts[2] = gs;
        counts[3] = ts;
    }
}

// Round 3: the same result with fully coalesced loads.  Above, lane q reads its four words as 32 CONTIGUOUS bytes, so each of its
// two load instructions covers the wave's 2 KiB with half-filled lines.  Here a wave takes 256 words and lane l loads the 16 bytes
// at 16 l of the first KiB and of the second KiB (two contiguous KiB per instruction pair), which leaves it with the distances of
// words 2l, 2l+1 and 128+2l, 128+2l+1; the four bytes a lane STORES (words 4l .. 4l+3) sit in two neighbouring lanes and arrive
// by two ds_bpermute_b32 (LDS crossbar, no LDS memory).  Only whole 256-word wave tiles; the tail goes through the kernel above.
template <bool QUERY>
__global__ void __launch_bounds__(kBlock)
hdist_words_coalesced_kernel(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b, unsigned long long query,
                             unsigned long long tiles /* of 256 words */, unsigned len, uint8_t *__restrict__ dist) {
    const unsigned long long mask = len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1); // scalar.rs:26-30
    const uint32_t mlo = (uint32_t)mask & 0x55555555u, mhi = (uint32_t)(mask >> 32) & 0x55555555u;
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const unsigned long long nwaves = (unsigned long long)gridDim.x * (kBlock / 64);
    const u32x4 q4 = {(uint32_t)query, (uint32_t)(query >> 32), (uint32_t)query, (uint32_t)(query >> 32)};
    const int src = (int)(((lane & 31u) << 1) << 2); // byte address of lane 2 (l & 31) for ds_bpermute
    for (unsigned long long t = wave; t < tiles; t += nwaves) {
        const u32x4 *pa = reinterpret_cast<const u32x4 *>(a + t * 256);
        const u32x4 a0 = __builtin_nontemporal_load(pa + lane), a1 = __builtin_nontemporal_load(pa + 64 + lane);
        u32x4 b0 = q4, b1 = q4;
        if constexpr (!QUERY) {
            const u32x4 *pb = reinterpret_cast<const u32x4 *>(b + t * 256);
            b0 = __builtin_nontemporal_load(pb + lane);
            b1 = __builtin_nontemporal_load(pb + 64 + lane);
        }
        const u32x4 x0 = a0 ^ b0, x1 = a1 ^ b1;
        const uint32_t d0 = __builtin_popcount((x0.x | (x0.x >> 1)) & mlo) + __builtin_popcount((x0.y | (x0.y >> 1)) & mhi);
        const uint32_t d1 = __builtin_popcount((x0.z | (x0.z >> 1)) & mlo) + __builtin_popcount((x0.w | (x0.w >> 1)) & mhi);
        const uint32_t d2 = __builtin_popcount((x1.x | (x1.x >> 1)) & mlo) + __builtin_popcount((x1.y | (x1.y >> 1)) & mhi);
        const uint32_t d3 = __builtin_popcount((x1.z | (x1.z >> 1)) & mlo) + __builtin_popcount((x1.w | (x1.w >> 1)) & mhi);
        const uint32_t mine = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24); // low half: words 2l, 2l+1; high half: words 128+2l, 128+2l+1
        const uint32_t v0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)mine), v1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src + 4, (int)mine);
        // lanes 0..31 store words 4l..4l+3 (first KiB: low halves), lanes 32..63 words 128 + 4(l-32).. (second KiB: high halves)
        const uint32_t out = lane < 32 ? ((v0 & 0xFFFFu) | (v1 << 16)) : ((v0 >> 16) | (v1 & 0xFFFF0000u));
        __builtin_nontemporal_store(out, reinterpret_cast<uint32_t *>(dist + t * 256) + lane);
    }
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

// split_packed (src/utils/functions/split.rs:15-99): cut a packed sequence at base c*32 + s/2.
//   left[j]  = e[j] for j < n_left-1, left[n_left-1] = e[n_left-1] & lmask_last
//   CANON:     right[j] = e[c+j] >> s | e[c+j+1] << (64-s)   (the funnel shift; last word & rmask_last)
//   as written: right[j] = e[c+j] >> s | e[c+j-1] << (64-s)  for j > 0 (split.rs:84-94 carries the
//              *previous* word's low bits), right[0] = e[c] >> s
// One lane -> two output words (one 16-byte store); the three source words a right-hand pair
// needs are one dwordx4 + one dwordx2 load, whichever of the two is 16-byte aligned (wave-uniform
// choice).  The last two words of each side (masks, buffer end) take the one-word path.
template <bool CANON>
__device__ __forceinline__ unsigned long long split_right_word(const unsigned long long *__restrict__ e, unsigned long long src_words,
                                                               unsigned long long c, unsigned s, unsigned long long j) {
    const unsigned long long k = c + j;
    unsigned long long v = e[k] >> s;
    if (s) {
        if constexpr (CANON) {
            if (k + 1 < src_words) v |= e[k + 1] << (64 - s);
        } else {
            if (j) v |= e[k - 1] << (64 - s);
        }
    }
    return v;
}

__device__ __forceinline__ unsigned long long u64_of(uint32_t lo, uint32_t hi) { return ((unsigned long long)hi << 32) | lo; }

template <bool CANON>
__global__ void __launch_bounds__(kBlock)
split_packed_kernel(const unsigned long long *__restrict__ e, unsigned long long src_words, unsigned long long c, unsigned s,
                    unsigned long long n_left, unsigned long long n_right, unsigned long long lmask_last,
                    unsigned long long rmask_last, unsigned long long *__restrict__ l, unsigned long long *__restrict__ r) {
    const bool fast = ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(l) | reinterpret_cast<uintptr_t>(r)) & 15) == 0;
    const unsigned long long lpairs = fast && n_left > 2 ? (n_left - 1) >> 1 : 0;  // pairs that end before the last left word
    const unsigned long long rpairs = fast && n_right > 2 ? (n_right - 1) >> 1 : 0;
    const unsigned long long ltail = n_left - 2 * lpairs, rtail = n_right - 2 * rpairs;
    const unsigned long long items = lpairs + rpairs + ltail + rtail;
    // first source word of a right pair is c + 2p (CANON) or c + 2p - 1 (as written): is that one 16-byte aligned?
    const bool first_aligned = (((CANON ? c : c + 1) & 1) == 0);
    const unsigned sh = s ? 64 - s : 0;
    const unsigned long long keep = s ? ~0ull : 0; // s == 0: nothing is carried in
    for (unsigned long long t = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; t < items;
         t += (unsigned long long)gridDim.x * kBlock) {
        if (t < lpairs) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(e) + t);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(l) + t);
        } else if (t < lpairs + rpairs) {
            const unsigned long long p = t - lpairs;
            // w0..w2 = the three consecutive source words starting at `base`
            const unsigned long long base = CANON ? c + 2 * p : c + 2 * p - 1; // as written, p == 0: e[c-1] is not used
            unsigned long long w0, w1, w2;
            if (first_aligned) { // as written this means c is odd, so e[c - 1] exists (and is ignored for p == 0)
                const u32x4 a = *reinterpret_cast<const u32x4 *>(e + base);
                w0 = u64_of(a.x, a.y);
                w1 = u64_of(a.z, a.w);
                w2 = e[base + 2];
            } else {
                w0 = (CANON || p) ? e[base] : 0;
                const u32x4 a = *reinterpret_cast<const u32x4 *>(e + (base + 1));
                w1 = u64_of(a.x, a.y);
                w2 = u64_of(a.z, a.w);
            }
            unsigned long long o0, o1;
            if constexpr (CANON) {
                o0 = (w0 >> s) | ((w1 << sh) & keep);
                o1 = (w1 >> s) | ((w2 << sh) & keep);
            } else {
                o0 = (w1 >> s) | (p ? (w0 << sh) & keep : 0);
                o1 = (w2 >> s) | ((w1 << sh) & keep);
            }
            __builtin_nontemporal_store(u32x4{(uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)o1, (uint32_t)(o1 >> 32)},
                                        reinterpret_cast<u32x4 *>(r) + p);
        } else if (t < lpairs + rpairs + ltail) {
            const unsigned long long j = 2 * lpairs + (t - lpairs - rpairs);
            unsigned long long v = e[j];
            if (j == n_left - 1) v &= lmask_last;
            l[j] = v;
        } else {
            const unsigned long long j = 2 * rpairs + (t - lpairs - rpairs - ltail);
            unsigned long long v = split_right_word<CANON>(e, src_words, c, s, j);
            if (CANON && j == n_right - 1) v &= rmask_last;
            r[j] = v;
        }
    }
}

} // namespace bitnuc_dev
