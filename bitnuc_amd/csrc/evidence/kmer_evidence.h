// kmer_evidence.h -- EVIDENCE BUILD ONLY (-DBITNUC_SWEEP_VARIANTS): k-mer / scan kernels that lost their A/B (profiles/README.md).
// Included at the end of kmer_device.h, inside namespace bitnuc_dev.  Nothing in the product library instantiates or even sees this file.
#pragma once

// ==== kmer_scan3_kernel (round 4): the scan with the fewest vector instructions -- a wave owns C CONSECUTIVE rounds and carries the planes
// of the round after its trip into the next trip (one plane build per round + one per C rounds, planes16_2lut as the shipped GEN 1).
// 10-20 % SLOWER than one trip per wave (profiles/r04_ab_scan3.txt): a wave that runs C / U dependent load -> compute -> store trips loses
// more than the saved build costs.  C / U should be odd (waves that start together stay in step; with C = 16 or 32 KiB they touch a
// quarter or an eighth of the 4 KiB blocks at a time). ====
template <bool NTLD, bool NTST, int U, int C>
__global__ void __launch_bounds__(kBlock)
kmer_scan3_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query, uint32_t ql, uint32_t qh,
                  uint8_t *__restrict__ dist, unsigned long long *__restrict__ slot) {
    static_assert(C % U == 0, "a chunk is whole trips");
    const unsigned long long nwin = n - k + 1;                        // host guarantees 1 <= k <= 32, n >= k
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0; // round r reads bytes [1024 r, 1024 r + 1056)
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    const uint32_t km = k == 32 ? ~0u : ((1u << k) - 1u);
    const uint32_t m63 = lane == 63 ? ~0u : 0u;
    const u32x4 kAs = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};

    const unsigned long long r_begin = wave * C;
    if (r_begin < rounds) { // wave-uniform
        const unsigned long long r_end = r_begin + C < rounds ? r_begin + C : rounds;
        uint32_t cur = 0;
        const u32x4 vfirst = load_group<NTLD, true>(ref + (r_begin << 10) + 16 * lane); // in flight together with the first trip's loads
        bool first = true;
        for (unsigned long long r0 = r_begin; r0 < r_end; r0 += U) {
            const unsigned m = r_end - r0 < (unsigned long long)U ? (unsigned)(r_end - r0) : (unsigned)U; // rounds of this trip (wave-uniform)
            // the rounds AFTER each of the trip's rounds: r0 + 1 .. r0 + m.  Inside the chunk they are whole KiB loads (and the next
            // round to compute); the one at the chunk's end only supplies the 30-base halo: two lanes load, the rest hold 'A's.
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = kAs;
                const unsigned long long ru = r0 + 1 + u;
                if ((unsigned)u < m && (ru < r_end || lane < 2)) v[u] = ru < r_end ? load_group<NTLD, true>(ref + (ru << 10) + 16 * lane) : load_group<false, true>(ref + (ru << 10) + 16 * lane);
            }
            if (first) { // wave-uniform
                first = false;
                uint32_t bad = 0;
                cur = planes16_2lut(vfirst, bad);
                if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(ref, (r_begin << 10) + 16 * lane, 16, slot);
            }
            uint32_t nx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t bad = 0;
                nx[u] = planes16_2lut(v[u], bad);
                // only the rounds this wave owns are validated here (a halo's bytes belong to the next chunk's wave or to the tail)
                if (__builtin_expect(residue_is_bad(bad) && (unsigned)u < m && r0 + 1 + u < r_end, 0)) rescan_bytes(ref, ((r0 + 1 + u) << 10) + 16 * lane, 16, slot);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((unsigned)u >= m) break; // wave-uniform
                const uint32_t pl = u == 0 ? cur : nx[u - 1], nxt = nx[u];
                const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)nxt, 0), h1 = (uint32_t)__builtin_amdgcn_readlane((int)nxt, 1);
                const uint32_t n1 = wave_shl1(pl) | (h0 & m63); // (lane 63 of a wave_shl is 0: see kmer_scan2_kernel)
                const uint32_t n2 = wave_shl1(n1) | (h1 & m63);
                const uint32_t Llo = __builtin_amdgcn_perm(n1, pl, 0x05040100u);
                const uint32_t Hlo = __builtin_amdgcn_perm(n1, pl, 0x07060302u);
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
                        acc |= (uint32_t)__builtin_popcount(((l ^ ql) | (h ^ qh)) & km) << (8 * b);
                    }
                    o[q] = acc;
                }
                const u32x4 ov = {o[0], o[1], o[2], o[3]};
                store_group<NTST, true>(dist + ((r0 + u) << 10) + 16 * lane, ov);
            }
            cur = nx[m - 1 < (unsigned)U ? m - 1 : 0]; // the planes of round r0 + m: the next trip's first round
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
        dist[i] = (uint8_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
}

