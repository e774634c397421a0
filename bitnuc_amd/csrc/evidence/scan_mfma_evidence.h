// scan_mfma_evidence.h -- EVIDENCE BUILD ONLY (-DBITNUC_SWEEP_VARIANTS): the forms of the matrix-core scan and fused count that lost their A/B (profiles/r05_ab_*.txt,
// DESIGN.md 3.4).  Included at the end of scan_mfma_device.h, inside namespace bitnuc_dev.  Nothing in the product library instantiates or even sees this file.
//   kmer_scan_mfma_kernel     the natural-layout tiling (six MFMAs per 1024 windows, results already in store order): round 5's first matrix-core form, with its
//                             operand (SHIFT), pack (PACK), trip (U) and grid (PERSIST) forms and the fused count on that tiling (COUNT)
//   kmer_count_mfma_kernel    the fused count on segments of 32 windows with FOUR channels per base (four MFMAs): the threshold compared per register (EMIT 0) or
//                             inside the product (1, 2) -- shipped until the three-channel form (kmer_count3_mfma_kernel) replaced it
//   kmer_scan_seg3_mfma_kernel the distance bytes with three channels per base (three MFMAs): no faster than the shipped four-channel scan
//   scan_count_finish_kernel  the second launch of the one-trip-per-wave fused count
#pragma once

constexpr int kScanPartials = 1024; // accumulators of the one-trip-per-wave fused count (context-owned, zero between calls)

// The query's side of the product, built on the host (kmer.hip: scan_mfma_table) and passed BY VALUE in the kernel arguments (1.5 KiB:
// a hipGraph node keeps its own copy).  w[rho][4 s + i]: K-step s = 2 b + e (load b, 8-position group e), dword i of the lane's 16-byte
// operand, for the row whose windows start rho bases into the lane's 16.  Byte t of dwords {0, 1} (2, 3) belongs to position
// 16 b + 8 e + t (+ 4): dword 0 / 2 holds channels A (low nibble) and C, dword 1 / 3 channels G and T; a nibble is 1.0 (0b0010)
// where the channel differs from the query's base at i = position - rho, 0 where it equals it or i is outside [0, k).
struct ScanMfmaTable { uint32_t w[17][24]; float c[4]; }; // row 16: zeros (SHIFT 6: the rows of the half a lane's K-block does not meet); c[r & 3]: where result register r's accumulator starts (PACK 1)

// SHIFT: where the two shifted operands (the next two lanes' 16 bytes) come from.
//   0 = two more global loads at +16 / +32 (round 5's first form: 3 KiB through the vector L1 per KiB; 360 us per 10^9 windows whatever the
//       VALU count -- with 40 KiB of rounds in flight per CU the 32 KiB L1 does not keep the lines, profiles/r05_ab_scan_mfma_v1_shifted_loads.txt)
//   1 = a wave-private LDS strip: the trip's U KiB + the 32-byte halo are written once (ds_write_b128), read back at +16 / +32
//   2 = DPP: wave_shl:1 per dword, lane 63 takes the halo dword from the scalar unit (s_load of the 32 bytes after the round)
//   3 = the ONE-HOT operands cross the strip instead of the bytes: every 16-byte group is expanded once (4 v_and + 8 v_perm) and
//       the six operands of a round are six ds_read_b128 -- a third of the expansion work of forms 0-2, which expand every group
//       three times (the count kernel is VALU-issue bound: each vector instruction per round costs 2 us per 10^9 windows)
//   4 = as 3, but a lane keeps the operands of its OWN 16 bytes in registers: four ds_read_b128 per round instead of six (the strip
//       costs 2 x 13 LDS cycles to write and 4 per read, MI355X_MICROARCH.md LDS table; 8 more registers per round of the trip)
//   6 = as 4 with less bookkeeping: the zero half of the block-diagonal A comes from a seventeenth, all-zero table row instead of 24 v_and per
//       trip, and the invalid-byte residue is OR-ed over the trip and tested once instead of once per round
//   5 = as 3 with the trip SOFTWARE-PIPELINED: rounds 0 and 1 are expanded up front, round u + 2 is expanded -- and round u - 1's results
//       are packed / counted -- in the same basic block as round u's six dependent MFMAs, so that the wave's vector work sits in the 24
//       issue cycles each 32-cycle MFMA leaves free instead of waiting for the chain to end (profiles/r05_pmc_scan_mfma.txt: in form 4
//       the vector ALUs are busy 59 % of the time and a wave waits 61 % of its cycles).  Whole trips only; a trip cut short by the end
//       of the input runs form 4's code.  One branch per trip for invalid bytes (after the trip), none inside it.
// PACK (how 16 f32 results become 16 bytes): 0 = v_cvt_pk_u8_f32 per window, accumulator starts at inline 0, no scales;
//   1 = the 2^23 bias + row scales (v_or3 + v_perm per 4 windows), the bias an untied C operand held in 16 registers;
//   2 = the same, the bias produced by a seventh instruction with constant operands (one nibble x ones, scale 2^23)
// PERSIST: true = a resident grid walks the trips (a wave builds its constant operand once; the next trip's loads are issued before the
//   current one is computed); false = one trip per wave, the hardware dispatcher walks the trips (how every streaming kernel of this
//   library runs fastest), the table loads overlap the trip's data loads.
// 16 results of a lane (windows 16 l .. 16 l + 15 of the round) -> the count of d <= tau, or 16 distance bytes
template <bool COUNT, int PACK, bool NTST>
__device__ __forceinline__ void scan_mfma_emit(const f32x16 &acc, float tauf, uint32_t &hits, uint8_t *dst) {
    if constexpr (COUNT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hits += (uint32_t)__builtin_popcountll(__ballot(acc[r] <= tauf));
    } else {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (PACK == 0) {
                uint32_t w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) w = __builtin_amdgcn_cvt_pk_u8_f32(acc[4 * q + j], j, w);
                o[q] = w;
            } else {
                // (__float_as_uint on a copy: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
                const float d0 = acc[4 * q], d1 = acc[4 * q + 1], d2 = acc[4 * q + 2], d3 = acc[4 * q + 3];
                o[q] = __builtin_amdgcn_perm(__float_as_uint(d3), __float_as_uint(d0) | __float_as_uint(d1) | __float_as_uint(d2), 0x04020100u);
            }
        }
        store_group<NTST, true>(dst, o);
    }
}

template <int POLICY, int U, bool COUNT, int PACK = 1, int SHIFT = 1, bool PERSIST = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) // >= 4 waves per SIMD (the strip already limits a CU to 16-20 waves): at most 128 registers
kmer_scan_mfma_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query, unsigned tau,
                      uint8_t *__restrict__ dist, unsigned long long *__restrict__ result, unsigned long long *__restrict__ total /* zero between launches */,
                      unsigned *__restrict__ ticket, unsigned long long *__restrict__ slot, const ScanMfmaTable tab) {
    constexpr bool NTLD = (POLICY & 1) != 0, NTST = (POLICY & 2) != 0;
    constexpr int kPlane = U * 1024 + 32;                    // a trip's bytes (SHIFT 1) or one of its two operand planes (SHIFT 3): U KiB + the halo
    constexpr bool PIPE = SHIFT == 5;
    constexpr bool LEAN = SHIFT == 6;
    constexpr bool ONEHOT = SHIFT == 3 || SHIFT == 4 || LEAN || PIPE, KEEP = SHIFT == 4 || LEAN; // (the pipelined form reads all six operands back: its registers go to the second accumulator)
    static_assert(!PIPE || U >= 2, "the pipelined trip expands two rounds ahead");
    constexpr int kStrip = ONEHOT ? 2 * kPlane : kPlane;
    constexpr bool LDS = SHIFT == 1 || ONEHOT;
    __shared__ __attribute__((aligned(16))) uint8_t strips[LDS ? kBlock / 64 : 1][LDS ? kStrip : 16];
    const unsigned long long nwin = n - k + 1;                              // host guarantees 1 <= k <= 32, n >= k
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;       // round r reads bytes [1024 r, 1024 r + 1056)
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    uint8_t *strip = strips[LDS ? wave_in_block() : 0];

    ScanTrip<U> cur;
    unsigned long long r0 = wave * U;
    if (r0 < rounds) scan_trip_load<U, SHIFT, NTLD>(ref, r0, rounds, lane, cur); // before the table: its loads overlap these

    // the wave's constant operand: row m = lane & 31 of K-block lane >> 5
    const unsigned m32 = lane & 31u, rho = (m32 & 3u) + 4u * (m32 >> 3);
    const uint32_t keep = ((m32 >> 2) & 1u) == (lane >> 5) ? ~0u : 0u; // A is block-diagonal: row half a only meets K-block a
    i32x8 A[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        A[s] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) A[s][i] = LEAN ? (int)tab.w[keep ? rho : 16u][4 * s + i] : (int)(tab.w[rho][4 * s + i] & keep);
    }
    if constexpr (PERSIST) { // the table's loads end BEFORE the loop: otherwise every trip waits for vmcnt(0) -- its own prefetch -- at its first MFMA (kmer_count_mfma_kernel's note)
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(A[s][0]), "+v"(A[s][1]), "+v"(A[s][2]), "+v"(A[s][3]));
    }
    constexpr bool BIAS = !COUNT && PACK != 0;
    const int scale_a = !BIAS ? 127 : 127 + 8 * (int)((m32 & 3u) == 3u ? 0u : (m32 & 3u)); // E8M0: 2^(8 (rho & 3)) for rho & 3 < 3
    f32x16 c0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c0[i] = BIAS && PACK == 1 ? tab.c[i & 3] : 0.f; // 2^23, or 2^23 + k 2^(8 (i & 3)) when the table counts matches DOWN from k (kmer.hip: scan_mfma_table)
    if constexpr (BIAS && PACK == 1) asm volatile("" : "+v"(c0)); // sixteen registers used as an untied C operand (a splat constant is re-materialised by 16 v_mov per round)
    const i32x8 bias_a = {lane < 32 ? 2 : 0, 0, 0, 0, 0, 0, 0, 0};                                   // PACK 2: nibble 0 of K-block 0 = 1.0, every row
    const i32x8 bias_b = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};              // ... x ones, scale 2^23
    const float tauf = (float)tau;
    const uint32_t m63 = lane == 63 ? ~0u : 0u;
    uint32_t hits = 0; // COUNT: wave-uniform until the tail

    while (r0 < rounds) {
        const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U; // valid rounds in this trip (wave-uniform)
        ScanTrip<U> nxt;
        const unsigned long long rn = r0 + nwaves * U;
        if constexpr (PERSIST) { if (rn < rounds) scan_trip_load<U, SHIFT, NTLD>(ref, rn, rounds, lane, nxt); }
        if constexpr (SHIFT == 1) {
            wave_lds_fence(); // the previous trip's readers are done
#pragma unroll
            for (int u = 0; u < U; ++u) *reinterpret_cast<u32x4 *>(strip + 1024 * u + 16 * lane) = cur.v[u][0];
            if (lane < 2) *reinterpret_cast<u32x4 *>(strip + 1024 * m + 16 * lane) = cur.hv; // after the last VALID round (a clamped copy may sit there: in-order LDS, the later write wins)
            wave_lds_fence();
        }
        i32x8 own[KEEP ? U : 1][2];
        if constexpr (PIPE) {
            if (m == (unsigned)U) { // a whole trip: the static schedule (wave-uniform branch)
                uint32_t badr[U];
                auto fill = [&](int u) { // round u of the trip: validity residue, one-hot operands of the lane's 16 bytes -> registers + both planes of the strip
                    const u32x4 x = cur.v[u][0];
                    uint32_t bad = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) bad |= x[i] ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x[i] & 0x07070707u);
                    badr[u] = bad;
                    const i32x8 e0 = onehot8(x.x, x.y), e1 = onehot8(x.z, x.w);
                    *reinterpret_cast<u32x4 *>(strip + 1024 * u + 16 * lane) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
                    *reinterpret_cast<u32x4 *>(strip + kPlane + 1024 * u + 16 * lane) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
                };
                auto fill_halo = [&]() {
                    if (lane < 2) {
                        const i32x8 e0 = onehot8(cur.hv.x, cur.hv.y), e1 = onehot8(cur.hv.z, cur.hv.w);
                        *reinterpret_cast<u32x4 *>(strip + 1024 * U + 16 * lane) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
                        *reinterpret_cast<u32x4 *>(strip + kPlane + 1024 * U + 16 * lane) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
                    }
                };
                wave_lds_fence(); // the previous trip's readers are done
                fill(0);
                fill(1);
                if constexpr (U == 2) fill_halo();
                f32x16 prev = c0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    wave_lds_fence(); // round u + 1's operands (or the halo) are in the strip
                    i32x8 B[6];
#pragma unroll
                    for (int s6 = 0; s6 < 6; ++s6) {
                        const u32x4 t = *reinterpret_cast<const u32x4 *>(strip + (s6 & 1) * kPlane + 1024 * u + 16 * lane + 16 * (s6 >> 1));
                        B[s6] = i32x8{(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
                    }
                    if (u + 2 < U) fill(u + 2);          // (compile-time after unrolling)
                    else if (u + 2 == U) fill_halo();
                    f32x16 acc = c0;
                    if constexpr (BIAS && PACK == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bias_a, bias_b, c0, 4, 4, 0, 127 + 23, 0, 127);
#pragma unroll
                    for (int s6 = 0; s6 < 6; ++s6) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[s6], B[s6], acc, 4, 4, 0, scale_a, 0, 127);
                    if (u > 0) scan_mfma_emit<COUNT, PACK, NTST>(prev, tauf, hits, dist + ((r0 + u - 1) << 10) + 16 * lane); // the previous round's results, beside this round's chain
                    prev = acc;
                }
                scan_mfma_emit<COUNT, PACK, NTST>(prev, tauf, hits, dist + ((r0 + U - 1) << 10) + 16 * lane);
                uint32_t any = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) any |= badr[u];
                if (__builtin_expect((any & 0xDFDFDFDFu) != 0u, 0)) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if ((badr[u] & 0xDFDFDFDFu) != 0u) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
                }
                if constexpr (!PERSIST) break;
                if (rn < rounds) cur = nxt;
                r0 = rn;
                continue;
            }
        }
        if constexpr (ONEHOT) {
            wave_lds_fence(); // the previous trip's readers are done
            uint32_t trip_bad = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4 x = cur.v[u][0];
                // validity of the lane's own 16 bytes, while they are in registers: a second LUT on the same index holds the upper-case byte
                // that index stands for (0xFF for the four indices no base has: their low bits never match), so x ^ t is 0 or the case bit
                uint32_t bad = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) bad |= x[i] ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x[i] & 0x07070707u);
                if constexpr (LEAN) trip_bad |= bad; // (a clamped copy repeats a round of this trip: nothing it could add)
                else if (__builtin_expect((bad & 0xDFDFDFDFu) != 0u && (unsigned)u < m, 0)) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
                const i32x8 e0 = onehot8(x.x, x.y), e1 = onehot8(x.z, x.w);
                if constexpr (KEEP) { own[u][0] = e0; own[u][1] = e1; }
                *reinterpret_cast<u32x4 *>(strip + 1024 * u + 16 * lane) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
                *reinterpret_cast<u32x4 *>(strip + kPlane + 1024 * u + 16 * lane) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
            }
            if (lane < 2) { // the halo after the last VALID round (a clamped copy may sit there: in-order LDS, the later write wins); validated by the round that owns it
                const i32x8 e0 = onehot8(cur.hv.x, cur.hv.y), e1 = onehot8(cur.hv.z, cur.hv.w);
                *reinterpret_cast<u32x4 *>(strip + 1024 * m + 16 * lane) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
                *reinterpret_cast<u32x4 *>(strip + kPlane + 1024 * m + 16 * lane) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
            }
            if constexpr (LEAN) {
                if (__builtin_expect((trip_bad & 0xDFDFDFDFu) != 0u, 0)) { // some lane of the trip holds an invalid byte: find the round
#pragma unroll 1
                    for (unsigned u = 0; u < m; ++u) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
                }
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((unsigned)u >= m) break; // wave-uniform
            const unsigned long long wb = (r0 + u) << 10;
            if constexpr (ONEHOT) {
                i32x8 B[6];
#pragma unroll
                for (int s6 = 0; s6 < 6; ++s6) {
                    if (KEEP && s6 < 2) { B[s6] = own[KEEP ? u : 0][s6]; continue; }
                    const u32x4 t = *reinterpret_cast<const u32x4 *>(strip + (s6 & 1) * kPlane + 1024 * u + 16 * lane + 16 * (s6 >> 1));
                    B[s6] = i32x8{(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
                }
                f32x16 acc = c0;
                if constexpr (BIAS && PACK == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bias_a, bias_b, c0, 4, 4, 0, 127 + 23, 0, 127);
#pragma unroll
                for (int s6 = 0; s6 < 6; ++s6) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[s6], B[s6], acc, 4, 4, 0, scale_a, 0, 127);
                scan_mfma_emit<COUNT, PACK, NTST>(acc, tauf, hits, dist + wb + 16 * lane);
                continue;
            }
            u32x4 sh[3];
            sh[0] = cur.v[u][0];
            if constexpr (SHIFT == 0) { sh[1] = cur.v[u][1]; sh[2] = cur.v[u][2]; }
            if constexpr (SHIFT == 1) {
                sh[1] = *reinterpret_cast<const u32x4 *>(strip + 1024 * u + 16 * lane + 16);
                sh[2] = *reinterpret_cast<const u32x4 *>(strip + 1024 * u + 16 * lane + 32);
            }
            if constexpr (SHIFT == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // lane 63 of a wave_shl is 0 (bound_ctrl): the halo goes in with one v_and_or (kmer_scan2_kernel's idiom)
                    sh[1][i] = wave_shl1(sh[0][i]) | (cur.hw[u][i] & m63);
                    sh[2][i] = wave_shl1(sh[1][i]) | (cur.hw[u][4 + i] & m63);
                }
            }
            uint32_t bad = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) bad |= __builtin_amdgcn_perm(0x42040453u, 0x41044004u, sh[0][i] & 0x07070707u) ^ (sh[0][i] & 0xD8D8D8D8u);
            if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(ref, wb + 16 * lane, 16, slot);
            f32x16 acc = c0;
            if constexpr (BIAS && PACK == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bias_a, bias_b, c0, 4, 4, 0, 127 + 23, 0, 127);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[2 * b], onehot8(sh[b].x, sh[b].y), acc, 4, 4, 0, scale_a, 0, 127);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[2 * b + 1], onehot8(sh[b].z, sh[b].w), acc, 4, 4, 0, scale_a, 0, 127);
            }
            scan_mfma_emit<COUNT, PACK, NTST>(acc, tauf, hits, dist + wb + 16 * lane);
        }
        if constexpr (!PERSIST) break;
        if (rn < rounds) cur = nxt; // (otherwise the loop ends: nothing was loaded into nxt)
        r0 = rn;
    }

    // tail: one window per thread, byte loads
    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const unsigned long long gt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    uint32_t tail_hits = 0;
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
        if constexpr (COUNT) tail_hits += d <= tau ? 1u : 0u;
        else dist[i] = (uint8_t)d;
    }
    if constexpr (COUNT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tail_hits += __shfl_xor(tail_hits, off);
        __shared__ uint32_t part[kBlock / 64];
        if (lane == 0) part[threadIdx.x >> 6] = hits + tail_hits;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long s = 0;
            for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += part[i];
            if constexpr (PERSIST) { // a resident grid: one arrival per workgroup at ONE accumulator + ticket, the last one publishes (device_prims.h)
                if (s) add_performed(total, s);
                if (draw_last_ticket(ticket)) *result = atomicExch(total, 0ull);
            } else {
                // one trip per wave = tens of thousands of workgroups: two returning atomics each on one address serialise at ~6 ns apiece
                // (0.77 ms per 10^9 windows, profiles/r05_ab_scan_mfma_v5*.txt).  Here a workgroup adds its count, fire-and-forget, to one of
                // kScanPartials accumulators; scan_count_finish_kernel -- the next launch on the stream -- sums them, zeroes them, publishes.
                if (s) atomicAdd(total + (blockIdx.x & (kScanPartials - 1)), s);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------

//
// EMIT: how 1024 f32 distances become a count.
//   0 = sixteen v_cmp_le_f32 into wave masks + s_bcnt1 + s_add each (round 5's first form: 16 vector and 32 scalar instructions per round; the
//       scalar unit is shared by the CU's four SIMDs, and its ~50 instructions per round cost as many issue slots as the ~55 vector ones)
//   1 = the threshold inside the product: A's entries are -1.0 and its rows carry the E8M0 scale 2^(6 j), j = row & 3 < 3, the accumulator starts at
//       2^23 + (32 + tau) 2^(6 j): a result's mantissa holds the 6-bit field 32 + tau - d of its row, whose top bit says d <= tau, and three rows OR
//       into one register.  Row j = 3 (entries +1.0, scale 2, start -(2 tau + 1)) holds 2 d - 2 tau - 1: an odd number below 64 -- six significant
//       bits, so mantissa bits 17 and below are zero and its SIGN says d <= tau.  (x | b3) & 0x80020820 then has one bit per hit of four windows:
//       v_or3 + v_bitop3 + v_bcnt (which accumulates) per four windows = 12 vector instructions per round and none on the scalar unit; every
//       partial sum is an integer below 2^24: exact.  The invalid-byte residue is OR-ed over the trip and tested once.
//   2 = as 1, and the NEXT trip's loads are issued after this trip's bytes have been expanded into the strip, into the same registers (no second
//       set of registers, no copy at the end of a trip; the matrix phase of the trip hides the loads)
template <int U, bool NTLD, int EMIT>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
kmer_count_mfma_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query, unsigned tau,
                       unsigned long long *__restrict__ result, unsigned long long *__restrict__ total /* zero between launches */,
                       unsigned *__restrict__ ticket, unsigned long long *__restrict__ slot, const CountMfmaTable tab) {
    // one (half, parity) region: 32 U entries + the halo's, padded so that the odd-parity region starts 16 banks (64 B mod 128) after the even one: a
    // ds_write_b128 serves 8 consecutive lanes at a time = 4 even groups (64 B of region 0) + 4 odd ones (64 B of region 1), which must not share a bank
    // (round 5's first padding, + 64 B, put them 20 banks apart: SQ_LDS_BANK_CONFLICT = 30 % of the LDS cycles, profiles/r05_pmc_scan_mfma_shipped_forms.txt)
    constexpr int kRegion = (32 * U + 1) * 16 + 48;
    static_assert(kRegion % 128 == 64, "the two parities of one store must land 16 banks apart");
    __shared__ __attribute__((aligned(16))) uint8_t strips[kBlock / 64][4 * kRegion];
    const unsigned long long nwin = n - k + 1;
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    uint8_t *strip = strips[wave_in_block()];

    ScanTrip<U> cur;
    unsigned long long r0 = wave * U;
    if (r0 < rounds) scan_trip_load<U, 3, NTLD>(ref, r0, rounds, lane, cur);
    const unsigned m32 = lane & 31u, hh = lane >> 5;
    i32x8 A[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[j] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) A[j][i] = (int)tab.w[m32 + 8u - 8u * hh][4 * j + i];
    }
    if constexpr (EMIT != 0) {
        // a use of the table's registers BEFORE the loop.  The table arrives by global loads (a lane-varying index into the kernel arguments);
        // left pending into the loop, they make the compiler wait for vmcnt(0) at the first MFMA of EVERY trip -- i.e. for the next trip's
        // loads, issued a few instructions earlier, whose whole point is to fly during the matrix phase (round 5's first form did that)
        asm volatile("" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2]), "+v"(A[1][3]),
                          "+v"(A[2][0]), "+v"(A[2][1]), "+v"(A[2][2]), "+v"(A[2][3]), "+v"(A[3][0]), "+v"(A[3][1]), "+v"(A[3][2]), "+v"(A[3][3]));
    }
    const float tauf = (float)tau;
    uint32_t hits = 0;      // EMIT 0: wave-uniform
    uint32_t lane_hits = 0; // EMIT 1, 2: per lane
    const unsigned jrow = m32 & 3u;
    const int scale_a = EMIT == 0 ? 127 : 127 + (jrow == 3u ? 1 : 6 * (int)jrow);
    f32x16 c0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c0[i] = EMIT == 0 ? 0.f : tab.c[i & 3]; // kmer.hip: count_mfma_table
    if constexpr (EMIT != 0) asm volatile("" : "+v"(c0)); // sixteen registers used as an untied C operand (scan_mfma_emit's note)
    // where group g of the trip lives: region (half e, parity g & 1), entry g >> 1
    const unsigned wr0 = (lane & 1u) * kRegion + 16u * (lane >> 1);                 // the lane's own group l of round u: + 2 kRegion e + 512 u
    const unsigned rd = hh * 2u * kRegion + 16u * m32;                              // lane (n, h), K-step j of round u: + (j & 1) kRegion + 16 (32 u + (j >> 1))

    while (r0 < rounds) {
        const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U;
        ScanTrip<U> nxt;
        const unsigned long long rn = r0 + nwaves * U;
        if constexpr (EMIT != 2) { if (rn < rounds) scan_trip_load<U, 3, NTLD>(ref, rn, rounds, lane, nxt); }
        wave_lds_fence(); // the previous trip's readers are done
        uint32_t trip_bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4 x = cur.v[u][0];
            uint32_t bad = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) bad |= x[i] ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x[i] & 0x07070707u);
            if constexpr (EMIT == 0) {
                if (__builtin_expect((bad & 0xDFDFDFDFu) != 0u && (unsigned)u < m, 0)) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
            } else {
                trip_bad |= bad; // (a clamped copy repeats a round of this trip: nothing it could add)
            }
            const i32x8 e0 = onehot8(x.x, x.y), e1 = onehot8(x.z, x.w);
            *reinterpret_cast<u32x4 *>(strip + wr0 + 512 * u) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
            *reinterpret_cast<u32x4 *>(strip + wr0 + 512 * u + 2 * kRegion) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
        }
        if (lane < 2) { // the halo: groups 64 m and 64 m + 1 (after the last VALID round; in-order LDS: the later write wins over a clamped copy)
            const i32x8 e0 = onehot8(cur.hv.x, cur.hv.y), e1 = onehot8(cur.hv.z, cur.hv.w);
            *reinterpret_cast<u32x4 *>(strip + lane * kRegion + 512 * m) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
            *reinterpret_cast<u32x4 *>(strip + lane * kRegion + 512 * m + 2 * kRegion) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
        }
        if constexpr (EMIT != 0) {
            if (__builtin_expect((trip_bad & 0xDFDFDFDFu) != 0u, 0)) { // some lane of the trip holds an invalid byte: find the round
#pragma unroll 1
                for (unsigned u = 0; u < m; ++u) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
            }
        }
        if constexpr (EMIT == 2) { if (rn < rounds) scan_trip_load<U, 3, NTLD>(ref, rn, rounds, lane, cur); } // cur's bytes are in the strip: its registers take the next trip
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((unsigned)u >= m) break; // wave-uniform
            i32x8 B[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 t = *reinterpret_cast<const u32x4 *>(strip + rd + (j & 1) * kRegion + 16 * (32 * u + (j >> 1)));
                B[j] = i32x8{(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
            }
            f32x16 acc = c0;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[j], B[j], acc, 4, 4, 0, scale_a, 0, 127);
            if constexpr (EMIT == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hits += (uint32_t)__builtin_popcountll(__ballot(acc[r] <= tauf));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (__float_as_uint on a copy: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
                    const float d0 = acc[4 * q], d1 = acc[4 * q + 1], d2 = acc[4 * q + 2], d3 = acc[4 * q + 3];
                    const uint32_t x = __float_as_uint(d0) | __float_as_uint(d1) | __float_as_uint(d2);
                    lane_hits += (uint32_t)__builtin_popcount((x | __float_as_uint(d3)) & 0x80020820u);
                }
            }
        }
        if constexpr (EMIT != 2) { if (rn < rounds) cur = nxt; } // (otherwise the loop ends: nothing was loaded into nxt)
        r0 = rn;
    }

    // tail: one window per thread, byte loads
    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const unsigned long long gt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    uint32_t tail_hits = 0;
    for (unsigned long long i = (rounds << 10) + gt; i < nwin; i += nthreads) {
        unsigned long long w = 0;
        bool flagged = false;
        for (unsigned b = 0; b < k; ++b) {
            const uint32_t byte = ref[i + b];
            if (!valid_base(byte) && !flagged) { latch_bad(slot, i + b, byte); flagged = true; }
            w |= (unsigned long long)code_of(byte) << (2 * b);
        }
        const unsigned long long x = (w ^ query) & kmask;
        tail_hits += (uint32_t)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull) <= tau ? 1u : 0u;
    }
    tail_hits += lane_hits;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tail_hits += __shfl_xor(tail_hits, off);
    __shared__ uint32_t part[kBlock / 64];
    if (lane == 0) part[threadIdx.x >> 6] = hits + tail_hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += part[i];
        if (s) add_performed(total, s);
        if (draw_last_ticket(ticket)) *result = atomicExch(total, 0ull);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// kmer_scan_seg3_mfma_kernel: the distance bytes with THREE channels per base (kmer_count3_mfma_kernel's operands: three MFMAs per 1024 windows instead of the shipped
// scan's four).  d = #(q_i != T) + sum (-1 | +1) x: the rows carry the pack's scales 2^(8 j) and the accumulators start at 2^23 + #(q_i != T) 2^(8 j).
// Measured against the shipped four-channel form: profiles/r05_ab_scan_seg3.txt.
template <int POLICY, int U>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
kmer_scan_seg3_mfma_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query, uint8_t *__restrict__ dist,
                           unsigned long long *__restrict__ slot, const Count3MfmaTable tab) {
    constexpr bool NTLD = (POLICY & 1) != 0, NTST = (POLICY & 2) != 0;
    constexpr int kAc = (32 * U + 1) * 16 + 48;
    static_assert(kAc % 128 == 64, "the two parities of one store must land 16 banks apart");
    constexpr int kG = (32 * U + 1) * 16;
    __shared__ __attribute__((aligned(16))) uint8_t strips[kBlock / 64][2 * kAc + kG];
    const unsigned long long nwin = n - k + 1;
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    uint8_t *strip = strips[wave_in_block()];
    const unsigned long long r0 = wave * U;
    if (r0 < rounds) {
        ScanTrip<U> cur;
        scan_trip_load<U, 3, NTLD>(ref, r0, rounds, lane, cur);
        const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U;
        const unsigned m32 = lane & 31u, hh = lane >> 5;
        i32x8 A[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[j] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) A[j][i] = (int)tab.w[lane][4 * j + i];
        }
        const int scale_a = 127 + 8 * (int)((m32 & 3u) == 3u ? 0u : (m32 & 3u));
        f32x16 c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c0[i] = tab.c[i & 3];
        asm volatile("" : "+v"(c0));
        const unsigned wr_ac = (lane & 1u) * kAc + 16u * (lane >> 1), wr_g = 2u * kAc + 8u * lane;
        const unsigned rd_ac = hh * kAc + 16u * m32, rd_g = 2u * kAc + 16u * (m32 + hh);
        uint32_t trip_bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4 x = cur.v[u][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) trip_bad |= x[i] ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x[i] & 0x07070707u);
            u32x4 ac;
            uint32_t g0, g1;
            expand3(x, ac, g0, g1);
            *reinterpret_cast<u32x4 *>(strip + wr_ac + 512 * u) = ac;
            *reinterpret_cast<u32x2 *>(strip + wr_g + 512 * u) = u32x2{g0, g1};
        }
        if (lane < 2) {
            u32x4 ac;
            uint32_t g0, g1;
            expand3(cur.hv, ac, g0, g1);
            *reinterpret_cast<u32x4 *>(strip + lane * kAc + 512 * m) = ac;
            *reinterpret_cast<u32x2 *>(strip + 2 * kAc + 512 * m + 8 * lane) = u32x2{g0, g1};
        }
        if (__builtin_expect((trip_bad & 0xDFDFDFDFu) != 0u, 0)) {
#pragma unroll 1
            for (unsigned u = 0; u < m; ++u) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((unsigned)u >= m) break; // wave-uniform
            i32x8 B[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4 t = *reinterpret_cast<const u32x4 *>(strip + (j < 2 ? rd_ac + 16 * j : rd_g) + 512 * u);
                B[j] = i32x8{(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
            }
            f32x16 acc = c0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[j], B[j], acc, 4, 4, 0, scale_a, 0, 127);
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d0 = acc[4 * q], d1 = acc[4 * q + 1], d2 = acc[4 * q + 2], d3 = acc[4 * q + 3];
                o[q] = __builtin_amdgcn_perm(__float_as_uint(d3), __float_as_uint(d0) | __float_as_uint(d1) | __float_as_uint(d2), 0x04020100u);
            }
            const auto s02 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
            store_group<NTST, true>(dist + ((r0 + u) << 10) + 16u * (2u * m32 + hh), u32x4{s02[0], s02[1], s13[0], s13[1]});
        }
    }
    scan_tail_windows<false>(ref, rounds << 10, nwin, k, query, 0u, dist, slot);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the second launch of the one-trip-per-wave fused count (evidence build: that form lost its A/B): sum + re-arm the partial accumulators (stream
// order makes the first launch's atomics visible)
__global__ void __launch_bounds__(kScanPartials)
scan_count_finish_kernel(unsigned long long *__restrict__ partials, unsigned long long *__restrict__ result) {
    __shared__ unsigned long long part[kScanPartials / 64];
    unsigned long long v = partials[threadIdx.x];
    partials[threadIdx.x] = 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int i = 0; i < kScanPartials / 64; ++i) s += part[i];
        *result = s;
    }
}
