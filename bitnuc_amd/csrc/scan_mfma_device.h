// scan_mfma_device.h -- BASELINE config 5 (sliding k-mer pack + Hamming distance to a query) as a contraction on the matrix cores.
//
// dist[j] = hdist_scalar(as_2bit(ref[j .. j+k]), query, k)   (packing/naive.rs:3-20 o hamming/scalar.rs:11-48)
//         = sum_i [ref[j+i] != q[i]] = sum_i sum_c onehot(ref[j+i])[c] * (1 - onehot(q[i])[c])
// is a Toeplitz product of the reference's one-hot code with a constant built from the query: exact in any format that holds 0 and 1.
// pack / unpack stay integer bit-twiddling (no contraction there); THIS row is one, and kmer_scan2_kernel spends 9.4 vector
// instructions per window on it (VALU-issue bound, clock-sensitive: DESIGN 3.4).  Here the VALU only builds the one-hot operand
// (two v_perm per ASCII dword) and packs the result; the 31 compare-and-add steps per window run on the matrix pipe.
//
// Instruction: v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands -- 32 x 32 outputs, K = 64 nibbles per instruction, 32 cycles
// per SIMD.  fp4 because a base is then 16 bits (4 channels x 4 bits, 1.0 = 0b0010) = one byte from each of two 8-entry v_perm LUTs
// keyed on b & 7 (the codec's LUT index); i8 would need four.  Layout (tools/exp/mfma_fp4_probe.hip checks it on the device with
// exact data): A lane l = row l & 31, K-block l >> 5; B lane l = column l & 31, K-block l >> 5; D lane l register r = column l & 31,
// row (r & 3) + 8 (r >> 2) + 4 (l >> 5).
//
// WHAT SHIPS (this file; the product library instantiates exactly these two -- the tilings, operand, pack and threshold forms that lost their A/B are in
// evidence/scan_mfma_evidence.h, compiled into the evidence build only: profiles/r05_ab_*.txt):
//   kmer_scan_seg_mfma_kernel<POLICY 3, U 4, BLOCK 64>  the distance bytes (workgroups of one wave): a column is a SEGMENT of 32 consecutive windows, a row one of its 32 shifts (four MFMAs per 1024
//                                              windows), two v_permlane32_swap put the packed results in store order, one trip of four rounds per wave
//   kmer_count3_mfma_kernel<U 4, nt loads>     the fused count of d <= tau: the same segments with THREE channels per base (three MFMAs per 1024 windows), the
//                                              threshold inside the product, a bounded grid with a ticketed reduction
//
// Common to both: a wave round is 1 KiB of windows at a 1 KiB aligned offset; lane l loads its natural 16 bytes at 16 l, expands them ONCE and writes the operands to
// a wave-private LDS strip, from which lane (n = l & 31, h = l >> 5) reads back the 16-byte operand of each K-step (aligned, conflict-free: even and odd 16-byte
// groups live in regions 16 banks apart).  A trip is U consecutive rounds + the 32-byte halo after them.  The query's side of the product is built on the host
// (kmer.hip) and passed BY VALUE in the kernel arguments (a hipGraph node keeps its own copy), together with the values the accumulators start at.
//
// Packing 16 f32 results into 16 bytes costs 2 instructions per 4 windows instead of 4: the accumulator starts at 2^23 (the integer d then sits in the low mantissa
// bits) and A's rows carry the E8M0 block scale 2^(8 (m & 3)) for m & 3 < 3, so three results OR together into bytes 0-2 and a v_perm drops the fourth into byte 3.
// Everything is an integer below 2^24: exact.  The count's threshold uses the same idea with 6-bit fields (kmer_count3_mfma_kernel).
// Invalid bytes: the lane's own 16 bytes against a second v_perm LUT on the same index (the upper-case byte the index stands for: x ^ t is 0 or the case bit for a
// valid byte), OR-ed over the trip and tested once; their one-hot is all zero, the call fails with INVALID_BASE anyway.
// Why four (three) matrix instructions and not the six of the tiling that shipped first: the matrix pipe's POWER is what makes a queue that starts on an idle chip
// dip (profiles/r05_ablate_count_parts.txt); with four the scan runs at the HBM plateau from its first launch (profiles/r05_ab_scan_seg.txt).
#pragma once
#include "device_prims.h"
#include "kmer_device.h" // wave_shl1

namespace bitnuc_dev {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 8 bases (two ASCII dwords) -> 32 one-hot nibbles in the same order
__device__ __forceinline__ i32x8 onehot8(uint32_t x0, uint32_t x1) {
    const uint32_t s0 = x0 & 0x07070707u, s1 = x1 & 0x07070707u; // A=1 C=3 T=4 G=7, case bit ignored
    i32x8 b = {0, 0, 0, 0, 0, 0, 0, 0};
    b[0] = (int)__builtin_amdgcn_perm(0u, 0x20000200u, s0);       // A -> 0x02, C -> 0x20
    b[1] = (int)__builtin_amdgcn_perm(0x02000020u, 0u, s0);       // G -> 0x02, T -> 0x20
    b[2] = (int)__builtin_amdgcn_perm(0u, 0x20000200u, s1);
    b[3] = (int)__builtin_amdgcn_perm(0x02000020u, 0u, s1);
    return b;
}

template <int U>
struct ScanTrip {
    u32x4 v[U][3]; // [round][shift]: shifts 1, 2 only with SHIFT 0
    u32x4 hv;      // SHIFT 1, 3: the 32 bytes after the trip's last round, lanes 0 and 1
    uint32_t hw[U][8]; // SHIFT 2: the 32 bytes after each round, wave-uniform
};

template <int U, int SHIFT, bool NTLD>
__device__ __forceinline__ void scan_trip_load(const uint8_t *__restrict__ ref, unsigned long long r0, unsigned long long rounds, unsigned lane, ScanTrip<U> &t) {
    const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U; // valid rounds (wave-uniform)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long r = (unsigned)u < m ? r0 + u : r0 + m - 1; // clamp: redundant but in bounds
        const uint8_t *p = ref + (r << 10) + 16 * lane;
        t.v[u][0] = load_group<NTLD, true>(p);
        if constexpr (SHIFT == 0) {
            t.v[u][1] = load_group<false, true>(p + 16);
            t.v[u][2] = load_group<false, true>(p + 32);
        }
        if constexpr (SHIFT == 2) {
            const uint32_t *hp = reinterpret_cast<const uint32_t *>(ref + ((r + 1) << 10));
#pragma unroll
            for (int i = 0; i < 8; ++i) t.hw[u][i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hp[i]);
        }
    }
    if constexpr (SHIFT == 1 || SHIFT >= 3) {
        t.hv = u32x4{0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
        if (lane < 2) t.hv = load_group<false, true>(ref + ((r0 + m) << 10) + 16 * lane);
    }
}

// The windows after the last whole round: one window per thread, byte loads (as the reference: naive.rs:3-20 then scalar.rs:33-47 per window)
template <bool COUNT>
__device__ __forceinline__ uint32_t scan_tail_windows(const uint8_t *__restrict__ ref, unsigned long long first, unsigned long long nwin, unsigned k, unsigned long long query, unsigned tau,
                                                     uint8_t *__restrict__ dist, unsigned long long *__restrict__ slot) {
    const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const unsigned long long gt = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    uint32_t hits = 0;
    for (unsigned long long i = first + gt; i < nwin; i += nthreads) {
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
    return hits;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The segment tiling with four channels per base (the scan; evidence: kmer_count_mfma_kernel).  D[m][n] = dist(window 32 n + m); the 63 positions a segment's
// windows cover are 4 K-steps of 16, nothing multiplies zeros: 4 MFMAs per 1024 windows.  Lane (n, h) of K-step j needs the one-hot operand of bases
// 32 n + 16 j + 8 h .. + 8: half h of the 16-byte group 2 n + j -- not the lane's own group, so all four operands come from the strip, which keeps the halves
// and the even / odd groups in separate regions (a K-step's 32 reads are then 32 consecutive 16-byte entries: conflict-free).
// w[delta + 8][4 j + i]: dword i of K-step j for the row with delta = m - 8 h (i = position - m only depends on it), built on the host (kmer.hip: count_mfma_table).
struct CountMfmaTable { uint32_t w[40][16]; float c[4]; }; // c[r & 3]: where result register r's accumulator starts

// ---------------------------------------------------------------------------------------------------------------------------------
// The distance bytes.  Lane (n, h) ends up holding windows 32 n + 8 q + 4 h + i (q = r >> 2, i = r & 3), i.e. after the 2^23-bias pack one dword per q with four
// consecutive distance bytes, and two v_permlane32_swap (lanes l and l + 32 exchange a register) give lane (n, 0) the dwords (q0, partner's q0, q1, partner's q1) =
// bytes 32 n .. 32 n + 15 and lane (n, 1) (partner's q2, q2, partner's q3, q3) = bytes 32 n + 16 .. 32 n + 31: one natural dwordx4 store per lane at 16 (2 n + h).
// One trip of U rounds per wave; the hardware dispatcher walks the trips (how every streaming kernel of this library runs fastest).  BLOCK: threads per workgroup --
// nothing is shared inside one (the strips are wave-private), and with one wave per workgroup 19 waves fit a CU's LDS instead of 16 (profiles/r05_ab_scan_block.txt).
template <int POLICY, int U, int BLOCK = kBlock>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
kmer_scan_seg_mfma_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query, uint8_t *__restrict__ dist,
                          unsigned long long *__restrict__ slot, const CountMfmaTable tab) {
    constexpr bool NTLD = (POLICY & 1) != 0, NTST = (POLICY & 2) != 0;
    // one (half, parity) region: 32 U entries + the halo's, padded so that the odd-parity region starts 16 banks (64 B mod 128) after the even one: a ds_write_b128
    // serves 8 consecutive lanes at a time = 4 even groups (64 B of region 0) + 4 odd ones (64 B of region 1), which must not share a bank
    constexpr int kRegion = (32 * U + 1) * 16 + 48;
    static_assert(kRegion % 128 == 64, "the two parities of one store must land 16 banks apart");
    __shared__ __attribute__((aligned(16))) uint8_t strips[BLOCK / 64][4 * kRegion];
    const unsigned long long nwin = n - k + 1;
    const unsigned long long rounds = n >= 1056 ? (n - 32) >> 10 : 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    uint8_t *strip = strips[wave_in_block()];
    const unsigned long long r0 = wave * U;
    if (r0 < rounds) {
        ScanTrip<U> cur;
        scan_trip_load<U, 3, NTLD>(ref, r0, rounds, lane, cur); // before the table: its loads overlap these
        const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U;
        const unsigned m32 = lane & 31u, hh = lane >> 5;
        i32x8 A[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A[j] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) A[j][i] = (int)tab.w[m32 + 8u - 8u * hh][4 * j + i];
        }
        const int scale_a = 127 + 8 * (int)((m32 & 3u) == 3u ? 0u : (m32 & 3u)); // E8M0: 2^(8 (row & 3)) for row & 3 < 3 (scan_mfma_emit's pack)
        f32x16 c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c0[i] = tab.c[i & 3]; // 2^23
        asm volatile("" : "+v"(c0)); // sixteen registers used as an untied C operand (a splat constant is re-materialised by 16 v_mov per round)
        const unsigned wr0 = (lane & 1u) * kRegion + 16u * (lane >> 1);
        const unsigned rd = hh * 2u * kRegion + 16u * m32;
        uint32_t trip_bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4 x = cur.v[u][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) trip_bad |= x[i] ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x[i] & 0x07070707u);
            const i32x8 e0 = onehot8(x.x, x.y), e1 = onehot8(x.z, x.w);
            *reinterpret_cast<u32x4 *>(strip + wr0 + 512 * u) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
            *reinterpret_cast<u32x4 *>(strip + wr0 + 512 * u + 2 * kRegion) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
        }
        if (lane < 2) { // the halo: groups 64 m and 64 m + 1 (after the last VALID round; in-order LDS: the later write wins over a clamped copy)
            const i32x8 e0 = onehot8(cur.hv.x, cur.hv.y), e1 = onehot8(cur.hv.z, cur.hv.w);
            *reinterpret_cast<u32x4 *>(strip + lane * kRegion + 512 * m) = u32x4{(uint32_t)e0[0], (uint32_t)e0[1], (uint32_t)e0[2], (uint32_t)e0[3]};
            *reinterpret_cast<u32x4 *>(strip + lane * kRegion + 512 * m + 2 * kRegion) = u32x4{(uint32_t)e1[0], (uint32_t)e1[1], (uint32_t)e1[2], (uint32_t)e1[3]};
        }
        if (__builtin_expect((trip_bad & 0xDFDFDFDFu) != 0u, 0)) { // some lane of the trip holds an invalid byte: find the round
#pragma unroll 1
            for (unsigned u = 0; u < m; ++u) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
        }
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
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (__float_as_uint on a copy: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
                const float d0 = acc[4 * q], d1 = acc[4 * q + 1], d2 = acc[4 * q + 2], d3 = acc[4 * q + 3];
                o[q] = __builtin_amdgcn_perm(__float_as_uint(d3), __float_as_uint(d0) | __float_as_uint(d1) | __float_as_uint(d2), 0x04020100u);
            }
            // v_permlane32_swap a, b: a's lanes 32-63 <-> b's lanes 0-31
            const auto s02 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
            store_group<NTST, true>(dist + ((r0 + u) << 10) + 16u * (2u * m32 + hh), u32x4{s02[0], s02[1], s13[0], s13[1]});
        }
    }

    scan_tail_windows<false>(ref, rounds << 10, nwin, k, query, 0u, dist, slot);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The fused count with THREE channels per base: three MFMAs per 1024 windows instead of four.  [b != q] is affine in a 3-channel code (A, C, G one-hot; T = 0):
// 1 - x_q for q in {A, C, G}, x_A + x_C + x_G for q = T, so d = #(q_i != T) + sum of (-1 | +1) entries times x.  63 positions x 3 channels = 189 nibbles fit the
// 192 of three K-steps, and because the ORDER of (position, channel) pairs inside the K dimension is free, every operand is an aligned 16-byte piece of one of two
// arrays the strip holds: the (A, C) byte of every base -- the low-LUT output, one v_perm per ASCII dword -- and the G nibble of every base, packed two to
// a byte (one v_perm per ASCII dword, one v_lshl_or per two: byte t of a dword holds bases t and t + 4).  K-step 0 / 1: lane (n, h) reads the (A, C) bytes of group 2 n + h / 2 n + 2 + h (even
// and odd groups in separate regions, 16 banks apart, as above); K-step 2: the G nibbles of positions 32 h .. 32 h + 31 of its segment.  Per round: three
// ds_read_b128 instead of four, ds_write_b128 + ds_write_b64 instead of two ds_write_b128, two more vector instructions for the nibble packing, a quarter fewer
// matrix instructions -- whose power is what lowers the clock (profiles/r05_ablate_count_parts.txt).  An invalid byte reads as T; the call fails anyway.
// w[lane][4 s + i]: the lane's 16-byte operand of K-step s (built on the host: kmer.hip count3_mfma_table); c as CountMfmaTable's.
// The threshold is inside the product: the entries are signed and A's rows carry the E8M0 scale 2^(6 j), j = row & 3 < 3, the accumulator starts at
// 2^23 + (32 + tau - #(q_i != T)) 2^(6 j): a result's mantissa holds the 6-bit field 32 + tau - d of its row, whose top bit says d <= tau, and three rows OR into
// one register.  Row j = 3 (scale 2, start 2 #(q_i != T) - 2 tau - 1) holds 2 d - 2 tau - 1: an odd number below 64 -- six significant bits, so mantissa bits 17 and
// below are zero and its SIGN says d <= tau.  (x | b3) & 0x80020820 then has one bit per hit of four windows: v_or3 + v_bitop3 + v_bcnt (which accumulates) per
// four windows = 12 vector instructions per round and none on the scalar unit (sixteen v_cmp + s_bcnt1 + s_add cost 16 + 32, and the scalar unit is shared by the
// CU's four SIMDs); every partial sum is an integer below 2^24: exact.  tau >= k (no window can miss) gets an all-zero table.
// A bounded grid (one arrival per workgroup at the accumulator's ticket) whose waves walk trips; the NEXT trip's loads are issued as soon as this trip's bytes
// are in the strip, into the same registers, and fly during the matrix phase.
struct Count3MfmaTable { uint32_t w[64][12]; float c[4]; };

// 16 bases (four ASCII dwords) -> their 16 (A, C) bytes and their 16 G nibbles (the three-channel operands)
__device__ __forceinline__ void expand3(const u32x4 &x, u32x4 &ac, uint32_t &g0, uint32_t &g1) {
    uint32_t g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t si = x[i] & 0x07070707u;
        ac[i] = __builtin_amdgcn_perm(0u, 0x20000200u, si);      // A -> 0x02, C -> 0x20
        g[i] = __builtin_amdgcn_perm(0x02000000u, 0u, si);        // G -> 0x02
    }
    g0 = (g[1] << 4) | g[0]; // byte t: low nibble = base t, high nibble = base t + 4 (the order inside the K dimension is free: the host's table follows it)
    g1 = (g[3] << 4) | g[2]; // ... bases 8 + t and 12 + t
}


template <int U, bool NTLD>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
kmer_count3_mfma_kernel(const uint8_t *__restrict__ ref, unsigned long long n, unsigned k, unsigned long long query, unsigned tau,
                        unsigned long long *__restrict__ result, unsigned long long *__restrict__ total /* zero between launches */,
                        unsigned *__restrict__ ticket, unsigned long long *__restrict__ slot, const Count3MfmaTable tab) {
    constexpr int kAc = (32 * U + 1) * 16 + 48; // one parity's (A, C) entries of a trip + the halo's; the odd region starts 16 banks after the even one
    static_assert(kAc % 128 == 64, "the two parities of one store must land 16 banks apart");
    constexpr int kG = (32 * U + 1) * 16;       // G nibbles: 16 bytes per 32 positions
    __shared__ __attribute__((aligned(16))) uint8_t strips[kBlock / 64][2 * kAc + kG];
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
    i32x8 A[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        A[j] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) A[j][i] = (int)tab.w[lane][4 * j + i];
    }
    // A use of the table's registers BEFORE the loop.  The table arrives by global loads (a lane-varying index into the kernel arguments); left pending into the
    // loop, they make the compiler wait for vmcnt(0) at the first MFMA of EVERY trip -- i.e. for the next trip's loads, issued a few instructions earlier, whose
    // whole point is to fly during the matrix phase.
    asm volatile("" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2]), "+v"(A[1][3]),
                      "+v"(A[2][0]), "+v"(A[2][1]), "+v"(A[2][2]), "+v"(A[2][3]));
    uint32_t lane_hits = 0;
    const unsigned jrow = m32 & 3u;
    const int scale_a = 127 + (jrow == 3u ? 1 : 6 * (int)jrow);
    f32x16 c0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c0[i] = tab.c[i & 3];
    asm volatile("" : "+v"(c0)); // sixteen registers used as an untied C operand
    const unsigned wr_ac = (lane & 1u) * kAc + 16u * (lane >> 1); // the lane's own group l of round u: + 512 u
    const unsigned wr_g = 2u * kAc + 8u * lane;                   // ... its 16 G nibbles: + 512 u
    const unsigned rd_ac = hh * kAc + 16u * m32;                  // K-step s < 2 of round u: + 16 s + 512 u
    const unsigned rd_g = 2u * kAc + 16u * (m32 + hh);            // K-step 2: + 512 u

    while (r0 < rounds) {
        const unsigned m = rounds - r0 < (unsigned long long)U ? (unsigned)(rounds - r0) : (unsigned)U;
        const unsigned long long rn = r0 + nwaves * U;
        wave_lds_fence(); // the previous trip's readers are done
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
        if (lane < 2) { // the halo: groups 64 m and 64 m + 1 (after the last VALID round; in-order LDS: the later write wins over a clamped copy)
            u32x4 ac;
            uint32_t g0, g1;
            expand3(cur.hv, ac, g0, g1);
            *reinterpret_cast<u32x4 *>(strip + lane * kAc + 512 * m) = ac;
            *reinterpret_cast<u32x2 *>(strip + 2 * kAc + 512 * m + 8 * lane) = u32x2{g0, g1};
        }
        if (__builtin_expect((trip_bad & 0xDFDFDFDFu) != 0u, 0)) { // some lane of the trip holds an invalid byte: find the round
#pragma unroll 1
            for (unsigned u = 0; u < m; ++u) rescan_bytes(ref, ((r0 + u) << 10) + 16 * lane, 16, slot);
        }
        if (rn < rounds) scan_trip_load<U, 3, NTLD>(ref, rn, rounds, lane, cur); // cur's bytes are in the strip: its registers take the next trip
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
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (__float_as_uint on a copy: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
                const float d0 = acc[4 * q], d1 = acc[4 * q + 1], d2 = acc[4 * q + 2], d3 = acc[4 * q + 3];
                const uint32_t x = __float_as_uint(d0) | __float_as_uint(d1) | __float_as_uint(d2);
                lane_hits += (uint32_t)__builtin_popcount((x | __float_as_uint(d3)) & 0x80020820u);
            }
        }
        r0 = rn;
    }

    uint32_t tail_hits = scan_tail_windows<true>(ref, rounds << 10, nwin, k, query, tau, nullptr, slot);
    tail_hits += lane_hits;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tail_hits += __shfl_xor(tail_hits, off);
    __shared__ uint32_t part[kBlock / 64];
    if (lane == 0) part[threadIdx.x >> 6] = tail_hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (unsigned i = 0; i < (blockDim.x >> 6); ++i) s += part[i];
        if (s) add_performed(total, s);
        if (draw_last_ticket(ticket)) *result = atomicExch(total, 0ull);
    }
}

#ifdef BITNUC_SWEEP_VARIANTS
#include "evidence/scan_mfma_evidence.h" // the tilings / operand / pack / threshold forms that lost their A/B: evidence build only
#endif

} // namespace bitnuc_dev
