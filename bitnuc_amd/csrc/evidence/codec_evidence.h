// codec_evidence.h -- EVIDENCE BUILD ONLY (-DBITNUC_SWEEP_VARIANTS): bulk-codec formulations that lost their A/B
// (profiles/NARRATIVE_r01_r03.md 3.1-3.2, profiles/README.md).  Included at the end of codec_device.h, inside namespace bitnuc_dev.
// Nothing in the product library instantiates or even sees this file.
#pragma once

// ---------------------------------------------------------------------------------
// encode, 16-byte stores by a register quad transpose (round 3, evidence build)
// ---------------------------------------------------------------------------------
// encode_kernel's stores are 4 bytes per lane (256 B per wave-instruction).  Here a wave owns 256 consecutive groups for 4
// rounds (round u: lane l loads group 64 u + l, one contiguous KiB per instruction, as before); the 4 x 4 matrix that the four
// lanes of a QUAD hold after the four rounds is transposed in registers (two DPP quad_perm butterflies, no LDS), after which
// lane 4 q + u owns the four consecutive results 64 u + 4 q .. + 3 and stores them as ONE dwordx4: a quarter of the store
// instructions, each wave-instruction still covering the same contiguous KiB (in a permuted lane order).
__device__ __forceinline__ uint32_t quad_xor1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true); }

template <int BLOCK, bool NTLD, bool NTST, bool XCD>
__global__ void __launch_bounds__(BLOCK)
encode_quad_kernel(const uint8_t *__restrict__ seq, uint32_t *__restrict__ out32, unsigned long long len,
                   unsigned long long *__restrict__ slot) {
    constexpr unsigned long long TILE = (unsigned long long)BLOCK * 4;
    const unsigned long long n16 = len >> 4;
    const unsigned long long full_tiles = n16 / TILE;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool odd1 = (lane & 1u) != 0u, odd2 = (lane & 2u) != 0u;
    for (unsigned long long tile = first_tile<XCD>(blockIdx.x, gridDim.x); tile < full_tiles; tile += gridDim.x) {
        const unsigned long long gw = tile * TILE + (unsigned long long)wave * 256;
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load_group<NTLD, true>(seq + ((gw + u * 64 + lane) << 4));
        uint32_t bad = 0, a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = enc16(v[u], bad);
        if (__builtin_expect(residue_is_bad(bad), 0)) {
#pragma unroll 1
            for (int u = 0; u < 4; ++u) rescan_bytes(seq, (gw + u * 64 + lane) << 4, 16, slot);
        }
        // transpose inside each quad: b[j] on lane (4 q + u) = a[u] of lane (4 q + j)
        // (every DPP move is pinned in wave-uniform control flow before the selects: hipcc may turn `c ? dpp(x) : y` into a branch
        //  that runs the DPP move with the other lanes masked off, and a DPP read from a disabled lane returns 0 -- kmer_scan2_kernel)
        uint32_t x0 = quad_xor1(a[0]), x1 = quad_xor1(a[1]), x2 = quad_xor1(a[2]), x3 = quad_xor1(a[3]);
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        const uint32_t s0 = odd1 ? x1 : a[0], s1 = odd1 ? a[1] : x0;
        const uint32_t s2 = odd1 ? x3 : a[2], s3 = odd1 ? a[3] : x2;
        uint32_t y0 = quad_xor2(s0), y1 = quad_xor2(s1), y2 = quad_xor2(s2), y3 = quad_xor2(s3);
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
        u32x4 r;
        r.x = odd2 ? y2 : s0;
        r.z = odd2 ? s2 : y0;
        r.y = odd2 ? y3 : s1;
        r.w = odd2 ? s3 : y1;
        store_group<NTST, true>(reinterpret_cast<uint8_t *>(out32 + gw + 64 * (lane & 3u) + 4 * (lane >> 2)), r);
    }
    if (blockIdx.x == (unsigned)(full_tiles % gridDim.x)) encode_tail<BLOCK, true>(seq, out32, len, full_tiles * TILE, slot);
}

// ---------------------------------------------------------------------------------
// encode, lane-per-base + wavefront ballot (the formulation north_star sketches)
// ---------------------------------------------------------------------------------
// One lane = one base: a wave loads 64 consecutive bytes (1 B/lane), each lane derives its
// 2-bit code, two __ballot()s turn the wave's code bits into two 64-bit planes (wave-uniform),
// a Morton interleave on the scalar unit merges the planes into the two u64 words of the 64
// bases, and a third ballot of the validity predicate gives the first invalid lane by ctz.
// Kept as a selectable variant for the record: at 1 B per lane per load it moves 16x fewer
// bytes per memory instruction than encode_kernel and measured far below it (profiles/).
__device__ __forceinline__ unsigned long long morton_spread32(unsigned long long x) { // bit i -> bit 2i, i < 32
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
encode_ballot_kernel(const uint8_t *__restrict__ seq, unsigned long long *__restrict__ out, unsigned long long len,
                     unsigned long long *__restrict__ slot) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long wave = ((unsigned long long)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * kBlock) >> 6;
    const unsigned long long nitems = (len + 63) >> 6; // 64 bases = 2 words per wave item
    for (unsigned long long it0 = wave * UNROLL; it0 < nitems; it0 += nwaves * UNROLL) {
        uint32_t b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned long long i = ((it0 + u) << 6) + lane;
            b[u] = i < len ? seq[i] : (uint32_t)'A'; // past the end: code 0, valid
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned long long base = (it0 + u) << 6;
            if (base >= len) break; // wave-uniform
            const uint32_t c = code_of(b[u]);
            const unsigned long long p0 = __ballot(c & 1u), p1 = __ballot(c >> 1);
            const unsigned long long invalid = __ballot(!valid_base(b[u]));
            if (invalid && lane == (unsigned)__builtin_ctzll(invalid)) latch_bad(slot, base + lane, b[u]); // the first invalid lane reports its own byte
            const unsigned long long w0 = morton_spread32(p0) | (morton_spread32(p1) << 1);
            const unsigned long long w1 = morton_spread32(p0 >> 32) | (morton_spread32(p1 >> 32) << 1);
            if (lane == 0) {
                out[base >> 5] = w0;
                if (base + 32 < len) out[(base >> 5) + 1] = w1;
            }
        }
    }
}


// ---------------------------------------------------------------------------------
// decode, 8-byte loads (round 2)
// ---------------------------------------------------------------------------------
// decode_kernel's loads are 4 bytes per lane (256 B per wave-instruction); the read side is only a fifth of the traffic but its
// requests are small.  Here a lane loads one whole u64 word (512 B per wave-instruction, one load per 2 KiB of output); the
// wave's 64 words cross a wave-private LDS strip once (ds_write_b64 at [lane], ds_read_b32 at [lane] and [64 + lane]: both
// conflict-free) so that each of the two stores is still one contiguous 1 KiB span.  WAVES_PER_ROUND independent words per
// lane are in flight.  Only whole 2 KiB wave tiles; the caller finishes the tail with decode_kernel.
template <int BLOCK, int UNROLL, bool NTLD, bool NTST>
__global__ void __launch_bounds__(BLOCK)
decode_x2_kernel(const unsigned long long *__restrict__ words, uint8_t *__restrict__ out, unsigned long long n_tiles /* of 64 words */) {
    __shared__ unsigned long long strips[BLOCK / 64][UNROLL][64];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    constexpr unsigned long long PER_BLOCK = (unsigned long long)(BLOCK / 64) * UNROLL;
    const unsigned long long t0 = (unsigned long long)blockIdx.x * PER_BLOCK + (unsigned long long)wave * UNROLL;
    unsigned long long w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const unsigned long long t = t0 + u < n_tiles ? t0 + u : n_tiles - 1;
        const unsigned long long *p = words + t * 64 + lane;
        if constexpr (NTLD) w[u] = __builtin_nontemporal_load(p); else w[u] = *p;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) strips[wave][u][lane] = w[u];
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        if (t0 + u >= n_tiles) break; // wave-uniform
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(strips[wave][u]);
        const uint32_t h0 = s32[lane], h1 = s32[64 + lane];
        uint8_t *dst = out + (t0 + u) * 2048;
        store_group<NTST, true>(dst + 16 * lane, dec16(h0));
        store_group<NTST, true>(dst + 16 * (lane + 64), dec16(h1));
    }
}


// The every-window kernel's access shape with no arithmetic (tools/ab_window_shape.py): a wave reads 1 KiB and writes 8 KiB per
// round.  MAP 0: the wave's eight 1 KiB stores are consecutive (kmer_slide2_kernel); MAP 1: the four waves of a workgroup
// interleave their stores KiB by KiB inside the workgroup's 32 KiB block (what probe_fill_kernel does at 4 KiB).
template <bool NTST, int U, int MAP>
__global__ void __launch_bounds__(kBlock)
probe_win_shape_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, unsigned long long rounds) {
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (kBlock / 64) + wv;
    const unsigned long long nwaves = (unsigned long long)gridDim.x * (kBlock / 64);
    for (unsigned long long r0 = wave * U; r0 < rounds; r0 += nwaves * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + ((r0 + u < rounds ? r0 + u : rounds - 1) << 6) + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r0 + u >= rounds) break;
            u32x4 *base;
            unsigned step;
            if constexpr (MAP == 0) { base = dst + ((r0 + u) << 9) + lane; step = 64; }                     // 8 consecutive KiB
            else { base = dst + (((r0 + u) & ~3ull) << 9) + (((r0 + u) & 3ull) << 6) + lane; step = 256; } // KiB (r & 3) + 4 j of the 4-round block
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x4 t = v[u];
                t.x += (uint32_t)i;
                if constexpr (NTST) __builtin_nontemporal_store(t, base + step * i); else base[step * i] = t;
            }
        }
    }
}
