// mfma_fp4_probe.hip -- what v_mfma_scale_f32_32x32x64_f8f6f4 does with fp4 (E2M1) operands, checked with exact integer data
// before the scan kernel (csrc/scan_mfma_device.h) relies on it (cdna_hip_programming.md 3: "check the map with exact integer data").
// Hypothesis H:
//   A lane l holds row l & 31, K-block l >> 5 (32 nibbles in 4 dwords); B lane l holds column l & 31, K-block l >> 5;
//   nibble e of A's K-block pairs with nibble e of B's K-block (same register bit position <-> same k);
//   scale_a (E8M0 byte 0 of the lane's scale register) multiplies that lane's row x K-block; scale_b likewise;
//   D lane l, register r = C + sum, at column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// The program fills A and B with random fp4 values from {0, 1.0} (and a second pass from {0, 0.5, 1, 1.5, 2, 3, 4, 6}), random
// row scales 2^{0, 8, 16}, C = 2^23, runs ONE instruction and compares all 1024 outputs bit for bit with the host's sum under H.
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/mfma_fp4_probe.hip -o /tmp/mfma_fp4_probe && /tmp/mfma_fp4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const uint32_t *a, const uint32_t *b, const uint32_t *sa, const uint32_t *sb, float c, float *d) {
    const int l = threadIdx.x;
    i32x8 A = {(int)a[4 * l], (int)a[4 * l + 1], (int)a[4 * l + 2], (int)a[4 * l + 3], 0, 0, 0, 0};
    i32x8 B = {(int)b[4 * l], (int)b[4 * l + 1], (int)b[4 * l + 2], (int)b[4 * l + 3], 0, 0, 0, 0};
    f32x16 C;
    for (int i = 0; i < 16; ++i) C[i] = c;
    f32x16 D = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 4, 4, 0, (int)sa[l], 0, (int)sb[l]);
    for (int i = 0; i < 16; ++i) d[16 * l + i] = D[i];
}

static const float kFp4[16] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f, -0.f, -.5f, -1.f, -1.5f, -2.f, -3.f, -4.f, -6.f};

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 20); }

int main() {
    uint32_t ha[256], hb[256], hsa[64], hsb[64];
    float hd[1024];
    uint32_t *da, *db, *dsa, *dsb;
    float *dd;
    if (hipMalloc(&da, sizeof ha) != hipSuccess) { printf("no device\n"); return 2; }
    hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dd, sizeof hd);
    int fails = 0;
    for (int pass = 0; pass < 4; ++pass) {
        // pass 0: {0, 1} x {0, 1}, row scales 2^{0,8,16}, C = 2^23 (the scan kernel's use); 1: all eight magnitudes, no scale; 2: scale_b too; 3: {0,1}, C = 0
        for (int i = 0; i < 256; ++i) {
            uint32_t wa = 0, wb = 0;
            for (int e = 0; e < 8; ++e) {
                const uint32_t na = pass == 1 || pass == 2 ? rnd() & 7 : (rnd() & 1 ? 2 : 0), nb = pass == 1 || pass == 2 ? rnd() & 7 : (rnd() & 1 ? 2 : 0);
                wa |= na << (4 * e); wb |= nb << (4 * e);
            }
            ha[i] = wa; hb[i] = wb;
        }
        int ea[32], eb[32]; // exponent of row i / column j (the same for both K-blocks of it, as the kernel sets it)
        for (int i = 0; i < 32; ++i) { ea[i] = pass == 1 ? 0 : 8 * (int)(rnd() % 3); eb[i] = pass == 2 ? (int)(rnd() % 3) : 0; }
        for (int l = 0; l < 64; ++l) { hsa[l] = 127 + ea[l & 31] | 0xAABBCC00u /* bytes 1-3 must be ignored with op_sel 0 */; hsb[l] = 127 + eb[l & 31]; }
        const float c = pass == 3 ? 0.f : 8388608.f;
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(da, db, dsa, dsb, c, dd);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int row = 0; row < 32; ++row)
            for (int col = 0; col < 32; ++col) {
                double s = 0;
                for (int h = 0; h < 2; ++h)
                    for (int e = 0; e < 32; ++e) {
                        const uint32_t na = (ha[4 * (row + 32 * h) + (e >> 3)] >> (4 * (e & 7))) & 15, nb = (hb[4 * (col + 32 * h) + (e >> 3)] >> (4 * (e & 7))) & 15;
                        s += (double)kFp4[na] * kFp4[nb];
                    }
                const float want = (float)((double)c + s * (double)(1u << ea[row]) * (double)(1u << eb[col]));
                const int r = (row & 3) + 4 * (row >> 3), lane = col + 32 * ((row >> 2) & 1);
                const float got = hd[16 * lane + r];
                if (memcmp(&want, &got, 4) != 0) {
                    if (bad < 8) printf("  pass %d row %d col %d: want %.1f (0x%08x) got %.1f (0x%08x)\n", pass, row, col, want, *(const uint32_t *)&want, got, *(const uint32_t *)&got);
                    ++bad;
                }
            }
        printf("pass %d: %s (%d of 1024 outputs differ from hypothesis H)\n", pass, bad ? "FAIL" : "ok", bad);
        fails += bad != 0;
    }
    if (fails) { // single-element responses: where does A[lane la, nibble 0] x B[all ones] land, and which B nibble pairs with A nibble e of lane 0?
        for (int la = 0; la < 64; la += 9) {
            memset(ha, 0, sizeof ha); ha[4 * la] = 2;
            for (int i = 0; i < 256; ++i) hb[i] = 0x22222222u;
            for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
            hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
            hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
            probe<<<1, 64>>>(da, db, dsa, dsb, 0.f, dd);
            hipDeviceSynchronize();
            hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
            printf("A lane %d nibble 0 x B ones: nonzero at (lane,reg):", la);
            int shown = 0;
            for (int i = 0; i < 1024 && shown < 6; ++i) if (hd[i] != 0.f) { printf(" (%d,%d)=%.0f", i >> 4, i & 15, hd[i]); ++shown; }
            printf("\n");
        }
    }
    printf(fails ? "PROBE FAIL\n" : "PROBE OK: hypothesis H holds\n");
    return fails ? 1 : 0;
}
