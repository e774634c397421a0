// Experiment: do unaligned ds_write_b32 / ds_read_b32 / ds_write_b128 work on gfx950 (HSA unaligned-access mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
__global__ void k(uint8_t* out, int off) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    // each lane writes 4 bytes at 5*lane + off (unaligned), pattern lane
    uint32_t v = 0x01010101u * (threadIdx.x + 1);
    *reinterpret_cast<u32_u*>(lds + 5 * threadIdx.x + off) = v;
    __syncthreads();
    // 16-byte unaligned write at 1024 + 17*lane + off
    u32x4 w = {v, v + 0x10101010u, v + 0x20202020u, v + 0x30303030u};
    *reinterpret_cast<u32x4_u*>(lds + 1024 + 17 * threadIdx.x + off) = w;
    __syncthreads();
    // unaligned read back
    uint32_t r = *reinterpret_cast<u32_u*>(lds + 5 * threadIdx.x + off + 1);
    lds[3000 + 4 * threadIdx.x + 0] = r; lds[3000 + 4 * threadIdx.x + 1] = r >> 8; lds[3000 + 4 * threadIdx.x + 2] = r >> 16; lds[3000 + 4*threadIdx.x+3] = r >> 24;
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = lds[i];
}
int main() {
    uint8_t* d; hipMalloc(&d, 4096);
    uint8_t h[4096], e[4096];
    int bad = 0;
    for (int off = 0; off < 4; ++off) {
        k<<<1, 64>>>(d, off);
        hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        memset(e, 0, sizeof e);
        for (int t = 0; t < 64; ++t) for (int b = 0; b < 4; ++b) e[5 * t + off + b] = (uint8_t)(t + 1);
        for (int t = 0; t < 64; ++t) for (int q = 0; q < 4; ++q) for (int b = 0; b < 4; ++b) e[1024 + 17 * t + off + 4 * q + b] = (uint8_t)(t + 1 + 0x10 * q);
        int m = 0;
        for (int i = 0; i < 3000; ++i) m += h[i] != e[i];
        // read check: r = bytes at 5t+off+1..+4
        for (int t = 0; t < 64; ++t) for (int b = 0; b < 4; ++b) m += h[3000 + 4 * t + b] != e[5 * t + off + 1 + b];
        printf("off %d mismatches %d\n", off, m);
        bad += m;
    }
    printf(bad ? "UNALIGNED LDS BROKEN\n" : "UNALIGNED LDS OK\n");
    return 0;
}
