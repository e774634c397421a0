// What SQ_LDS_BANK_CONFLICT counts on gfx950 for a few wave64 LDS access patterns (one wave-instruction per
// kernel trip; 1024 workgroups x 256 threads x 64 trips).  Build + run under rocprofv3:
//   hipcc --offload-arch=gfx950 -O3 -o lds_counter_probe tools/exp/lds_counter_probe.hip
//   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d out -- ./lds_counter_probe
// Expected per wave-instruction if the counter is "extra LDS-array cycles": stride-1 b32 = 0, stride-2 b32 = 2 (one per
// 32-lane group).  The question: what do conflict-free ds_or_b32 (atomics) report?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kTrips = 64;
template <int MODE>
__global__ void __launch_bounds__(256) probe(uint32_t *out) {
    __shared__ uint32_t lds[4][512];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *mine = lds[wave];
    for (int i = lane; i < 512; i += 64) mine[i] = i;
    __builtin_amdgcn_s_waitcnt(0);
    uint32_t acc = 0;
    for (int t = 0; t < kTrips; ++t) {
        if (MODE == 0) acc += ((volatile uint32_t *)mine)[lane];                        // ds_read_b32, stride 1
        if (MODE == 1) acc += ((volatile uint32_t *)mine)[2 * lane];                    // ds_read_b32, stride 2: 2-way
        if (MODE == 2) ((volatile uint32_t *)mine)[lane] = acc + t;                     // ds_write_b32, stride 1
        if (MODE == 3) ((volatile uint32_t *)mine)[2 * lane] = acc + t;                 // ds_write_b32, stride 2: 2-way
        if (MODE == 4) atomicOr(mine + lane, 1u << (t & 31));                           // ds_or_b32, stride 1 (no return)
        if (MODE == 5) atomicOr(mine + 2 * lane, 1u << (t & 31));                       // ds_or_b32, stride 2: 2-way
        if (MODE == 6) atomicAdd(mine + lane, 1u);                                      // ds_add_u32, stride 1 (no return)
        if (MODE == 7) acc += atomicOr(mine + lane, 1u << (t & 31));                    // ds_or_rtn_b32, stride 1
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc + mine[lane];
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 1024 * 256 * 4);
    probe<0><<<1024, 256>>>(d);
    probe<1><<<1024, 256>>>(d);
    probe<2><<<1024, 256>>>(d);
    probe<3><<<1024, 256>>>(d);
    probe<4><<<1024, 256>>>(d);
    probe<5><<<1024, 256>>>(d);
    probe<6><<<1024, 256>>>(d);
    probe<7><<<1024, 256>>>(d);
    hipError_t rc = hipDeviceSynchronize();
    printf("probe rc=%d; wave-instructions of the pattern per launch: %d\n", (int)rc, 1024 * 4 * kTrips);
    return rc != hipSuccess;
}
