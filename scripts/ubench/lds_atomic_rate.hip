// LDS atomic throughput on gfx950: ns per wave64 instruction per CU-resident wave set, for ds_add_u32 / ds_add_f32 with
// (a) 64 distinct addresses, (b) 4 distinct addresses (16 lanes each), (c) all lanes one address.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_atomic_rate.hip -o build/lds_atomic_rate && build/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2000
template <int KIND, int PATTERN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t idx = PATTERN == 0 ? (w * 64 + lane) : (PATTERN == 1 ? (w * 64 + (lane >> 4)) : w * 64);
    uint32_t addr = idx * 4u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
            else if (KIND == 1) asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(1.0f) : "memory");
            else { uint32_t t; asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr), "v"(1u) : "memory"); addr ^= (t & 0u); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)lds[threadIdx.x];
}
template <int KIND, int PATTERN>
static void run(const char* name) {
    float* out; const int blocks = 256 * 4;  // 4 workgroups (16 waves) per CU
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, ITERS);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)ITERS * 8 * 16;   // wave-instructions per CU
    printf("%-44s %.1f ns per wave-instruction per CU (%.0f cycles @2.4 GHz)\n", name, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    (void)hipFree(out);
}
int main() {
    run<0, 0>("ds_add_u32, 64 distinct addresses");
    run<0, 1>("ds_add_u32, 4 addresses x 16 lanes");
    run<0, 2>("ds_add_u32, 1 address x 64 lanes");
    run<1, 0>("ds_add_f32, 64 distinct addresses");
    run<1, 1>("ds_add_f32, 4 addresses x 16 lanes");
    run<1, 2>("ds_add_f32, 1 address x 64 lanes");
    run<2, 0>("ds_add_rtn_u32 (waited), 64 distinct");
    run<2, 1>("ds_add_rtn_u32 (waited), 4 x 16");
    return 0;
}
