// Second issue-rate microbenchmark for gfx950: the integer / select / compare instruction kinds of the K6/K7 mask walk.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate2.hip -o build/valu_rate2 && build/valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 400
#define X8(s) s s s s s s s s

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 1.0001f;
    uint64_t m = 0xF0F0F0F0F0F0F0F0ull ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
        if (KIND == 0) {   // cndmask VOP2, vcc
            asm volatile(X8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n") : OPS : : "vcc");
        } else if (KIND == 1) {   // cndmask e64, sgpr pair mask
            asm volatile(X8("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n") : OPS : : "s20", "s21");
        } else if (KIND == 2) {   // cndmask with inline constants
            asm volatile(X8("v_cndmask_b32 %0, 0, %1, vcc\n v_cndmask_b32 %2, 0, %3, vcc\n v_cndmask_b32 %4, 0, %5, vcc\n v_cndmask_b32 %6, 0, %7, vcc\n") : OPS : : "vcc");
        } else if (KIND == 3) {   // v_cmp writes vcc
            asm volatile(X8("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %6, %7\n") : OPS : : "vcc");
        } else if (KIND == 4) {   // v_cmp -> v_cndmask pairs (dependent through vcc)
            asm volatile(X8("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n") : OPS : : "vcc");
        } else if (KIND == 5) {   // v_and_b32
            asm volatile(X8("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7\n") : OPS);
        } else if (KIND == 6) {   // v_ffbl_b32
            asm volatile(X8("v_ffbl_b32 %0, %1\n v_ffbl_b32 %2, %3\n v_ffbl_b32 %4, %5\n v_ffbl_b32 %6, %7\n") : OPS);
        } else if (KIND == 7) {   // v_min3_u32
            asm volatile(X8("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %2, %2, %3, %4\n v_min3_u32 %4, %4, %5, %6\n v_min3_u32 %6, %6, %7, %0\n") : OPS);
        } else if (KIND == 8) {   // v_lshl_add_u64 (the m-1 of take_bit)
            asm volatile(X8("v_lshl_add_u64 %0, %0, 0, -1\n v_lshl_add_u64 %0, %0, 0, -1\n v_lshl_add_u64 %0, %0, 0, -1\n v_lshl_add_u64 %0, %0, 0, -1\n") : "+v"(m));
        } else if (KIND == 9) {   // v_fmac_f32 (2 sources + accumulator)
            asm volatile(X8("v_fmac_f32 %0, %1, %8\n v_fmac_f32 %2, %3, %8\n v_fmac_f32 %4, %5, %8\n v_fmac_f32 %6, %7, %8\n") : OPS : "v"(b));
        } else if (KIND == 10) {  // v_fma_f32 with one constant
            asm volatile(X8("v_fma_f32 %0, %0, %1, 1.0\n v_fma_f32 %2, %2, %3, 1.0\n v_fma_f32 %4, %4, %5, 1.0\n v_fma_f32 %6, %6, %7, 1.0\n") : OPS);
        } else if (KIND == 11) {  // v_cmp_ne_u64
            asm volatile(X8("v_cmp_ne_u64 vcc, 0, %0\n v_cmp_ne_u64 vcc, 0, %0\n v_cmp_ne_u64 vcc, 0, %0\n v_cmp_ne_u64 vcc, 0, %0\n") : "+v"(m) : : "vcc");
        } else if (KIND == 12) {  // v_add_u32
            asm volatile(X8("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n") : OPS);
        } else if (KIND == 13) {  // v_max_f32
            asm volatile(X8("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3\n v_max_f32 %4, %4, %5\n v_max_f32 %6, %6, %7\n") : OPS);
        } else if (KIND == 14) {  // quad_perm DPP mov
            asm volatile(X8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : OPS);
        } else if (KIND == 15) {  // v_cmp e64 writing an SGPR pair
            asm volatile(X8("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %2, %3\n v_cmp_lt_f32_e64 s[20:21], %4, %5\n v_cmp_lt_f32_e64 s[22:23], %6, %7\n") : OPS : : "s20", "s21", "s22", "s23");
        } else if (KIND == 16) {  // v_mul then independent cndmask: mixed stream 3 mul : 1 cndmask
            asm volatile(X8("v_mul_f32 %0, %0, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %4, %4, %8\n v_cndmask_b32 %6, %6, %7, vcc\n") : OPS : "v"(b) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)m;
}

template <int KIND>
static void run(const char* name, int waves_per_simd) {
    float* out;
    const int blocks = 256 * waves_per_simd;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, ITERS);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)ITERS * 32 * waves_per_simd;
    printf("%-34s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name,
           waves_per_simd, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    (void)hipFree(out);
}

int main() {
    for (int w : {2, 8}) {
        run<0>("v_cndmask_b32 vcc", w);
        run<1>("v_cndmask_b32_e64 sgpr mask", w);
        run<2>("v_cndmask_b32 0, v, vcc", w);
        run<3>("v_cmp_lt_f32 -> vcc", w);
        run<15>("v_cmp_lt_f32_e64 -> sgpr", w);
        run<4>("v_cmp + v_cndmask pairs", w);
        run<16>("3 v_mul : 1 v_cndmask", w);
        run<5>("v_and_b32", w);
        run<6>("v_ffbl_b32", w);
        run<7>("v_min3_u32", w);
        run<8>("v_lshl_add_u64 (dependent)", w);
        run<9>("v_fmac_f32", w);
        run<10>("v_fma_f32 v,v,const", w);
        run<11>("v_cmp_ne_u64", w);
        run<12>("v_add_u32", w);
        run<13>("v_max_f32", w);
        run<14>("v_mov_b32_dpp quad_perm", w);
    }
    return 0;
}
