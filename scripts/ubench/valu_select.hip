// valu_select.hip — what does a SELECT cost on gfx950?  (round 6; follow-up of valu_rate.hip's 22-cycle v_cndmask_b32 reading)
// K6 / K7 predicate with v_cndmask_b32 (7.5 per K7 iteration, 2.75 per K6 iteration); valu_rate.hip's chain of
// `v_cndmask_b32 vA, vA, vB, vcc` retires one per ~22 shader cycles per SIMD whatever the number of resident waves, ten times
// a v_fma_f32.  This probe separates the encodings (VOP2 + implicit vcc, VOP3 + SGPR pair), the operand kinds (two VGPRs,
// inline constant) and the alternatives a select can be rewritten into (v_max / v_min / v_med3 / v_and / v_bfi / a multiply
// by a 0/1 mask), plus the other non-fma kinds of the K7 loop (v_mov, v_cmp into an SGPR pair, v_lshl_add_u64, quad_perm DPP,
// v_readlane).  Cycles are READ (s_memtime / s_memrealtime), per wave-instruction per SIMD, at 1 / 2 / 4 / 5 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_select.hip -o generativedensification_amd/lib/valu_select
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define ITERS 100
#define BLOCKS_PER_ITER 32   // asm blocks of 8 instructions per loop iteration

#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk, float bb) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = bb, c = 1e-7f;
    uint32_t ib = __float_as_uint(bb) | 0x3f000000u;
    // an SGPR-pair mask and vcc with a lane pattern (half the lanes set)
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x33333333\n s_mov_b64 vcc, s[20:21]" ::: "s20", "s21", "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < BLOCKS_PER_ITER; ++r) {
            if (KIND == 0) {
#define OP(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
                asm volatile(R8(OP) : REGS : "v"(b), "v"(c));
#undef OP
            } else if (KIND == 1) {   // VOP2, implicit vcc, both sources VGPRs
#define OP(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
                asm volatile(R8(OP) : REGS : "v"(b) : );
#undef OP
            } else if (KIND == 2) {   // VOP3, SGPR-pair mask
#define OP(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
                asm volatile(R8(OP) : REGS : "v"(b) : );
#undef OP
            } else if (KIND == 3) {   // VOP2, inline constant 0 as src0 (the `hit ? x : 0` form)
#define OP(i) "v_cndmask_b32_e32 %" #i ", 0, %" #i ", vcc\n"
                asm volatile(R8(OP) : REGS : : );
#undef OP
            } else if (KIND == 4) {   // VOP3, destination differs from both sources
                float d0, d1, d2, d3, d4, d5, d6, d7;
                asm volatile("v_cndmask_b32_e64 %0, %8, %16, s[20:21]\n v_cndmask_b32_e64 %1, %9, %16, s[20:21]\n v_cndmask_b32_e64 %2, %10, %16, s[20:21]\n"
                             "v_cndmask_b32_e64 %3, %11, %16, s[20:21]\n v_cndmask_b32_e64 %4, %12, %16, s[20:21]\n v_cndmask_b32_e64 %5, %13, %16, s[20:21]\n"
                             "v_cndmask_b32_e64 %6, %14, %16, s[20:21]\n v_cndmask_b32_e64 %7, %15, %16, s[20:21]\n"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b));
                a0 = d0; a1 = d1; a2 = d2; a3 = d3; a4 = d4; a5 = d5; a6 = d6; a7 = d7;
            } else if (KIND == 5) {
#define OP(i) "v_max_f32_e32 %" #i ", %" #i ", %8\n"
                asm volatile(R8(OP) : REGS : "v"(b));
#undef OP
            } else if (KIND == 6) {
#define OP(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
                asm volatile(R8(OP) : REGS : "v"(b), "v"(c));
#undef OP
            } else if (KIND == 7) {
#define OP(i) "v_and_b32_e32 %" #i ", %8, %" #i "\n"
                asm volatile(R8(OP) : REGS : "v"(ib));
#undef OP
            } else if (KIND == 8) {
#define OP(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
                asm volatile(R8(OP) : REGS : "v"(ib), "v"(c));
#undef OP
            } else if (KIND == 9) {
#define OP(i) "v_mov_b32_e32 %" #i ", %8\n"
                asm volatile(R8(OP) : REGS : "v"(b));
#undef OP
            } else if (KIND == 10) {  // compare into an SGPR pair (VOP3)
#define OP(i) "v_cmp_gt_f32_e64 s[22:23], %" #i ", %8\n"
                asm volatile(R8(OP) : REGS : "v"(b) : "s22", "s23");
#undef OP
            } else if (KIND == 11) {  // compare into vcc (VOPC) then the select that reads it
                asm volatile("v_cmp_gt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_gt_f32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                             "v_cmp_gt_f32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_gt_f32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                             : REGS : "v"(b) : "vcc");
            } else if (KIND == 12) {
#define OP(i) "v_mul_f32_e32 %" #i ", %8, %" #i "\n"
                asm volatile(R8(OP) : REGS : "v"(b));
#undef OP
            } else if (KIND == 13) {
#define OP(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
                asm volatile(R8(OP) : REGS : "v"(b), "v"(c));
#undef OP
            } else if (KIND == 14) {
#define OP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                asm volatile(R8(OP) : REGS);
#undef OP
            } else if (KIND == 15) {
                unsigned long long p0 = __float_as_uint(a0), p1 = __float_as_uint(a1), p2 = __float_as_uint(a2), p3 = __float_as_uint(a3);
                asm volatile("v_lshl_add_u64 %0, %0, 4, %1\n v_lshl_add_u64 %1, %1, 4, %2\n v_lshl_add_u64 %2, %2, 4, %3\n v_lshl_add_u64 %3, %3, 4, %0\n"
                             "v_lshl_add_u64 %0, %0, 4, %1\n v_lshl_add_u64 %1, %1, 4, %2\n v_lshl_add_u64 %2, %2, 4, %3\n v_lshl_add_u64 %3, %3, 4, %0\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                a0 += (float)(uint32_t)p0; a1 += (float)(uint32_t)p1;
            } else if (KIND == 16) {  // select rewritten as multiply-add by a 0/1 lane mask held in a VGPR: x = x*m + y*(1-m) -> fma chain
                asm volatile("v_mul_f32_e32 %0, %8, %0\n v_fmac_f32_e32 %0, %9, %1\n v_mul_f32_e32 %2, %8, %2\n v_fmac_f32_e32 %2, %9, %3\n"
                             "v_mul_f32_e32 %4, %8, %4\n v_fmac_f32_e32 %4, %9, %5\n v_mul_f32_e32 %6, %8, %6\n v_fmac_f32_e32 %6, %9, %7\n"
                             : REGS : "v"(b), "v"(c));
            } else if (KIND == 17) {  // VOP3 select with an SGPR pair written by a VOP3 compare just before (K7's `hit ? a : 0`)
                asm volatile("v_cmp_gt_f32_e64 s[22:23], %0, %8\n v_cndmask_b32_e64 %1, 0, %1, s[22:23]\n v_cmp_gt_f32_e64 s[24:25], %2, %8\n v_cndmask_b32_e64 %3, 0, %3, s[24:25]\n"
                             "v_cmp_gt_f32_e64 s[22:23], %4, %8\n v_cndmask_b32_e64 %5, 0, %5, s[22:23]\n v_cmp_gt_f32_e64 s[24:25], %6, %8\n v_cndmask_b32_e64 %7, 0, %7, s[24:25]\n"
                             : REGS : "v"(b) : "s22", "s23", "s24", "s25");
            } else if (KIND == 20) {  // ONE VALU compare into vcc, then seven selects reading it
                asm volatile("v_cmp_gt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                             : REGS : "v"(b) : "vcc");
            } else if (KIND == 21) {  // K7's pattern: compare into vcc, a scalar branch on vcc (never taken), two selects
                asm volatile("v_cmp_gt_f32_e32 vcc, %0, %8\n s_cbranch_vccz 1f\n v_cndmask_b32_e32 %1, 0, %1, vcc\n v_cndmask_b32_e32 %2, 0, %2, vcc\n"
                             "1:\n v_cmp_gt_f32_e32 vcc, %4, %8\n s_cbranch_vccz 2f\n v_cndmask_b32_e32 %5, 0, %5, vcc\n v_cndmask_b32_e32 %6, 0, %6, vcc\n 2:\n"
                             : REGS : "v"(b) : "vcc");
            } else if (KIND == 22) {  // ONE compare into an SGPR pair, then seven VOP3 selects reading it
                asm volatile("v_cmp_gt_f32_e64 s[22:23], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n"
                             "v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n"
                             : REGS : "v"(b) : "s22", "s23");
            } else if (KIND == 23) {  // s_and_b64 vcc (SALU write), then three selects: the `a && b` condition pattern
                asm volatile("s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n"
                             "s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n"
                             : REGS : "v"(b) : "vcc", "scc");   // (s_and_b64 writes SCC: undeclared, the loop's own s_cmp / s_cbranch_scc never terminated)
            } else if (KIND == 24) {  // K6's pattern: compare into vcc, three other VALU ops, THEN the select (8 VALU)
                asm volatile("v_cmp_gt_f32_e32 vcc, %0, %8\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                             "v_cndmask_b32_e32 %4, 0, %4, vcc\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n"
                             : REGS : "v"(b) : "vcc");
            } else if (KIND == 25) {  // compare into vcc, s_nop 1 (what the compiler's hazard recogniser inserts), the select, six other VALU ops
                asm volatile("v_cmp_gt_f32_e32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32_e32 %4, 0, %4, vcc\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                             "v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n"
                             : REGS : "v"(b) : "vcc");
            } else if (KIND == 26) {  // the same two with the VOP3 forms: compare into an SGPR pair, three VALU ops, select
                asm volatile("v_cmp_gt_f32_e64 s[22:23], %0, %8\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                             "v_cndmask_b32_e64 %4, 0, %4, s[22:23]\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n"
                             : REGS : "v"(b) : "s22", "s23");
            } else if (KIND == 27) {  // eight plain multiplies with ONE compare into vcc and no select: the baseline of 24 / 25
                asm volatile("v_cmp_gt_f32_e32 vcc, %0, %8\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                             "v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n"
                             : REGS : "v"(b) : "vcc");
            } else if (KIND == 18) {
#define OP(i) "v_exp_f32_e32 %" #i ", %" #i "\n"
                asm volatile(R8(OP) : REGS);
#undef OP
            } else if (KIND == 19) {
#define OP(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %5\n"
                asm volatile(OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3)
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
                             : "v"(*(const double*)&b), "v"(*(const double*)&c));
#undef OP
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if ((threadIdx.x & 63u) == 0u) {
        clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char* name) {
    printf("%-44s", name);
    for (int w : {1, 2, 4, 5}) {
        const int blocks = 256 * w;   // 256 CUs x w workgroups of 4 waves (one per SIMD)
        float* out; unsigned long long* clk;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
        (void)hipMalloc(&clk, (size_t)blocks * 4 * 2 * sizeof(unsigned long long));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 2, clk, 1.0001f);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, ITERS, clk, 1.0001f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
        (void)hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double ticks = 0, ref = 0;
        for (size_t i = 0; i < h.size() / 2; ++i) { ticks += (double)h[2 * i]; ref += (double)h[2 * i + 1]; }
        ticks /= (double)(h.size() / 2); ref /= (double)(h.size() / 2);
        const double per_wave = (double)ITERS * BLOCKS_PER_ITER * 8;
        printf("  w%d %6.2f cyc (%.2f GHz)", w, ticks / (per_wave * w), ticks / (ref / 100e6) * 1e-9);
        (void)hipFree(out); (void)hipFree(clk);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) {   // one of the round-6 follow-up kinds alone (each under its own `timeout` on the GPU box)
        const int kind = atoi(argv[1]);
        if (kind == 20) run<20>("1 v_cmp_e32 vcc + 7 v_cndmask_e32 vcc");
        if (kind == 21) run<21>("v_cmp vcc; s_cbranch_vccz; 2 v_cndmask (x2, 6 VALU / 8)");
        if (kind == 22) run<22>("1 v_cmp_e64 s[..] + 7 v_cndmask_e64 s[..]");
        if (kind == 24) run<24>("v_cmp vcc; 3 v_mul; v_cndmask_e32 vcc; 3 v_mul");
        if (kind == 25) run<25>("v_cmp vcc; s_nop 1; v_cndmask_e32 vcc; 6 v_mul");
        if (kind == 26) run<26>("v_cmp_e64 s[..]; 3 v_mul; v_cndmask_e64; 3 v_mul");
        if (kind == 27) run<27>("v_cmp vcc; 7 v_mul (no select)");
        if (kind == 23) run<23>("s_and_b64 vcc + 3 v_cndmask_e32 (x2, 6 VALU / 8)");
        return 0;
    }
    printf("shader cycles per wave64 instruction per SIMD (s_memtime), w = waves resident per SIMD; clock from s_memrealtime\n");
    run<0>("v_fma_f32 (VOP3, 3 VGPR sources)");
    run<13>("v_fmac_f32_e32 (VOP2)");
    run<12>("v_mul_f32_e32");
    run<9>("v_mov_b32_e32");
    run<7>("v_and_b32_e32");
    run<5>("v_max_f32_e32");
    run<6>("v_med3_f32");
    run<8>("v_bfi_b32");
    run<1>("v_cndmask_b32_e32 vD, vD, vB, vcc");
    run<3>("v_cndmask_b32_e32 vD, 0, vD, vcc");
    run<2>("v_cndmask_b32_e64 vD, vD, vB, s[20:21]");
    run<4>("v_cndmask_b32_e64 vD, vA, vB, s[20:21]");
    run<10>("v_cmp_gt_f32_e64 s[22:23], vA, vB");
    run<11>("v_cmp_gt_f32_e32 vcc + v_cndmask_e32 (pairs)");
    run<17>("v_cmp_e64 s[..] + v_cndmask_e64 0, v, s[..]");
    run<16>("select as v_mul + v_fmac by a 0/1 VGPR (pairs)");
    run<14>("v_add_f32_dpp quad_perm");
    run<15>("v_lshl_add_u64");
    run<18>("v_exp_f32");
    run<19>("v_pk_fma_f32");
    return 0;
}
