// atomic_probe.hip — cost model of global_atomic_add_f32 on gfx950 for K7's publish patterns (round 4).
// Every wave issues `iters` atomic instructions; pattern selects which lanes are active and which 64-byte records
// they hit.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ubench/atomic_probe.hip -o <out>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// group = lanes per record (16 or 8); act = active lanes per group; same = 1: all groups of a wave hit ONE record;
// twice = 1: the instruction is issued twice on the same records; local = 1: records drawn from a window private to
// the workgroup (no other workgroup touches them), 0: from the whole buffer
template <int LDS>
__global__ __launch_bounds__(256) void probe(float* buf, uint32_t nrec, int iters, int group, int act, int same, int twice,
                                             int local, int valu, int wide) {
    __shared__ float lacc[LDS ? 256 * 12 : 1];
    if (LDS) for (int k = threadIdx.x; k < 256 * 12; k += 256) lacc[k] = 0.f;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t g = lane / (uint32_t)group, li = lane % (uint32_t)group;
    float x = (float)lane;
    for (int i = 0; i < iters; ++i) {
        uint32_t h = mix(wave * 2654435761u + (uint32_t)i * 40503u + (same ? 0u : g * 97u));
        uint32_t rec = local ? ((blockIdx.x * 64u + (h & 63u)) & (nrec - 1u)) : (h & (nrec - 1u));
        for (int k = 0; k < valu; ++k) x = fmaf(x, 1.0001f, 0.5f);   // independent VALU work between the atomics
        if (li < (uint32_t)act) {
            if (LDS) {
                atomicAdd(&lacc[(h & 255u) * 12u + li % 12u], x);
            } else {
                if (wide) {   // 128-byte records: one instruction over `act` <= 32 adjacent words, or (twice) 16 + 4 in two
                    if (!twice) atomicAdd(buf + 32 * (size_t)(rec >> 1) + li, x);
                    else { if (li < 16u) atomicAdd(buf + 32 * (size_t)(rec >> 1) + li, x); if (li < 4u) atomicAdd(buf + 32 * (size_t)(rec >> 1) + 16u + li, x); }
                } else {
                atomicAdd(buf + 16 * (size_t)rec + li, x);
                if (twice) atomicAdd(buf + 16 * (size_t)rec + (li + 4u) % 16u, x);
                }
            }
        }
    }
    if (LDS) { __syncthreads(); if (lacc[threadIdx.x] == 12345.f) buf[0] = 1.f; }
    if (x == 12345.f) buf[1] = x;
}

int main() {
    const uint32_t nrec = 1u << 21;   // (power of two: no division in the loop)
    float* buf; hipMalloc(&buf, (size_t)nrec * 64); hipMemset(buf, 0, (size_t)nrec * 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 2500 * 4, iters = 200;
    struct P { const char* name; int lds, group, act, same, twice, local, valu, wide; } ps[] = {
        {"rows16 x12 lanes, 4 records / instr", 0, 16, 12, 0, 0, 0, 0},
        {"rows16 x12, 4 records, +90 VALU", 0, 16, 12, 0, 0, 0, 90},
        {"rows16 x12, all rows ONE record", 0, 16, 12, 1, 0, 0, 0},
        {"rows16 x12, ONE record, +90 VALU", 0, 16, 12, 1, 0, 0, 90},
        {"rows16 x12, 4 records, workgroup-local window", 0, 16, 12, 0, 0, 1, 0},
        {"rows16 x12, 4 records, issued twice", 0, 16, 12, 0, 1, 0, 0},
        {"half8 x8 lanes, 8 records / instr", 0, 8, 8, 0, 0, 0, 0},
        {"half8 x8, 8 records, +90 VALU", 0, 8, 8, 0, 0, 0, 90},
        {"half8 x8, issued twice", 0, 8, 8, 0, 1, 0, 0},
        {"half8 x4 lanes, 8 records", 0, 8, 4, 0, 0, 0, 0},
        {"rows16 x16 lanes, 4 records", 0, 16, 16, 0, 0, 0, 0},
        {"rows16 x4 lanes, 4 records", 0, 16, 4, 0, 0, 0, 0},
        {"rows16 x1 lane, 4 records", 0, 16, 1, 0, 0, 0, 0},
        {"64 lanes, 64 records (group 1)", 0, 1, 1, 0, 0, 0, 0},
        {"128 B records: 32 lanes x 32 adjacent words, 2 / instr", 0, 32, 32, 0, 0, 0, 0, 1},
        {"128 B records: 32-lane groups, 20 adjacent words", 0, 32, 20, 0, 0, 0, 0, 1},
        {"128 B records: rows16, 16 words + 4 words (2 instr)", 0, 16, 16, 0, 1, 0, 0, 1},
        {"only VALU 90", 0, 16, 0, 0, 0, 0, 90},
        {"no atomics, no VALU (loop overhead)", 0, 16, 0, 0, 0, 0, 0},
        {"LDS ds_add rows16 x12, 4 entries, +90 VALU", 1, 16, 12, 0, 0, 0, 90},
        {"LDS ds_add rows16 x1 lane", 1, 16, 1, 0, 0, 0, 0},
        {"LDS ds_add rows16 x4 lanes", 1, 16, 4, 0, 0, 0, 0},
        {"LDS ds_add 64 lanes, 64 entries", 1, 1, 1, 0, 0, 0, 0},
        {"LDS ds_add rows16 x12, 4 entries", 1, 16, 12, 0, 0, 0, 0},
        {"LDS ds_add rows16 x12, ONE entry", 1, 16, 12, 1, 0, 0, 0},
        {"LDS ds_add half8 x8, 8 entries", 1, 8, 8, 0, 0, 0, 0},
    };
    for (auto& p : ps) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (p.lds) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, buf, nrec, iters, p.group, p.act, p.same, p.twice, p.local, p.valu, p.wide);
            else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, buf, nrec, iters, p.group, p.act, p.same, p.twice, p.local, p.valu, p.wide);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 1) {
                const double instr = (double)blocks * 4 * iters * (p.twice ? 2 : 1);
                const int groups = p.act ? 64 / p.group : 0;                 // lane groups = record lines (LDS: entries) per instruction
                const double lines = instr * (p.same ? 1 : groups);
                printf("%-50s %8.3f ms  %6.2f G wave-instr/s  %6.2f G %s/s  %6.1f G lanes/s\n", p.name, ms, instr / ms * 1e-6,
                       lines / ms * 1e-6, p.lds ? "LDS entries" : "record lines", instr * groups * p.act / ms * 1e-6);
            }
        }
    }
    return 0;
}
