// device_math.h — per-Gaussian device helpers shared by preprocess.hip (3DGS path) and preprocess_surfel.hip
// (2DGS path): camera block, SH basis + gradient (expression trees identical to the oracle's), quaternion ->
// rotation, and the adaptor activations that can be folded into K1/K9.  Include inside a translation unit
// compiled with -ffp-contract=off when integer intermediates must match the oracle bit for bit.
#pragma once
#include "gdr_common.h"

namespace gdr {
namespace {

#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

struct Cam {
    float v[16];
    float p[16];
    float c[3];
};

__device__ __forceinline__ void load_cam(Cam& cam, const float* __restrict__ view,
                                         const float* __restrict__ proj,
                                         const float* __restrict__ campos) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { cam.v[k] = view[k]; cam.p[k] = proj[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) cam.c[k] = campos ? campos[k] : 0.f;
}

// SH basis values b_k(x,y,z); expression trees identical to oracle sh_basis().
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
    b[0] = SH_C0;
    if (DEG >= 1) {
        b[1] = -SH_C1 * y;
        b[2] = SH_C1 * z;
        b[3] = -SH_C1 * x;
    }
    if (DEG >= 2) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = SH_C2_0 * xy;
        b[5] = SH_C2_1 * yz;
        b[6] = SH_C2_2 * (2.f * zz - xx - yy);
        b[7] = SH_C2_3 * xz;
        b[8] = SH_C2_4 * (xx - yy);
        if (DEG >= 3) {
            b[9] = SH_C3_0 * y * (3.f * xx - yy);
            b[10] = SH_C3_1 * xy * z;
            b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
            b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
            b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
            b[14] = SH_C3_5 * z * (xx - yy);
            b[15] = SH_C3_6 * x * (xx - 3.f * yy);
        }
    }
}

template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float* bx, float* by,
                                              float* bz) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
#pragma unroll
    for (int k = 0; k < NB; ++k) bx[k] = by[k] = bz[k] = 0.f;
    if (DEG >= 1) {
        by[1] = -SH_C1;
        bz[2] = SH_C1;
        bx[3] = -SH_C1;
    }
    if (DEG >= 2) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        bx[4] = SH_C2_0 * y; by[4] = SH_C2_0 * x;
        by[5] = SH_C2_1 * z; bz[5] = SH_C2_1 * y;
        bx[6] = SH_C2_2 * (-2.f * x); by[6] = SH_C2_2 * (-2.f * y); bz[6] = SH_C2_2 * (4.f * z);
        bx[7] = SH_C2_3 * z; bz[7] = SH_C2_3 * x;
        bx[8] = SH_C2_4 * (2.f * x); by[8] = SH_C2_4 * (-2.f * y);
        if (DEG >= 3) {
            bx[9] = SH_C3_0 * (6.f * xy); by[9] = SH_C3_0 * (3.f * xx - 3.f * yy);
            bx[10] = SH_C3_1 * yz; by[10] = SH_C3_1 * xz; bz[10] = SH_C3_1 * xy;
            bx[11] = SH_C3_2 * (-2.f * xy); by[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); bz[11] = SH_C3_2 * (8.f * yz);
            bx[12] = SH_C3_3 * (-6.f * xz); by[12] = SH_C3_3 * (-6.f * yz); bz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
            bx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); by[13] = SH_C3_4 * (-2.f * xy); bz[13] = SH_C3_4 * (8.f * xz);
            bx[14] = SH_C3_5 * (2.f * xz); by[14] = SH_C3_5 * (-2.f * yz); bz[14] = SH_C3_5 * (xx - yy);
            bx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); by[15] = SH_C3_6 * (-6.f * xy);
        }
    }
}

__device__ __forceinline__ void quat_to_R(float r, float x, float y, float z, float* R) {
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - r * z);
    R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);
    R[7] = 2.f * (y * z + r * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

// Activations of the render adaptor (lightning/renderer.py:225-230: sigmoid / exp / F.normalize),
// optionally folded into K1/K9 (gdr_inputs.flags) so the caller's raw tensors are read once.
__device__ __forceinline__ float act_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float4 act_normalize(float4 q, float* inv_norm) {
    const float n = sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
    const float inv = 1.f / fmaxf(n, 1e-12f);  // F.normalize(eps=1e-12)
    *inv_norm = inv;
    return make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
}


// ---- coalesced access to per-Gaussian rows of ROWF floats (the SH block: 3 (deg+1)^2) -------------------------
// One thread owns one Gaussian, so its row (192 bytes at degree 3) is a stride-ROWF access pattern: every wave-level
// load touches 64 different cache lines and the lines are re-fetched from L2 several times before a lane has consumed
// them (PMC on K9: 45 M L2 hits for 15 M misses).  Instead the workgroup copies its 256 consecutive rows — one
// contiguous 48 KB span — with fully coalesced float4 accesses through LDS.  Row stride in LDS = ROWF + 4 or + 8
// floats, chosen so that (stride / 4) is odd: 16-byte row reads/writes of consecutive lanes then rotate through all
// banks.
template <int ROWF>
struct RowStage {
    static constexpr int Q = ROWF / 4;                               // float4 per row
    static constexpr int STRIDE = ROWF + ((Q % 2 == 0) ? 4 : 8);     // floats
    static constexpr int LDS_FLOATS = GDR_BLOCK * STRIDE;
};

// global rows [row0, row0 + nrows) -> LDS (all GDR_BLOCK threads call it; nrows <= GDR_BLOCK)
template <int ROWF>
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ g, int row0, int nrows, float* lds) {
    constexpr int Q = RowStage<ROWF>::Q, STRIDE = RowStage<ROWF>::STRIDE;
    const float4* src = reinterpret_cast<const float4*>(g + (size_t)row0 * ROWF);
    const int total = nrows * Q;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int idx = (int)threadIdx.x + GDR_BLOCK * j;
        if (idx < total) {
            const int row = idx / Q, c = idx - row * Q;
            *reinterpret_cast<float4*>(lds + row * STRIDE + 4 * c) = src[idx];
        }
    }
}

// LDS -> global rows (optionally added to what is there), same mapping
template <int ROWF>
__device__ __forceinline__ void stage_rows_out(float* __restrict__ g, int row0, int nrows, const float* lds,
                                               bool accumulate) {
    constexpr int Q = RowStage<ROWF>::Q, STRIDE = RowStage<ROWF>::STRIDE;
    float4* dst = reinterpret_cast<float4*>(g + (size_t)row0 * ROWF);
    const int total = nrows * Q;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int idx = (int)threadIdx.x + GDR_BLOCK * j;
        if (idx < total) {
            const int row = idx / Q, c = idx - row * Q;
            float4 v = *reinterpret_cast<const float4*>(lds + row * STRIDE + 4 * c);
            if (accumulate) {
                const float4 o = dst[idx];
                v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
            }
            dst[idx] = v;
        }
    }
}

}  // namespace
}  // namespace gdr
