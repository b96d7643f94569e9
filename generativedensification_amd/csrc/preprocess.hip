// preprocess.hip — per-Gaussian stages of the rasterizer for gfx950.
//   K1  preprocess forward : project, EWA 2D covariance, conic, radius, tile rect,
//                            SH -> RGB, block partial sums of tiles_touched
//   K8/K9 preprocess backward: conic/mean2D/colour/depth partials -> means3D, SH,
//                            scales, rotations (or cov3D)
//   K10 mark_visible
// Behaviour: SURVEY.md Appendix A.1 / A.5 (the un-vendored CUDA rasterizer the reference
// installs from third_party/diff-gaussian-rasterization, /root/reference/.gitmodules:1-3).
//
// This translation unit is compiled with -ffp-contract=off and IEEE divide/sqrt: every
// float operation rounds exactly once, in the order written, so that radii, tile rects,
// depth key bits and therefore the whole sorted (tile,depth) list are bit-identical to
// the CPU oracle (oracle/gdr_oracle.c, built with the same contraction setting).
// These kernels are HBM-bound (236 B read + ~90 B written per Gaussian at SH degree 3);
// the extra VALU ops from not fusing multiply-adds are hidden behind memory.
#include <stdlib.h>

#include "gdr_common.h"
#include "device_math.h"

#pragma clang fp contract(off)

namespace gdr {

namespace {

// conservative half-extent (pixels) of {alpha >= 1/255} for conic (cx,cy,cz) and opacity o:
// exact bounding box of the ellipse power >= -ln(255 o), widened by 0.2 % + 0.02 px.  K6/K7 only
// use it to skip entries that would fail the alpha test at every pixel of a 4x4 block.
__device__ __forceinline__ float2 alpha_extent(const float4 co) {
    const float t = 255.f * co.w;
    if (!(t > 1.f)) return make_float2(-1.f, -1.f);  // alpha <= o < 1/255 everywhere (also NaN)
    const float det = co.x * co.z - co.y * co.y;
    if (!(det > 0.f)) return make_float2(1e30f, 1e30f);  // degenerate: never cull
    const float tau2 = 2.f * __logf(t) / det;            // 2 ln(255 o) / det(conic)
    return make_float2(sqrtf(tau2 * co.z) * 1.002f + 0.02f, sqrtf(tau2 * co.x) * 1.002f + 0.02f);
}
__device__ __forceinline__ void write_rec(float4* __restrict__ rec, int i, float2 pxy, float depth, float4 con_o,
                                          float4 rgbd) {
    const float2 ext = alpha_extent(con_o);
    rec[4 * i + 0] = make_float4(pxy.x, pxy.y, depth, 0.f);
    rec[4 * i + 1] = con_o;
    rec[4 * i + 2] = rgbd;
    rec[4 * i + 3] = make_float4(ext.x, ext.y, 0.f, 0.f);
}

// EWA projection pieces shared by forward and backward (Appendix A.1-4).
struct Ewa {
    float tx, ty, tz, xmul, ymul;
    float A0[3], A1[3], v0[3], v1[3];
    float a, b, c;
};

__device__ __forceinline__ void ewa(const Cam& cam, float pvx, float pvy, float pvz,
                                    const float* c6, float focal_x, float focal_y, float tanx,
                                    float tany, Ewa& e) {
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    float txtz = pvx / pvz, tytz = pvy / pvz;
    e.tz = pvz;
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * pvz;
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * pvz;
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    float J00 = focal_x / pvz, J02 = -(focal_x * e.tx) / (pvz * pvz);
    float J11 = focal_y / pvz, J12 = -(focal_y * e.ty) / (pvz * pvz);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e.A0[k] = J00 * cam.v[4 * k + 0] + J02 * cam.v[4 * k + 2];
        e.A1[k] = J11 * cam.v[4 * k + 1] + J12 * cam.v[4 * k + 2];
    }
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        e.v0[r] = (S[3 * r] * e.A0[0] + S[3 * r + 1] * e.A0[1]) + S[3 * r + 2] * e.A0[2];
        e.v1[r] = (S[3 * r] * e.A1[0] + S[3 * r + 1] * e.A1[1]) + S[3 * r + 2] * e.A1[2];
    }
    e.a = ((e.A0[0] * e.v0[0] + e.A0[1] * e.v0[1]) + e.A0[2] * e.v0[2]) + 0.3f;
    e.b = (e.A0[0] * e.v1[0] + e.A0[1] * e.v1[1]) + e.A0[2] * e.v1[2];
    e.c = ((e.A1[0] * e.v1[0] + e.A1[1] * e.v1[1]) + e.A1[2] * e.v1[2]) + 0.3f;
}

// ---------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(GDR_BLOCK) void preprocess_fwd_kernel(
    int N, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
    float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos, int W, int H, float tanx,
    float tany, float focal_x, float focal_y, int32_t* __restrict__ radii, float* __restrict__ g_depths,
    float4* __restrict__ g_rec,
    float* __restrict__ g_cov3D, int4* __restrict__ g_rect, uint32_t* __restrict__ g_tiles,
    uint8_t* __restrict__ g_clamped, uint32_t* __restrict__ block_sums, uint32_t flags,
    uint32_t* __restrict__ block_offs, uint32_t* __restrict__ num_rendered) {
    Cam cam;
    load_cam(cam, view, proj, campos);
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const int gx = (W + GDR_TILE - 1) / GDR_TILE, gy = (H + GDR_TILE - 1) / GDR_TILE;
    constexpr int REC_STRIDE = 20;
    __shared__ float rec_out[GDR_BLOCK * REC_STRIDE];

    uint32_t tiles = 0;
    if (i < N) {
        int rad = 0;
        float depth = 0.f;
        float2 pxy = make_float2(0.f, 0.f);
        float4 con_o = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 rgbd = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 rect = make_int4(0, 0, 0, 0);
        uint32_t clampbits = 0;
        float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        const float px_ = means3D[3 * i], py_ = means3D[3 * i + 1], pz_ = means3D[3 * i + 2];
        const float pvx = cam.v[0] * px_ + cam.v[4] * py_ + cam.v[8] * pz_ + cam.v[12];
        const float pvy = cam.v[1] * px_ + cam.v[5] * py_ + cam.v[9] * pz_ + cam.v[13];
        const float pvz = cam.v[2] * px_ + cam.v[6] * py_ + cam.v[10] * pz_ + cam.v[14];
        bool ok = pvz > 0.2f;  // near cull (A.1-1); no x/y frustum test when prefiltered=False
        if (ok) {
            const float phx = cam.p[0] * px_ + cam.p[4] * py_ + cam.p[8] * pz_ + cam.p[12];
            const float phy = cam.p[1] * px_ + cam.p[5] * py_ + cam.p[9] * pz_ + cam.p[13];
            const float phw = cam.p[3] * px_ + cam.p[7] * py_ + cam.p[11] * pz_ + cam.p[15];
            const float p_w = 1.0f / (phw + 0.0000001f);
            const float ppx = phx * p_w, ppy = phy * p_w;

            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
            } else {
                float4 q = reinterpret_cast<const float4*>(rotations)[i];
                float sc0 = scales[3 * i], sc1 = scales[3 * i + 1], sc2 = scales[3 * i + 2];
                if (flags & GDR_IN_RAW_ROTATIONS) { float inv_n; q = act_normalize(q, &inv_n); }
                if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); sc2 = expf(sc2); }
                float R[9], Mm[9];
                quat_to_R(q.x, q.y, q.z, q.w, R);
                const float s[3] = {scale_modifier * sc0, scale_modifier * sc1, scale_modifier * sc2};
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) Mm[r * 3 + k] = R[r * 3 + k] * s[k];
#define SIG(a_, b_) ((Mm[a_ * 3 + 0] * Mm[b_ * 3 + 0] + Mm[a_ * 3 + 1] * Mm[b_ * 3 + 1]) + Mm[a_ * 3 + 2] * Mm[b_ * 3 + 2])
                c6[0] = SIG(0, 0); c6[1] = SIG(0, 1); c6[2] = SIG(0, 2);
                c6[3] = SIG(1, 1); c6[4] = SIG(1, 2); c6[5] = SIG(2, 2);
#undef SIG
            }
            Ewa e;
            ewa(cam, pvx, pvy, pvz, c6, focal_x, focal_y, tanx, tany, e);
            const float det = e.a * e.c - e.b * e.b;
            ok = det != 0.f;
            if (ok) {
                const float det_inv = 1.f / det;
                const float mid = 0.5f * (e.a + e.c);
                const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda1 = mid + disc, lambda2 = mid - disc;
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                const float sx = ((ppx + 1.0f) * (float)W - 1.0f) * 0.5f;
                const float sy = ((ppy + 1.0f) * (float)H - 1.0f) * 0.5f;
                const int r_i = (int)my_radius;
                const float rf = (float)r_i;
                rect.x = min(gx, max(0, (int)((sx - rf) / (float)GDR_TILE)));
                rect.y = min(gy, max(0, (int)((sy - rf) / (float)GDR_TILE)));
                rect.z = min(gx, max(0, (int)((sx + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                rect.w = min(gy, max(0, (int)((sy + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                tiles = (uint32_t)((rect.z - rect.x) * (rect.w - rect.y));
                ok = tiles != 0;
                if (ok) {
                    rad = r_i;
                    depth = pvz;
                    pxy = make_float2(sx, sy);
                    const float op = (flags & GDR_IN_RAW_OPACITY) ? act_sigmoid(opacities[i]) : opacities[i];
                    con_o = make_float4(e.c * det_inv, -e.b * det_inv, e.a * det_inv, op);
                    if (colors_precomp) {
                        rgbd.x = colors_precomp[3 * i];
                        rgbd.y = colors_precomp[3 * i + 1];
                        rgbd.z = colors_precomp[3 * i + 2];
                    } else {
                        float dx = px_ - cam.c[0], dy = py_ - cam.c[1], dz = pz_ - cam.c[2];
                        const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
                        dx *= inv; dy *= inv; dz *= inv;
                        constexpr int NB = (DEG + 1) * (DEG + 1);
                        float bk[NB];
                        sh_basis<DEG>(dx, dy, dz, bk);
                        const float* sh = shs + (size_t)i * M * 3;
                        float acc[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) acc[ch] = bk[0] * sh[ch];
#pragma unroll
                        for (int k = 1; k < NB; ++k)
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) acc[ch] = acc[ch] + bk[k] * sh[3 * k + ch];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            acc[ch] = acc[ch] + 0.5f;
                            if (acc[ch] < 0.f) clampbits |= (1u << ch);
                            acc[ch] = fmaxf(acc[ch], 0.f);
                        }
                        rgbd.x = acc[0]; rgbd.y = acc[1]; rgbd.z = acc[2];
                    }
                    rgbd.w = depth;
                } else {
                    rect = make_int4(0, 0, 0, 0);
                }
            }
            if (!ok) { tiles = 0; }
        }
        radii[i] = rad;
        g_depths[i] = depth;
        {   // this lane's record into its wave's OUT slice (stored coalesced below: see preprocess_fwd_views_kernel)
            const float2 ext = alpha_extent(con_o);
            float* mine = rec_out + (int)threadIdx.x * REC_STRIDE;
            *reinterpret_cast<float4*>(mine) = make_float4(pxy.x, pxy.y, depth, 0.f);
            *reinterpret_cast<float4*>(mine + 4) = con_o;
            *reinterpret_cast<float4*>(mine + 8) = rgbd;
            *reinterpret_cast<float4*>(mine + 12) = make_float4(ext.x, ext.y, 0.f, 0.f);
        }
        if (!cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) g_cov3D[6 * i + k] = c6[k];
        }
        g_rect[i] = rect;
        g_tiles[i] = tiles;
        g_clamped[i] = (uint8_t)clampbits;
    }
    {   // 64 records = 4 KB contiguous: four coalesced 1 KB stores per wave instead of 64-line-stride float4 stores per lane
        const int wave0 = (int)(threadIdx.x & ~63u), lane = (int)(threadIdx.x & 63u);
        const int first = blockIdx.x * GDR_BLOCK + wave0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int nrec = min(GDR_WAVE, N - first);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = lane + GDR_WAVE * j, r = q >> 2, c = q & 3;
            if (r < nrec) g_rec[4 * (size_t)first + q] = *reinterpret_cast<const float4*>(rec_out + (wave0 + r) * REC_STRIDE + 4 * c);
        }
    }
    // block partial sum of tiles_touched -> block_sums[blockIdx.x] (feeds the scan, K2)
    uint32_t v = tiles;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __shared__ uint32_t wsum[GDR_BLOCK / GDR_WAVE];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t bs = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        block_sums[blockIdx.x] = bs;
        block_offs[blockIdx.x] = bs ? atomicAdd(num_rendered, bs) : 0u;  // this block's slice of the D duplicates
    }
}

// ---------------------------------------------------------------------------------
// K8 + K9 fused: one pass over the Gaussians.
// grad_rec layout per Gaussian (16 floats = one 64-byte line, accumulated by K7):
//   [0..3] dL/dmean2D x, y (signed, NDC units), sum|x-term|, sum|y-term|
//   [4..6] dL/dconic.x, .y, .z (true partials)   [7] dL/ddepth
//   [8..10] dL/dcolour r, g, b                   [11] dL/dopacity     [12..15] unused
// ---------------------------------------------------------------------------------
// STAGED (shs given, M == NB, degrees 1 and 3): SH rows in / SH-gradient rows out through LDS, coalesced
// (device_math.h RowStage); one barrier after body() for every thread
template <int DEG, bool STAGED>
__global__ __launch_bounds__(GDR_BLOCK) void preprocess_bwd_kernel(
    int N, int M, const float* __restrict__ means3D, const int32_t* __restrict__ radii,
    const float* __restrict__ shs, const uint8_t* __restrict__ g_clamped,
    const float* __restrict__ scales, const float* __restrict__ rotations, float scale_modifier,
    const float* __restrict__ cov3D, int cov_precomp, int colors_precomp,
    const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ campos, int W, int H, float tanx, float tany, float focal_x,
    float focal_y, const float4* __restrict__ grad_rec, float4* __restrict__ dL_dmean2D,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dscale, float4* __restrict__ dL_drot,
    const float4* __restrict__ g_rec, uint32_t flags, int accumulate) {
    Cam cam;
    load_cam(cam, view, proj, campos);
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int ROWF = 3 * NB;
    using RS = RowStage<STAGED ? ROWF : 4>;
    __shared__ float lds_rows[STAGED ? RS::LDS_FLOATS : 1];
    const int row0 = blockIdx.x * GDR_BLOCK, nrows = min(GDR_BLOCK, N - row0);
    const int i = row0 + threadIdx.x;
    float* my_row = lds_rows + (STAGED ? (int)threadIdx.x * RS::STRIDE : 0);
    if (STAGED) {
        stage_rows_in<STAGED ? ROWF : 4>(shs, row0, nrows, lds_rows);
        __syncthreads();
    }
    auto zero_row = [&]() {
        if (STAGED) {
#pragma unroll
            for (int c = 0; c < ROWF / 4; ++c) *reinterpret_cast<float4*>(my_row + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto body = [&]() __attribute__((always_inline)) {
    if (i >= N) return;
    const bool vis = radii[i] > 0;
    if (accumulate && !vis) { zero_row(); return; }  // += 0 everywhere: nothing to do for a culled Gaussian
    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float4 drot = make_float4(0.f, 0.f, 0.f, 0.f);
    float* dsh = dL_dsh ? dL_dsh + (size_t)i * M * 3 : nullptr;

    if (vis) {
        const float px_ = means3D[3 * i], py_ = means3D[3 * i + 1], pz_ = means3D[3 * i + 2];
        const float4 g2 = grad_rec[4 * i];          // mean2D x, y, |x|, |y|
        float4 gconic = grad_rec[4 * i + 1];        // conic.xyz (K7 leaves the exact factors -1/2, -1, -1/2 to us), ddepth
        gconic.x *= -0.5f; gconic.y = -gconic.y; gconic.z *= -0.5f;
        const float4 gcolor = grad_rec[4 * i + 2];  // rgb, opacity
        {
            float dop = gcolor.w;
            if (flags & GDR_IN_RAW_OPACITY) {  // d sigmoid = o (1 - o), o as stored by K1
                const float o = g_rec[4 * i + 1].w;
                dop = dop * (o * (1.f - o));
            }
            if (accumulate) {
                const float4 old = dL_dmean2D[i];
                dL_dmean2D[i] = make_float4(old.x + g2.x, old.y + g2.y, old.z + g2.z, old.w + g2.w);
                dL_dopacity[i] += dop;
            } else {
                dL_dmean2D[i] = g2;
                dL_dopacity[i] = dop;
            }
        }
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cov3D[6 * i + k];

        const float pvx = cam.v[0] * px_ + cam.v[4] * py_ + cam.v[8] * pz_ + cam.v[12];
        const float pvy = cam.v[1] * px_ + cam.v[5] * py_ + cam.v[9] * pz_ + cam.v[13];
        const float pvz = cam.v[2] * px_ + cam.v[6] * py_ + cam.v[10] * pz_ + cam.v[14];
        Ewa e;
        ewa(cam, pvx, pvy, pvz, c6, focal_x, focal_y, tanx, tany, e);
        const float a = e.a, b = e.b, c = e.c;
        const float det = a * c - b * b;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (det * det != 0.f) {
            const float d2 = 1.f / (det * det);
            dL_da = d2 * (-c * c * gconic.x + b * c * gconic.y - b * b * gconic.z);
            dL_db = d2 * (2.f * b * c * gconic.x - (a * c + b * b) * gconic.y + 2.f * a * b * gconic.z);
            dL_dc = d2 * (-b * b * gconic.x + a * b * gconic.y - a * a * gconic.z);
            const float* A0 = e.A0;
            const float* A1 = e.A1;
            dcov[0] = A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
            dcov[3] = A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
            dcov[5] = A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
            dcov[1] = 2.f * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + 2.f * A1[0] * A1[1] * dL_dc;
            dcov[2] = 2.f * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + 2.f * A1[0] * A1[2] * dL_dc;
            dcov[4] = 2.f * A0[1] * A0[2] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + 2.f * A1[1] * A1[2] * dL_dc;
        }
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float dA0 = 2.f * dL_da * e.v0[k] + dL_db * e.v1[k];
            const float dA1 = 2.f * dL_dc * e.v1[k] + dL_db * e.v0[k];
            dJ00 += dA0 * cam.v[4 * k + 0];
            dJ02 += dA0 * cam.v[4 * k + 2];
            dJ11 += dA1 * cam.v[4 * k + 1];
            dJ12 += dA1 * cam.v[4 * k + 2];
        }
        const float tz1 = 1.f / e.tz, tz2 = tz1 * tz1, tz3 = tz2 * tz1;
        const float dtx = e.xmul * (-focal_x * tz2 * dJ02);
        const float dty = e.ymul * (-focal_y * tz2 * dJ12);
        const float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 +
                          (2.f * focal_x * e.tx) * tz3 * dJ02 + (2.f * focal_y * e.ty) * tz3 * dJ12;
        // projection of the mean (A.5-ii) and depth path (A.5-iii)
        const float mhx = cam.p[0] * px_ + cam.p[4] * py_ + cam.p[8] * pz_ + cam.p[12];
        const float mhy = cam.p[1] * px_ + cam.p[5] * py_ + cam.p[9] * pz_ + cam.p[13];
        const float mhw = cam.p[3] * px_ + cam.p[7] * py_ + cam.p[11] * pz_ + cam.p[15];
        const float m_w = 1.f / (mhw + 0.0000001f);
        const float mul1 = mhx * m_w * m_w, mul2 = mhy * m_w * m_w;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dmean[k] = cam.v[4 * k + 0] * dtx + cam.v[4 * k + 1] * dty + cam.v[4 * k + 2] * dtz;
            dmean[k] += (cam.p[4 * k + 0] * m_w - cam.p[4 * k + 3] * mul1) * g2.x +
                        (cam.p[4 * k + 1] * m_w - cam.p[4 * k + 3] * mul2) * g2.y;
            if (!(flags & GDR_IN_NO_DEPTH_TO_MEAN)) dmean[k] += cam.v[4 * k + 2] * gconic.w;  // risk R1 switch
        }
        // SH backward (A.5-iv)
        if (!colors_precomp) {
            float dx = px_ - cam.c[0], dy = py_ - cam.c[1], dz = pz_ - cam.c[2];
            const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
            const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
            float bk[NB], bx[NB], by[NB], bz[NB];
            sh_basis<DEG>(ux, uy, uz, bk);
            sh_basis_grad<DEG>(ux, uy, uz, bx, by, bz);
            const uint32_t cl = g_clamped[i];
            const float g[3] = {(cl & 1u) ? 0.f : gcolor.x, (cl & 2u) ? 0.f : gcolor.y,
                                (cl & 4u) ? 0.f : gcolor.z};
            const float* sh_g = shs + (size_t)i * M * 3;
            float sh[NB * 3];
            if (STAGED) {
#pragma unroll
                for (int c = 0; c < ROWF / 4; ++c) {
                    const float4 t = *reinterpret_cast<const float4*>(my_row + 4 * c);
                    sh[4 * c] = t.x; sh[4 * c + 1] = t.y; sh[4 * c + 2] = t.z; sh[4 * c + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NB * 3; ++k) sh[k] = sh_g[k];
            }
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float sg = sh[3 * k + ch] * g[ch];
                    ddx += bx[k] * sg;
                    ddy += by[k] * sg;
                    ddz += bz[k] * sg;
                }
            }
            // dL/dsh[k][ch] = b_k g_ch: written (or added) as whole 16-byte chunks when the SH row
            // of a Gaussian is 16-byte aligned (M*12 % 16 == 0: deg 1 and 3), else scalar
            if (STAGED) {  // the gradient row replaces the SH row this thread owns; written out coalesced after body()
#pragma unroll
                for (int c = 0; c < ROWF / 4; ++c)
                    *reinterpret_cast<float4*>(my_row + 4 * c) =
                        make_float4(bk[(4 * c) / 3] * g[(4 * c) % 3], bk[(4 * c + 1) / 3] * g[(4 * c + 1) % 3],
                                    bk[(4 * c + 2) / 3] * g[(4 * c + 2) % 3], bk[(4 * c + 3) / 3] * g[(4 * c + 3) % 3]);
            } else if ((3 * NB) % 4 == 0 && M == NB) {
                float4* d4 = reinterpret_cast<float4*>(dsh);
#pragma unroll
                for (int c = 0; c < (3 * NB) / 4; ++c) {
                    float4 v = make_float4(bk[(4 * c) / 3] * g[(4 * c) % 3], bk[(4 * c + 1) / 3] * g[(4 * c + 1) % 3],
                                           bk[(4 * c + 2) / 3] * g[(4 * c + 2) % 3], bk[(4 * c + 3) / 3] * g[(4 * c + 3) % 3]);
                    if (accumulate) {
                        const float4 o = d4[c];
                        v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
                    }
                    d4[c] = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NB; ++k)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        if (accumulate) dsh[3 * k + ch] += bk[k] * g[ch];
                        else dsh[3 * k + ch] = bk[k] * g[ch];
                    }
                if (!accumulate)
                    for (int k = NB; k < M; ++k)
                        for (int ch = 0; ch < 3; ++ch) dsh[3 * k + ch] = 0.f;
            }
            const float dot = ux * ddx + uy * ddy + uz * ddz;
            dmean[0] += (ddx - ux * dot) * inv;
            dmean[1] += (ddy - uy * dot) * inv;
            dmean[2] += (ddz - uz * dot) * inv;
        } else if (dL_dcolors) {
            if (accumulate) {
                dL_dcolors[3 * i] += gcolor.x; dL_dcolors[3 * i + 1] += gcolor.y; dL_dcolors[3 * i + 2] += gcolor.z;
            } else {
                dL_dcolors[3 * i] = gcolor.x; dL_dcolors[3 * i + 1] = gcolor.y; dL_dcolors[3 * i + 2] = gcolor.z;
            }
        }
        // cov3D -> scale / quaternion (A.5-v)
        if (!cov_precomp) {
            float4 q = reinterpret_cast<const float4*>(rotations)[i];
            float sc0 = scales[3 * i], sc1 = scales[3 * i + 1], sc2 = scales[3 * i + 2];
            float inv_n = 1.f;
            if (flags & GDR_IN_RAW_ROTATIONS) q = act_normalize(q, &inv_n);
            if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); sc2 = expf(sc2); }
            float R[9];
            quat_to_R(q.x, q.y, q.z, q.w, R);
            const float s[3] = {scale_modifier * sc0, scale_modifier * sc1, scale_modifier * sc2};
            const float Gs[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3],
                                 0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
            float dR[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float ds = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float acc = 0.f;
#pragma unroll
                    for (int l = 0; l < 3; ++l) acc += Gs[3 * r + l] * (R[3 * l + k] * s[k]);
                    const float dM = 2.f * acc;
                    ds += R[3 * r + k] * dM;
                    dR[3 * r + k] = s[k] * dM;
                }
                dscale[k] = scale_modifier * ds;
            }
            const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
#define G_(r_, c_) dR[3 * (r_) + (c_)]
            drot.x = 2.f * (-qz * G_(0, 1) + qy * G_(0, 2) + qz * G_(1, 0) - qx * G_(1, 2) - qy * G_(2, 0) + qx * G_(2, 1));
            drot.y = 2.f * (qy * G_(0, 1) + qz * G_(0, 2) + qy * G_(1, 0) - 2.f * qx * G_(1, 1) - qr * G_(1, 2) + qz * G_(2, 0) + qr * G_(2, 1) - 2.f * qx * G_(2, 2));
            drot.z = 2.f * (-2.f * qy * G_(0, 0) + qx * G_(0, 1) + qr * G_(0, 2) + qx * G_(1, 0) + qz * G_(1, 2) - qr * G_(2, 0) + qz * G_(2, 1) - 2.f * qy * G_(2, 2));
            drot.w = 2.f * (-2.f * qz * G_(0, 0) - qr * G_(0, 1) + qx * G_(0, 2) + qr * G_(1, 0) - 2.f * qz * G_(1, 1) + qy * G_(1, 2) + qx * G_(2, 0) + qy * G_(2, 1));
#undef G_
            if (flags & GDR_IN_RAW_SCALES) {  // d exp = s
                dscale[0] *= sc0; dscale[1] *= sc1; dscale[2] *= sc2;
            }
            if (flags & GDR_IN_RAW_ROTATIONS) {  // d (q/|q|) = (I - q^ q^T) / |q|
                const float dot = (q.x * drot.x + q.y * drot.y) + (q.z * drot.z + q.w * drot.w);
                drot = make_float4((drot.x - q.x * dot) * inv_n, (drot.y - q.y * dot) * inv_n,
                                   (drot.z - q.z * dot) * inv_n, (drot.w - q.w * dot) * inv_n);
            }
        }
    } else {
        dL_dmean2D[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dL_dopacity[i] = 0.f;
        zero_row();
        if (dsh && !STAGED)
            for (int k = 0; k < 3 * M; ++k) dsh[k] = 0.f;
        if (colors_precomp && dL_dcolors) {
            dL_dcolors[3 * i] = 0.f; dL_dcolors[3 * i + 1] = 0.f; dL_dcolors[3 * i + 2] = 0.f;
        }
    }
    if (accumulate) {
        dL_dmeans3D[3 * i] += dmean[0];
        dL_dmeans3D[3 * i + 1] += dmean[1];
        dL_dmeans3D[3 * i + 2] += dmean[2];
        if (cov_precomp) {
            if (dL_dcov3D)
#pragma unroll
                for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] += dcov[k];
        } else {
            dL_dscale[3 * i] += dscale[0];
            dL_dscale[3 * i + 1] += dscale[1];
            dL_dscale[3 * i + 2] += dscale[2];
            const float4 old = dL_drot[i];
            dL_drot[i] = make_float4(old.x + drot.x, old.y + drot.y, old.z + drot.z, old.w + drot.w);
        }
        return;
    }
    dL_dmeans3D[3 * i] = dmean[0];
    dL_dmeans3D[3 * i + 1] = dmean[1];
    dL_dmeans3D[3 * i + 2] = dmean[2];
    if (cov_precomp) {
        if (dL_dcov3D)
#pragma unroll
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dcov[k];
    } else {
        dL_dscale[3 * i] = dscale[0];
        dL_dscale[3 * i + 1] = dscale[1];
        dL_dscale[3 * i + 2] = dscale[2];
        dL_drot[i] = drot;
    }
    };
    body();
    if (STAGED) {
        __syncthreads();
        stage_rows_out<STAGED ? ROWF : 4>(dL_dsh, row0, nrows, lds_rows, accumulate != 0);
    }
}


// =================================================================================
// Multi-view variants: the callers render V views of ONE Gaussian set back to
// back (network.py:826-838).  One thread still owns one Gaussian but loops over the views, so
// the view-independent work is done once: inputs (236 B at SH degree 3) are read once instead
// of V times, cov3D is computed/stored once, and in the backward the per-view partial
// gradients are summed in registers and every output is written once (no read-modify-write
// passes, no V-fold SH gradient traffic).  Per-view arithmetic is expression-for-expression the
// single-view kernels', so integer intermediates stay bit-identical.
// =================================================================================
struct FwdView {
    const float* view; const float* proj; const float* campos;
    float tanx, tany, fx, fy;
    int32_t* radii; float* depths; float4* rec; int4* rect;
    uint32_t* tiles; uint8_t* clamped; uint32_t* block_sums; uint32_t* block_offs; uint32_t* num_rendered;
};
struct FwdViewsArgs { int V; FwdView v[GDR_MAX_VIEWS]; };

// STAGED (M == NB and 3 NB a multiple of 4: degrees 1 and 3; round 5): the per-Gaussian rows move through LDS with
// coalesced global accesses, as K9's do (device_math.h RowStage).  A thread owns a Gaussian, so its SH row (192 bytes at
// degree 3) and the 64-byte render record it writes per view are stride accesses: every wave-level float4 load / store touches
// 64 different cache lines.  IN: the workgroup copies its 256 consecutive SH rows (one contiguous 48 KB span) into LDS with
// lane-contiguous float4 loads, every thread takes its row into registers, and the same LDS then serves OUT: each wave
// transposes its 64 records (a contiguous 4 KB span per view) through a wave-private slice and stores them as four fully
// coalesced 1 KB float4 stores.  Arithmetic untouched: same values, same bits.
template <int DEG, bool STAGED = false>
__global__ __launch_bounds__(GDR_BLOCK) void preprocess_fwd_views_kernel(
    int N, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
    float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
    const float* __restrict__ shs, int W, int H, float* __restrict__ g_cov3D, uint32_t flags,
    const FwdViewsArgs a) {
    __shared__ uint32_t wsum[GDR_MAX_VIEWS][GDR_BLOCK / GDR_WAVE];
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int ROWF = 3 * NB;
    using RS = RowStage<STAGED ? ROWF : 4>;
    constexpr int REC_STRIDE = 20;   // floats per record in the OUT slice: (stride / 4) odd -> 16-byte accesses rotate through the banks
    // IN staging only where the rows are small (degree 1: 12 floats, 20 KB per workgroup).  At degree 3 the 53 KB it takes
    // leave two workgroups per CU instead of three and K1 runs 30 % SLOWER than with the strided loads (same-box A/B,
    // profiles/r05_ab_k1_staged.txt: C4 286 -> 378 us, C2 46 -> 54 us): the OUT staging alone is kept there.
    constexpr bool STAGE_SH = STAGED && DEG == 1;
    constexpr int LDS_FLOATS = STAGED ? (STAGE_SH && RS::LDS_FLOATS > GDR_BLOCK * REC_STRIDE ? RS::LDS_FLOATS : GDR_BLOCK * REC_STRIDE) : 1;
    __shared__ float lds_rows[LDS_FLOATS];
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const int gx = (W + GDR_TILE - 1) / GDR_TILE, gy = (H + GDR_TILE - 1) / GDR_TILE;
    const bool valid = i < N;
    float px_ = 0.f, py_ = 0.f, pz_ = 0.f, op = 0.f;
    float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float sh[NB * 3];
    bool sh_loaded = false;
    if (STAGE_SH) {
        const int row0 = blockIdx.x * GDR_BLOCK;
        stage_rows_in<STAGED ? ROWF : 4>(shs, row0, min(GDR_BLOCK, N - row0), lds_rows);
        __syncthreads();
        if (valid) {
            const float* my_row = lds_rows + (int)threadIdx.x * RS::STRIDE;
#pragma unroll
            for (int c = 0; c < ROWF / 4; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(my_row + 4 * c);
                sh[4 * c] = t.x; sh[4 * c + 1] = t.y; sh[4 * c + 2] = t.z; sh[4 * c + 3] = t.w;
            }
        }
        sh_loaded = true;
        __syncthreads();     // the rows are in registers: the LDS is the waves' OUT slices from here on
    }
    if (valid) {
        px_ = means3D[3 * i]; py_ = means3D[3 * i + 1]; pz_ = means3D[3 * i + 2];
        op = (flags & GDR_IN_RAW_OPACITY) ? act_sigmoid(opacities[i]) : opacities[i];
        float4 q = reinterpret_cast<const float4*>(rotations)[i];
        float sc0 = scales[3 * i], sc1 = scales[3 * i + 1], sc2 = scales[3 * i + 2];
        if (flags & GDR_IN_RAW_ROTATIONS) { float inv_n; q = act_normalize(q, &inv_n); }
        if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); sc2 = expf(sc2); }
        float R[9], Mm[9];
        quat_to_R(q.x, q.y, q.z, q.w, R);
        const float s[3] = {scale_modifier * sc0, scale_modifier * sc1, scale_modifier * sc2};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) Mm[r * 3 + k] = R[r * 3 + k] * s[k];
#define SIG(a_, b_) ((Mm[a_ * 3 + 0] * Mm[b_ * 3 + 0] + Mm[a_ * 3 + 1] * Mm[b_ * 3 + 1]) + Mm[a_ * 3 + 2] * Mm[b_ * 3 + 2])
        c6[0] = SIG(0, 0); c6[1] = SIG(0, 1); c6[2] = SIG(0, 2);
        c6[3] = SIG(1, 1); c6[4] = SIG(1, 2); c6[5] = SIG(2, 2);
#undef SIG
#pragma unroll
        for (int k = 0; k < 6; ++k) g_cov3D[6 * i + k] = c6[k];
    }
    for (int v = 0; v < a.V; ++v) {
        const FwdView& fv = a.v[v];
        Cam cam;
        load_cam(cam, fv.view, fv.proj, fv.campos);
        uint32_t tiles = 0;
        if (valid) {
            int rad = 0;
            float depth = 0.f;
            float2 pxy = make_float2(0.f, 0.f);
            float4 con_o = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 rgbd = make_float4(0.f, 0.f, 0.f, 0.f);
            int4 rect = make_int4(0, 0, 0, 0);
            uint32_t clampbits = 0;
            const float pvx = cam.v[0] * px_ + cam.v[4] * py_ + cam.v[8] * pz_ + cam.v[12];
            const float pvy = cam.v[1] * px_ + cam.v[5] * py_ + cam.v[9] * pz_ + cam.v[13];
            const float pvz = cam.v[2] * px_ + cam.v[6] * py_ + cam.v[10] * pz_ + cam.v[14];
            bool ok = pvz > 0.2f;
            if (ok) {
                const float phx = cam.p[0] * px_ + cam.p[4] * py_ + cam.p[8] * pz_ + cam.p[12];
                const float phy = cam.p[1] * px_ + cam.p[5] * py_ + cam.p[9] * pz_ + cam.p[13];
                const float phw = cam.p[3] * px_ + cam.p[7] * py_ + cam.p[11] * pz_ + cam.p[15];
                const float p_w = 1.0f / (phw + 0.0000001f);
                const float ppx = phx * p_w, ppy = phy * p_w;
                Ewa e;
                ewa(cam, pvx, pvy, pvz, c6, fv.fx, fv.fy, fv.tanx, fv.tany, e);
                const float det = e.a * e.c - e.b * e.b;
                ok = det != 0.f;
                if (ok) {
                    const float det_inv = 1.f / det;
                    const float mid = 0.5f * (e.a + e.c);
                    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                    const float lambda1 = mid + disc, lambda2 = mid - disc;
                    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                    const float sx = ((ppx + 1.0f) * (float)W - 1.0f) * 0.5f;
                    const float sy = ((ppy + 1.0f) * (float)H - 1.0f) * 0.5f;
                    const int r_i = (int)my_radius;
                    const float rf = (float)r_i;
                    rect.x = min(gx, max(0, (int)((sx - rf) / (float)GDR_TILE)));
                    rect.y = min(gy, max(0, (int)((sy - rf) / (float)GDR_TILE)));
                    rect.z = min(gx, max(0, (int)((sx + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                    rect.w = min(gy, max(0, (int)((sy + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                    tiles = (uint32_t)((rect.z - rect.x) * (rect.w - rect.y));
                    ok = tiles != 0;
                    if (ok) {
                        rad = r_i;
                        depth = pvz;
                        pxy = make_float2(sx, sy);
                        con_o = make_float4(e.c * det_inv, -e.b * det_inv, e.a * det_inv, op);
                        if (!sh_loaded) {  // first view in which this Gaussian is visible
                            const float* src = shs + (size_t)i * M * 3;
                            if ((M * 3) % 4 == 0) {
#pragma unroll
                                for (int c = 0; c < (NB * 3) / 4; ++c) {
                                    const float4 t = reinterpret_cast<const float4*>(src)[c];
                                    sh[4 * c] = t.x; sh[4 * c + 1] = t.y; sh[4 * c + 2] = t.z; sh[4 * c + 3] = t.w;
                                }
#pragma unroll
                                for (int k = ((NB * 3) / 4) * 4; k < NB * 3; ++k) sh[k] = src[k];
                            } else {
#pragma unroll
                                for (int k = 0; k < NB * 3; ++k) sh[k] = src[k];
                            }
                            sh_loaded = true;
                        }
                        float dx = px_ - cam.c[0], dy = py_ - cam.c[1], dz = pz_ - cam.c[2];
                        const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
                        dx *= inv; dy *= inv; dz *= inv;
                        float bk[NB];
                        sh_basis<DEG>(dx, dy, dz, bk);
                        float acc[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) acc[ch] = bk[0] * sh[ch];
#pragma unroll
                        for (int k = 1; k < NB; ++k)
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) acc[ch] = acc[ch] + bk[k] * sh[3 * k + ch];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            acc[ch] = acc[ch] + 0.5f;
                            if (acc[ch] < 0.f) clampbits |= (1u << ch);
                            acc[ch] = fmaxf(acc[ch], 0.f);
                        }
                        rgbd = make_float4(acc[0], acc[1], acc[2], depth);
                    } else {
                        rect = make_int4(0, 0, 0, 0);
                    }
                }
                if (!ok) tiles = 0;
            }
            fv.radii[i] = rad;
            fv.depths[i] = depth;
            if (!STAGED) write_rec(fv.rec, i, pxy, depth, con_o, rgbd);
            fv.rect[i] = rect;
            fv.tiles[i] = tiles;
            fv.clamped[i] = (uint8_t)clampbits;
            if (STAGED) {       // this lane's record into the wave's OUT slice
                const float2 ext = alpha_extent(con_o);
                float* mine = lds_rows + (int)threadIdx.x * REC_STRIDE;
                *reinterpret_cast<float4*>(mine) = make_float4(pxy.x, pxy.y, depth, 0.f);
                *reinterpret_cast<float4*>(mine + 4) = con_o;
                *reinterpret_cast<float4*>(mine + 8) = rgbd;
                *reinterpret_cast<float4*>(mine + 12) = make_float4(ext.x, ext.y, 0.f, 0.f);
            }
        }
        if (STAGED) {           // 64 records = 4 KB contiguous in the view's record array: four coalesced 1 KB stores per wave
            const int wave0 = (int)(threadIdx.x & ~63u), lane = (int)(threadIdx.x & 63u);
            const int first = blockIdx.x * GDR_BLOCK + wave0;          // first Gaussian of this wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int nrec = min(GDR_WAVE, N - first);                   // (<= 0 for a wave past the end)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = lane + GDR_WAVE * j;                       // float4 index inside the wave's 4 KB span
                const int r = q >> 2, c = q & 3;
                if (r < nrec)
                    fv.rec[4 * (size_t)first + q] = *reinterpret_cast<const float4*>(lds_rows + (wave0 + r) * REC_STRIDE + 4 * c);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();                             // the slice is free for the next view
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        uint32_t t = tiles;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if ((threadIdx.x & 63) == 0) wsum[v][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.V) {
        const uint32_t bs = wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
        a.v[threadIdx.x].block_sums[blockIdx.x] = bs;
        a.v[threadIdx.x].block_offs[blockIdx.x] = bs ? atomicAdd(a.v[threadIdx.x].num_rendered, bs) : 0u;
    }
}

struct BwdView {
    const float* view; const float* proj; const float* campos;
    float tanx, tany, fx, fy;
    const int32_t* radii; const uint8_t* clamped; const float4* grad_rec;
};
struct BwdViewsArgs { int V; BwdView v[GDR_MAX_VIEWS]; };

// STAGED (M == NB and 3 NB a multiple of 4: degrees 1 and 3): SH rows in, SH gradient rows out through LDS with
// coalesced accesses (device_math.h RowStage)
template <int DEG, bool STAGED, bool PREFETCH = false>
__global__ __launch_bounds__(GDR_BLOCK) void preprocess_bwd_views_kernel(
    int N, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ opacities, float scale_modifier, const float* __restrict__ cov3D, int W,
    int H, uint32_t flags, int accumulate, float4* __restrict__ dL_dmean2D,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscale, float4* __restrict__ dL_drot, const BwdViewsArgs a) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int ROWF = 3 * NB;
    using RS = RowStage<STAGED ? ROWF : 4>;
    __shared__ float lds_rows[STAGED ? RS::LDS_FLOATS : 1];
    const int row0 = blockIdx.x * GDR_BLOCK;
    const int nrows = min(GDR_BLOCK, N - row0);
    const int i = row0 + threadIdx.x;
    if (STAGED) {
        stage_rows_in<STAGED ? ROWF : 4>(shs, row0, nrows, lds_rows);
        __syncthreads();
    }
    const bool in_range = i < N;
    if (!STAGED && !in_range) return;
    float* my_row = lds_rows + (STAGED ? (int)threadIdx.x * RS::STRIDE : 0);
    if (in_range) {
    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 dm2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float dop = 0.f;
    float dsh[NB * 3];
#pragma unroll
    for (int k = 0; k < NB * 3; ++k) dsh[k] = 0.f;
    bool any_vis = false;
    const float px_ = means3D[3 * i], py_ = means3D[3 * i + 1], pz_ = means3D[3 * i + 2];
    float c6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = cov3D[6 * i + k];
    const float* sh_g = shs + (size_t)i * M * 3;
    auto sh = [&](int idx) -> float { return STAGED ? my_row[idx] : sh_g[idx]; };

    // The gradient record of view v+1 is fetched while view v is processed: with 2 waves per SIMD (219 VGPRs) a
    // dependent radii -> record load per view left ~3 TB/s worth of bytes in flight.
    int n_rad = 0;
    float4 n_g2 = make_float4(0.f, 0.f, 0.f, 0.f), n_gconic = n_g2, n_gcolor = n_g2;
    uint32_t n_cl = 0u;
    auto fetch_view = [&](int v) __attribute__((always_inline)) {
        const BwdView& b = a.v[v];
        n_rad = b.radii[i];
        if (n_rad > 0) {
            n_g2 = b.grad_rec[4 * i]; n_gconic = b.grad_rec[4 * i + 1]; n_gcolor = b.grad_rec[4 * i + 2];
            n_cl = b.clamped[i];
        }
    };
    if (PREFETCH) fetch_view(0);
    for (int v = 0; v < a.V; ++v) {
        const BwdView& bv = a.v[v];
        int rad;
        float4 g2, gconic, gcolor;
        uint32_t cl_v = 0u;
        if (PREFETCH) {
            rad = n_rad; g2 = n_g2; gconic = n_gconic; gcolor = n_gcolor; cl_v = n_cl;
            if (v + 1 < a.V) fetch_view(v + 1);
        } else {
            rad = bv.radii[i];
        }
        if (rad <= 0) continue;
        any_vis = true;
        Cam cam;
        load_cam(cam, bv.view, bv.proj, bv.campos);
        if (!PREFETCH) {
            g2 = bv.grad_rec[4 * i]; gconic = bv.grad_rec[4 * i + 1]; gcolor = bv.grad_rec[4 * i + 2];
            cl_v = bv.clamped[i];
        }
        gconic.x *= -0.5f; gconic.y = -gconic.y; gconic.z *= -0.5f;   // (K7 leaves these exact factors to us)
        dm2 = make_float4(dm2.x + g2.x, dm2.y + g2.y, dm2.z + g2.z, dm2.w + g2.w);
        dop += gcolor.w;
        const float pvx = cam.v[0] * px_ + cam.v[4] * py_ + cam.v[8] * pz_ + cam.v[12];
        const float pvy = cam.v[1] * px_ + cam.v[5] * py_ + cam.v[9] * pz_ + cam.v[13];
        const float pvz = cam.v[2] * px_ + cam.v[6] * py_ + cam.v[10] * pz_ + cam.v[14];
        Ewa e;
        ewa(cam, pvx, pvy, pvz, c6, bv.fx, bv.fy, bv.tanx, bv.tany, e);
        const float ea = e.a, eb = e.b, ec = e.c;
        const float det = ea * ec - eb * eb;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (det * det != 0.f) {
            const float d2 = 1.f / (det * det);
            dL_da = d2 * (-ec * ec * gconic.x + eb * ec * gconic.y - eb * eb * gconic.z);
            dL_db = d2 * (2.f * eb * ec * gconic.x - (ea * ec + eb * eb) * gconic.y + 2.f * ea * eb * gconic.z);
            dL_dc = d2 * (-eb * eb * gconic.x + ea * eb * gconic.y - ea * ea * gconic.z);
            const float* A0 = e.A0;
            const float* A1 = e.A1;
            dcov[0] += A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
            dcov[3] += A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
            dcov[5] += A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
            dcov[1] += 2.f * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + 2.f * A1[0] * A1[1] * dL_dc;
            dcov[2] += 2.f * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + 2.f * A1[0] * A1[2] * dL_dc;
            dcov[4] += 2.f * A0[1] * A0[2] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + 2.f * A1[1] * A1[2] * dL_dc;
        }
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float dA0 = 2.f * dL_da * e.v0[k] + dL_db * e.v1[k];
            const float dA1 = 2.f * dL_dc * e.v1[k] + dL_db * e.v0[k];
            dJ00 += dA0 * cam.v[4 * k + 0];
            dJ02 += dA0 * cam.v[4 * k + 2];
            dJ11 += dA1 * cam.v[4 * k + 1];
            dJ12 += dA1 * cam.v[4 * k + 2];
        }
        const float tz1 = 1.f / e.tz, tz2 = tz1 * tz1, tz3 = tz2 * tz1;
        const float dtx = e.xmul * (-bv.fx * tz2 * dJ02);
        const float dty = e.ymul * (-bv.fy * tz2 * dJ12);
        const float dtz = -bv.fx * tz2 * dJ00 - bv.fy * tz2 * dJ11 + (2.f * bv.fx * e.tx) * tz3 * dJ02 +
                          (2.f * bv.fy * e.ty) * tz3 * dJ12;
        const float mhx = cam.p[0] * px_ + cam.p[4] * py_ + cam.p[8] * pz_ + cam.p[12];
        const float mhy = cam.p[1] * px_ + cam.p[5] * py_ + cam.p[9] * pz_ + cam.p[13];
        const float mhw = cam.p[3] * px_ + cam.p[7] * py_ + cam.p[11] * pz_ + cam.p[15];
        const float m_w = 1.f / (mhw + 0.0000001f);
        const float mul1 = mhx * m_w * m_w, mul2 = mhy * m_w * m_w;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dmean[k] += cam.v[4 * k + 0] * dtx + cam.v[4 * k + 1] * dty + cam.v[4 * k + 2] * dtz;
            dmean[k] += (cam.p[4 * k + 0] * m_w - cam.p[4 * k + 3] * mul1) * g2.x +
                        (cam.p[4 * k + 1] * m_w - cam.p[4 * k + 3] * mul2) * g2.y;
            if (!(flags & GDR_IN_NO_DEPTH_TO_MEAN)) dmean[k] += cam.v[4 * k + 2] * gconic.w;  // risk R1 switch
        }
        {   // SH backward
            float dx = px_ - cam.c[0], dy = py_ - cam.c[1], dz = pz_ - cam.c[2];
            const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
            const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
            float bk[NB], bx[NB], by[NB], bz[NB];
            sh_basis<DEG>(ux, uy, uz, bk);
            sh_basis_grad<DEG>(ux, uy, uz, bx, by, bz);
            const uint32_t cl = cl_v;
            const float g[3] = {(cl & 1u) ? 0.f : gcolor.x, (cl & 2u) ? 0.f : gcolor.y,
                                (cl & 4u) ? 0.f : gcolor.z};
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
            float shr[NB * 3];  // this view's copy of the SH row: 16-byte LDS reads when staged
            if (STAGED) {
#pragma unroll
                for (int c = 0; c < ROWF / 4; ++c) {
                    const float4 t = *reinterpret_cast<const float4*>(my_row + 4 * c);
                    shr[4 * c] = t.x; shr[4 * c + 1] = t.y; shr[4 * c + 2] = t.z; shr[4 * c + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NB * 3; ++k) shr[k] = sh(k);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float sg = shr[3 * k + ch] * g[ch];
                    dsh[3 * k + ch] += bk[k] * g[ch];
                    ddx += bx[k] * sg;
                    ddy += by[k] * sg;
                    ddz += bz[k] * sg;
                }
            }
            const float dot = ux * ddx + uy * ddy + uz * ddz;
            dmean[0] += (ddx - ux * dot) * inv;
            dmean[1] += (ddy - uy * dot) * inv;
            dmean[2] += (ddz - uz * dot) * inv;
        }
    }
    if (!STAGED && accumulate && !any_vis) return;
    float dscale[3] = {0.f, 0.f, 0.f};
    float4 drot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any_vis) {
        if (flags & GDR_IN_RAW_OPACITY) {
            const float o = act_sigmoid(opacities[i]);
            dop = dop * (o * (1.f - o));
        }
        float4 q = reinterpret_cast<const float4*>(rotations)[i];
        float sc0 = scales[3 * i], sc1 = scales[3 * i + 1], sc2 = scales[3 * i + 2];
        float inv_n = 1.f;
        if (flags & GDR_IN_RAW_ROTATIONS) q = act_normalize(q, &inv_n);
        if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); sc2 = expf(sc2); }
        float R[9];
        quat_to_R(q.x, q.y, q.z, q.w, R);
        const float s[3] = {scale_modifier * sc0, scale_modifier * sc1, scale_modifier * sc2};
        const float Gs[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3],
                             0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
        float dR[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float ds = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int l = 0; l < 3; ++l) acc += Gs[3 * r + l] * (R[3 * l + k] * s[k]);
                const float dM = 2.f * acc;
                ds += R[3 * r + k] * dM;
                dR[3 * r + k] = s[k] * dM;
            }
            dscale[k] = scale_modifier * ds;
        }
        const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
#define G_(r_, c_) dR[3 * (r_) + (c_)]
        drot.x = 2.f * (-qz * G_(0, 1) + qy * G_(0, 2) + qz * G_(1, 0) - qx * G_(1, 2) - qy * G_(2, 0) + qx * G_(2, 1));
        drot.y = 2.f * (qy * G_(0, 1) + qz * G_(0, 2) + qy * G_(1, 0) - 2.f * qx * G_(1, 1) - qr * G_(1, 2) + qz * G_(2, 0) + qr * G_(2, 1) - 2.f * qx * G_(2, 2));
        drot.z = 2.f * (-2.f * qy * G_(0, 0) + qx * G_(0, 1) + qr * G_(0, 2) + qx * G_(1, 0) + qz * G_(1, 2) - qr * G_(2, 0) + qz * G_(2, 1) - 2.f * qy * G_(2, 2));
        drot.w = 2.f * (-2.f * qz * G_(0, 0) - qr * G_(0, 1) + qx * G_(0, 2) + qr * G_(1, 0) - 2.f * qz * G_(1, 1) + qy * G_(1, 2) + qx * G_(2, 0) + qy * G_(2, 1));
#undef G_
        if (flags & GDR_IN_RAW_SCALES) { dscale[0] *= sc0; dscale[1] *= sc1; dscale[2] *= sc2; }
        if (flags & GDR_IN_RAW_ROTATIONS) {
            const float dot = (q.x * drot.x + q.y * drot.y) + (q.z * drot.z + q.w * drot.w);
            drot = make_float4((drot.x - q.x * dot) * inv_n, (drot.y - q.y * dot) * inv_n,
                               (drot.z - q.z * dot) * inv_n, (drot.w - q.w * dot) * inv_n);
        }
    }
    float* o_sh = dL_dsh + (size_t)i * M * 3;
    if (STAGED && accumulate && !any_vis) {
        // nothing to add for this Gaussian (zeros go through the staged SH rows below)
    } else {
    if (accumulate) {
        const float4 om = dL_dmean2D[i];
        dm2 = make_float4(dm2.x + om.x, dm2.y + om.y, dm2.z + om.z, dm2.w + om.w);
        dop += dL_dopacity[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) { dmean[k] += dL_dmeans3D[3 * i + k]; dscale[k] += dL_dscale[3 * i + k]; }
        const float4 orot = dL_drot[i];
        drot = make_float4(drot.x + orot.x, drot.y + orot.y, drot.z + orot.z, drot.w + orot.w);
    }
    dL_dmean2D[i] = dm2;
    dL_dopacity[i] = dop;
#pragma unroll
    for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = dmean[k]; dL_dscale[3 * i + k] = dscale[k]; }
    dL_drot[i] = drot;
    }
    if (STAGED) {
        // every lane is past its last read of its SH row (it owns that row exclusively): overwrite it with the gradient
#pragma unroll
        for (int c = 0; c < ROWF / 4; ++c)
            *reinterpret_cast<float4*>(my_row + 4 * c) = make_float4(dsh[4 * c], dsh[4 * c + 1], dsh[4 * c + 2], dsh[4 * c + 3]);
    } else if ((M * 3) % 4 == 0 && M == NB) {
        float4* d4 = reinterpret_cast<float4*>(o_sh);
#pragma unroll
        for (int c = 0; c < (3 * NB) / 4; ++c) {
            float4 t = make_float4(dsh[4 * c], dsh[4 * c + 1], dsh[4 * c + 2], dsh[4 * c + 3]);
            if (accumulate) {
                const float4 o = d4[c];
                t = make_float4(t.x + o.x, t.y + o.y, t.z + o.z, t.w + o.w);
            }
            d4[c] = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) {
            if (accumulate) o_sh[k] += dsh[k];
            else o_sh[k] = dsh[k];
        }
        if (!accumulate)
            for (int k = NB * 3; k < M * 3; ++k) o_sh[k] = 0.f;
    }
    }  // in_range
    if (STAGED) {
        __syncthreads();
        stage_rows_out<STAGED ? ROWF : 4>(dL_dsh, row0, nrows, lds_rows, accumulate != 0);
    }
}

__global__ __launch_bounds__(GDR_BLOCK) void mark_visible_kernel(int N, const float* __restrict__ means3D,
                                                                  const float* __restrict__ view,
                                                                  uint8_t* __restrict__ present) {
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    if (i >= N) return;
    const float z = view[2] * means3D[3 * i] + view[6] * means3D[3 * i + 1] + view[10] * means3D[3 * i + 2] + view[14];
    present[i] = z > 0.2f ? 1 : 0;
}

}  // namespace

#define LAUNCH_DEG(KID, KERNEL, deg, grid, st, ...)                                          \
    switch (deg) {                                                                          \
        case 0: GDR_LAUNCH(KID, KERNEL<0>, dim3(grid), dim3(GDR_BLOCK), st, __VA_ARGS__); break; \
        case 1: GDR_LAUNCH(KID, KERNEL<1>, dim3(grid), dim3(GDR_BLOCK), st, __VA_ARGS__); break; \
        case 2: GDR_LAUNCH(KID, KERNEL<2>, dim3(grid), dim3(GDR_BLOCK), st, __VA_ARGS__); break; \
        default: GDR_LAUNCH(KID, KERNEL<3>, dim3(grid), dim3(GDR_BLOCK), st, __VA_ARGS__); break; \
    }

hipError_t launch_preprocess_fwd(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* g,
                                 int32_t* radii, hipStream_t st) {
    const int N = in->N;
    if (N == 0) return hipSuccess;
    const int W = s->image_width, H = s->image_height;
    const float focal_x = (float)W / (2.f * s->tanfovx), focal_y = (float)H / (2.f * s->tanfovy);
    const int grid = div_up(N, GDR_BLOCK);
    const int deg = in->shs ? s->sh_degree : 0;
    LAUNCH_DEG(GDR_K_PREPROCESS_FWD, preprocess_fwd_kernel, deg, grid, st, N, in->M, in->means3D, in->scales,
               s->scale_modifier, in->rotations, in->opacities, in->shs, in->colors_precomp,
               in->cov3D_precomp, s->viewmatrix, s->projmatrix, s->campos, W, H, s->tanfovx,
               s->tanfovy, focal_x, focal_y, radii, g->depths, (float4*)g->rec, g->cov3D, (int4*)g->rect,
               g->tiles_touched, g->clamped, g->block_sums, in->flags, g->block_offs, g->num_rendered);
    return hipGetLastError();
}

hipError_t launch_preprocess_bwd(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* g,
                                 const int32_t* radii, const gdr_grad_outputs* go, hipStream_t st) {
    const int N = in->N;
    if (N == 0) return hipSuccess;
    const int W = s->image_width, H = s->image_height;
    const float focal_x = (float)W / (2.f * s->tanfovx), focal_y = (float)H / (2.f * s->tanfovy);
    const int grid = div_up(N, GDR_BLOCK);
    const int deg = in->shs ? s->sh_degree : 0;
    const float* cov3D = in->cov3D_precomp ? in->cov3D_precomp : g->cov3D;
#define GDR_K9(DEG_, ST_) GDR_LAUNCH(GDR_K_PREPROCESS_BWD, (preprocess_bwd_kernel<DEG_, ST_>), dim3(grid), dim3(GDR_BLOCK), st, N, in->M, in->means3D, radii, in->shs, \
               g->clamped, in->scales, in->rotations, s->scale_modifier, cov3D, \
               in->cov3D_precomp ? 1 : 0, in->colors_precomp ? 1 : 0, s->viewmatrix, \
               s->projmatrix, s->campos, W, H, s->tanfovx, s->tanfovy, focal_x, focal_y, \
               (const float4*)go->scratch, (float4*)go->dL_dmeans2D, go->dL_dopacities, go->dL_dmeans3D, \
               go->dL_dcov3D, go->dL_dshs, go->dL_dcolors, go->dL_dscales, \
               (float4*)go->dL_drotations, (const float4*)g->rec, in->flags, go->accumulate)
    {
        const int nb_ = (deg + 1) * (deg + 1);
        const bool staged = in->shs && go->dL_dshs && in->M == nb_ && (3 * nb_) % 4 == 0;
        switch (deg) {
            case 0: GDR_K9(0, false); break;
            case 1: if (staged) GDR_K9(1, true); else GDR_K9(1, false); break;
            case 2: GDR_K9(2, false); break;
            default: if (staged) GDR_K9(3, true); else GDR_K9(3, false); break;
        }
    }
#undef GDR_K9
    return hipGetLastError();
}

hipError_t launch_preprocess_fwd_views(int V, const gdr_settings* s, const gdr_inputs* in,
                                       const gdr_geom* geoms, int32_t* const* radii, hipStream_t st) {
    const int N = in->N;
    if (N == 0) return hipSuccess;
    const int W = s[0].image_width, H = s[0].image_height;
    FwdViewsArgs a;
    a.V = V;
    for (int v = 0; v < V; ++v) {
        FwdView& f = a.v[v];
        f.view = s[v].viewmatrix; f.proj = s[v].projmatrix; f.campos = s[v].campos;
        f.tanx = s[v].tanfovx; f.tany = s[v].tanfovy;
        f.fx = (float)W / (2.f * s[v].tanfovx); f.fy = (float)H / (2.f * s[v].tanfovy);
        f.radii = radii[v]; f.depths = geoms[v].depths; f.rec = (float4*)geoms[v].rec;
        f.rect = (int4*)geoms[v].rect; f.tiles = geoms[v].tiles_touched; f.clamped = geoms[v].clamped;
        f.block_sums = geoms[v].block_sums; f.block_offs = geoms[v].block_offs;
        f.num_rendered = geoms[v].num_rendered;
    }
    const int grid = div_up(N, GDR_BLOCK);
    const int deg = s[0].sh_degree, nb = (deg + 1) * (deg + 1);
    static const bool stage = getenv("GDR_K1_STAGED") ? atoi(getenv("GDR_K1_STAGED")) != 0 : true;   // (developer A/B)
    const bool staged = stage && in->M == nb && (3 * nb) % 4 == 0 && (deg == 1 || deg == 3);
#define GDR_K1V(DEG_, ST_)                                                                                              \
    GDR_LAUNCH(GDR_K_PREPROCESS_FWD, (preprocess_fwd_views_kernel<DEG_, ST_>), dim3(grid), dim3(GDR_BLOCK), st, N, in->M,    \
               in->means3D, in->scales, s[0].scale_modifier, in->rotations, in->opacities, in->shs, W, H, geoms[0].cov3D,   \
               in->flags, a)
    if (staged && deg == 3) GDR_K1V(3, true);
    else if (staged && deg == 1) GDR_K1V(1, true);
    else
        LAUNCH_DEG(GDR_K_PREPROCESS_FWD, preprocess_fwd_views_kernel, s[0].sh_degree, grid, st, N, in->M,
                   in->means3D, in->scales, s[0].scale_modifier, in->rotations, in->opacities, in->shs, W, H,
                   geoms[0].cov3D, in->flags, a);
#undef GDR_K1V
    return hipGetLastError();
}

hipError_t launch_preprocess_bwd_views(int V, const gdr_settings* s, const gdr_inputs* in,
                                       const gdr_geom* geoms, const int32_t* const* radii,
                                       float* const* grad_recs, const gdr_grad_outputs* go,
                                       hipStream_t st) {
    const int N = in->N;
    if (N == 0) return hipSuccess;
    const int W = s[0].image_width, H = s[0].image_height;
    BwdViewsArgs a;
    a.V = V;
    for (int v = 0; v < V; ++v) {
        BwdView& b = a.v[v];
        b.view = s[v].viewmatrix; b.proj = s[v].projmatrix; b.campos = s[v].campos;
        b.tanx = s[v].tanfovx; b.tany = s[v].tanfovy;
        b.fx = (float)W / (2.f * s[v].tanfovx); b.fy = (float)H / (2.f * s[v].tanfovy);
        b.radii = radii[v]; b.clamped = geoms[v].clamped; b.grad_rec = (const float4*)grad_recs[v];
    }
    const int grid = div_up(N, GDR_BLOCK);
    // record prefetch (see the kernel): degree 3 only — 404 -> 381 us per 4 views at 2 M Gaussians with the same 2 waves
    // per SIMD; at degree 1 the extra registers cost a wave per SIMD (C3: 82 -> 90 us)
    static const bool prefetch = getenv("GDR_K9_PREFETCH") ? atoi(getenv("GDR_K9_PREFETCH")) != 0 : true;
#define GDR_K9V(DEG_, ST_)                                                                                  \
    if (prefetch && V > 1 && DEG_ == 3)                                                                     \
        GDR_LAUNCH(GDR_K_PREPROCESS_BWD, (preprocess_bwd_views_kernel<DEG_, ST_, true>), dim3(grid), dim3(GDR_BLOCK), st, N,  \
               in->M, in->means3D, in->shs, in->scales, in->rotations, in->opacities, s[0].scale_modifier,     \
               geoms[0].cov3D, W, H, in->flags, go->accumulate, (float4*)go->dL_dmeans2D, go->dL_dopacities,   \
               go->dL_dmeans3D, go->dL_dshs, go->dL_dscales, (float4*)go->dL_drotations, a);                  \
    else                                                                                                    \
    GDR_LAUNCH(GDR_K_PREPROCESS_BWD, (preprocess_bwd_views_kernel<DEG_, ST_>), dim3(grid), dim3(GDR_BLOCK), st, N,  \
               in->M, in->means3D, in->shs, in->scales, in->rotations, in->opacities, s[0].scale_modifier,     \
               geoms[0].cov3D, W, H, in->flags, go->accumulate, (float4*)go->dL_dmeans2D, go->dL_dopacities,   \
               go->dL_dmeans3D, go->dL_dshs, go->dL_dscales, (float4*)go->dL_drotations, a)
    const int deg = s[0].sh_degree, nb = (deg + 1) * (deg + 1);
    const bool staged = in->M == nb && (3 * nb) % 4 == 0;
    switch (deg) {
        case 0: GDR_K9V(0, false); break;
        case 1: if (staged) GDR_K9V(1, true); else GDR_K9V(1, false); break;
        case 2: GDR_K9V(2, false); break;
        default: if (staged) GDR_K9V(3, true); else GDR_K9V(3, false); break;
    }
#undef GDR_K9V
    return hipGetLastError();
}

hipError_t launch_mark_visible(int N, const float* means3D, const float* view, uint8_t* present,
                               hipStream_t st) {
    if (N == 0) return hipSuccess;
    GDR_LAUNCH(GDR_K_MARK_VISIBLE, mark_visible_kernel, dim3(div_up(N, GDR_BLOCK)), dim3(GDR_BLOCK), st, N,
                       means3D, view, present);
    return hipGetLastError();
}

}  // namespace gdr
