// preprocess_surfel.hip — per-surfel stages of the 2D-Gaussian (surfel) rasterizer (include/gsr.h):
//   K1s  splat-to-pixel homography T (rows Tu, Tv, Tw), view-space normal facing the camera, 3-sigma bounding
//        box -> radius / tile rect / tiles_touched, SH colour, the 96-byte render record (incl. a conservative box
//        around {alpha >= 1/255} for K6s/K7s' block culling) and the block's slice of the duplicate list;
//   K9s  gradient record (dL/dT, low-pass centre, normal, colour, opacity) -> means3D, scales (N,2), rotations,
//        SH, opacity and the (N,4) screen-space densification signal.
// What the reference obtains from `diff_surfel_rasterization` at /root/reference/lightning/renderer_2dgs.py:224-234;
// arithmetic as restated in oracle/gsr_oracle.c.  This translation unit is compiled with -ffp-contract=off and
// mirrors the oracle's expression trees, so radii / rects / tiles_touched / T / depths are bit-identical.
// HBM-bound like the 3DGS K1/K9: ~230 B read + ~130 B written per surfel at SH degree 3.
#include "gdr_common.h"
#include "device_math.h"
#include "../../include/gsr.h"

#pragma clang fp contract(off)

namespace gdr {

namespace {

#define GSR_FILTER_SIZE 0.707106f
#define GSR_ALPHA_MIN_INV 255.0

struct SurfelT {
    float Tu[3], Tv[3], Tw[3];
};

// clip = [v, w] @ proj with the oracle's association ((m0 x + m4 y) + m8 z) (+ m12)
__device__ __forceinline__ void clip_of(const float* v, bool w, const float* m, float* c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = (m[j] * v[0] + m[4 + j] * v[1]) + m[8 + j] * v[2];
        c[j] = w ? a + m[12 + j] : a;
    }
}

__device__ __forceinline__ void surfel_transmat(const float* p, float s0, float s1, const float* R, const float* proj,
                                                int W, int H, SurfelT& T) {
    const float L0[3] = {s0 * R[0], s0 * R[3], s0 * R[6]}, L1[3] = {s1 * R[1], s1 * R[4], s1 * R[7]};
    const float hw = (float)W / 2.f, hh = (float)H / 2.f, cw = (float)(W - 1) / 2.f, ch = (float)(H - 1) / 2.f;
    float c[4];
    clip_of(L0, false, proj, c);
    T.Tu[0] = c[0] * hw + c[3] * cw; T.Tv[0] = c[1] * hh + c[3] * ch; T.Tw[0] = c[3];
    clip_of(L1, false, proj, c);
    T.Tu[1] = c[0] * hw + c[3] * cw; T.Tv[1] = c[1] * hh + c[3] * ch; T.Tw[1] = c[3];
    clip_of(p, true, proj, c);
    T.Tu[2] = c[0] * hw + c[3] * cw; T.Tv[2] = c[1] * hh + c[3] * ch; T.Tw[2] = c[3];
}

// conservative box around {pixels where alpha = o G can reach 1/255}: G >= 1/(255 o) <=> rho <= c2 = 2 ln(255 o),
// rho = min(rho3d, rho2d) => union of the projected c-sigma disc (exact box of the conic, evaluated in fp64: the
// fp32 form cancels catastrophically off-centre) and the low-pass disc of radius sqrt(c2/2) around the centre.
// lo > hi: never visible; (-inf, inf): no bound (the c-sigma disc crosses the camera plane).
__device__ __forceinline__ void alpha_box(const SurfelT& T, float cx, float cy, float opacity, float* lo, float* hi) {
    const double t = GSR_ALPHA_MIN_INV * (double)opacity;
    if (!(t > 1.0)) { lo[0] = lo[1] = INFINITY; hi[0] = hi[1] = -INFINITY; return; }
    const double c2 = 2.0 * log(t);
    const double w0 = T.Tw[0], w1 = T.Tw[1], w2 = T.Tw[2];
    const double d = c2 * (w0 * w0 + w1 * w1) - w2 * w2;
    if (!(d < 0.0)) { lo[0] = lo[1] = -INFINITY; hi[0] = hi[1] = INFINITY; return; }
    const double f0 = c2 / d, f2 = -1.0 / d;
    const double rB = sqrt(0.5 * c2);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float* Tr = a == 0 ? T.Tu : T.Tv;
        const double u0 = Tr[0], u1 = Tr[1], u2 = Tr[2];
        const double p = f0 * (u0 * w0 + u1 * w1) + f2 * u2 * w2;
        const double h = sqrt(fmax(p * p - (f0 * (u0 * u0 + u1 * u1) + f2 * u2 * u2), 0.0));
        const double c = a == 0 ? (double)cx : (double)cy;
        const double l = fmin(p - h, c - rB), u = fmax(p + h, c + rB);
        const double margin = 0.02 + 1e-3 * (u - l);
        lo[a] = (float)(l - margin);
        hi[a] = (float)(u + margin);
    }
}

template <int DEG>
__global__ __launch_bounds__(GDR_BLOCK) void surfel_preprocess_fwd_kernel(
    int N, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, const float* __restrict__ transMat_precomp,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos, int W, int H,
    int32_t* __restrict__ radii, float* __restrict__ g_depths, float4* __restrict__ g_rec, int4* __restrict__ g_rect,
    uint32_t* __restrict__ g_tiles, uint8_t* __restrict__ g_clamped, uint32_t* __restrict__ block_sums,
    uint32_t flags, uint32_t* __restrict__ block_offs, uint32_t* __restrict__ num_rendered) {
    Cam cam;
    load_cam(cam, view, proj, campos);
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const int gx = (W + GDR_TILE - 1) / GDR_TILE, gy = (H + GDR_TILE - 1) / GDR_TILE;

    uint32_t tiles = 0;
    if (i < N) {
        int rad = 0;
        float depth = 0.f;
        int4 rect = make_int4(0, 0, 0, 0);
        uint32_t clampbits = 0;
        SurfelT T;
#pragma unroll
        for (int k = 0; k < 3; ++k) T.Tu[k] = T.Tv[k] = T.Tw[k] = 0.f;
        float nv[3] = {0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f}, cxy[2] = {0.f, 0.f}, op = 0.f;
        float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};

        const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
        const float pvx = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
        const float pvy = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
        const float pvz = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
        bool ok = pvz > 0.2f;
        if (ok) {
            SurfelT Tl;
            float n[3];
            if (transMat_precomp) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Tl.Tu[k] = transMat_precomp[9 * i + k];
                    Tl.Tv[k] = transMat_precomp[9 * i + 3 + k];
                    Tl.Tw[k] = transMat_precomp[9 * i + 6 + k];
                }
                n[0] = 0.f; n[1] = 0.f; n[2] = 1.f;
            } else {
                float4 q = reinterpret_cast<const float4*>(rotations)[i];
                float sc0 = scales[2 * i], sc1 = scales[2 * i + 1];
                if (flags & GDR_IN_RAW_ROTATIONS) { float inv_n; q = act_normalize(q, &inv_n); }
                if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); }
                float R[9];
                quat_to_R(q.x, q.y, q.z, q.w, R);
                surfel_transmat(p, scale_modifier * sc0, scale_modifier * sc1, R, cam.p, W, H, Tl);
                n[0] = (cam.v[0] * R[2] + cam.v[4] * R[5]) + cam.v[8] * R[8];
                n[1] = (cam.v[1] * R[2] + cam.v[5] * R[5]) + cam.v[9] * R[8];
                n[2] = (cam.v[2] * R[2] + cam.v[6] * R[5]) + cam.v[10] * R[8];
            }
            const float cosv = -((pvx * n[0] + pvy * n[1]) + pvz * n[2]);
            ok = cosv != 0.f;
            const float mult = cosv > 0.f ? 1.f : -1.f;
            // 3-sigma bounding box (oracle surfel_aabb)
            const float t0 = 9.f, t2 = -1.f;
            const float d = (t0 * Tl.Tw[0] * Tl.Tw[0] + t0 * Tl.Tw[1] * Tl.Tw[1]) + t2 * Tl.Tw[2] * Tl.Tw[2];
            ok = ok && d != 0.f;
            if (ok) {
                const float inv_d = 1.f / d;
                const float f0 = t0 * inv_d, f2 = t2 * inv_d;
                const float cx = (f0 * Tl.Tu[0] * Tl.Tw[0] + f0 * Tl.Tu[1] * Tl.Tw[1]) + f2 * Tl.Tu[2] * Tl.Tw[2];
                const float cy = (f0 * Tl.Tv[0] * Tl.Tw[0] + f0 * Tl.Tv[1] * Tl.Tw[1]) + f2 * Tl.Tv[2] * Tl.Tw[2];
                const float hx0 = cx * cx - ((f0 * Tl.Tu[0] * Tl.Tu[0] + f0 * Tl.Tu[1] * Tl.Tu[1]) + f2 * Tl.Tu[2] * Tl.Tu[2]);
                const float hy0 = cy * cy - ((f0 * Tl.Tv[0] * Tl.Tv[0] + f0 * Tl.Tv[1] * Tl.Tv[1]) + f2 * Tl.Tv[2] * Tl.Tv[2]);
                const float ex = sqrtf(fmaxf(1e-4f, hx0)), ey = sqrtf(fmaxf(1e-4f, hy0));
                const float my_radius = ceilf(fmaxf(fmaxf(ex, ey), 3.f * GSR_FILTER_SIZE));
                const int r_i = (int)my_radius;
                const float rf = (float)r_i;
                rect.x = min(gx, max(0, (int)((cx - rf) / (float)GDR_TILE)));
                rect.y = min(gy, max(0, (int)((cy - rf) / (float)GDR_TILE)));
                rect.z = min(gx, max(0, (int)((cx + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                rect.w = min(gy, max(0, (int)((cy + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                tiles = (uint32_t)((rect.z - rect.x) * (rect.w - rect.y));
                ok = tiles != 0;
                if (ok) {
                    rad = r_i;
                    depth = pvz;
                    T = Tl;
                    cxy[0] = cx; cxy[1] = cy;
                    nv[0] = mult * n[0]; nv[1] = mult * n[1]; nv[2] = mult * n[2];
                    op = (flags & GDR_IN_RAW_OPACITY) ? act_sigmoid(opacities[i]) : opacities[i];
                    alpha_box(T, cx, cy, op, lo, hi);
                    if (colors_precomp) {
                        rgb[0] = colors_precomp[3 * i]; rgb[1] = colors_precomp[3 * i + 1]; rgb[2] = colors_precomp[3 * i + 2];
                    } else {
                        float dx = p[0] - cam.c[0], dy = p[1] - cam.c[1], dz = p[2] - cam.c[2];
                        const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
                        dx *= inv; dy *= inv; dz *= inv;
                        constexpr int NB = (DEG + 1) * (DEG + 1);
                        float bk[NB];
                        sh_basis<DEG>(dx, dy, dz, bk);
                        const float* sh = shs + (size_t)i * M * 3;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) rgb[ch] = bk[0] * sh[ch];
#pragma unroll
                        for (int k = 1; k < NB; ++k)
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) rgb[ch] = rgb[ch] + bk[k] * sh[3 * k + ch];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            rgb[ch] = rgb[ch] + 0.5f;
                            if (rgb[ch] < 0.f) clampbits |= (1u << ch);
                            rgb[ch] = fmaxf(rgb[ch], 0.f);
                        }
                    }
                } else {
                    rect = make_int4(0, 0, 0, 0);
                }
            }
            if (!ok) tiles = 0;
        }
        radii[i] = rad;
        g_depths[i] = depth;
        float4* r = g_rec + 6 * (size_t)i;
        r[0] = make_float4(T.Tu[0], T.Tu[1], T.Tu[2], cxy[0]);
        r[1] = make_float4(T.Tv[0], T.Tv[1], T.Tv[2], cxy[1]);
        r[2] = make_float4(T.Tw[0], T.Tw[1], T.Tw[2], op);
        r[3] = make_float4(nv[0], nv[1], nv[2], rgb[0]);
        r[4] = make_float4(rgb[1], rgb[2], lo[0], lo[1]);
        r[5] = make_float4(hi[0], hi[1], 0.f, 0.f);
        g_rect[i] = rect;
        g_tiles[i] = tiles;
        g_clamped[i] = (uint8_t)clampbits;
    }
    uint32_t v = tiles;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __shared__ uint32_t wsum[GDR_BLOCK / GDR_WAVE];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t bs = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        block_sums[blockIdx.x] = bs;
        block_offs[blockIdx.x] = bs ? atomicAdd(num_rendered, bs) : 0u;
    }
}

template <typename T>
__device__ __forceinline__ void put(T* p, T v, int accumulate) {
    *p = accumulate ? *p + v : v;
}

// K9s.  grad_rec: (N,32) floats accumulated by K7s, see include/gsr.h.  STAGED (M == NB, degrees 1 and 3): SH rows
// in and SH-gradient rows out (incl. the read-modify-write of accumulate mode) through LDS with coalesced accesses
// (device_math.h RowStage).
template <int DEG, bool STAGED>
__global__ __launch_bounds__(GDR_BLOCK) void surfel_preprocess_bwd_kernel(
    int N, int M, const float* __restrict__ means3D, const int32_t* __restrict__ radii, const float* __restrict__ shs,
    const uint8_t* __restrict__ g_clamped, const float* __restrict__ scales, const float* __restrict__ rotations,
    float scale_modifier, int transmat_precomp, int colors_precomp, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos, int W, int H,
    const float4* __restrict__ grad_rec, const float4* __restrict__ g_rec, float4* __restrict__ dL_dmean2D,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dtransMat,
    float* __restrict__ dL_dsh, float* __restrict__ dL_dcolors, float* __restrict__ dL_dscale,
    float4* __restrict__ dL_drot, uint32_t flags, int accumulate) {
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
    // STAGED: every thread reaches the ONE barrier + staged write-out after body(); its row then holds the SH gradient
    // (zeros if culled)
    auto zero_row = [&]() {
        if (STAGED) {
#pragma unroll
            for (int c = 0; c < ROWF / 4; ++c) *reinterpret_cast<float4*>(my_row + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto body = [&]() __attribute__((always_inline)) {
    if (i >= N) return;
    const bool vis = radii[i] > 0;
    if (accumulate && !vis) { zero_row(); return; }
    float* dsh = dL_dsh ? dL_dsh + (size_t)i * M * 3 : nullptr;
    if (!vis) {
        zero_row();
        dL_dmean2D[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dL_dopacity[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0.f;
        if (dsh && !STAGED)
            for (int k = 0; k < 3 * M; ++k) dsh[k] = 0.f;
        if (colors_precomp && dL_dcolors) { dL_dcolors[3 * i] = 0.f; dL_dcolors[3 * i + 1] = 0.f; dL_dcolors[3 * i + 2] = 0.f; }
        if (transmat_precomp) {
            if (dL_dtransMat)
                for (int k = 0; k < 9; ++k) dL_dtransMat[9 * i + k] = 0.f;
        } else {
            dL_dscale[2 * i] = 0.f; dL_dscale[2 * i + 1] = 0.f;
            dL_drot[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float4 g0 = grad_rec[8 * (size_t)i], g1 = grad_rec[8 * (size_t)i + 1], g2 = grad_rec[8 * (size_t)i + 2];
    const float4 g3 = grad_rec[8 * (size_t)i + 3], g4 = grad_rec[8 * (size_t)i + 4];
    float dT[9] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x};
    float dop = g2.y;
    const float gcol[3] = {g2.z, g2.w, g3.x};
    const float gax = g3.y, gay = g3.z, glx = g3.w;       // record words 13..15 (include/gsr.h)
    const float gnrm[3] = {g4.x, g4.y, g4.z};
    const float gly = g4.w;
    const float4 r0 = g_rec[6 * (size_t)i], r1 = g_rec[6 * (size_t)i + 1], r2 = g_rec[6 * (size_t)i + 2];
    const float Tu[3] = {r0.x, r0.y, r0.z}, Tv[3] = {r1.x, r1.y, r1.z}, Tw[3] = {r2.x, r2.y, r2.z};
    const float depth = Tw[2];
    {   // densification signal (uses the RAW render-stage dL/dT) and opacity
        const float4 m2 = make_float4(dT[2] * depth * 0.5f * (float)W, dT[5] * depth * 0.5f * (float)H,
                                      gax * depth * 0.5f * (float)W, gay * depth * 0.5f * (float)H);
        if (flags & GDR_IN_RAW_OPACITY) { const float o = r2.w; dop = dop * (o * (1.f - o)); }
        if (accumulate) {
            const float4 old = dL_dmean2D[i];
            dL_dmean2D[i] = make_float4(old.x + m2.x, old.y + m2.y, old.z + m2.z, old.w + m2.w);
            dL_dopacity[i] += dop;
        } else {
            dL_dmean2D[i] = m2;
            dL_dopacity[i] = dop;
        }
    }
    // low-pass branch: centre = 3-sigma box centre(T)
    if (glx != 0.f || gly != 0.f) {
        const float t[3] = {9.f, 9.f, -1.f};
        const float d = (t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1]) + t[2] * Tw[2] * Tw[2];
        const float inv_d = 1.f / d;
        float f[3], dfdot = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f[k] = t[k] * inv_d;
            dT[0 + k] += glx * f[k] * Tw[k];
            dT[3 + k] += gly * f[k] * Tw[k];
            dT[6 + k] += glx * f[k] * Tu[k] + gly * f[k] * Tv[k];
            dfdot += (glx * Tu[k] * Tw[k] + gly * Tv[k] * Tw[k]) * f[k];
        }
        const float dL_dd = -dfdot * inv_d;
#pragma unroll
        for (int k = 0; k < 3; ++k) dT[6 + k] += dL_dd * 2.f * t[k] * Tw[k];
    }
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float dmean[3] = {0.f, 0.f, 0.f};
    if (transmat_precomp) {
        if (dL_dtransMat)
#pragma unroll
            for (int k = 0; k < 9; ++k) put(dL_dtransMat + 9 * i + k, dT[k], accumulate);
    } else {
        const float hw = (float)W / 2.f, hh = (float)H / 2.f, cw = (float)(W - 1) / 2.f, ch_ = (float)(H - 1) / 2.f;
        float dv[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float dc0 = dT[0 + a] * hw, dc1 = dT[3 + a] * hh, dc3 = dT[0 + a] * cw + dT[3 + a] * ch_ + dT[6 + a];
#pragma unroll
            for (int r = 0; r < 3; ++r) dv[a][r] = cam.p[4 * r + 0] * dc0 + cam.p[4 * r + 1] * dc1 + cam.p[4 * r + 3] * dc3;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) dmean[r] += dv[2][r];
        float4 q = reinterpret_cast<const float4*>(rotations)[i];
        float sc0 = scales[2 * i], sc1 = scales[2 * i + 1];
        float inv_n = 1.f;
        if (flags & GDR_IN_RAW_ROTATIONS) q = act_normalize(q, &inv_n);
        if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); }
        float R[9];
        quat_to_R(q.x, q.y, q.z, q.w, R);
        const float s0 = scale_modifier * sc0, s1 = scale_modifier * sc1;
        // n_view = mult * (n_world @ view3x3): the sign K1s chose
        const float pvx = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
        const float pvy = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
        const float pvz = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
        const float n0 = (cam.v[0] * R[2] + cam.v[4] * R[5]) + cam.v[8] * R[8];
        const float n1 = (cam.v[1] * R[2] + cam.v[5] * R[5]) + cam.v[9] * R[8];
        const float n2 = (cam.v[2] * R[2] + cam.v[6] * R[5]) + cam.v[10] * R[8];
        const float cosv = -((pvx * n0 + pvy * n1) + pvz * n2);
        const float mult = cosv > 0.f ? 1.f : -1.f;
        float dR[9], ds0 = 0.f, ds1 = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dR[3 * r + 0] = s0 * dv[0][r];
            dR[3 * r + 1] = s1 * dv[1][r];
            dR[3 * r + 2] = mult * (cam.v[4 * r + 0] * gnrm[0] + cam.v[4 * r + 1] * gnrm[1] + cam.v[4 * r + 2] * gnrm[2]);
            ds0 += R[3 * r + 0] * dv[0][r];
            ds1 += R[3 * r + 1] * dv[1][r];
        }
        float dscale0 = scale_modifier * ds0, dscale1 = scale_modifier * ds1;
        const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
        float4 drot;
#define G_(r_, c_) dR[3 * (r_) + (c_)]
        drot.x = 2.f * (-qz * G_(0, 1) + qy * G_(0, 2) + qz * G_(1, 0) - qx * G_(1, 2) - qy * G_(2, 0) + qx * G_(2, 1));
        drot.y = 2.f * (qy * G_(0, 1) + qz * G_(0, 2) + qy * G_(1, 0) - 2.f * qx * G_(1, 1) - qr * G_(1, 2) + qz * G_(2, 0) + qr * G_(2, 1) - 2.f * qx * G_(2, 2));
        drot.z = 2.f * (-2.f * qy * G_(0, 0) + qx * G_(0, 1) + qr * G_(0, 2) + qx * G_(1, 0) + qz * G_(1, 2) - qr * G_(2, 0) + qz * G_(2, 1) - 2.f * qy * G_(2, 2));
        drot.w = 2.f * (-2.f * qz * G_(0, 0) - qr * G_(0, 1) + qx * G_(0, 2) + qr * G_(1, 0) - 2.f * qz * G_(1, 1) + qy * G_(1, 2) + qx * G_(2, 0) + qy * G_(2, 1));
#undef G_
        if (flags & GDR_IN_RAW_SCALES) { dscale0 *= sc0; dscale1 *= sc1; }
        if (flags & GDR_IN_RAW_ROTATIONS) {
            const float dot = (q.x * drot.x + q.y * drot.y) + (q.z * drot.z + q.w * drot.w);
            drot = make_float4((drot.x - q.x * dot) * inv_n, (drot.y - q.y * dot) * inv_n, (drot.z - q.z * dot) * inv_n,
                               (drot.w - q.w * dot) * inv_n);
        }
        put(dL_dscale + 2 * i, dscale0, accumulate);
        put(dL_dscale + 2 * i + 1, dscale1, accumulate);
        if (accumulate) {
            const float4 old = dL_drot[i];
            drot = make_float4(old.x + drot.x, old.y + drot.y, old.z + drot.z, old.w + drot.w);
        }
        dL_drot[i] = drot;
    }
    if (!colors_precomp) {
        float dx = p[0] - cam.c[0], dy = p[1] - cam.c[1], dz = p[2] - cam.c[2];
        const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
        const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
        float bk[NB], bx[NB], by[NB], bz[NB];
        sh_basis<DEG>(ux, uy, uz, bk);
        sh_basis_grad<DEG>(ux, uy, uz, bx, by, bz);
        const uint32_t cl = g_clamped[i];
        const float g[3] = {(cl & 1u) ? 0.f : gcol[0], (cl & 2u) ? 0.f : gcol[1], (cl & 4u) ? 0.f : gcol[2]};
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
                ddx += bx[k] * sg; ddy += by[k] * sg; ddz += bz[k] * sg;
            }
        }
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
                for (int ch = 0; ch < 3; ++ch) put(dsh + 3 * k + ch, bk[k] * g[ch], accumulate);
            if (!accumulate)
                for (int k = NB; k < M; ++k)
                    for (int ch = 0; ch < 3; ++ch) dsh[3 * k + ch] = 0.f;
        }
        const float dot = ux * ddx + uy * ddy + uz * ddz;
        dmean[0] += (ddx - ux * dot) * inv;
        dmean[1] += (ddy - uy * dot) * inv;
        dmean[2] += (ddz - uz * dot) * inv;
    } else if (dL_dcolors) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) put(dL_dcolors + 3 * i + ch, gcol[ch], accumulate);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) put(dL_dmeans3D + 3 * i + k, dmean[k], accumulate);
    };
    body();
    if (STAGED) {
        __syncthreads();
        stage_rows_out<STAGED ? ROWF : 4>(dL_dsh, row0, nrows, lds_rows, accumulate != 0);
    }
}

// =================================================================================
// Multi-view variants (SURVEY §8f-1 for the surfel path): V views of ONE surfel set per launch.  One thread owns one
// surfel and loops over the views: position / scales / quaternion / opacity (and their activations, the rotation
// matrix) are read and computed once, the SH row is loaded once; in the backward the per-view gradients of the three
// world-space vectors that build T (s_u t_u, s_v t_v, p), of the normal and of the SH block are summed in registers
// and the scale / quaternion chain runs once.  Per-view arithmetic is expression-for-expression the single-view
// kernels', so integer intermediates stay bit-identical.
// =================================================================================
struct SFwdView {
    const float* view; const float* proj; const float* campos;
    int32_t* radii; float* depths; float4* rec; int4* rect; uint32_t* tiles; uint8_t* clamped;
    uint32_t* block_sums; uint32_t* block_offs; uint32_t* num_rendered;
};
struct SFwdViewsArgs { int V; SFwdView v[GDR_MAX_VIEWS]; };

template <int DEG>
__global__ __launch_bounds__(GDR_BLOCK) void surfel_preprocess_fwd_views_kernel(
    int N, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs, int W, int H,
    uint32_t flags, const SFwdViewsArgs a) {
    __shared__ uint32_t wsum[GDR_MAX_VIEWS][GDR_BLOCK / GDR_WAVE];
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const int gx = (W + GDR_TILE - 1) / GDR_TILE, gy = (H + GDR_TILE - 1) / GDR_TILE;
    const bool valid = i < N;
    float p[3] = {0.f, 0.f, 0.f}, R[9], s0 = 0.f, s1 = 0.f, op = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = 0.f;
    float sh[NB * 3];
    bool sh_loaded = false;
    if (valid) {
        p[0] = means3D[3 * i]; p[1] = means3D[3 * i + 1]; p[2] = means3D[3 * i + 2];
        op = (flags & GDR_IN_RAW_OPACITY) ? act_sigmoid(opacities[i]) : opacities[i];
        float4 q = reinterpret_cast<const float4*>(rotations)[i];
        float sc0 = scales[2 * i], sc1 = scales[2 * i + 1];
        if (flags & GDR_IN_RAW_ROTATIONS) { float inv_n; q = act_normalize(q, &inv_n); }
        if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); }
        quat_to_R(q.x, q.y, q.z, q.w, R);
        s0 = scale_modifier * sc0; s1 = scale_modifier * sc1;
    }
    for (int v = 0; v < a.V; ++v) {
        const SFwdView& fv = a.v[v];
        uint32_t tiles = 0;
        if (valid) {
            Cam cam;
            load_cam(cam, fv.view, fv.proj, fv.campos);
            int rad = 0;
            float depth = 0.f;
            int4 rect = make_int4(0, 0, 0, 0);
            uint32_t clampbits = 0;
            SurfelT T;
#pragma unroll
            for (int k = 0; k < 3; ++k) T.Tu[k] = T.Tv[k] = T.Tw[k] = 0.f;
            float nv[3] = {0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f}, cxy[2] = {0.f, 0.f}, opv = 0.f;
            float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
            const float pvx = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
            const float pvy = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
            const float pvz = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
            bool ok = pvz > 0.2f;
            if (ok) {
                SurfelT Tl;
                surfel_transmat(p, s0, s1, R, cam.p, W, H, Tl);
                const float n[3] = {(cam.v[0] * R[2] + cam.v[4] * R[5]) + cam.v[8] * R[8],
                                    (cam.v[1] * R[2] + cam.v[5] * R[5]) + cam.v[9] * R[8],
                                    (cam.v[2] * R[2] + cam.v[6] * R[5]) + cam.v[10] * R[8]};
                const float cosv = -((pvx * n[0] + pvy * n[1]) + pvz * n[2]);
                ok = cosv != 0.f;
                const float mult = cosv > 0.f ? 1.f : -1.f;
                const float t0 = 9.f, t2 = -1.f;
                const float d = (t0 * Tl.Tw[0] * Tl.Tw[0] + t0 * Tl.Tw[1] * Tl.Tw[1]) + t2 * Tl.Tw[2] * Tl.Tw[2];
                ok = ok && d != 0.f;
                if (ok) {
                    const float inv_d = 1.f / d;
                    const float f0 = t0 * inv_d, f2 = t2 * inv_d;
                    const float cx = (f0 * Tl.Tu[0] * Tl.Tw[0] + f0 * Tl.Tu[1] * Tl.Tw[1]) + f2 * Tl.Tu[2] * Tl.Tw[2];
                    const float cy = (f0 * Tl.Tv[0] * Tl.Tw[0] + f0 * Tl.Tv[1] * Tl.Tw[1]) + f2 * Tl.Tv[2] * Tl.Tw[2];
                    const float hx0 = cx * cx - ((f0 * Tl.Tu[0] * Tl.Tu[0] + f0 * Tl.Tu[1] * Tl.Tu[1]) + f2 * Tl.Tu[2] * Tl.Tu[2]);
                    const float hy0 = cy * cy - ((f0 * Tl.Tv[0] * Tl.Tv[0] + f0 * Tl.Tv[1] * Tl.Tv[1]) + f2 * Tl.Tv[2] * Tl.Tv[2]);
                    const float ex = sqrtf(fmaxf(1e-4f, hx0)), ey = sqrtf(fmaxf(1e-4f, hy0));
                    const float my_radius = ceilf(fmaxf(fmaxf(ex, ey), 3.f * GSR_FILTER_SIZE));
                    const int r_i = (int)my_radius;
                    const float rf = (float)r_i;
                    rect.x = min(gx, max(0, (int)((cx - rf) / (float)GDR_TILE)));
                    rect.y = min(gy, max(0, (int)((cy - rf) / (float)GDR_TILE)));
                    rect.z = min(gx, max(0, (int)((cx + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                    rect.w = min(gy, max(0, (int)((cy + rf + (float)(GDR_TILE - 1)) / (float)GDR_TILE)));
                    tiles = (uint32_t)((rect.z - rect.x) * (rect.w - rect.y));
                    ok = tiles != 0;
                    if (ok) {
                        rad = r_i;
                        depth = pvz;
                        T = Tl;
                        cxy[0] = cx; cxy[1] = cy;
                        nv[0] = mult * n[0]; nv[1] = mult * n[1]; nv[2] = mult * n[2];
                        opv = op;
                        alpha_box(T, cx, cy, op, lo, hi);
                        if (!sh_loaded) {  // first view in which this surfel is visible
                            const float* src = shs + (size_t)i * M * 3;
#pragma unroll
                            for (int k = 0; k < NB * 3; ++k) sh[k] = src[k];
                            sh_loaded = true;
                        }
                        float dx = p[0] - cam.c[0], dy = p[1] - cam.c[1], dz = p[2] - cam.c[2];
                        const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
                        dx *= inv; dy *= inv; dz *= inv;
                        float bk[NB];
                        sh_basis<DEG>(dx, dy, dz, bk);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) rgb[ch] = bk[0] * sh[ch];
#pragma unroll
                        for (int k = 1; k < NB; ++k)
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) rgb[ch] = rgb[ch] + bk[k] * sh[3 * k + ch];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            rgb[ch] = rgb[ch] + 0.5f;
                            if (rgb[ch] < 0.f) clampbits |= (1u << ch);
                            rgb[ch] = fmaxf(rgb[ch], 0.f);
                        }
                    } else {
                        rect = make_int4(0, 0, 0, 0);
                    }
                }
                if (!ok) tiles = 0;
            }
            fv.radii[i] = rad;
            fv.depths[i] = depth;
            float4* r = fv.rec + 6 * (size_t)i;
            r[0] = make_float4(T.Tu[0], T.Tu[1], T.Tu[2], cxy[0]);
            r[1] = make_float4(T.Tv[0], T.Tv[1], T.Tv[2], cxy[1]);
            r[2] = make_float4(T.Tw[0], T.Tw[1], T.Tw[2], opv);
            r[3] = make_float4(nv[0], nv[1], nv[2], rgb[0]);
            r[4] = make_float4(rgb[1], rgb[2], lo[0], lo[1]);
            r[5] = make_float4(hi[0], hi[1], 0.f, 0.f);
            fv.rect[i] = rect;
            fv.tiles[i] = tiles;
            fv.clamped[i] = (uint8_t)clampbits;
        }
        uint32_t t = tiles;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if ((threadIdx.x & 63) == 0) wsum[v][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.V) {
        const SFwdView& fv = a.v[threadIdx.x];
        const uint32_t bs = wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
        fv.block_sums[blockIdx.x] = bs;
        fv.block_offs[blockIdx.x] = bs ? atomicAdd(fv.num_rendered, bs) : 0u;
    }
}

struct SBwdView {
    const float* view; const float* proj; const float* campos;
    const int32_t* radii; const uint8_t* clamped; const float4* grad_rec; const float4* rec;
};
struct SBwdViewsArgs { int V; SBwdView v[GDR_MAX_VIEWS]; };

template <int DEG>
__global__ __launch_bounds__(GDR_BLOCK) void surfel_preprocess_bwd_views_kernel(
    int N, int M, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ opacities, float scale_modifier, int W, int H,
    uint32_t flags, int accumulate, float4* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
    float4* __restrict__ dL_drot, const SBwdViewsArgs a) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int ROWF = 3 * NB;
    constexpr bool STAGED = (ROWF % 4) == 0;  // launcher guarantees M == NB when ROWF % 4 == 0 is used (else DEG fallback)
    using RS = RowStage<STAGED ? ROWF : 4>;
    __shared__ float lds_rows[STAGED ? RS::LDS_FLOATS : 1];
    const int row0 = blockIdx.x * GDR_BLOCK, nrows = min(GDR_BLOCK, N - row0);
    const int i = row0 + threadIdx.x;
    float* my_row = lds_rows + (STAGED ? (int)threadIdx.x * RS::STRIDE : 0);
    if (STAGED) {
        stage_rows_in<STAGED ? ROWF : 4>(shs, row0, nrows, lds_rows);
        __syncthreads();
    }
    auto body = [&]() __attribute__((always_inline)) {
        float dsh[NB * 3];
#pragma unroll
        for (int k = 0; k < NB * 3; ++k) dsh[k] = 0.f;
        auto put_row = [&]() {
            if (STAGED) {
#pragma unroll
                for (int c = 0; c < ROWF / 4; ++c)
                    *reinterpret_cast<float4*>(my_row + 4 * c) = make_float4(dsh[4 * c], dsh[4 * c + 1], dsh[4 * c + 2], dsh[4 * c + 3]);
            }
        };
        if (i >= N) return;
        const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
        float4 q = reinterpret_cast<const float4*>(rotations)[i];
        float sc0 = scales[2 * i], sc1 = scales[2 * i + 1];
        float inv_n = 1.f;
        if (flags & GDR_IN_RAW_ROTATIONS) q = act_normalize(q, &inv_n);
        if (flags & GDR_IN_RAW_SCALES) { sc0 = expf(sc0); sc1 = expf(sc1); }
        float R[9];
        quat_to_R(q.x, q.y, q.z, q.w, R);
        const float s0 = scale_modifier * sc0, s1 = scale_modifier * sc1;
        const float hw = (float)W / 2.f, hh = (float)H / 2.f, cw = (float)(W - 1) / 2.f, ch_ = (float)(H - 1) / 2.f;
        float dv[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // dL/d(s_u t_u), dL/d(s_v t_v), dL/dp (T path)
        float dn[3] = {0.f, 0.f, 0.f}, dmean[3] = {0.f, 0.f, 0.f}, dop = 0.f;
        float4 dm2 = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any_vis = false;
        const float* sh_g = shs + (size_t)i * M * 3;
        for (int v = 0; v < a.V; ++v) {
            const SBwdView& bv = a.v[v];
            if (bv.radii[i] <= 0) continue;
            any_vis = true;
            Cam cam;
            load_cam(cam, bv.view, bv.proj, bv.campos);
            const float4* gr = bv.grad_rec + 8 * (size_t)i;
            const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4];
            float dT[9] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x};
            const float gcol[3] = {g2.z, g2.w, g3.x};
            const float gax = g3.y, gay = g3.z, glx = g3.w;       // record words 13..15 (include/gsr.h)
            const float gnrm[3] = {g4.x, g4.y, g4.z};
            const float gly = g4.w;
            const float4 r0 = bv.rec[6 * (size_t)i], r1 = bv.rec[6 * (size_t)i + 1], r2 = bv.rec[6 * (size_t)i + 2];
            const float Tu[3] = {r0.x, r0.y, r0.z}, Tv[3] = {r1.x, r1.y, r1.z}, Tw[3] = {r2.x, r2.y, r2.z};
            const float depth = Tw[2];
            dm2.x += dT[2] * depth * 0.5f * (float)W; dm2.y += dT[5] * depth * 0.5f * (float)H;
            dm2.z += gax * depth * 0.5f * (float)W;   dm2.w += gay * depth * 0.5f * (float)H;
            dop += g2.y;
            if (glx != 0.f || gly != 0.f) {
                const float t[3] = {9.f, 9.f, -1.f};
                const float d = (t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1]) + t[2] * Tw[2] * Tw[2];
                const float inv_d = 1.f / d;
                float f[3], dfdot = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    f[k] = t[k] * inv_d;
                    dT[0 + k] += glx * f[k] * Tw[k];
                    dT[3 + k] += gly * f[k] * Tw[k];
                    dT[6 + k] += glx * f[k] * Tu[k] + gly * f[k] * Tv[k];
                    dfdot += (glx * Tu[k] * Tw[k] + gly * Tv[k] * Tw[k]) * f[k];
                }
                const float dL_dd = -dfdot * inv_d;
#pragma unroll
                for (int k = 0; k < 3; ++k) dT[6 + k] += dL_dd * 2.f * t[k] * Tw[k];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float dc0 = dT[0 + c] * hw, dc1 = dT[3 + c] * hh, dc3 = dT[0 + c] * cw + dT[3 + c] * ch_ + dT[6 + c];
#pragma unroll
                for (int r = 0; r < 3; ++r) dv[c][r] += cam.p[4 * r + 0] * dc0 + cam.p[4 * r + 1] * dc1 + cam.p[4 * r + 3] * dc3;
            }
            const float pvx = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
            const float pvy = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
            const float pvz = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
            const float n0 = (cam.v[0] * R[2] + cam.v[4] * R[5]) + cam.v[8] * R[8];
            const float n1 = (cam.v[1] * R[2] + cam.v[5] * R[5]) + cam.v[9] * R[8];
            const float n2 = (cam.v[2] * R[2] + cam.v[6] * R[5]) + cam.v[10] * R[8];
            const float mult = -((pvx * n0 + pvy * n1) + pvz * n2) > 0.f ? 1.f : -1.f;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                dn[r] += mult * (cam.v[4 * r + 0] * gnrm[0] + cam.v[4 * r + 1] * gnrm[1] + cam.v[4 * r + 2] * gnrm[2]);
            {   // SH backward of this view
                float dx = p[0] - cam.c[0], dy = p[1] - cam.c[1], dz = p[2] - cam.c[2];
                const float inv = 1.f / sqrtf((dx * dx + dy * dy) + dz * dz);
                const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
                float bk[NB], bx[NB], by[NB], bz[NB];
                sh_basis<DEG>(ux, uy, uz, bk);
                sh_basis_grad<DEG>(ux, uy, uz, bx, by, bz);
                const uint32_t cl = bv.clamped[i];
                const float g[3] = {(cl & 1u) ? 0.f : gcol[0], (cl & 2u) ? 0.f : gcol[1], (cl & 4u) ? 0.f : gcol[2]};
                float shr[NB * 3];
                if (STAGED) {
#pragma unroll
                    for (int c = 0; c < ROWF / 4; ++c) {
                        const float4 t = *reinterpret_cast<const float4*>(my_row + 4 * c);
                        shr[4 * c] = t.x; shr[4 * c + 1] = t.y; shr[4 * c + 2] = t.z; shr[4 * c + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NB * 3; ++k) shr[k] = sh_g[k];
                }
                float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float sg = shr[3 * k + ch] * g[ch];
                        dsh[3 * k + ch] += bk[k] * g[ch];
                        ddx += bx[k] * sg; ddy += by[k] * sg; ddz += bz[k] * sg;
                    }
                }
                const float dot = ux * ddx + uy * ddy + uz * ddz;
                dmean[0] += (ddx - ux * dot) * inv;
                dmean[1] += (ddy - uy * dot) * inv;
                dmean[2] += (ddz - uz * dot) * inv;
            }
        }
        if (accumulate && !any_vis) { put_row(); return; }
        // scale / quaternion chain once, from the view-summed world-space gradients
#pragma unroll
        for (int r = 0; r < 3; ++r) dmean[r] += dv[2][r];
        float dR[9], ds0 = 0.f, ds1 = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dR[3 * r + 0] = s0 * dv[0][r];
            dR[3 * r + 1] = s1 * dv[1][r];
            dR[3 * r + 2] = dn[r];
            ds0 += R[3 * r + 0] * dv[0][r];
            ds1 += R[3 * r + 1] * dv[1][r];
        }
        float dscale0 = scale_modifier * ds0, dscale1 = scale_modifier * ds1;
        const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
        float4 drot;
#define G_(r_, c_) dR[3 * (r_) + (c_)]
        drot.x = 2.f * (-qz * G_(0, 1) + qy * G_(0, 2) + qz * G_(1, 0) - qx * G_(1, 2) - qy * G_(2, 0) + qx * G_(2, 1));
        drot.y = 2.f * (qy * G_(0, 1) + qz * G_(0, 2) + qy * G_(1, 0) - 2.f * qx * G_(1, 1) - qr * G_(1, 2) + qz * G_(2, 0) + qr * G_(2, 1) - 2.f * qx * G_(2, 2));
        drot.z = 2.f * (-2.f * qy * G_(0, 0) + qx * G_(0, 1) + qr * G_(0, 2) + qx * G_(1, 0) + qz * G_(1, 2) - qr * G_(2, 0) + qz * G_(2, 1) - 2.f * qy * G_(2, 2));
        drot.w = 2.f * (-2.f * qz * G_(0, 0) - qr * G_(0, 1) + qx * G_(0, 2) + qr * G_(1, 0) - 2.f * qz * G_(1, 1) + qy * G_(1, 2) + qx * G_(2, 0) + qy * G_(2, 1));
#undef G_
        if (flags & GDR_IN_RAW_SCALES) { dscale0 *= sc0; dscale1 *= sc1; }
        if (flags & GDR_IN_RAW_ROTATIONS) {
            const float dot = (q.x * drot.x + q.y * drot.y) + (q.z * drot.z + q.w * drot.w);
            drot = make_float4((drot.x - q.x * dot) * inv_n, (drot.y - q.y * dot) * inv_n, (drot.z - q.z * dot) * inv_n,
                               (drot.w - q.w * dot) * inv_n);
        }
        if (flags & GDR_IN_RAW_OPACITY) { const float o = act_sigmoid(opacities[i]); dop = dop * (o * (1.f - o)); }
        if (accumulate) {
            const float4 om = dL_dmean2D[i];
            dm2 = make_float4(dm2.x + om.x, dm2.y + om.y, dm2.z + om.z, dm2.w + om.w);
            dop += dL_dopacity[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) dmean[k] += dL_dmeans3D[3 * i + k];
            dscale0 += dL_dscale[2 * i]; dscale1 += dL_dscale[2 * i + 1];
            const float4 orot = dL_drot[i];
            drot = make_float4(drot.x + orot.x, drot.y + orot.y, drot.z + orot.z, drot.w + orot.w);
        }
        dL_dmean2D[i] = dm2;
        dL_dopacity[i] = dop;
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];
        dL_dscale[2 * i] = dscale0; dL_dscale[2 * i + 1] = dscale1;
        dL_drot[i] = drot;
        if (STAGED) {
            put_row();
        } else {
            float* o_sh = dL_dsh + (size_t)i * M * 3;
#pragma unroll
            for (int k = 0; k < NB * 3; ++k) {
                if (accumulate) o_sh[k] += dsh[k];
                else o_sh[k] = dsh[k];
            }
            if (!accumulate)
                for (int k = NB * 3; k < M * 3; ++k) o_sh[k] = 0.f;
        }
    };
    body();
    if (STAGED) {
        __syncthreads();
        stage_rows_out<STAGED ? ROWF : 4>(dL_dsh, row0, nrows, lds_rows, accumulate != 0);
    }
}

#define LAUNCH_DEG_S(KID, KERNEL, deg, grid, st, ...)                                          \
    switch (deg) {                                                                            \
        case 0: GDR_LAUNCH(KID, KERNEL<0>, grid, dim3(GDR_BLOCK), st, __VA_ARGS__); break;    \
        case 1: GDR_LAUNCH(KID, KERNEL<1>, grid, dim3(GDR_BLOCK), st, __VA_ARGS__); break;    \
        case 2: GDR_LAUNCH(KID, KERNEL<2>, grid, dim3(GDR_BLOCK), st, __VA_ARGS__); break;    \
        default: GDR_LAUNCH(KID, KERNEL<3>, grid, dim3(GDR_BLOCK), st, __VA_ARGS__); break;   \
    }

}  // namespace

hipError_t launch_surfel_preprocess_fwd(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* g,
                                        int32_t* radii, hipStream_t st) {
    if (in->N == 0) return hipSuccess;
    const dim3 grid(div_up(in->N, GDR_BLOCK));
    const int deg = in->shs ? s->sh_degree : 0;
    LAUNCH_DEG_S(GDR_K_PREPROCESS_FWD, surfel_preprocess_fwd_kernel, deg, grid, st, in->N, in->M, in->means3D,
                 in->scales, s->scale_modifier, in->rotations, in->opacities, in->shs, in->colors_precomp,
                 in->transMat_precomp, s->viewmatrix, s->projmatrix, s->campos, s->image_width, s->image_height,
                 radii, g->depths, (float4*)g->rec, (int4*)g->rect, g->tiles_touched, g->clamped, g->block_sums,
                 in->flags, g->block_offs, g->num_rendered);
    return hipGetLastError();
}

hipError_t launch_surfel_preprocess_bwd(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* g,
                                        const int32_t* radii, const gsr_grad_outputs* go, hipStream_t st) {
    if (in->N == 0) return hipSuccess;
    const dim3 grid(div_up(in->N, GDR_BLOCK));
    const int deg = in->shs ? s->sh_degree : 0;
#define GSR_K9(DEG_, ST_)                                                                                      \
    GDR_LAUNCH(GDR_K_PREPROCESS_BWD, (surfel_preprocess_bwd_kernel<DEG_, ST_>), grid, dim3(GDR_BLOCK), st, in->N,      \
               in->M, in->means3D, radii, in->shs, g->clamped, in->scales, in->rotations, s->scale_modifier,       \
               in->transMat_precomp ? 1 : 0, in->colors_precomp ? 1 : 0, s->viewmatrix, s->projmatrix, s->campos,  \
               s->image_width, s->image_height, (const float4*)go->scratch, (const float4*)g->rec,                 \
               (float4*)go->dL_dmeans2D, go->dL_dopacities, go->dL_dmeans3D, go->dL_dtransMat, go->dL_dshs,        \
               go->dL_dcolors, go->dL_dscales, (float4*)go->dL_drotations, in->flags, go->accumulate)
    const int nb = (deg + 1) * (deg + 1);
    const bool staged = in->shs && in->M == nb && (3 * nb) % 4 == 0;
    switch (deg) {
        case 0: GSR_K9(0, false); break;
        case 1: if (staged) GSR_K9(1, true); else GSR_K9(1, false); break;
        case 2: GSR_K9(2, false); break;
        default: if (staged) GSR_K9(3, true); else GSR_K9(3, false); break;
    }
#undef GSR_K9
    return hipGetLastError();
}

hipError_t launch_surfel_preprocess_fwd_views(int V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                              int32_t* const* radii, hipStream_t st) {
    if (in->N == 0) return hipSuccess;
    SFwdViewsArgs a;
    a.V = V;
    for (int v = 0; v < V; ++v) {
        SFwdView& f = a.v[v];
        f.view = s[v].viewmatrix; f.proj = s[v].projmatrix; f.campos = s[v].campos;
        f.radii = radii[v]; f.depths = geoms[v].depths; f.rec = (float4*)geoms[v].rec; f.rect = (int4*)geoms[v].rect;
        f.tiles = geoms[v].tiles_touched; f.clamped = geoms[v].clamped; f.block_sums = geoms[v].block_sums;
        f.block_offs = geoms[v].block_offs; f.num_rendered = geoms[v].num_rendered;
    }
    const dim3 grid(div_up(in->N, GDR_BLOCK));
    LAUNCH_DEG_S(GDR_K_PREPROCESS_FWD, surfel_preprocess_fwd_views_kernel, s[0].sh_degree, grid, st, in->N, in->M,
                 in->means3D, in->scales, s[0].scale_modifier, in->rotations, in->opacities, in->shs, s[0].image_width,
                 s[0].image_height, in->flags, a);
    return hipGetLastError();
}

// requires M == (deg+1)^2 for degrees 1 and 3 (the staged SH rows); the caller falls back to per-view launches otherwise
hipError_t launch_surfel_preprocess_bwd_views(int V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                              const int32_t* const* radii, float* const* grad_recs,
                                              const gsr_grad_outputs* go, hipStream_t st) {
    if (in->N == 0) return hipSuccess;
    SBwdViewsArgs a;
    a.V = V;
    for (int v = 0; v < V; ++v) {
        SBwdView& b = a.v[v];
        b.view = s[v].viewmatrix; b.proj = s[v].projmatrix; b.campos = s[v].campos;
        b.radii = radii[v]; b.clamped = geoms[v].clamped; b.grad_rec = (const float4*)grad_recs[v];
        b.rec = (const float4*)geoms[v].rec;
    }
    const dim3 grid(div_up(in->N, GDR_BLOCK));
    LAUNCH_DEG_S(GDR_K_PREPROCESS_BWD, surfel_preprocess_bwd_views_kernel, s[0].sh_degree, grid, st, in->N, in->M,
                 in->means3D, in->shs, in->scales, in->rotations, in->opacities, s[0].scale_modifier, s[0].image_width,
                 s[0].image_height, in->flags, go->accumulate, (float4*)go->dL_dmeans2D, go->dL_dopacities,
                 go->dL_dmeans3D, go->dL_dshs, go->dL_dscales, (float4*)go->dL_drotations, a);
    return hipGetLastError();
}

}  // namespace gdr
