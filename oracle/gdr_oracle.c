/*
 * oracle/gdr_oracle.c — CPU restatement of the differentiable Gaussian-splatting
 * rasterizer behind `diff_gaussian_rasterization.GaussianRasterizer`.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker.
 * The product path (generativedensification_amd/, include/gdr.h) never links,
 * imports or falls back to anything in oracle/.
 *
 * *** PARITY UNPINNED. ***  The arithmetic of this path lives in a third-party
 * submodule (Xiangyu1Sun/diff-gaussian-rasterization-GDM, /root/reference/.gitmodules:1-3)
 * whose directory is EMPTY in /root/reference; no source, golden vector or test of
 * it exists there.  This file restates the published algorithm of that lineage
 * (3DGS tile rasterizer + depth/alpha outputs + AbsGS abs-gradient channels) as
 * written down in SURVEY.md Appendix A, constrained by the reference's call sites:
 *   lightning/renderer.py:106-126   (12 settings fields)
 *   lightning/renderer.py:250-259   (4-tuple color, radii, depth, alpha)
 *   lightning/network.py:867-878    ((N,4) means2D grads, abs in [:,2:4])
 *   lightning/network.py:743-752    (depth = camera-space z, alpha = coverage)
 *   lightning/renderer.py:17-19     (C0, rgb = C0*sh0 + 0.5)
 *   lightning/utils.py:5-48         (row-vector matrix conventions)
 *
 * Precision: `real` = float (libgdr_oracle_f32.so: mirrors the HIP kernels op for op
 * in the preprocess stage so that every integer intermediate is bit-identical) or
 * double with -DGDR_REAL_DOUBLE (libgdr_oracle_f64.so: "truth" for finite
 * differences and error attribution).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: NO fused multiply-add,
 * so the f32 build rounds exactly like the HIP preprocess kernel, which is also
 * compiled with -ffp-contract=off).
 */
#include "oracle_common.h"

int oracle_real_bytes(void) { return (int)sizeof(real); }

const int32_t* g_oracle_tile_sel = NULL;
int g_oracle_tile_sel_n = 0;
/* tiles = NULL: every tile (default).  The array must stay alive until the selection is reset. */
void oracle_select_tiles(const int32_t* tiles, int n) {
    g_oracle_tile_sel = n > 0 ? tiles : NULL;
    g_oracle_tile_sel_n = n > 0 ? n : 0;
}

/* ------------------------------------------------------------------------- */
/* A.1 preprocess forward.  One Gaussian at a time (K1).                      */
/* rect = (minx, miny, maxx, maxy) in tile units.                            */
/* ------------------------------------------------------------------------- */
void oracle_preprocess_fwd(int N, int deg, int M, const real* means3D, const real* scales,
                           real scale_modifier, const real* rotations, const real* opacities,
                           const real* shs, const real* colors_precomp, const real* cov3D_precomp,
                           const real* view, const real* proj, const real* campos, int W, int H,
                           real tan_fovx, real tan_fovy, int prefiltered, int32_t* radii, real* xy,
                           real* depths, real* cov3D, real* rgb, real* conic_opacity,
                           uint32_t* tiles_touched, int32_t* rect, uint8_t* clamped, int nthreads) {
    (void)nthreads;
    (void)prefiltered; /* A.1-1: no x/y frustum test when prefiltered=False (always, renderer.py:122) */
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const real focal_x = (real)W / (RC(2) * tan_fovx), focal_y = (real)H / (RC(2) * tan_fovy);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        depths[i] = 0;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0;
        for (int k = 0; k < 4; ++k) rect[4 * i + k] = 0;
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        if (!cov3D_precomp) for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0;

        const real* p = means3D + 3 * i;
        real pv[3];
        xform4x3(p, view, pv);
        if (pv[2] <= RC(0.2)) continue; /* near cull, A.1-1 */

        real ph[4];
        xform4x4(p, proj, ph);
        real p_w = RC(1) / (ph[3] + RC(0.0000001));
        real pp[3] = {ph[0] * p_w, ph[1] * p_w, ph[2] * p_w};

        const real* c6;
        if (cov3D_precomp) {
            c6 = cov3D_precomp + 6 * i;
        } else {
            compute_cov3D(scales + 3 * i, scale_modifier, rotations + 4 * i, cov3D + 6 * i);
            c6 = cov3D + 6 * i;
        }

        /* A.1-4 EWA cov2D */
        real tx = pv[0], ty = pv[1], tz = pv[2];
        const real limx = RC(1.3) * tan_fovx, limy = RC(1.3) * tan_fovy;
        real txtz = tx / tz, tytz = ty / tz;
        tx = rmin(limx, rmax(-limx, txtz)) * tz;
        ty = rmin(limy, rmax(-limy, tytz)) * tz;
        real J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
        real J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
        /* A = J * Wm, Wm_rk = view[4k + r] */
        real A0[3], A1[3];
        for (int k = 0; k < 3; ++k) {
            A0[k] = J00 * view[4 * k + 0] + J02 * view[4 * k + 2];
            A1[k] = J11 * view[4 * k + 1] + J12 * view[4 * k + 2];
        }
        /* v0 = Sigma A0, v1 = Sigma A1 */
        real S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        real v0[3], v1[3];
        for (int r = 0; r < 3; ++r) {
            v0[r] = (S[3 * r] * A0[0] + S[3 * r + 1] * A0[1]) + S[3 * r + 2] * A0[2];
            v1[r] = (S[3 * r] * A1[0] + S[3 * r + 1] * A1[1]) + S[3 * r + 2] * A1[2];
        }
        real a = ((A0[0] * v0[0] + A0[1] * v0[1]) + A0[2] * v0[2]) + RC(0.3);
        real b = (A0[0] * v1[0] + A0[1] * v1[1]) + A0[2] * v1[2];
        real c = ((A1[0] * v1[0] + A1[1] * v1[1]) + A1[2] * v1[2]) + RC(0.3);

        real det = a * c - b * b;
        if (det == RC(0)) continue;
        real det_inv = RC(1) / det;
        real conic[3] = {c * det_inv, -b * det_inv, a * det_inv};

        real mid = RC(0.5) * (a + c);
        real disc = R_SQRT(rmax(RC(0.1), mid * mid - det));
        real lambda1 = mid + disc, lambda2 = mid - disc;
        real my_radius = R_CEIL(RC(3) * R_SQRT(rmax(lambda1, lambda2)));
        real px = ((pp[0] + RC(1)) * (real)W - RC(1)) * RC(0.5);
        real py = ((pp[1] + RC(1)) * (real)H - RC(1)) * RC(0.5);
        int rad = (int)my_radius;
        /* A.1-8 tile rect, C int truncation */
        int rminx = imin(gx, imax(0, (int)((px - (real)rad) / (real)BLOCK_X)));
        int rminy = imin(gy, imax(0, (int)((py - (real)rad) / (real)BLOCK_Y)));
        int rmaxx = imin(gx, imax(0, (int)((px + (real)rad + (real)(BLOCK_X - 1)) / (real)BLOCK_X)));
        int rmaxy = imin(gy, imax(0, (int)((py + (real)rad + (real)(BLOCK_Y - 1)) / (real)BLOCK_Y)));
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;

        if (colors_precomp) {
            for (int ch = 0; ch < 3; ++ch) rgb[3 * i + ch] = colors_precomp[3 * i + ch];
        } else {
            real dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
            real inv = RC(1) / R_SQRT((dx * dx + dy * dy) + dz * dz);
            dx *= inv; dy *= inv; dz *= inv;
            real bk[16];
            sh_basis(deg, dx, dy, dz, bk);
            int nb = (deg + 1) * (deg + 1);
            const real* sh = shs + (size_t)i * M * 3;
            for (int ch = 0; ch < 3; ++ch) {
                real acc = bk[0] * sh[ch];
                for (int k = 1; k < nb; ++k) acc = acc + bk[k] * sh[3 * k + ch];
                acc = acc + RC(0.5);
                clamped[3 * i + ch] = (acc < RC(0)) ? 1 : 0;
                rgb[3 * i + ch] = rmax(acc, RC(0));
            }
        }
        depths[i] = pv[2];
        radii[i] = rad;
        xy[2 * i] = px;
        xy[2 * i + 1] = py;
        conic_opacity[4 * i + 0] = conic[0];
        conic_opacity[4 * i + 1] = conic[1];
        conic_opacity[4 * i + 2] = conic[2];
        conic_opacity[4 * i + 3] = opacities[i];
        rect[4 * i + 0] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy;
        tiles_touched[i] = (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
    }
}

/* K10: visibility mask (upstream markVisible; never called by the reference). */
void oracle_mark_visible(int N, const real* means3D, const real* view, uint8_t* present) {
    for (int i = 0; i < N; ++i) {
        real pv[3];
        xform4x3(means3D + 3 * i, view, pv);
        present[i] = pv[2] > RC(0.2) ? 1 : 0;
    }
}

/* ------------------------------------------------------------------------- */
/* A.2 binning: scan, duplicate with keys, stable sort, tile ranges.          */
/* ------------------------------------------------------------------------- */
uint64_t oracle_scan(int N, const uint32_t* tiles_touched, uint32_t* offsets) {
    uint64_t acc = 0;
    for (int i = 0; i < N; ++i) {
        acc += tiles_touched[i];
        offsets[i] = (uint32_t)acc; /* inclusive */
    }
    return acc;
}

static uint32_t depth_bits(real d) {
    float f = (float)d;
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* keys_unsorted/vals_unsorted may be NULL. Sort is a stable LSD radix sort on the
 * full 64-bit key (a superset of the bits [0, 32+msb(tiles)) upstream sorts on; the
 * bits above are zero, so the order is identical).  Equal keys keep emission
 * order = ascending Gaussian index. */
void oracle_bin(int N, int W, int H, const int32_t* radii, const int32_t* rect, const real* depths,
                const uint32_t* offsets, uint64_t D, uint64_t* keys_unsorted, uint32_t* vals_unsorted,
                uint64_t* keys_sorted, uint32_t* vals_sorted, uint32_t* ranges) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    uint64_t* k0 = (uint64_t*)malloc(sizeof(uint64_t) * (D ? D : 1));
    uint32_t* v0 = (uint32_t*)malloc(sizeof(uint32_t) * (D ? D : 1));
    uint64_t* k1 = (uint64_t*)malloc(sizeof(uint64_t) * (D ? D : 1));
    uint32_t* v1 = (uint32_t*)malloc(sizeof(uint32_t) * (D ? D : 1));
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        uint64_t off = (i == 0) ? 0 : offsets[i - 1];
        uint32_t db = depth_bits(depths[i]);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
                uint64_t key = (uint64_t)(y * gx + x);
                key = (key << 32) | db;
                k0[off] = key;
                v0[off] = (uint32_t)i;
                ++off;
            }
    }
    if (keys_unsorted) memcpy(keys_unsorted, k0, sizeof(uint64_t) * D);
    if (vals_unsorted) memcpy(vals_unsorted, v0, sizeof(uint32_t) * D);
    for (int pass = 0; pass < 8; ++pass) {
        size_t hist[257];
        memset(hist, 0, sizeof(hist));
        int sh = 8 * pass;
        for (uint64_t e = 0; e < D; ++e) hist[((k0[e] >> sh) & 0xFF) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (uint64_t e = 0; e < D; ++e) {
            size_t dst = hist[(k0[e] >> sh) & 0xFF]++;
            k1[dst] = k0[e];
            v1[dst] = v0[e];
        }
        uint64_t* tk = k0; k0 = k1; k1 = tk;
        uint32_t* tv = v0; v0 = v1; v1 = tv;
    }
    memcpy(keys_sorted, k0, sizeof(uint64_t) * D);
    memcpy(vals_sorted, v0, sizeof(uint32_t) * D);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (uint64_t e = 0; e < D; ++e) {
        uint32_t t = (uint32_t)(k0[e] >> 32);
        if (e == 0) ranges[2 * t] = 0;
        else {
            uint32_t tp = (uint32_t)(k0[e - 1] >> 32);
            if (tp != t) { ranges[2 * tp + 1] = (uint32_t)e; ranges[2 * t] = (uint32_t)e; }
        }
        if (e == D - 1) ranges[2 * t + 1] = (uint32_t)D;
    }
    free(k0); free(v0); free(k1); free(v1);
}

/* ------------------------------------------------------------------------- */
/* A.3 render forward (K6).                                                   */
/* ------------------------------------------------------------------------- */
void oracle_render_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                       const real* xy, const real* colors, const real* conic_opacity,
                       const real* depths, const real* bg, real* out_color, real* out_depth,
                       real* out_alpha, uint32_t* n_contrib, real* final_T, int nthreads) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int it = 0; it < ORACLE_TILE_COUNT(gx * gy); ++it) {
        const int t = ORACLE_TILE_AT(it);
        if (t < 0 || t >= gx * gy) continue;
        int tx0 = (t % gx) * BLOCK_X, ty0 = (t / gx) * BLOCK_Y;
        uint32_t r0 = ranges[2 * t], r1 = ranges[2 * t + 1];
        for (int ly = 0; ly < BLOCK_Y; ++ly)
            for (int lx = 0; lx < BLOCK_X; ++lx) {
                int px = tx0 + lx, py = ty0 + ly;
                if (px >= W || py >= H) continue;
                real pxf = (real)px, pyf = (real)py;
                real T = 1, C[3] = {0, 0, 0}, Dp = 0, Wt = 0;
                uint32_t contributor = 0, last = 0;
                for (uint32_t e = r0; e < r1; ++e) {
                    contributor++;
                    uint32_t j = point_list[e];
                    real dx = xy[2 * j] - pxf, dy = xy[2 * j + 1] - pyf;
                    const real* co = conic_opacity + 4 * j;
                    real power = RC(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > RC(0)) continue;
                    real alpha = rmin(RC(0.99), co[3] * R_EXP(power));
                    if (alpha < RC(1.0 / 255.0)) continue;
                    real test_T = T * (RC(1) - alpha);
                    if (test_T < RC(0.0001)) break;
                    real w = alpha * T;
                    for (int ch = 0; ch < 3; ++ch) C[ch] += colors[3 * j + ch] * w;
                    Dp += depths[j] * w;
                    Wt += w;
                    T = test_T;
                    last = contributor;
                }
                size_t pix = (size_t)py * W + px;
                final_T[pix] = T;
                n_contrib[pix] = last;
                for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
                out_depth[pix] = Dp;
                out_alpha[pix] = Wt;
            }
    }
}

/* ------------------------------------------------------------------------- */
/* A.4 render backward (K7).  Per-Gaussian partial grads:                     */
/*   dL_dmean2D (N,4): .xy signed (NDC units: x 0.5W / 0.5H), .zw = sum |term| */
/*   dL_dconic  (N,4): d/d(conic.x, conic.y, conic.z) true partials, [3] unused */
/* ------------------------------------------------------------------------- */

void oracle_render_bwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                       const real* bg, const real* xy, const real* conic_opacity,
                       const real* colors, const real* depths, const real* final_T,
                       const uint32_t* n_contrib, const real* dL_dpix, const real* dL_ddepthpix,
                       const real* dL_dalphapix, real* dL_dmean2D, real* dL_dconic,
                       real* dL_dopacity, real* dL_dcolor, real* dL_ddepth, int nthreads) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int atomic = nthreads > 1;
    const real ddelx_dx = RC(0.5) * (real)W, ddely_dy = RC(0.5) * (real)H;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int it = 0; it < ORACLE_TILE_COUNT(gx * gy); ++it) {
        const int t = ORACLE_TILE_AT(it);
        if (t < 0 || t >= gx * gy) continue;
        int tx0 = (t % gx) * BLOCK_X, ty0 = (t / gx) * BLOCK_Y;
        uint32_t r0 = ranges[2 * t];
        /* Summation structure (round 3): the per-Gaussian sums are formed as the GPU forms them — the 16 pixel terms of a
         * 4x4 pixel block are summed first (K7: one DPP row), the block totals are then added to the Gaussian's
         * accumulators (K7: one atomic per row) — instead of one long sequential fp32 sum over all pixels of all tiles.
         * The reference's CUDA adds every pixel term with atomicAdd in no particular order, so any order restates it;
         * this one keeps the float32 restatement's rounding error at the level of the GPU's (a screen-filling Gaussian
         * is a 65 k-term sum).  blk: NS partial sums per list position of this tile. */
        enum { NS = 12 };
        const uint32_t Ltile = ranges[2 * t + 1] - r0;
        real* blk = (real*)calloc((size_t)(Ltile ? Ltile : 1) * NS, sizeof(real));
        for (int b4 = 0; b4 < (BLOCK_X / 4) * (BLOCK_Y / 4); ++b4) {
            uint32_t kmax = 0;
            for (int q4 = 0; q4 < 16; ++q4) {
                const int lx = (b4 % (BLOCK_X / 4)) * 4 + (q4 & 3), ly = (b4 / (BLOCK_X / 4)) * 4 + (q4 >> 2);
                int px = tx0 + lx, py = ty0 + ly;
                if (px >= W || py >= H) continue;
                size_t pix = (size_t)py * W + px;
                real pxf = (real)px, pyf = (real)py;
                const real T_final = final_T[pix];
                real T = T_final;
                uint32_t last = n_contrib[pix];
                if (last > kmax) kmax = last;
                real gC[3] = {dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[(size_t)2 * H * W + pix]};
                real gD = dL_ddepthpix[pix], gA = dL_dalphapix[pix];
                real accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
                real accum_depth = 0, last_depth = 0, accum_alpha = 0, last_alpha = 0;
                real bg_dot = (bg[0] * gC[0] + bg[1] * gC[1]) + bg[2] * gC[2];
                for (uint32_t k = last; k-- > 0;) {
                    uint32_t j = point_list[r0 + k];
                    real dx = xy[2 * j] - pxf, dy = xy[2 * j + 1] - pyf;
                    const real* co = conic_opacity + 4 * j;
                    real power = RC(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > RC(0)) continue;
                    real G = R_EXP(power);
                    real alpha = rmin(RC(0.99), co[3] * G);
                    if (alpha < RC(1.0 / 255.0)) continue;
                    T = T / (RC(1) - alpha);
                    real w = alpha * T;
                    real dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ++ch) {
                        real c = colors[3 * j + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (RC(1) - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * gC[ch];
                        blk[NS * k + 8 + ch] += w * gC[ch];
                    }
                    real dep = depths[j];
                    accum_depth = last_alpha * last_depth + (RC(1) - last_alpha) * accum_depth;
                    last_depth = dep;
                    dL_dalpha += (dep - accum_depth) * gD;
                    blk[NS * k + 7] += w * gD;
                    accum_alpha = last_alpha + (RC(1) - last_alpha) * accum_alpha;
                    dL_dalpha += (RC(1) - accum_alpha) * gA;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (RC(1) - alpha)) * bg_dot;
                    /* straight-through the min(0.99, .) as upstream 3DGS does */
                    real dL_dG = co[3] * dL_dalpha;
                    real gdx = G * dx, gdy = G * dy;
                    real dG_ddelx = -gdx * co[0] - gdy * co[1];
                    real dG_ddely = -gdy * co[2] - gdx * co[1];
                    real mx = dL_dG * dG_ddelx * ddelx_dx, my = dL_dG * dG_ddely * ddely_dy;
                    blk[NS * k + 0] += mx;
                    blk[NS * k + 1] += my;
                    blk[NS * k + 2] += R_FABS(mx);
                    blk[NS * k + 3] += R_FABS(my);
                    blk[NS * k + 4] += RC(-0.5) * gdx * dx * dL_dG;
                    blk[NS * k + 5] += -gdx * dy * dL_dG;
                    blk[NS * k + 6] += RC(-0.5) * gdy * dy * dL_dG;
                    blk[NS * k + 11] += G * dL_dalpha;
                }
            }
            for (uint32_t k = 0; k < kmax; ++k) {   /* block totals -> the Gaussian's accumulators */
                real* a = blk + (size_t)NS * k;
                const uint32_t j = point_list[r0 + k];
                real* const dst[NS] = {dL_dmean2D + 4 * j, dL_dmean2D + 4 * j + 1, dL_dmean2D + 4 * j + 2, dL_dmean2D + 4 * j + 3,
                                       dL_dconic + 4 * j, dL_dconic + 4 * j + 1, dL_dconic + 4 * j + 2, dL_ddepth + j,
                                       dL_dcolor + 3 * j, dL_dcolor + 3 * j + 1, dL_dcolor + 3 * j + 2, dL_dopacity + j};
                for (int s_ = 0; s_ < NS; ++s_)
                    if (a[s_] != RC(0)) { accum(dst[s_], a[s_], atomic); a[s_] = RC(0); }
            }
        }
        free(blk);
    }
}

/* ------------------------------------------------------------------------- */
/* A.5 preprocess backward (K8 + K9).                                         */
/* ------------------------------------------------------------------------- */
void oracle_preprocess_bwd(int N, int deg, int M, const real* means3D, const int32_t* radii,
                           const real* shs, const uint8_t* clamped, const real* scales,
                           const real* rotations, real scale_modifier, const real* cov3D,
                           int cov3D_is_precomp, int colors_is_precomp, const real* view, const real* proj,
                           const real* campos, int W, int H, real tan_fovx, real tan_fovy,
                           const real* dL_dmean2D, const real* dL_dconic, const real* dL_dcolor,
                           const real* dL_ddepth, real* dL_dmeans3D, real* dL_dcov3D, real* dL_dsh,
                           real* dL_dscale, real* dL_drot, int nthreads) {
    const real focal_x = (real)W / (RC(2) * tan_fovx), focal_y = (real)H / (RC(2) * tan_fovy);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        const real* p = means3D + 3 * i;
        const real* c6 = cov3D + 6 * i;
        real dmean[3] = {0, 0, 0};

        /* (i) conic -> cov2D -> cov3D, mean (K8) */
        real pv[3];
        xform4x3(p, view, pv);
        real tx = pv[0], ty = pv[1], tz = pv[2];
        const real limx = RC(1.3) * tan_fovx, limy = RC(1.3) * tan_fovy;
        real txtz = tx / tz, tytz = ty / tz;
        tx = rmin(limx, rmax(-limx, txtz)) * tz;
        ty = rmin(limy, rmax(-limy, tytz)) * tz;
        real x_grad_mul = (txtz < -limx || txtz > limx) ? RC(0) : RC(1);
        real y_grad_mul = (tytz < -limy || tytz > limy) ? RC(0) : RC(1);
        real J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
        real J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
        real A0[3], A1[3];
        for (int k = 0; k < 3; ++k) {
            A0[k] = J00 * view[4 * k + 0] + J02 * view[4 * k + 2];
            A1[k] = J11 * view[4 * k + 1] + J12 * view[4 * k + 2];
        }
        real S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        real v0[3], v1[3];
        for (int r = 0; r < 3; ++r) {
            v0[r] = (S[3 * r] * A0[0] + S[3 * r + 1] * A0[1]) + S[3 * r + 2] * A0[2];
            v1[r] = (S[3 * r] * A1[0] + S[3 * r + 1] * A1[1]) + S[3 * r + 2] * A1[2];
        }
        real a = ((A0[0] * v0[0] + A0[1] * v0[1]) + A0[2] * v0[2]) + RC(0.3);
        real b = (A0[0] * v1[0] + A0[1] * v1[1]) + A0[2] * v1[2];
        real c = ((A1[0] * v1[0] + A1[1] * v1[1]) + A1[2] * v1[2]) + RC(0.3);
        real det = a * c - b * b;
        real gx_ = dL_dconic[4 * i + 0], gy_ = dL_dconic[4 * i + 1], gz_ = dL_dconic[4 * i + 2];
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real dcov[6] = {0, 0, 0, 0, 0, 0};
        if (det * det != RC(0)) {
            real d2inv = RC(1) / (det * det);
            dL_da = d2inv * (-c * c * gx_ + b * c * gy_ - b * b * gz_);
            dL_db = d2inv * (RC(2) * b * c * gx_ - (a * c + b * b) * gy_ + RC(2) * a * b * gz_);
            dL_dc = d2inv * (-b * b * gx_ + a * b * gy_ - a * a * gz_);
            /* cov2D = A Sigma A^T */
            dcov[0] = A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
            dcov[3] = A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
            dcov[5] = A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
            dcov[1] = RC(2) * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + RC(2) * A1[0] * A1[1] * dL_dc;
            dcov[2] = RC(2) * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + RC(2) * A1[0] * A1[2] * dL_dc;
            dcov[4] = RC(2) * A0[1] * A0[2] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + RC(2) * A1[1] * A1[2] * dL_dc;
        }
        /* dL/dA rows */
        real dA0[3], dA1[3];
        for (int k = 0; k < 3; ++k) {
            dA0[k] = RC(2) * dL_da * v0[k] + dL_db * v1[k];
            dA1[k] = RC(2) * dL_dc * v1[k] + dL_db * v0[k];
        }
        real dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
        for (int k = 0; k < 3; ++k) {
            dJ00 += dA0[k] * view[4 * k + 0];
            dJ02 += dA0[k] * view[4 * k + 2];
            dJ11 += dA1[k] * view[4 * k + 1];
            dJ12 += dA1[k] * view[4 * k + 2];
        }
        real tz1 = RC(1) / tz, tz2 = tz1 * tz1, tz3 = tz2 * tz1;
        real dtx = x_grad_mul * (-focal_x * tz2 * dJ02);
        real dty = y_grad_mul * (-focal_y * tz2 * dJ12);
        real dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (RC(2) * focal_x * tx) * tz3 * dJ02 +
                   (RC(2) * focal_y * ty) * tz3 * dJ12;
        /* t = Wm p + trans  => dL/dp_k = sum_r Wm_rk dL/dt_r, Wm_rk = view[4k+r] */
        for (int k = 0; k < 3; ++k)
            dmean[k] += view[4 * k + 0] * dtx + view[4 * k + 1] * dty + view[4 * k + 2] * dtz;

        /* (ii) mean2D (NDC) -> mean3D through projmatrix and 1/(w+1e-7) */
        real mh[4];
        xform4x4(p, proj, mh);
        real m_w = RC(1) / (mh[3] + RC(0.0000001));
        real mul1 = mh[0] * m_w * m_w, mul2 = mh[1] * m_w * m_w;
        real g2x = dL_dmean2D[4 * i + 0], g2y = dL_dmean2D[4 * i + 1];
        for (int k = 0; k < 3; ++k)
            dmean[k] += (proj[4 * k + 0] * m_w - proj[4 * k + 3] * mul1) * g2x +
                        (proj[4 * k + 1] * m_w - proj[4 * k + 3] * mul2) * g2y;

        /* (iii) depth_i = p_view.z -> mean3D (risk R1: present) */
        real gdep = dL_ddepth[i];
        for (int k = 0; k < 3; ++k) dmean[k] += view[4 * k + 2] * gdep;

        /* (iv) SH backward */
        if (!colors_is_precomp) {
            real dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
            real len2 = (dx * dx + dy * dy) + dz * dz;
            real inv = RC(1) / R_SQRT(len2);
            real ux = dx * inv, uy = dy * inv, uz = dz * inv;
            real bk[16], bx[16], by[16], bz[16];
            sh_basis(deg, ux, uy, uz, bk);
            sh_basis_grad(deg, ux, uy, uz, bx, by, bz);
            int nb = (deg + 1) * (deg + 1);
            const real* sh = shs + (size_t)i * M * 3;
            real* dsh = dL_dsh + (size_t)i * M * 3;
            real ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
                real g = clamped[3 * i + ch] ? RC(0) : dL_dcolor[3 * i + ch];
                for (int k = 0; k < nb; ++k) {
                    dsh[3 * k + ch] = bk[k] * g;
                    ddir[0] += bx[k] * sh[3 * k + ch] * g;
                    ddir[1] += by[k] * sh[3 * k + ch] * g;
                    ddir[2] += bz[k] * sh[3 * k + ch] * g;
                }
            }
            real dot = ux * ddir[0] + uy * ddir[1] + uz * ddir[2];
            dmean[0] += (ddir[0] - ux * dot) * inv;
            dmean[1] += (ddir[1] - uy * dot) * inv;
            dmean[2] += (ddir[2] - uz * dot) * inv;
        }
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];

        /* (v) cov3D -> scale, quaternion */
        if (cov3D_is_precomp) {
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dcov[k];
        } else {
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dcov[k];
            real R[9];
            const real* q = rotations + 4 * i;
            quat_to_R(q, R);
            real s[3] = {scale_modifier * scales[3 * i], scale_modifier * scales[3 * i + 1],
                         scale_modifier * scales[3 * i + 2]};
            /* full symmetric dL/dSigma */
            real Gs[9] = {dcov[0], RC(0.5) * dcov[1], RC(0.5) * dcov[2],
                          RC(0.5) * dcov[1], dcov[3], RC(0.5) * dcov[4],
                          RC(0.5) * dcov[2], RC(0.5) * dcov[4], dcov[5]};
            /* Sigma = Mm Mm^T, Mm = R diag(s) ; dL/dMm = 2 Gs Mm */
            real dM[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) {
                    real acc = 0;
                    for (int l = 0; l < 3; ++l) acc += Gs[3 * r + l] * (R[3 * l + k] * s[k]);
                    dM[3 * r + k] = RC(2) * acc;
                }
            real dR[9];
            for (int k = 0; k < 3; ++k) {
                real ds = 0;
                for (int r = 0; r < 3; ++r) {
                    ds += R[3 * r + k] * dM[3 * r + k];
                    dR[3 * r + k] = s[k] * dM[3 * r + k];
                }
                dL_dscale[3 * i + k] = scale_modifier * ds;
            }
            real qr = q[0], qx = q[1], qy = q[2], qz = q[3];
#define G_(r, c) dR[3 * (r) + (c)]
            dL_drot[4 * i + 0] = RC(2) * (-qz * G_(0, 1) + qy * G_(0, 2) + qz * G_(1, 0) - qx * G_(1, 2) - qy * G_(2, 0) + qx * G_(2, 1));
            dL_drot[4 * i + 1] = RC(2) * (qy * G_(0, 1) + qz * G_(0, 2) + qy * G_(1, 0) - RC(2) * qx * G_(1, 1) - qr * G_(1, 2) + qz * G_(2, 0) + qr * G_(2, 1) - RC(2) * qx * G_(2, 2));
            dL_drot[4 * i + 2] = RC(2) * (-RC(2) * qy * G_(0, 0) + qx * G_(0, 1) + qr * G_(0, 2) + qx * G_(1, 0) + qz * G_(1, 2) - qr * G_(2, 0) + qz * G_(2, 1) - RC(2) * qy * G_(2, 2));
            dL_drot[4 * i + 3] = RC(2) * (-RC(2) * qz * G_(0, 0) - qr * G_(0, 1) + qx * G_(0, 2) + qr * G_(1, 0) - RC(2) * qz * G_(1, 1) + qy * G_(1, 2) + qx * G_(2, 0) + qy * G_(2, 1));
#undef G_
        }
    }
}
