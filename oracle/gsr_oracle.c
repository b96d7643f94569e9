/*
 * oracle/gsr_oracle.c — CPU restatement of the differentiable 2D-Gaussian (surfel) rasterizer behind
 * `diff_surfel_rasterization.GaussianRasterizer` (BASELINE config 5).
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Same rule as gdr_oracle.c: only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker.
 *
 * *** PARITY UNPINNED. ***  `diff_surfel_rasterization` is imported by /root/reference/lightning/renderer_2dgs.py:7-10
 * but is in neither .gitmodules nor the tree: no source, test or golden vector of it
 * exists in /root/reference.  This file restates the published 2DGS algorithm (Huang et al. 2024, the
 * hbb1/diff-surfel-rasterization lineage the adaptor's 3-tuple `image, radii, allmap[7,H,W]` belongs to) from
 * its paper and public description, constrained by the reference's call sites:
 *   lightning/renderer_2dgs.py:111-126   12 settings fields (same record as the 3DGS path)
 *   lightning/renderer_2dgs.py:224-234   3-tuple (image (3,H,W), radii (N,), allmap (7,H,W))
 *   lightning/renderer_2dgs.py:241-257   allmap channels: 0 expected depth (sum w z), 1 alpha, 2-4 normal
 *                                        (VIEW space: the adaptor rotates it by world_view[:3,:3].T), 5 median
 *                                        depth, 6 depth distortion
 *   lightning/renderer_2dgs.py:92-96     scales are (N,2)
 *   lightning/renderer_2dgs.py:207-208   means2D carrier is (N,4)
 *
 * Algorithm (per surfel): splat-to-pixel homography T = [s_u t_u | s_v t_v | p]^T · projmatrix · ndc2pix
 * (rows Tu, Tv, Tw = x_h, y_h, w coefficients of (u,v,1)); bounding box of the 3-sigma ellipse from T; per
 * pixel the ray–splat intersection (u,v) = cross(x·Tw − Tu, y·Tw − Tv) dehomogenised, G = exp(−½ min(u²+v²,
 * 2|pix − centre|²)) (object-space Gaussian with a screen-space low-pass floor), alpha compositing front to
 * back with the same skip rules as the 3DGS path, plus depth / normal / median-depth / distortion outputs.
 * Constants: cutoff 3 sigma, low-pass FilterSize = 0.707106, FilterInvSquare = 2, near_n = 0.2, far_n = 100.
 * Build-defined choices where nothing pins the behaviour (documented in DESIGN.md):
 *   * dual-visible surfels: the normal is flipped to face the camera;
 *   * means2D gradient (N,4): cols 0-1 = dL/dT[0][2], dL/dT[1][2] scaled by depth·0.5·W (resp. H) — the
 *     densification signal the lineage returns — cols 2-3 the same with per-pixel |.| accumulation (AbsGS
 *     analogue, so that renderer_2dgs.py's (N,4) carrier is filled the way renderer.py's is).
 */
#include "oracle_common.h"

#define NEAR_N RC(0.2)
#define FAR_N RC(100.0)
#define FILTER_SIZE RC(0.707106)
#define FILTER_INV_SQUARE RC(2.0)

/* clip = [v, w] @ proj, fixed association ((m0 x + m4 y) + m8 z) (+ m12) */
static inline void clip_of(const real* v, int w, const real* m, real* c) {
    for (int j = 0; j < 4; ++j) {
        real a = (m[j] * v[0] + m[4 + j] * v[1]) + m[8 + j] * v[2];
        c[j] = w ? a + m[12 + j] : a;
    }
}

/* T rows (Tu, Tv, Tw) of one surfel; nrm = third rotation axis in WORLD space */
static inline void surfel_transmat(const real* p, const real* scale2, real mod, const real* q, const real* proj,
                                   int W, int H, real* T9, real* nrm, real* Rout) {
    real R[9];
    quat_to_R(q, R);
    const real s0 = mod * scale2[0], s1 = mod * scale2[1];
    real L0[3] = {s0 * R[0], s0 * R[3], s0 * R[6]}, L1[3] = {s1 * R[1], s1 * R[4], s1 * R[7]};
    nrm[0] = R[2]; nrm[1] = R[5]; nrm[2] = R[8];
    const real hw = (real)W / RC(2), hh = (real)H / RC(2), cw = (real)(W - 1) / RC(2), ch = (real)(H - 1) / RC(2);
    const real* vs[3] = {L0, L1, p};
    for (int i = 0; i < 3; ++i) {
        real c[4];
        clip_of(vs[i], i == 2, proj, c);
        T9[0 + i] = c[0] * hw + c[3] * cw;
        T9[3 + i] = c[1] * hh + c[3] * ch;
        T9[6 + i] = c[3];
    }
    if (Rout) memcpy(Rout, R, sizeof(R));
}

/* bounding box of the `cutoff`-sigma ellipse: centre, half extent; 0 if degenerate */
static inline int surfel_aabb(const real* T9, real cutoff, real* centre, real* extent) {
    const real* Tu = T9; const real* Tv = T9 + 3; const real* Tw = T9 + 6;
    const real t[3] = {cutoff * cutoff, cutoff * cutoff, RC(-1)};
    const real d = (t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1]) + t[2] * Tw[2] * Tw[2];
    if (d == RC(0)) return 0;
    const real inv_d = RC(1) / d;
    const real f[3] = {t[0] * inv_d, t[1] * inv_d, t[2] * inv_d};
    const real px = (f[0] * Tu[0] * Tw[0] + f[1] * Tu[1] * Tw[1]) + f[2] * Tu[2] * Tw[2];
    const real py = (f[0] * Tv[0] * Tw[0] + f[1] * Tv[1] * Tw[1]) + f[2] * Tv[2] * Tw[2];
    const real hx0 = px * px - ((f[0] * Tu[0] * Tu[0] + f[1] * Tu[1] * Tu[1]) + f[2] * Tu[2] * Tu[2]);
    const real hy0 = py * py - ((f[0] * Tv[0] * Tv[0] + f[1] * Tv[1] * Tv[1]) + f[2] * Tv[2] * Tv[2]);
    centre[0] = px; centre[1] = py;
    extent[0] = R_SQRT(rmax(RC(1e-4), hx0));
    extent[1] = R_SQRT(rmax(RC(1e-4), hy0));
    return 1;
}

/* ------------------------------------------------------------------------- */
/* preprocess forward (one surfel at a time)                                  */
/* ------------------------------------------------------------------------- */
void oracle_surfel_preprocess_fwd(int N, int deg, int M, const real* means3D, const real* scales,
                                  real scale_modifier, const real* rotations, const real* opacities,
                                  const real* shs, const real* colors_precomp, const real* transMat_precomp,
                                  const real* view, const real* proj, const real* campos, int W, int H,
                                  int32_t* radii, real* xy, real* depths, real* transMats, real* rgb,
                                  real* normal_opacity, uint32_t* tiles_touched, int32_t* rect, uint8_t* clamped,
                                  int nthreads) {
    (void)nthreads;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        depths[i] = 0;
        for (int k = 0; k < 4; ++k) { normal_opacity[4 * i + k] = 0; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 9; ++k) transMats[9 * i + k] = 0;

        const real* p = means3D + 3 * i;
        real pv[3];
        xform4x3(p, view, pv);
        if (pv[2] <= RC(0.2)) continue;

        real T9[9], nw[3], nv[3];
        if (transMat_precomp) {
            memcpy(T9, transMat_precomp + 9 * i, sizeof(T9));
            nv[0] = 0; nv[1] = 0; nv[2] = 1;
        } else {
            surfel_transmat(p, scales + 2 * i, scale_modifier, rotations + 4 * i, proj, W, H, T9, nw, NULL);
            nv[0] = (view[0] * nw[0] + view[4] * nw[1]) + view[8] * nw[2];
            nv[1] = (view[1] * nw[0] + view[5] * nw[1]) + view[9] * nw[2];
            nv[2] = (view[2] * nw[0] + view[6] * nw[1]) + view[10] * nw[2];
        }
        const real cosv = -((pv[0] * nv[0] + pv[1] * nv[1]) + pv[2] * nv[2]);
        if (cosv == RC(0)) continue;
        const real mult = cosv > RC(0) ? RC(1) : RC(-1);

        real centre[2], extent[2];
        if (!surfel_aabb(T9, RC(3), centre, extent)) continue;
        const real my_radius = R_CEIL(rmax(rmax(extent[0], extent[1]), RC(3) * FILTER_SIZE));
        const int rad = (int)my_radius;
        const real px = centre[0], py = centre[1];
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
        for (int k = 0; k < 9; ++k) transMats[9 * i + k] = T9[k];
        normal_opacity[4 * i + 0] = mult * nv[0];
        normal_opacity[4 * i + 1] = mult * nv[1];
        normal_opacity[4 * i + 2] = mult * nv[2];
        normal_opacity[4 * i + 3] = opacities[i];
        rect[4 * i + 0] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy;
        tiles_touched[i] = (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
    }
}

/* one pixel–surfel evaluation shared by forward and backward */
typedef struct {
    real k[3], l[3], pz, sx, sy, rho3d, rho2d, dx, dy, depth, G, alpha;
    int use3d;
} SurfelHit;

static inline int surfel_eval(const real* T9, const real* xy, real opac, real pxf, real pyf, SurfelHit* h) {
    const real* Tu = T9; const real* Tv = T9 + 3; const real* Tw = T9 + 6;
    for (int c = 0; c < 3; ++c) { h->k[c] = pxf * Tw[c] - Tu[c]; h->l[c] = pyf * Tw[c] - Tv[c]; }
    const real cx = h->k[1] * h->l[2] - h->k[2] * h->l[1];
    const real cy = h->k[2] * h->l[0] - h->k[0] * h->l[2];
    const real cz = h->k[0] * h->l[1] - h->k[1] * h->l[0];
    if (cz == RC(0)) return 0;
    h->pz = cz;
    h->sx = cx / cz; h->sy = cy / cz;
    h->rho3d = h->sx * h->sx + h->sy * h->sy;
    h->dx = xy[0] - pxf; h->dy = xy[1] - pyf;
    h->rho2d = FILTER_INV_SQUARE * (h->dx * h->dx + h->dy * h->dy);
    h->use3d = h->rho3d <= h->rho2d;
    const real rho = h->use3d ? h->rho3d : h->rho2d;
    h->depth = h->use3d ? (h->sx * Tw[0] + h->sy * Tw[1]) + Tw[2] : Tw[2];
    if (h->depth < NEAR_N) return 0;
    const real power = RC(-0.5) * rho;
    if (power > RC(0)) return 0;
    h->G = R_EXP(power);
    h->alpha = rmin(RC(0.99), opac * h->G);
    if (h->alpha < RC(1.0 / 255.0)) return 0;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* render forward.  others = allmap (7,H,W); n_contrib (2,H,W): last, median (1-based, 0 = none);  */
/* final_T (3,H,W): T, M1 = sum w m, M2 = sum w m^2 (m = normalised depth), kept for backward.      */
/* ------------------------------------------------------------------------- */
void oracle_surfel_render_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const real* xy,
                              const real* colors, const real* transMats, const real* normal_opacity,
                              const real* bg, real* out_color, real* out_others, uint32_t* n_contrib,
                              real* final_T, int nthreads) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const size_t HW = (size_t)H * W;
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
                real T = 1, C[3] = {0, 0, 0}, Nn[3] = {0, 0, 0}, D = 0, M1 = 0, M2 = 0, dist = 0, med = 0;
                uint32_t contributor = 0, last = 0, med_c = 0;
                for (uint32_t e = r0; e < r1; ++e) {
                    contributor++;
                    uint32_t j = point_list[e];
                    const real* no = normal_opacity + 4 * j;
                    SurfelHit h;
                    if (!surfel_eval(transMats + 9 * j, xy + 2 * j, no[3], pxf, pyf, &h)) continue;
                    real test_T = T * (RC(1) - h.alpha);
                    if (test_T < RC(0.0001)) break;
                    real w = h.alpha * T;
                    real A = RC(1) - T;
                    real m = FAR_N / (FAR_N - NEAR_N) * (RC(1) - NEAR_N / h.depth);
                    dist += (m * m * A + M2 - RC(2) * m * M1) * w;
                    D += h.depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                    if (T > RC(0.5)) { med = h.depth; med_c = contributor; }
                    for (int ch = 0; ch < 3; ++ch) Nn[ch] += no[ch] * w;
                    for (int ch = 0; ch < 3; ++ch) C[ch] += colors[3 * j + ch] * w;
                    T = test_T;
                    last = contributor;
                }
                size_t pix = (size_t)py * W + px;
                final_T[pix] = T; final_T[HW + pix] = M1; final_T[2 * HW + pix] = M2;
                n_contrib[pix] = last; n_contrib[HW + pix] = med_c;
                for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * HW + pix] = C[ch] + T * bg[ch];
                out_others[0 * HW + pix] = D;
                out_others[1 * HW + pix] = RC(1) - T;
                for (int ch = 0; ch < 3; ++ch) out_others[(2 + ch) * HW + pix] = Nn[ch];
                out_others[5 * HW + pix] = med;
                out_others[6 * HW + pix] = dist;
            }
    }
}

/* ------------------------------------------------------------------------- */
/* render backward.  Per-surfel partials: dL_dtransMat (N,9), dL_dmean2D (N,4): .xy = low-pass-branch   */
/* d/d centre (pixels), .zw = sum over pixels of |dL/dTu.z|, |dL/dTv.z|; dL_dnormal (N,3) view space.   */
/* ------------------------------------------------------------------------- */
void oracle_surfel_render_bwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const real* bg,
                              const real* xy, const real* normal_opacity, const real* transMats,
                              const real* colors, const real* final_T, const uint32_t* n_contrib,
                              const real* dL_dpix, const real* dL_dothers, real* dL_dtransMat, real* dL_dmean2D,
                              real* dL_dnormal, real* dL_dopacity, real* dL_dcolor, int nthreads) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const size_t HW = (size_t)H * W;
    const int atomic = nthreads > 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int it = 0; it < ORACLE_TILE_COUNT(gx * gy); ++it) {
        const int t = ORACLE_TILE_AT(it);
        if (t < 0 || t >= gx * gy) continue;
        int tx0 = (t % gx) * BLOCK_X, ty0 = (t / gx) * BLOCK_Y;
        uint32_t r0 = ranges[2 * t];
        /* Summation structure (round 3, as oracle_render_bwd in gdr_oracle.c): the 16 pixel terms of a 4x4 pixel block
         * are summed first (K7s: one DPP row), the block totals are then added to the surfel's accumulators (K7s: one
         * atomic per row) — any order restates the reference's per-pixel atomicAdd. */
        enum { NS = 20 };   /* transMat 0..8, mean2D 9..12, normal 13..15, colour 16..18, opacity 19 */
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
                const real T_final = final_T[pix], final_D = final_T[HW + pix], final_D2 = final_T[2 * HW + pix];
                const real final_A = RC(1) - T_final;
                real T = T_final;
                uint32_t last = n_contrib[pix], med_c = n_contrib[HW + pix];
                if (last > kmax) kmax = last;
                real gC[3] = {dL_dpix[pix], dL_dpix[HW + pix], dL_dpix[2 * HW + pix]};
                const real gDepth = dL_dothers[0 * HW + pix], gAlpha = dL_dothers[1 * HW + pix];
                const real gN[3] = {dL_dothers[2 * HW + pix], dL_dothers[3 * HW + pix], dL_dothers[4 * HW + pix]};
                const real gMed = dL_dothers[5 * HW + pix], gReg = dL_dothers[6 * HW + pix];
                real accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
                real accum_depth = 0, last_depth = 0, accum_alpha = 0, last_alpha = 0;
                real accum_nrm[3] = {0, 0, 0}, last_nrm[3] = {0, 0, 0};
                real last_dL_dT = 0;
                real bg_dot = (bg[0] * gC[0] + bg[1] * gC[1]) + bg[2] * gC[2];
                for (uint32_t k = last; k-- > 0;) {
                    uint32_t j = point_list[r0 + k];
                    const real* no = normal_opacity + 4 * j;
                    const real* T9 = transMats + 9 * j;
                    SurfelHit h;
                    if (!surfel_eval(T9, xy + 2 * j, no[3], pxf, pyf, &h)) continue;
                    const real alpha = h.alpha, G = h.G;
                    T = T / (RC(1) - alpha);
                    const real w = alpha * T;
                    real dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ++ch) {
                        real c = colors[3 * j + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (RC(1) - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * gC[ch];
                        blk[NS * k + 16 + ch] += w * gC[ch];
                    }
                    real dL_dz = 0;
                    const real c_d = h.depth;
                    const real m_d = FAR_N / (FAR_N - NEAR_N) * (RC(1) - NEAR_N / c_d);
                    const real dmd_dd = (FAR_N * NEAR_N) / ((FAR_N - NEAR_N) * c_d * c_d);
                    if (k + 1 == med_c) dL_dz += gMed;
                    const real dL_dweight = (final_D2 + m_d * m_d * final_A - RC(2) * m_d * final_D) * gReg;
                    dL_dalpha += dL_dweight - last_dL_dT;
                    last_dL_dT = dL_dweight * alpha + (RC(1) - alpha) * last_dL_dT;
                    const real dL_dmd = RC(2) * w * (m_d * final_A - final_D) * gReg;
                    dL_dz += dL_dmd * dmd_dd;
                    accum_depth = last_alpha * last_depth + (RC(1) - last_alpha) * accum_depth;
                    last_depth = c_d;
                    dL_dalpha += (c_d - accum_depth) * gDepth;
                    accum_alpha = last_alpha + (RC(1) - last_alpha) * accum_alpha;
                    dL_dalpha += (RC(1) - accum_alpha) * gAlpha;
                    for (int ch = 0; ch < 3; ++ch) {
                        accum_nrm[ch] = last_alpha * last_nrm[ch] + (RC(1) - last_alpha) * accum_nrm[ch];
                        last_nrm[ch] = no[ch];
                        dL_dalpha += (no[ch] - accum_nrm[ch]) * gN[ch];
                        blk[NS * k + 13 + ch] += w * gN[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (RC(1) - alpha)) * bg_dot;
                    const real dL_dG = no[3] * dL_dalpha; /* straight-through the min(0.99, .) */
                    dL_dz += w * gDepth;
                    if (h.use3d) {
                        const real* Tw = T9 + 6;
                        const real dsx = dL_dG * -G * h.sx + dL_dz * Tw[0];
                        const real dsy = dL_dG * -G * h.sy + dL_dz * Tw[1];
                        const real dsx_pz = dsx / h.pz, dsy_pz = dsy / h.pz;
                        const real dp[3] = {dsx_pz, dsy_pz, -(dsx_pz * h.sx + dsy_pz * h.sy)};
                        /* p = cross(k, l): dL/dk = cross(l, dp), dL/dl = cross(dp, k) */
                        const real* kk = h.k; const real* ll = h.l;
                        const real dk[3] = {ll[1] * dp[2] - ll[2] * dp[1], ll[2] * dp[0] - ll[0] * dp[2], ll[0] * dp[1] - ll[1] * dp[0]};
                        const real dl[3] = {dp[1] * kk[2] - dp[2] * kk[1], dp[2] * kk[0] - dp[0] * kk[2], dp[0] * kk[1] - dp[1] * kk[0]};
                        const real dz_dTw[3] = {h.sx, h.sy, RC(1)};
                        for (int c = 0; c < 3; ++c) {
                            blk[NS * k + 0 + c] += -dk[c];
                            blk[NS * k + 3 + c] += -dl[c];
                            blk[NS * k + 6 + c] += pxf * dk[c] + pyf * dl[c] + dL_dz * dz_dTw[c];
                        }
                        blk[NS * k + 11] += R_FABS(dk[2]);
                        blk[NS * k + 12] += R_FABS(dl[2]);
                    } else {
                        const real dG_ddelx = -G * FILTER_INV_SQUARE * h.dx, dG_ddely = -G * FILTER_INV_SQUARE * h.dy;
                        blk[NS * k + 9] += dL_dG * dG_ddelx;
                        blk[NS * k + 10] += dL_dG * dG_ddely;
                        blk[NS * k + 8] += dL_dz;
                    }
                    blk[NS * k + 19] += G * dL_dalpha;
                }
            }
            for (uint32_t k = 0; k < kmax; ++k) {   /* block totals -> the surfel's accumulators */
                real* a = blk + (size_t)NS * k;
                const uint32_t j = point_list[r0 + k];
                for (int s_ = 0; s_ < NS; ++s_) {
                    if (a[s_] == RC(0)) continue;
                    real* dst = s_ < 9 ? dL_dtransMat + 9 * j + s_ : s_ < 13 ? dL_dmean2D + 4 * j + (s_ - 9)
                              : s_ < 16 ? dL_dnormal + 3 * j + (s_ - 13) : s_ < 19 ? dL_dcolor + 3 * j + (s_ - 16) : dL_dopacity + j;
                    accum(dst, a[s_], atomic);
                    a[s_] = RC(0);
                }
            }
        }
        free(blk);
    }
}

/* ------------------------------------------------------------------------- */
/* preprocess backward: T, low-pass centre, normal, colour -> means3D, scales (N,2), rotations, SH;   */
/* dL_dmean2D_out (N,4) = the densification signal described in the header.                           */
/* ------------------------------------------------------------------------- */
void oracle_surfel_preprocess_bwd(int N, int deg, int M, const real* means3D, const int32_t* radii,
                                  const real* shs, const uint8_t* clamped, const real* scales,
                                  const real* rotations, real scale_modifier, const real* transMats,
                                  int transmat_is_precomp, int colors_is_precomp, const real* view,
                                  const real* proj, const real* campos, int W, int H, const real* dL_dtransMat_in,
                                  const real* dL_dmean2D_in, const real* dL_dnormal, const real* dL_dcolor,
                                  real* dL_dmeans3D, real* dL_dtransMat_out, real* dL_dsh, real* dL_dscale,
                                  real* dL_drot, real* dL_dmean2D_out, int nthreads) {
    (void)nthreads;
    const real hw = (real)W / RC(2), hh = (real)H / RC(2), cw = (real)(W - 1) / RC(2), ch_ = (real)(H - 1) / RC(2);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        const real* p = means3D + 3 * i;
        const real* T9 = transMats + 9 * i;
        const real* Tu = T9; const real* Tv = T9 + 3; const real* Tw = T9 + 6;
        real dT[9];
        for (int k = 0; k < 9; ++k) dT[k] = dL_dtransMat_in[9 * i + k];
        const real depth = T9[8];
        dL_dmean2D_out[4 * i + 0] = dT[2] * depth * RC(0.5) * (real)W;
        dL_dmean2D_out[4 * i + 1] = dT[5] * depth * RC(0.5) * (real)H;
        dL_dmean2D_out[4 * i + 2] = dL_dmean2D_in[4 * i + 2] * depth * RC(0.5) * (real)W;
        dL_dmean2D_out[4 * i + 3] = dL_dmean2D_in[4 * i + 3] * depth * RC(0.5) * (real)H;

        /* low-pass branch: centre = aabb centre(T) */
        const real gx_ = dL_dmean2D_in[4 * i + 0], gy_ = dL_dmean2D_in[4 * i + 1];
        if (gx_ != RC(0) || gy_ != RC(0)) {
            const real t[3] = {RC(9), RC(9), RC(-1)};
            const real d = (t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1]) + t[2] * Tw[2] * Tw[2];
            const real inv_d = RC(1) / d;
            real f[3], dfdot = 0;
            for (int k = 0; k < 3; ++k) {
                f[k] = t[k] * inv_d;
                dT[0 + k] += gx_ * f[k] * Tw[k];
                dT[3 + k] += gy_ * f[k] * Tw[k];
                dT[6 + k] += gx_ * f[k] * Tu[k] + gy_ * f[k] * Tv[k];
                dfdot += (gx_ * Tu[k] * Tw[k] + gy_ * Tv[k] * Tw[k]) * f[k];
            }
            const real dL_dd = -dfdot * inv_d;
            for (int k = 0; k < 3; ++k) dT[6 + k] += dL_dd * RC(2) * t[k] * Tw[k];
        }
        real dmean[3] = {0, 0, 0};
        if (transmat_is_precomp) {
            for (int k = 0; k < 9; ++k) dL_dtransMat_out[9 * i + k] = dT[k];
        } else {
            /* rows of T -> clip-space images c_i of (L0, L1, p) -> world vectors */
            real dv[3][3];
            for (int a = 0; a < 3; ++a) {
                const real dc[4] = {dT[0 + a] * hw, dT[3 + a] * hh, RC(0), dT[0 + a] * cw + dT[3 + a] * ch_ + dT[6 + a]};
                for (int r = 0; r < 3; ++r)
                    dv[a][r] = proj[4 * r + 0] * dc[0] + proj[4 * r + 1] * dc[1] + proj[4 * r + 3] * dc[3];
            }
            for (int r = 0; r < 3; ++r) dmean[r] += dv[2][r];
            real R[9];
            const real* q = rotations + 4 * i;
            quat_to_R(q, R);
            const real s0 = scale_modifier * scales[2 * i], s1 = scale_modifier * scales[2 * i + 1];
            /* normal: n_view = mult * (n_world @ view3x3) */
            real pv[3];
            xform4x3(p, view, pv);
            real nv[3] = {(view[0] * R[2] + view[4] * R[5]) + view[8] * R[8], (view[1] * R[2] + view[5] * R[5]) + view[9] * R[8],
                          (view[2] * R[2] + view[6] * R[5]) + view[10] * R[8]};
            const real cosv = -((pv[0] * nv[0] + pv[1] * nv[1]) + pv[2] * nv[2]);
            const real mult = cosv > RC(0) ? RC(1) : RC(-1);
            real dn[3];
            for (int r = 0; r < 3; ++r)
                dn[r] = mult * (view[4 * r + 0] * dL_dnormal[3 * i + 0] + view[4 * r + 1] * dL_dnormal[3 * i + 1] +
                                view[4 * r + 2] * dL_dnormal[3 * i + 2]);
            real dR[9];
            real ds0 = 0, ds1 = 0;
            for (int r = 0; r < 3; ++r) {
                dR[3 * r + 0] = s0 * dv[0][r];
                dR[3 * r + 1] = s1 * dv[1][r];
                dR[3 * r + 2] = dn[r];
                ds0 += R[3 * r + 0] * dv[0][r];
                ds1 += R[3 * r + 1] * dv[1][r];
            }
            dL_dscale[2 * i + 0] = scale_modifier * ds0;
            dL_dscale[2 * i + 1] = scale_modifier * ds1;
            real qr = q[0], qx = q[1], qy = q[2], qz = q[3];
#define G_(r, c) dR[3 * (r) + (c)]
            dL_drot[4 * i + 0] = RC(2) * (-qz * G_(0, 1) + qy * G_(0, 2) + qz * G_(1, 0) - qx * G_(1, 2) - qy * G_(2, 0) + qx * G_(2, 1));
            dL_drot[4 * i + 1] = RC(2) * (qy * G_(0, 1) + qz * G_(0, 2) + qy * G_(1, 0) - RC(2) * qx * G_(1, 1) - qr * G_(1, 2) + qz * G_(2, 0) + qr * G_(2, 1) - RC(2) * qx * G_(2, 2));
            dL_drot[4 * i + 2] = RC(2) * (-RC(2) * qy * G_(0, 0) + qx * G_(0, 1) + qr * G_(0, 2) + qx * G_(1, 0) + qz * G_(1, 2) - qr * G_(2, 0) + qz * G_(2, 1) - RC(2) * qy * G_(2, 2));
            dL_drot[4 * i + 3] = RC(2) * (-RC(2) * qz * G_(0, 0) - qr * G_(0, 1) + qx * G_(0, 2) + qr * G_(1, 0) - RC(2) * qz * G_(1, 1) + qy * G_(1, 2) + qx * G_(2, 0) + qy * G_(2, 1));
#undef G_
        }
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
    }
}

/* ------------------------------------------------------------------------- */
/* simple_knn.distCUDA2 (renderer_2dgs.py:11,92-96): brute-force restatement.  out[i] = (d1 + d2 + d3) / 3 with d_k   */
/* the squared distances of the three nearest OTHER points (by index), +inf terms when fewer than four points exist.   */
/* ------------------------------------------------------------------------- */
void oracle_knn_mean_dist2(int N, const real* pts, real* out, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < N; ++i) {
        real b0 = (real)INFINITY, b1 = (real)INFINITY, b2 = (real)INFINITY;
        const real px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            const real dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
            real d = dx * dx + dy * dy + dz * dz;
            if (d < b0) { real t = b0; b0 = d; d = t; }
            if (d < b1) { real t = b1; b1 = d; d = t; }
            if (d < b2) b2 = d;
        }
        out[i] = (b0 + b1 + b2) / RC(3);
    }
}
