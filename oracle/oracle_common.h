/*
 * oracle/oracle_common.h — helpers shared by the two CPU restatements (gdr_oracle.c: 3DGS path,
 * gsr_oracle.c: 2DGS surfel path).  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED — see the headers of those files.
 */
#ifndef ORACLE_COMMON_H
#define ORACLE_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef GDR_REAL_DOUBLE
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define R_FABS fabs
#else
typedef float real;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#define R_FABS fabsf
#endif
#define RC(x) ((real)(x))

#define BLOCK_X 16
#define BLOCK_Y 16

/* SH constants: SURVEY Appendix A.1-9; C0 = lightning/renderer.py:17 */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};


static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* [p,1] @ M for a 4x4 stored as 16 contiguous values (torch row-major of the
 * row-vector-convention matrix, lightning/utils.py:37-47).  Appendix A preamble. */
static inline void xform4x3(const real* p, const real* m, real* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const real* p, const real* m, real* o) {
    xform4x3(p, m, o);
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Appendix A.1-3: Sigma = R diag(mod*s)^2 R^T, quaternion (r,x,y,z) used as given. */
static inline void quat_to_R(const real* q, real R[9]) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = RC(1) - RC(2) * (y * y + z * z);
    R[1] = RC(2) * (x * y - r * z);
    R[2] = RC(2) * (x * z + r * y);
    R[3] = RC(2) * (x * y + r * z);
    R[4] = RC(1) - RC(2) * (x * x + z * z);
    R[5] = RC(2) * (y * z - r * x);
    R[6] = RC(2) * (x * z - r * y);
    R[7] = RC(2) * (y * z + r * x);
    R[8] = RC(1) - RC(2) * (x * x + y * y);
}
static inline void compute_cov3D(const real* scale, real mod, const real* q, real* cov6) {
    real R[9], Mm[9];
    quat_to_R(q, R);
    real s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) Mm[i * 3 + k] = R[i * 3 + k] * s[k];
    /* Sigma_ij = (M_i0 M_j0 + M_i1 M_j1) + M_i2 M_j2 */
#define SIG(i, j) ((Mm[i * 3 + 0] * Mm[j * 3 + 0] + Mm[i * 3 + 1] * Mm[j * 3 + 1]) + Mm[i * 3 + 2] * Mm[j * 3 + 2])
    cov6[0] = SIG(0, 0);
    cov6[1] = SIG(0, 1);
    cov6[2] = SIG(0, 2);
    cov6[3] = SIG(1, 1);
    cov6[4] = SIG(1, 2);
    cov6[5] = SIG(2, 2);
#undef SIG
}

/* SH basis b_k(dir), k < 16.  Appendix A.1-9. */
static inline void sh_basis(int deg, real x, real y, real z, real* b) {
    b[0] = RC(SH_C0);
    if (deg < 1) return;
    b[1] = -RC(SH_C1) * y;
    b[2] = RC(SH_C1) * z;
    b[3] = -RC(SH_C1) * x;
    if (deg < 2) return;
    real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = RC(SH_C2[0]) * xy;
    b[5] = RC(SH_C2[1]) * yz;
    b[6] = RC(SH_C2[2]) * (RC(2) * zz - xx - yy);
    b[7] = RC(SH_C2[3]) * xz;
    b[8] = RC(SH_C2[4]) * (xx - yy);
    if (deg < 3) return;
    b[9] = RC(SH_C3[0]) * y * (RC(3) * xx - yy);
    b[10] = RC(SH_C3[1]) * xy * z;
    b[11] = RC(SH_C3[2]) * y * (RC(4) * zz - xx - yy);
    b[12] = RC(SH_C3[3]) * z * (RC(2) * zz - RC(3) * xx - RC(3) * yy);
    b[13] = RC(SH_C3[4]) * x * (RC(4) * zz - xx - yy);
    b[14] = RC(SH_C3[5]) * z * (xx - yy);
    b[15] = RC(SH_C3[6]) * x * (xx - RC(3) * yy);
}
/* d b_k / d(x,y,z) */
static inline void sh_basis_grad(int deg, real x, real y, real z, real* bx, real* by, real* bz) {
    for (int k = 0; k < 16; ++k) bx[k] = by[k] = bz[k] = 0;
    if (deg < 1) return;
    by[1] = -RC(SH_C1);
    bz[2] = RC(SH_C1);
    bx[3] = -RC(SH_C1);
    if (deg < 2) return;
    real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    bx[4] = RC(SH_C2[0]) * y;  by[4] = RC(SH_C2[0]) * x;
    by[5] = RC(SH_C2[1]) * z;  bz[5] = RC(SH_C2[1]) * y;
    bx[6] = RC(SH_C2[2]) * (-RC(2) * x); by[6] = RC(SH_C2[2]) * (-RC(2) * y); bz[6] = RC(SH_C2[2]) * (RC(4) * z);
    bx[7] = RC(SH_C2[3]) * z;  bz[7] = RC(SH_C2[3]) * x;
    bx[8] = RC(SH_C2[4]) * (RC(2) * x); by[8] = RC(SH_C2[4]) * (-RC(2) * y);
    if (deg < 3) return;
    bx[9] = RC(SH_C3[0]) * (RC(6) * xy);  by[9] = RC(SH_C3[0]) * (RC(3) * xx - RC(3) * yy);
    bx[10] = RC(SH_C3[1]) * yz; by[10] = RC(SH_C3[1]) * xz; bz[10] = RC(SH_C3[1]) * xy;
    bx[11] = RC(SH_C3[2]) * (-RC(2) * xy); by[11] = RC(SH_C3[2]) * (RC(4) * zz - xx - RC(3) * yy); bz[11] = RC(SH_C3[2]) * (RC(8) * yz);
    bx[12] = RC(SH_C3[3]) * (-RC(6) * xz); by[12] = RC(SH_C3[3]) * (-RC(6) * yz); bz[12] = RC(SH_C3[3]) * (RC(6) * zz - RC(3) * xx - RC(3) * yy);
    bx[13] = RC(SH_C3[4]) * (RC(4) * zz - RC(3) * xx - yy); by[13] = RC(SH_C3[4]) * (-RC(2) * xy); bz[13] = RC(SH_C3[4]) * (RC(8) * xz);
    bx[14] = RC(SH_C3[5]) * (RC(2) * xz); by[14] = RC(SH_C3[5]) * (-RC(2) * yz); bz[14] = RC(SH_C3[5]) * (xx - yy);
    bx[15] = RC(SH_C3[6]) * (RC(3) * xx - RC(3) * yy); by[15] = RC(SH_C3[6]) * (-RC(6) * xy);
}

/* += with an OpenMP atomic when several tiles run concurrently */
/* Tile selection of the four render loops (oracle_select_tiles): NULL = every tile; otherwise only the listed tiles are
 * composited / differentiated, so that a comparison at BASELINE sizes (2 M Gaussians, 800x800) costs a fraction of a
 * full render on a host with few cores.  Pixels outside the selection keep whatever the output arrays held (the
 * Python front-end zero-fills); a backward over a selection equals the full backward with the upstream pixel
 * gradients zeroed outside the selected tiles. */
extern const int32_t* g_oracle_tile_sel;
extern int g_oracle_tile_sel_n;
#define ORACLE_TILE_COUNT(all) (g_oracle_tile_sel ? g_oracle_tile_sel_n : (all))
#define ORACLE_TILE_AT(it) (g_oracle_tile_sel ? (int)g_oracle_tile_sel[it] : (it))

static inline void accum(real* p, real v, int atomic) {
    if (atomic) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        *p += v;
    } else {
        *p += v;
    }
}

#endif
