/*
 * include/gsr.h — C ABI of the 2D-Gaussian (surfel) rasterizer in libgdr_hip.so (MI355X, gfx950): the native
 * half of `diff_surfel_rasterization`, the package /root/reference/lightning/renderer_2dgs.py:7-10 imports
 * (BASELINE config 5).  What the reference binds:
 *   - settings record:   /root/reference/lightning/renderer_2dgs.py:111-124 (the same 12 fields as the 3DGS path)
 *   - forward call:      /root/reference/lightning/renderer_2dgs.py:224-234
 *                        (-> rendered_image (3,H,W), radii (N), allmap (7,H,W))
 *   - allmap channels:   /root/reference/lightning/renderer_2dgs.py:241-257  0 expected depth (sum w z), 1 alpha,
 *                        2-4 normal (view space), 5 median depth, 6 depth distortion
 *   - inputs:            scales (N,2) (renderer_2dgs.py:92-96), means2D carrier (N,4) (:207-208)
 * The package itself is in neither .gitmodules nor the tree, so the arithmetic follows the published 2DGS
 * algorithm as restated in oracle/gsr_oracle.c (PARITY UNPINNED).  Conventions, ownership, error codes and the
 * stream rule are those of include/gdr.h, whose settings / binning / workspace structs are reused:
 *   gdr_geom   — `rec` holds 24 floats (96 B) per surfel instead of 16:
 *                  [0..2] Tu  [3] centre x      [4..6] Tv  [7] centre y     [8..10] Tw  [11] opacity
 *                  [12..14] normal (view space, facing the camera)  [15] r   [16] g  [17] b
 *                  [18..19] lower, [20..21] upper corner of a conservative box around {alpha >= 1/255}
 *                where (Tu, Tv, Tw) are the rows of the splat-to-pixel homography: (u,v,1) . (Tu,Tv,Tw) =
 *                (x w, y w, w).  `cov3D` is unused.
 *   gdr_image  — n_contrib is (2,H,W): last contributor, median contributor (1-based, 0 = none);
 *                final_T is (3,H,W): T, M1 = sum w m, M2 = sum w m^2 (m = normalised depth).
 * Use gsr_geom_bytes / gsr_image_bytes + gsr_*_carve for these layouts; binning is gdr_binning_bytes / _carve.
 */
#ifndef GSR_H
#define GSR_H

#include "gdr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_REC_FLOATS 24  /* floats per surfel in gdr_geom.rec          */
#define GSR_GRAD_FLOATS 32 /* floats per surfel in gsr_grad_outputs.scratch */

/* Per-surfel inputs of GaussianRasterizer.forward (renderer_2dgs.py:224-234).  Exactly one of shs /
 * colors_precomp and one of (scales, rotations) / transMat_precomp is non-NULL.  flags: GDR_IN_RAW_* of gdr.h
 * (sigmoid / exp / normalize of renderer_2dgs.py:190-199 folded into the kernels). */
typedef struct gsr_inputs {
    int32_t N;
    int32_t M;                     /* SH coefficients per surfel stored in `shs` (>= (deg+1)^2) */
    const float* means3D;          /* (N,3) */
    const float* opacities;        /* (N)   */
    const float* shs;              /* (N,M,3) or NULL */
    const float* colors_precomp;   /* (N,3)   or NULL */
    const float* scales;           /* (N,2)   or NULL */
    const float* rotations;        /* (N,4)   or NULL, (r,x,y,z) */
    const float* transMat_precomp; /* (N,9)   or NULL: rows Tu, Tv, Tw (the adaptor's `cov3D_precomp` argument) */
    uint32_t flags;
    uint32_t reserved;
} gsr_inputs;

typedef struct gsr_outputs {
    float* color;   /* (3,H,W) */
    float* allmap;  /* (7,H,W) */
    int32_t* radii; /* (N)     */
} gsr_outputs;

typedef struct gsr_grad_inputs {
    const float* dL_dcolor;  /* (3,H,W) */
    const float* dL_dallmap; /* (7,H,W) or NULL (= zeros) */
} gsr_grad_inputs;

/* Every buffer is fully written (zeros for culled surfels) unless accumulate != 0.  dL_dmeans2D is (N,4):
 * columns 0-1 = dL/dTu.z, dL/dTv.z scaled by depth * 0.5 W (resp. H) — the densification signal of the lineage —
 * columns 2-3 the same with per-pixel |.| accumulation.  scratch: N * GSR_GRAD_FLOATS floats (128-byte records:
 * [0..8] dL/dT, [9] opacity, [10..12] colour, [13..14] |dTu.z|, |dTv.z|, [15] low-pass centre x | [16..18] normal,
 * [19] low-pass centre y — v15: the first 64-byte line holds everything an image-only loss produces; K7s touches the second
 * line only where a total is non-zero). */
typedef struct gsr_grad_outputs {
    float* dL_dmeans3D;   /* (N,3) */
    float* dL_dmeans2D;   /* (N,4) */
    float* dL_dshs;       /* (N,M,3) or NULL when colors_precomp was used */
    float* dL_dcolors;    /* (N,3)  or NULL when shs was used */
    float* dL_dopacities; /* (N)   */
    float* dL_dscales;    /* (N,2) or NULL when transMat_precomp was used */
    float* dL_drotations; /* (N,4) or NULL when transMat_precomp was used */
    float* dL_dtransMat;  /* (N,9) or NULL when scales/rotations were used */
    float* scratch;       /* (N*32) floats, 128-byte aligned */
    int32_t accumulate;
    int32_t reserved;
} gsr_grad_outputs;

size_t gsr_geom_bytes(int32_t N);
size_t gsr_image_bytes(int32_t H, int32_t W);
int gsr_geom_carve(void* base, int32_t N, gdr_geom* out);
int gsr_image_carve(void* base, int32_t H, int32_t W, gdr_image* out);

/* Stage 1: per-surfel homography, bounding box, SH colour, tile rect, total D (same contract as
 * gdr_preprocess_forward).  Stage 2: duplicate / sort / ranges (the 3DGS path's kernels) + surfel compositing. */
int gsr_preprocess_forward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, int32_t* radii,
                           uint32_t* num_rendered_host, void* stream);
int gsr_render_forward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, gdr_binning* bin,
                       const gdr_image* img, uint64_t D, const gsr_outputs* out, void* stream);
/* K6s alone (after gdr_binning_forward on the surfel geometry); see gdr.h for the two-stream use */
int gsr_composite_forward(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                          const gsr_outputs* out, void* stream);
int gsr_forward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, gdr_binning* bin,
                const gdr_image* img, uint64_t D_cap, const gsr_outputs* out, uint32_t* num_rendered_host,
                void* stream);
/* K7s of V <= GDR_MAX_VIEWS views of one image size in ONE launch (v14; see gdr_render_backward_views): every view writes its
 * own N*GSR_GRAD_FLOATS record (cleared here unless bins[v].grad_rec_cleared). */
int gsr_render_backward_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms, const gdr_binning* bins,
                              const gdr_image* imgs, const gsr_grad_inputs* gins, float* const* grad_recs,
                              int32_t interleave, void* stream);
/* The (N,4) means2D gradient of ONE view from its K7s record (v14): columns 0-1 = dL/dTu.z, dL/dTv.z x depth x W/2 | H/2
 * (the densification signal K9s returns summed over the views), columns 2-3 its per-pixel-|.| twin; zero where radii <= 0.
 * For callers that give every view its own means2D carrier (/root/reference/lightning/renderer_2dgs.py:209-222) while one
 * gsr_preprocess_backward_views serves all the views. */
int gsr_means2d_of_view(const gdr_settings* s, int32_t N, const gdr_geom* geom, const int32_t* radii,
                        const float* grad_rec, float* dL_dmean2D, void* stream);
/* one forward call per view (v14): the surfel twin of gdr_forward_view — plan with gdr_view_plan_for(N, H, W, 1, ...);
 * replaces the forward of `diff_surfel_rasterization._C.rasterize_gaussians` reached from
 * /root/reference/lightning/renderer_2dgs.py:224-234 in one native call */
int gsr_forward_view(const gdr_settings* s, const gsr_inputs* in, const gdr_view_plan* plan, void* workspace,
                     const gdr_view_opts* opts, const gdr_same_as* same, const gsr_outputs* out, gdr_view_state* state,
                     void* stream);
int gsr_backward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, const gdr_binning* bin,
                 const gdr_image* img, uint64_t D, const int32_t* radii, const gsr_grad_inputs* gin,
                 const gsr_grad_outputs* gout, void* stream);

/* ---- multi-view entry points (one surfel set, V <= GDR_MAX_VIEWS views of one image size; as gdr.h's) ----------------
 * gsr_preprocess_forward_views: K1s for V views in one launch (inputs and activations once); read the V values
 * geoms[v].num_rendered, then per view gdr_binning_forward + gsr_composite_forward (or gsr_render_forward).
 * gsr_render_backward: memset + K7s of one view into its own N*GSR_GRAD_FLOATS record.
 * gsr_preprocess_backward_views: K9s for V views at once (per-view world-space gradients summed in registers, the scale /
 * quaternion chain once); needs shs + scales + rotations and, at SH degree 1 or 3, M == (sh_degree+1)^2
 * (GDR_ERR_UNSUPPORTED otherwise: fall back to gsr_backward per view with accumulate). */
int gsr_preprocess_forward_views(int32_t V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                 int32_t* const* radii, void* stream);
int gsr_render_backward(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                        const gdr_image* img, const gsr_grad_inputs* gin, float* grad_rec, void* stream);
int gsr_preprocess_backward_views(int32_t V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                  const int32_t* const* radii, float* const* grad_recs, const gsr_grad_outputs* gout,
                                  void* stream);

/* ---- the adaptor's per-pixel maps, fused (renderer_2dgs.py:241-278; SURVEY §8f-3 "depth_to_normal fused") ------
 * forward:  allmap (7,H,W), rays (H,W,6) = origin | direction (dataLoader/utils.py:21-34), viewmatrix (16),
 *           depth_ratio r  ->  depth (H,W,1) = (1-r) nan0(allmap[0]/alpha) + r nan0(allmap[5]); acc_map (H,W) = alpha;
 *           rend_normal (H,W,3) = allmap[2:5] rotated to world space; depth_normal (H,W,3) = unit normal of the
 *           unprojected depth map (central differences, zero border) times alpha; rend_dist (H,W) = allmap[6].
 * backward: upstream gradients of the five maps (any may be NULL = zeros; alpha is detached inside depth_normal as in
 *           the reference) -> dL_dallmap (7,H,W), fully written.  scratch: 6*H*W floats (needed iff g_depth_normal). */
int gsr_maps_forward(const float* allmap, const float* rays, const float* viewmatrix, int32_t H, int32_t W,
                     float depth_ratio, float* depth, float* acc_map, float* rend_normal, float* depth_normal,
                     float* rend_dist, void* stream);
int gsr_maps_backward(const float* allmap, const float* rays, const float* viewmatrix, int32_t H, int32_t W,
                      float depth_ratio, const float* g_depth, const float* g_acc_map, const float* g_rend_normal,
                      const float* g_depth_normal, const float* g_rend_dist, float* scratch, float* dL_dallmap,
                      void* stream);

/* ---- fused image-space loss of the surfel path ----------------------------------------------------
 * loss += mean_{c,p}(clamp(color,0,1) - target)^2 + w_dist mean(rend_dist)
 *       + w_normal mean((1 - <rend_normal, depth_normal>) acc_map.detach()) + w_depth mean(depth) + w_alpha mean(acc_map)
 * on the maps gsr_maps_forward would produce (never materialised): renderer_2dgs.py:236 clamp, loss.py:37-38 MSE,
 * loss.py:49-61 distortion (x1000) and normal consistency (x0.2); the depth / alpha means are the measurement loss's
 * coverage terms.  color, target: (3,H,W); loss: device float the caller zeroes; g: device pointer to the
 * upstream scalar (NULL = 1); scratch: 9*H*W floats. */
int gsr_view_loss_forward(const float* color, const float* allmap, const float* rays, const float* viewmatrix,
                          const float* target, int32_t H, int32_t W, float depth_ratio, float w_dist, float w_normal,
                          float w_depth, float w_alpha, float* loss, void* stream);
int gsr_view_loss_backward(const float* color, const float* allmap, const float* rays, const float* viewmatrix,
                           const float* target, int32_t H, int32_t W, float depth_ratio, float w_dist, float w_normal,
                           float w_depth, float w_alpha, const float* g, float* scratch, float* dL_dcolor,
                           float* dL_dallmap, void* stream);

/* ---- simple_knn._C.distCUDA2 (renderer_2dgs.py:11,92-96) -----------------------------------------
 * out[i] = mean of the squared distances from point i to its three nearest OTHER points (fp32).  Two stages around
 * host plumbing (sort by cell, cell populations -> exclusive prefix):
 *   gsr_knn_cells:      cell[i] = (cz*G + cy)*G + cx of point i in the G^3 grid over bbox (device float[6]: min, max)
 *   gsr_knn_mean_dist2: points_sorted = the points in ascending cell order, cell_start = (G^3 + 1) exclusive prefix;
 *                       out is in the SAME (sorted) order.  Exact (growing cubic shells until the third best is closer
 *                       than the searched cube's nearest face). */
int gsr_knn_cells(const float* points, int32_t N, const float* bbox, int32_t G, int32_t* cell, void* stream);
int gsr_knn_mean_dist2(const float* points_sorted, int32_t N, const float* bbox, int32_t G, const int32_t* cell_start,
                       float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
