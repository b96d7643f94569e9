// gdr_common.h — shared declarations of the gfx950 rasterizer kernels (internal).
// Public ABI: include/gdr.h.  Everything here is CDNA4-only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gdr.h"
#include "../../include/gsr.h"

#define GDR_WAVE 64
#define GDR_BLOCK 256          // threads per workgroup (4 waves) everywhere
#define GDR_TILE_PIX (GDR_TILE * GDR_TILE)

// radix sort geometry (binning.hip)
#define GDR_SORT_ITEMS 16                               // keys per thread
#define GDR_SORT_TILE (GDR_BLOCK * GDR_SORT_ITEMS)      // keys per workgroup
#define GDR_RADIX_BITS 8
#define GDR_RADIX (1 << GDR_RADIX_BITS)

// kernel ids for the opt-in per-kernel timing (gdr_profile_*, include/gdr.h)
enum {
    GDR_K_PREPROCESS_FWD = 0,
    GDR_K_SCAN,
    GDR_K_DUPLICATE,
    GDR_K_SORT_HIST,
    GDR_K_SORT_ROWSCAN,
    GDR_K_SORT_SCATTER,
    GDR_K_RANGES,
    GDR_K_RENDER_FWD,
    GDR_K_RENDER_BWD,
    GDR_K_PREPROCESS_BWD,
    GDR_K_MARK_VISIBLE,
    GDR_K_TILE_ORDER,
    GDR_K_TILE_SORT,
    GDR_K_TILE_SORT_LONG,
    GDR_K_VIEW_LOSS,
    GDR_K_SURFEL_MAPS,
    GDR_K_KNN,
    GDR_K_SELECT,
    GDR_K_RENDER_FWD_DEEP,
    GDR_K_TILE_COUNT,
    GDR_K_TILE_SCAN,
    GDR_K_TILE_SCATTER,
    GDR_K_COUNT
};

namespace gdr {

// no-ops unless gdr_profile_enable(1) was called: bracket one launch with HIP events on `st`
void prof_begin(int kernel_id, hipStream_t st);
void prof_end(int kernel_id, hipStream_t st);

#define GDR_LAUNCH(KID, KERNEL, GRID, BLOCK, ST, ...)                     \
    do {                                                                  \
        ::gdr::prof_begin(KID, ST);                                       \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, ST, __VA_ARGS__);      \
        ::gdr::prof_end(KID, ST);                                         \
    } while (0)

void set_error(const char* what, hipError_t e);

static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int tile_grid_x(int W) { return (W + GDR_TILE - 1) / GDR_TILE; }
static inline int tile_grid_y(int H) { return (H + GDR_TILE - 1) / GDR_TILE; }

// number of key bits that must be sorted: 32 depth bits + bits of (tiles-1)
static inline int key_bits(int tiles) {
    int b = 0;
    while ((1u << b) < (unsigned)tiles) ++b;
    return 32 + b;
}

// ---- launchers (one per translation unit) ----------------------------------------
hipError_t launch_preprocess_fwd(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* g,
                                 int32_t* radii, hipStream_t st);
hipError_t launch_preprocess_bwd(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* g,
                                 const int32_t* radii, const gdr_grad_outputs* go, hipStream_t st);
hipError_t launch_preprocess_fwd_views(int V, const gdr_settings* s, const gdr_inputs* in,
                                       const gdr_geom* geoms, int32_t* const* radii, hipStream_t st);
hipError_t launch_preprocess_bwd_views(int V, const gdr_settings* s, const gdr_inputs* in,
                                       const gdr_geom* geoms, const int32_t* const* radii,
                                       float* const* grad_recs, const gdr_grad_outputs* go,
                                       hipStream_t st);
hipError_t launch_mark_visible(int N, const float* means3D, const float* view, uint8_t* present,
                               hipStream_t st);
hipError_t launch_scan_block_sums(const gdr_geom* g, int N, hipStream_t st);
// ---- binning (binning.hip): every kernel works on a table of V <= GDR_MAX_VIEWS views, view = blockIdx.y ------------
// The binning chain of one view is ~13 short dependent launches (duplicate, 2 x (hist, row scan, scatter), ranges,
// tile order, 3 x tile sort) that leave most of the chip idle and cost the host ~5-10 us each: a multi-view node issues
// the chain ONCE for all its views (kernel timeline of a C4 step, scripts/gpu_timeline.sh: 52 launches spread over
// 700 us before, host-launch bound).
struct BinView {
    const int32_t* radii; const float* depths; const int4* rect; const uint32_t* tiles_touched; const uint32_t* block_offs;
    uint64_t* keys[2]; uint32_t* vals[2]; uint32_t* hist; uint32_t* scratch32;
    uint2* ranges; uint32_t* tile_order; uint32_t* seg_base; uint2* seg_extra; uint32_t* seg_count;
    uint64_t D; uint32_t nblk; int32_t seg_len, seg_cap; uint32_t deep_max_busy, deep_min_mean;
    const uint32_t* d_dev;   // gdr_binning.d_dev: the duplicate count stays on the device, D/nblk above are CAPACITIES
    uint32_t* stats_out;     // gdr_binning.stats_out (tile_order_kernel writes it)
    int32_t hint_long, hint_medium;   // host side only: grid sizes of the tile sort's long / medium class
    uint32_t* tile_hist; int32_t hist_width;   // direct tile binning: (hist_width rows, tiles rounded up to 64) counts + a totals row
    int32_t from_totals;                       // tile_order_kernel: ranges are formed from the totals row first
};
#define GDR_BIN_MAX_TILES 16384     // LDS histogram of the direct tile binning: 64 KB (beyond: radix partition on the tile bits)
#define GDR_BIN_MAX_WIDTH 256       // workgroups of tile_count / tile_scatter = rows of the count matrix
// tile sort size classes (list entries): one workgroup per tile up to SMALL, small grids walking the longer tiles
#define GDR_TSORT_SMALL 2048
#define GDR_TSORT_MEDIUM 4096
#define GDR_TSORT_LARGE 16384
struct BinViews { BinView v[GDR_MAX_VIEWS]; };
void fill_bin_views(BinViews* vs, int V, const gdr_geom* geoms, const gdr_binning* bins, const gdr_image* imgs,
                    const uint64_t* D, const int32_t* const* radii);
hipError_t launch_duplicate_views(const BinViews& vs, int V, int N, int W, int H, hipStream_t st);
hipError_t launch_sort_views(const BinViews& vs, int V, int lo, int hi, int* sorted, hipStream_t st);
hipError_t launch_ranges_views(const BinViews& vs, int V, int cur, int tiles, hipStream_t st);
hipError_t launch_tile_sort_views(const BinViews& vs, int V, int in, int tiles, bool packed, hipStream_t st);
hipError_t launch_tile_count_scan(const BinViews& vs, int V, int N, int W, int H, hipStream_t st);
hipError_t launch_tile_scatter(const BinViews& vs, int V, int N, int W, int H, hipStream_t st);
hipError_t launch_tile_order_views(const BinViews& vs, int V, int tiles, hipStream_t st);
// per-pixel compositing state saved at a cut of a long tile list, 256 pixels each: 3DGS T, colour x3, depth, alpha
// sums (6 used); surfels T, colour x3, normal x3, depth, M1, M2 (10)
#define GDR_SEG_STATE_FLOATS (10 * GDR_BLOCK)
#define GDR_DEEP_MIN_MEAN 2560   /* mean list length of the busy tiles from which the deep forward applies (render.hip) */
hipError_t launch_render_fwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_outputs* out, hipStream_t st);
hipError_t launch_render_fwd_loss(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                  const gdr_image* img, const gdr_outputs* out, const float* target, float w_depth,
                                  float w_alpha, float* loss, hipStream_t st);
hipError_t launch_render_fwd_lossgrad(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                      const gdr_image* img, const float* target, float go_scale, float* loss,
                                      float* dL_dcolor, hipStream_t st);
hipError_t launch_render_fwd_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin, const gdr_image* img,
                                   const gdr_outputs* outs, int loss_mode, const float* const* targets, float w_depth,
                                   float w_alpha, float go_scale, float* losses, int interleave, hipStream_t st);
hipError_t launch_topk_absgrad(int N, const float* grad, const uint8_t* cand, int k, void* workspace,
                               uint8_t* mask, int32_t* idx, hipStream_t st);
size_t select_workspace_bytes();
void surfel_render_bwd_set_pairs(int pairs);
void render_bwd_set_pairs(int pairs);   // K7 variant of this thread's next launches: 0 rows, 1 row pairs where they pay
hipError_t launch_render_bwd_loss(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                  const gdr_image* img, const float* color, const float* target, float w_depth,
                                  float w_alpha, const float* go, float* grad_rec, hipStream_t st);
hipError_t launch_render_bwd_mean2d_loss(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                         const gdr_image* img, const float* color, const float* target, const float* go,
                                         float* dL_dmean2D, hipStream_t st);
hipError_t launch_render_bwd_mean2d(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                    const gdr_image* img, const float* dL_dcolor, float* dL_dmean2D,
                                    hipStream_t st);
hipError_t launch_render_bwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_grad_inputs* gi,
                             const gdr_grad_outputs* go, hipStream_t st);

hipError_t launch_render_bwd_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                   const gdr_image* img, const gdr_grad_inputs* gi, float* const* grad_recs,
                                   int interleave, hipStream_t st);
hipError_t launch_render_bwd_loss_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                        const gdr_image* img, const float* const* colors, const float* const* targets,
                                        float w_depth, float w_alpha, const float* go, float* const* grad_recs,
                                        int interleave, hipStream_t st);
hipError_t launch_render_bwd_mean2d_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                          const gdr_image* img, const float* const* dL_dcolors, float* dL_dmean2D,
                                          int interleave, hipStream_t st);

hipError_t launch_view_loss_fwd(const float* color, const float* depth, const float* alpha, const float* target,
                                int P, float w_depth, float w_alpha, float* loss, hipStream_t st);
hipError_t launch_view_loss_bwd(const float* color, const float* target, int P, float w_depth, float w_alpha,
                                const float* g, float* d_color, float* d_depth, float* d_alpha, hipStream_t st);

// 2DGS surfel path (preprocess_surfel.hip, render_surfel.hip)
hipError_t launch_surfel_preprocess_fwd(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* g,
                                        int32_t* radii, hipStream_t st);
hipError_t launch_surfel_preprocess_bwd(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* g,
                                        const int32_t* radii, const gsr_grad_outputs* go, hipStream_t st);
hipError_t launch_surfel_preprocess_fwd_views(int V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                              int32_t* const* radii, hipStream_t st);
hipError_t launch_surfel_preprocess_bwd_views(int V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                              const int32_t* const* radii, float* const* grad_recs,
                                              const gsr_grad_outputs* go, hipStream_t st);
hipError_t launch_surfel_render_fwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                    const gdr_image* img, const gsr_outputs* out, hipStream_t st);
hipError_t launch_surfel_render_bwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                    const gdr_image* img, const gsr_grad_inputs* gi, float* grad_rec,
                                    hipStream_t st);

hipError_t launch_surfel_render_bwd_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                          const gdr_image* img, const gsr_grad_inputs* gi, float* const* grad_recs, int interleave,
                                          hipStream_t st);
hipError_t launch_surfel_means2d_view(int N, const gdr_settings* s, const gdr_geom* g, const int32_t* radii,
                                      const float* grad_rec, float* out, hipStream_t st);

hipError_t launch_surfel_maps_fwd(const float* allmap, const float* rays, const float* view, int H, int W, float r,
                                  float* depth, float* acc, float* rend_normal, float* depth_normal, float* rend_dist,
                                  hipStream_t st);
hipError_t launch_surfel_maps_bwd(const float* allmap, const float* rays, const float* view, int H, int W, float r,
                                  const float* g_depth, const float* g_acc, const float* g_rn, const float* g_dn,
                                  const float* g_dist, float* scratch, float* dL_dallmap, hipStream_t st);

hipError_t launch_surfel_loss_fwd(const float* color, const float* allmap, const float* rays, const float* view,
                                  const float* target, int H, int W, float r, float w_dist, float w_normal, float w_depth,
                                  float w_alpha, float* loss, hipStream_t st);
hipError_t launch_surfel_loss_bwd(const float* color, const float* allmap, const float* rays, const float* view,
                                  const float* target, int H, int W, float r, float w_dist, float w_normal, float w_depth,
                                  float w_alpha, const float* g, float* scratch, float* dL_dcolor, float* dL_dallmap,
                                  hipStream_t st);

hipError_t launch_knn_cells(const float* pts, int N, const float* bbox, int G, int32_t* cell, hipStream_t st);
hipError_t launch_knn_mean_dist2(const float* pts_sorted, int N, const float* bbox, int G, const int32_t* cell_start,
                                 float* out, hipStream_t st);

size_t sort_hist_bytes(uint64_t D);
hipError_t launch_words_differ(const void* a, const void* b, uint64_t n_bytes, uint32_t* flag, hipStream_t st);
#define GDR_DIFFER_MAX 8   /* = the arrays of gdr_same_as (include/gdr.h) */
hipError_t launch_words_differ_multi(int n, const void* const* a, const void* const* b, const uint64_t* n_bytes, uint32_t* flag,
                                     hipStream_t st);
// out[c] = 1 if the bg / viewmatrix / projmatrix / campos words of candidate c equal those of `cur` bit for bit (c < n <=
// GDR_REUSE_MAX; candidates whose bit in `eligible` is clear are not read: out[c] = 0)
hipError_t launch_settings_match(const gdr_settings* cur, int n, const gdr_settings* cand, uint64_t eligible, uint32_t* out,
                                 hipStream_t st);

}  // namespace gdr
