/*
 * include/gdr.h — C ABI of libgdr_hip.so, the MI355X (gfx950) differentiable
 * Gaussian-splatting rasterizer for the Generative Densification render path.
 *
 * This is the drop-in boundary: the entry points below are what a binding of the
 * reference's rasterizer extension would call.  The reference reaches its CUDA
 * extension through the Python package `diff_gaussian_rasterization`
 *   - import:            /root/reference/lightning/renderer.py:10-13
 *                        /root/reference/lightning/point_decoder/layers/gaussian_renderer.py:14
 *   - settings record:   /root/reference/lightning/renderer.py:111-124 (12 fields)
 *   - forward call:      /root/reference/lightning/renderer.py:250-259
 *                        (-> color(3,H,W), radii(N), depth(1,H,W), alpha(1,H,W))
 *   - backward contract: /root/reference/lightning/network.py:867-878
 *                        ((N,4) means2D gradient, |.|-accumulated in columns 2-3)
 * whose native half (`_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`,
 * `_C.mark_visible`; un-vendored submodule, /root/reference/.gitmodules:1-3) is what
 * these functions replace.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / HIP types in any signature
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream);
 *   - every pointer is a DEVICE pointer to fp32/int32 data unless it says "host";
 *   - matrices are 16 contiguous floats in the reference's row-vector convention
 *     (/root/reference/lightning/utils.py:37-47): p_view = [p,1] @ viewmatrix;
 *   - the caller owns every buffer; the library allocates no device memory.  Process-wide state it DOES keep (all
 *     mutex-guarded; results never depend on any of it except where said): (1) the per-shape view history behind
 *     gdr_view_plan_for — duplicates per Gaussian and launch-size reports of recent calls, gdr_view_history_*;
 *     (2) the K7 choice per launch shape, gdr_k7_tune_* — decided per shape by timing (once + one confirmation), so WHICH of two
 *     kernels sums a gradient (the same terms in another order: differences at the fp32 rounding level) can differ
 *     between processes unless gdr_k7_tune_override pins it; (3) pooled pinned words / events (gdr_host_copy_*,
 *     gdr_forward_view(s), gdr_view_reuse_probe); (4) the opt-in timing facility at the end of this header.
 *     Re-entrant, thread-safe for distinct workspaces, one call per stream;
 *   - all work is enqueued on `stream`; the only host synchronisation is the
 *     optional read-back of num_rendered in gdr_preprocess_forward;
 *   - return value: 0 = GDR_OK, negative = error (never throws across the ABI).
 */
#ifndef GDR_H
#define GDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDR_ABI_VERSION 17

#define GDR_OK 0
#define GDR_ERR_INVALID_ARG (-1)  /* NULL / inconsistent arguments                     */
#define GDR_ERR_HIP (-2)          /* a HIP call failed: see gdr_last_error()           */
#define GDR_ERR_UNSUPPORTED (-3)  /* sh_degree > 3, image too large for the key layout */
#define GDR_ERR_WORKSPACE (-4)    /* caller workspace smaller than required            */

/* Size limits of this build (checked at every entry point, GDR_ERR_UNSUPPORTED beyond them):
 * per-Gaussian element offsets are computed in 32-bit registers up to 9*N (SH rows use 64-bit
 * offsets), and num_rendered is a 32-bit counter as in the reference's int num_rendered. */
#define GDR_MAX_GAUSSIANS (1 << 27) /* 134 M Gaussians: 32 GB of degree-3 inputs               */
#define GDR_MAX_RENDERED 0xFFFFFFFFull /* D = sum of tiles_touched of one view                  */

#define GDR_DEFAULT_DEEP_MAX_BUSY 768 /* gdr_binning.deep_max_busy as carved: 3/4 of the 1024 resident K6 workgroups */
#define GDR_DEFAULT_SEG_LEN 256 /* gdr_binning.seg_len as carved (the library reads no environment variable); callers may
                                 * raise it after carving.  2048 in round 1; short segments balance K7's (tile, segment)
                                 * workgroups: 512 vs 2048 C3 shell 2130 -> 2640, C2 shell 1970 -> 2460 views/s at the same
                                 * speed on uniform scenes; 256 gains another 4-6 % on 512x512 images and loses 2 % at
                                 * 2 M Gaussians 800x800, so the Python host raises it to 512 for >= 2000 tiles. */
#define GDR_TILE 16 /* tile edge in pixels (BLOCK_X = BLOCK_Y = 16) */

/* The 12 fields of GaussianRasterizationSettings (renderer.py:111-124), flattened.
 * bg / viewmatrix / projmatrix / campos stay device tensors exactly as the caller
 * holds them (no host copies, no sync). */
typedef struct gdr_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;   /* active degree, 0..3 */
    int32_t prefiltered; /* accepted, unused (reference always passes False) */
    int32_t debug;       /* !=0: synchronise + check after every kernel */
    const float* bg;         /* (3)  */
    const float* viewmatrix; /* (16) */
    const float* projmatrix; /* (16) */
    const float* campos;     /* (3)  */
} gdr_settings;

/* Per-Gaussian inputs of GaussianRasterizer.forward (renderer.py:250-259).
 * Exactly one of shs / colors_precomp and one of (scales, rotations) / cov3D_precomp
 * is non-NULL. */
typedef struct gdr_inputs {
    int32_t N;                   /* number of Gaussians */
    int32_t M;                   /* SH coefficients per Gaussian stored in `shs` (>= (deg+1)^2) */
    const float* means3D;        /* (N,3) */
    const float* opacities;      /* (N)   activated, [0,1] */
    const float* shs;            /* (N,M,3) or NULL */
    const float* colors_precomp; /* (N,3)   or NULL */
    const float* scales;         /* (N,3)   or NULL, activated */
    const float* rotations;      /* (N,4)   or NULL, (r,x,y,z), used as given */
    const float* cov3D_precomp;  /* (N,6)   or NULL */
    uint32_t flags;              /* GDR_IN_RAW_*: fold the adaptor's activations into K1/K9 */
    uint32_t reserved;
} gdr_inputs;

/* gdr_inputs.flags — the render adaptor (lightning/renderer.py:225-230) applies sigmoid to the
 * opacity logits, exp to the log-scales and F.normalize to the quaternions before calling the
 * rasterizer.  With a flag set the corresponding pointer holds the RAW tensor, the activation
 * runs inside K1 and its derivative inside K9 (the returned gradient is w.r.t. the raw tensor). */
#define GDR_IN_RAW_OPACITY 1u
#define GDR_IN_RAW_SCALES 2u
#define GDR_IN_RAW_ROTATIONS 4u
/* parity-risk switch R1: the CUDA fork's source is unavailable, so whether its backward sends
 * dL/d(depth image) into the Gaussian centres (depth_i = z of the centre in view space) is unknown; default = it
 * does (ashawkey lineage).  With this flag the depth gradient still reaches the opacities / conics / 2D means
 * through the blend weights, but no longer moves the centres along the view axis. */
#define GDR_IN_NO_DEPTH_TO_MEAN 8u

/* Geometry state written by the forward and re-read by the backward
 * (upstream "geomBuffer").  Carved from one caller allocation by gdr_geom_carve. */
typedef struct gdr_geom {
    float* depths;           /* (N)   camera-space z                         */
    float* rec;              /* (N,16) render record, ONE 64-byte line per Gaussian so that
                              * K6/K7 gather one line instead of three:
                              *   [0..1] pixel-space mean xy   [2] depth   [3] -
                              *   [4..6] inverse 2D cov (xx,xy,yy)         [7] opacity
                              *   [8..10] colour rgb            [11] depth
                              *   [12..13] half-extent of {alpha >= 1/255} (px; < 0: never)  */
    float* cov3D;            /* (N,6)                                        */
    int32_t* rect;           /* (N,4) tile rect minx,miny,maxx,maxy          */
    uint32_t* tiles_touched; /* (N)                                          */
    uint8_t* clamped;        /* (N)   bit ch set <=> colour channel clamped  */
    uint32_t* block_sums;    /* (ceil(N/256)+1) per-block sums of tiles_touched (global-sort path) */
    uint32_t* block_offs;    /* (ceil(N/256))   per-block duplicate offsets drawn from num_rendered
                              * with one atomic per block (emission order is free: the default
                              * sort orders ties by Gaussian id explicitly)                     */
    uint32_t* num_rendered;  /* (1)   D = sum tiles_touched                  */
} gdr_geom;

/* Binning state (upstream "binningBuffer"): D (key,value) pairs, double-buffered. */
typedef struct gdr_binning {
    uint64_t* keys[2];   /* (D) each.  Direct tile binning (default): keys[0] holds ONE packed word per entry of the partitioned
                          * list, id << 32 | float_bits(depth), which the per-tile depth sort consumes; keys[1] is unused.  Radix
                          * partition / global sort: the (tile << 32) | float_bits(depth) keys, double-buffered.  The sorted
                          * 64-bit keys of the reference are never materialised on the default path (the parity tests rebuild
                          * them from ranges + sorted ids + depths) */
    uint32_t* values[2]; /* (D) Gaussian index                              */
    uint32_t* hist;      /* radix-sort scratch                              */
    int32_t sorted;      /* which of the two buffers holds the sorted list  */
    int32_t global_sort; /* !=0: one global LSD radix sort over all key bits instead of the default
                          * tile partition + per-tile LDS depth sort (same result; A/B and tests) */
    uint32_t* scratch32; /* (2*D) depth-key ping-pong for tile lists that do not fit in LDS */
    /* Segments of long tile lists (K7 walks the segments of one tile in parallel workgroups):
     * a tile whose sorted list is longer than seg_len entries is cut every seg_len entries;
     * K6 saves the per-pixel compositing state at every cut and at the end of the list. */
    uint32_t* seg_extra; /* (seg_cap,2) (tile, segment) of every segment but the last of its tile */
    uint32_t* seg_count; /* (4) rows of seg_extra, state slots, "deep forward" flag (few busy tiles), - */
    float* seg_state;    /* (2*seg_cap, 10, 256) K6 -> K7: per pixel of the tile, in front of each cut and at
                          * the end of the list: T, colour x3, depth, alpha sums (gdr); T, colour x3,
                          * normal x3, depth, M1, M2 (gsr)                                          */
    int32_t seg_len;     /* entries per segment (multiple of 256); 0 = lists are never cut.  gdr_binning_carve sets
                          * GDR_DEFAULT_SEG_LEN and sizes the tables for it; a caller may RAISE it or set 0 afterwards */
    int32_t seg_cap;     /* D / seg_len + 1                                                         */
    int32_t deep_max_busy; /* K6 renders the CUT tiles with 16 instead of 64 pixels per wave ("deep" forward, 4 workgroups
                          * per tile) when at most this many tiles hold >= 64 entries AND their lists average >= 2560
                          * entries — an object in front of an empty background leaves most CUs with one workgroup
                          * walking a long list.  gdr_binning_carve sets GDR_DEFAULT_DEEP_MAX_BUSY; 0 = never. */
    int32_t deep_min_mean; /* ... and their lists average at least this many entries; <= 0 = the library default (2560) */
    const uint32_t* d_dev; /* NULL (gdr_binning_carve): the D passed to the binning entry points is the duplicate count,
                          * read back from geom->num_rendered by the caller.  Non-NULL = a DEVICE-SIZED call: the binning
                          * kernels read the count from this device word (geom->num_rendered) themselves and the D passed
                          * to gdr_binning_carve / gdr_binning_forward* is only the CAPACITY the buffers were carved for —
                          * the caller can enqueue binning + K6 behind K1 without waiting for K1, and compares the count
                          * with the capacity afterwards.  A count above the capacity is clamped (nothing is written out
                          * of bounds; the images are then incomplete and the view must be repeated with enough room). */
    /* Launch-size feedback between calls that render the same kind of scene (all optional, results never depend on it):
     * the binning stage reports what it found, the caller passes it back into the next call of that shape. */
    uint32_t* stats_out; /* NULL, or 4 device-writable words (pinned host memory works): {tiles in the tile sort's long
                          * class (> 4096 entries), tiles in its medium class (2049..4096), "deep forward" flag, busy tiles} */
    int32_t hint_long;   /* > 0: workgroups for the tile sort's long class; 0 = sized for the worst case; < 0: the long class
                          * (16 waves, 144 KB of LDS per workgroup) is not launched, the medium class takes any longer list
                          * through its global-memory bucket pass (same result, slower for such a list).  Every value >= 1 */
    int32_t hint_medium; /*      is correct (the classes walk their tiles with a grid stride), a wrong one only costs time   */
    int32_t hint_no_deep;/* != 0: the deep forward is not launched; K6 renders every tile the standard way (same images)     */
    int32_t grad_rec_cleared; /* != 0: the caller has already zero-filled the gradient record it passes to the K7 entry points
                          * for this view (e.g. early, on a side stream, while the forward still runs): they skip their own
                          * clear of N*64 bytes */
    uint32_t* tile_hist; /* direct tile binning (gdr_binning_carve_for): the (hist_width rows x hist_tiles columns) count matrix of
                          * the counting sort on the tile id + one row of tile totals; NULL (gdr_binning_carve, or an image
                          * of more than 16384 tiles): the radix partition on the tile bits is used instead (same lists) */
    int32_t hist_width;  /* rows of tile_hist = workgroups of the count / scatter kernels, <= 256 */
    int32_t hist_tiles;  /* columns of tile_hist = the tile count it was carved for, rounded up to 64; a binning call on an image
                          * with more tiles than that uses the radix partition instead (same lists) — was reserved1 until v14 */
    int32_t k7_class;    /* v16: duplicates per Gaussian of the view as a quarter-octave class (1..63), set by gdr_forward_view(s)
                          * from the exact count; 0 = unknown.  Part of the key under which the library remembers which K7
                          * serves a scene (gdr_k7_tune_*) — nothing else reads it */
    int32_t reserved2;
} gdr_binning;

/* Image state (upstream "imgBuffer"). */
typedef struct gdr_image {
    uint32_t* ranges;    /* (tiles,2) [first,last) into the sorted list     */
    uint32_t* n_contrib; /* (H*W) 1-based index of the last contributor     */
    float* final_T;      /* (H*W) transmittance after the last contributor  */
    uint32_t* tile_order; /* (tiles) tile ids, longest sorted list first (launch order of K6/K7) */
    uint32_t* seg_base;   /* (tiles) first seg_state slot of a tile whose list is cut, 0xFFFFFFFF otherwise */
} gdr_image;

typedef struct gdr_outputs {
    float* color;   /* (3,H,W) */
    float* depth;   /* (1,H,W) sum_i w_i z_i (not normalised) */
    float* alpha;   /* (1,H,W) sum_i w_i = 1 - T_final        */
    int32_t* radii; /* (N)     0 = culled                     */
} gdr_outputs;

typedef struct gdr_grad_inputs {
    const float* dL_dcolor; /* (3,H,W) */
    const float* dL_ddepth; /* (H,W) or NULL (= zeros) */
    const float* dL_dalpha; /* (H,W) or NULL (= zeros) */
} gdr_grad_inputs;

/* Gradient outputs; every buffer is fully written (zeros for culled Gaussians), the
 * caller does not need to clear anything.  dL_dmeans2D is (N,4): columns 0-1 the signed
 * NDC-space gradient, columns 2-3 the sum over pixels of its absolute per-pixel terms
 * (network.py:876-878).  scratch: (N*16) floats — one 64-byte gradient record per Gaussian
 * that K7 accumulates into (mean2D, conic, depth, colour, opacity partials). */
typedef struct gdr_grad_outputs {
    float* dL_dmeans3D;   /* (N,3) */
    float* dL_dmeans2D;   /* (N,4) */
    float* dL_dshs;       /* (N,M,3) or NULL when colors_precomp was used */
    float* dL_dcolors;    /* (N,3)  or NULL when shs was used (then taken from scratch) */
    float* dL_dopacities; /* (N)   */
    float* dL_dscales;    /* (N,3) or NULL when cov3D_precomp was used */
    float* dL_drotations; /* (N,4) or NULL when cov3D_precomp was used */
    float* dL_dcov3D;     /* (N,6) or NULL when scales/rotations were used */
    float* scratch;       /* (N*16) floats, 64-byte aligned, contents undefined on return */
    int32_t accumulate;   /* !=0: ADD into the output buffers (sum over views of one Gaussian
                           * set, network.py:826-838) instead of overwriting them */
    int32_t reserved;
} gdr_grad_outputs;

/* ---- sizes and carving ---------------------------------------------------------- */
int gdr_abi_version(void);
const char* gdr_last_error(void); /* thread-local, host string */
/* "release" for the product build.  Anything else marks a measurement / experimental build of the library (compiled with
 * -DGDR_BUILD_TAG=...): the Python loader refuses to load such a library unless GDR_ALLOW_EXPERIMENTAL_LIB=1 is set. */
const char* gdr_build_tag(void);

size_t gdr_geom_bytes(int32_t N);
size_t gdr_binning_bytes(uint64_t D);
size_t gdr_image_bytes(int32_t H, int32_t W);
/* base must be 256-byte aligned and at least gdr_*_bytes(...) long. */
int gdr_geom_carve(void* base, int32_t N, gdr_geom* out);
int gdr_binning_carve(void* base, uint64_t D, gdr_binning* out);
/* The same for a known scene shape: the cut-list tables (seg_extra, seg_state: 2 * (D / seg_len + 1) slots of 10 KB —
 * 80 bytes per duplicate at seg_len = 256, 40 at 512) sized for the segment length the caller is going to use (seg_len =
 * a multiple of 256, or 0: lists are never cut, no tables; out->seg_len = seg_len), and — given N Gaussians and the
 * image's tile count — the count matrix of the direct tile binning (min(256, N / 1024) + 1 rows of tiles words; tiles = 0 or
 * > 16384: none, the radix partition is used). */
size_t gdr_binning_bytes_for(uint64_t D, int32_t seg_len, int32_t N, int32_t tiles);
int gdr_binning_carve_for(void* base, uint64_t D, int32_t seg_len, int32_t N, int32_t tiles, gdr_binning* out);
int gdr_image_carve(void* base, int32_t H, int32_t W, gdr_image* out);

/* ---- forward ---------------------------------------------------------------------
 * gdr_preprocess_forward + gdr_render_forward (or gdr_forward = both) replace the forward half of the extension,
 * `_C.rasterize_gaussians`, reached from `rasterizer(means3D=…, …)` at /root/reference/lightning/renderer.py:250-259 and
 * /root/reference/lightning/point_decoder/layers/gaussian_renderer.py:88-108.
 * Stage 1 (K1 + K2): per-Gaussian projection, EWA covariance, SH colour, tile rect,
 * and the scan total.  radii is written here.  If num_rendered_host != NULL the
 * stream is synchronised and D is returned through it (the one host read the
 * upstream extension also performs); otherwise D stays in geom->num_rendered. */
int gdr_preprocess_forward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                           int32_t* radii, uint32_t* num_rendered_host, void* stream);

/* Stage 2 (K3..K6): duplicate with keys, radix sort on (tile, depth), tile ranges,
 * per-tile alpha-composited render.  D is the capacity of `bin` and must be >= the
 * value stage 1 produced.  bin->sorted is set. */
int gdr_render_forward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                       gdr_binning* bin, const gdr_image* img, uint64_t D,
                       const gdr_outputs* out, void* stream);

/* The two halves of stage 2, for callers that overlap the binning of one view with the compositing of another on
 * separate streams (binning is latency-bound with few workgroups, compositing is VALU-bound):
 * gdr_binning_forward = K3..K5 + tile order + tile sort (also valid for the surfel geometry of gsr.h),
 * gdr_composite_forward = K6.  gdr_render_forward is exactly one after the other on one stream.
 * The binning stage builds the cut-list tables of `bin` (seg_extra, seg_count, img->seg_base); K6 fills seg_state
 * and every backward entry point below reads the seg_state of the LATEST compositing call on that (bin, img) pair. */
int gdr_binning_forward(const gdr_settings* s, int32_t N, const gdr_geom* geom, gdr_binning* bin, const gdr_image* img,
                        uint64_t D, const int32_t* radii, void* stream);
int gdr_composite_forward(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                          const gdr_outputs* out, void* stream);

/* K6 / K7 with the image loss folded in ("clamp + MSE + per-view loss reduction folded into the K6/K7
 * epilogue/prologue"; renderer.py:261 clamp, loss.py:37-38 MSE, plus the depth / alpha means of the measurement loss):
 *   *loss += mean_{c,p}(clamp(color,0,1) - target)^2 + w_depth mean(depth) + w_alpha mean(alpha)
 * gdr_composite_forward_loss = K6 that also accumulates the view's loss (one atomic per tile; the caller zeroes *loss);
 * the images are still written.  gdr_render_backward_loss = K7 whose per-pixel upstream gradients are computed from
 * `color` (what K6 wrote) and `target` times the upstream scalar *g (device) instead of being read — no loss kernels,
 * no dL/dimage tensors.  target, color: (3,H,W). */
int gdr_composite_forward_loss(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                               const gdr_outputs* out, const float* target, float w_depth, float w_alpha, float* loss,
                               void* stream);
int gdr_render_backward_loss(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                             const gdr_image* img, const float* color, const float* target, float w_depth, float w_alpha,
                             const float* g, float* grad_rec, void* stream);

/* K6 of V <= GDR_MAX_VIEWS views of one image size in ONE launch (v14; SURVEY section 7 step 5 "grid.z = view"): the
 * workgroups of all views in one grid (interleave != 0: the same launch-order slot of consecutive views next to each other).
 * Every view must have finished its binning stage on `stream`'s dependencies.  loss_mode 0 = gdr_composite_forward,
 * 1 = gdr_composite_forward_loss (losses[v] accumulated: the caller zeroes the V floats), 2 = gdr_composite_forward_lossgrad
 * (outs[v].color receives d loss / d colour, depth / alpha are not written and may be NULL). */
int gdr_composite_forward_views(int32_t V, const gdr_settings* s, const gdr_geom* geoms, const gdr_binning* bins,
                                const gdr_image* imgs, const gdr_outputs* outs, int32_t loss_mode,
                                const float* const* targets, float w_depth, float w_alpha, float go_scale, float* losses,
                                int32_t interleave, void* stream);

/* Stages 1+2 with a caller-provided binning capacity D_cap.  Synchronises once to
 * read D; returns GDR_ERR_WORKSPACE (and *num_rendered_host = required D) if
 * D > D_cap, in which case only stage 1 has run. */
int gdr_forward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                gdr_binning* bin, const gdr_image* img, uint64_t D_cap, const gdr_outputs* out,
                uint32_t* num_rendered_host, void* stream);

/* ---- one forward call per view (v14) ------------------------------------------------------------------------------
 * The reference's extension does its whole forward in ONE native call (`_C.rasterize_gaussians`, reached from
 * /root/reference/lightning/renderer.py:250-259 inside the per-view loops of network.py:827-838).  gdr_forward_view is
 * that call: it carves ONE caller allocation into the geometry / image / binning state, runs K1, starts the duplicate
 * count on its way to pinned host memory, enqueues binning + K6 sized by the device counter, waits for the count and
 * compares it with the capacity — everything the Python boundary did with ~20 ABI calls until round 3.
 *   gdr_view_plan_for   sizes the allocation: capacity = exact_D if given, else 1.5 x the largest recent duplicate count
 *                       per Gaussian of this scene shape (device, power-of-two bucket of N, H, W) x N + 4096 — the library
 *                       keeps that small per-shape history (process-wide, mutex-guarded; gdr_view_history_reset clears it)
 *                       together with the launch-size feedback of gdr_binning.stats_out / hint_*.  No history yet:
 *                       plan.have_binning = 0 — the forward then stops behind K1 with GDR_ERR_WORKSPACE and state.D set;
 *   gdr_forward_view    returns GDR_OK with `state` filled (the structs every backward entry point takes, and D), or
 *                       GDR_ERR_WORKSPACE with state.D = the count: plan again with exact_D = state.D, allocate, call again
 *                       (the first call of a shape, or a scene that grew past the slack: nothing was written out of bounds).
 * opts (NULL = defaults): test / A-B overrides, each -1 / 0 = the library's policy.  same (NULL = none): up to 8 buffer
 * pairs compared bit for bit next to K1 (see gdr_words_differ; state.differ != 0 if any differs) — the equality check of a
 * render group rides on the count's copy.  Thread-safe for distinct workspaces. */
typedef struct gdr_view_plan {
    uint64_t capacity;     /* duplicates the binning state is carved for */
    uint64_t bytes;        /* size of the one allocation (256-byte aligned base) */
    int32_t seg_len;       /* gdr_binning.seg_len of the carve */
    int32_t deferred;      /* != 0: capacity is a guess -> device-sized call */
    int32_t have_binning;  /* 0: no history and no exact_D: geometry + image state only */
    int32_t reserved;
} gdr_view_plan;
typedef struct gdr_view_opts {
    int32_t seg_len;         /* >= 0: segment length of cut lists (0 = never cut); -1: policy (256; 512 on busy 800x800 images) */
    int32_t deep_max_busy;   /* >= 0: gdr_binning.deep_max_busy; -1: default */
    int32_t deep_min_mean;   /* >= 0: gdr_binning.deep_min_mean; -1: default */
    int32_t global_sort;     /* != 0: one global radix sort (tested fallback) */
    int32_t radix_partition; /* != 0: radix partition on the tile bits instead of the direct tile binning (tested fallback) */
    int32_t no_hints;        /* != 0: no launch-size feedback */
} gdr_view_opts;
#define GDR_SAME_AS_MAX 8
typedef struct gdr_same_as {
    int32_t n;               /* <= GDR_SAME_AS_MAX (4 until v15; a caller that re-activates all five inputs per call has 5 pairs) */
    int32_t reserved;
    const void* a[GDR_SAME_AS_MAX]; const void* b[GDR_SAME_AS_MAX]; uint64_t n_bytes[GDR_SAME_AS_MAX];
} gdr_same_as;
typedef struct gdr_view_state {
    gdr_geom geom; gdr_binning bin; gdr_image img;
    uint64_t D;              /* duplicates of the view (the reference's num_rendered) */
    uint32_t differ;         /* != 0: a same-as pair differed */
    uint32_t reserved;
} gdr_view_state;
int gdr_view_plan_for(int32_t N, int32_t H, int32_t W, int32_t surfel, uint64_t exact_D, const gdr_view_opts* opts,
                      gdr_view_plan* plan);
int gdr_forward_view(const gdr_settings* s, const gdr_inputs* in, const gdr_view_plan* plan, void* workspace,
                     const gdr_view_opts* opts, const gdr_same_as* same, const gdr_outputs* out, gdr_view_state* state,
                     void* stream);
/* Does the view about to be rendered repeat an earlier view of the same Gaussians (v16)?  The reference renders the first
 * n_views_sel views of the coarse Gaussians twice per sample — /root/reference/lightning/network.py:827-838, then again inside
 * the function `vjp` differentiates at :848-856, with the same c2w / bg_color — and a render group (viewgroup.py) that knows
 * the Gaussians are the same only has to know that the 12 settings fields are, too, to hand out the first forward's results
 * instead of running K1, binning and K6 again.  The host scalars of `s` are compared here on the host; bg / viewmatrix /
 * projmatrix / campos (device tensors the caller rebuilds per call: MiniCam at network.py:851) are compared bit for bit on
 * the device against up to GDR_REUSE_MAX candidates in one small launch, together with the `same` pairs of the render group
 * (gdr_same_as: the activated tensors of this call against the group's).  BLOCKING: *match = index of the first candidate
 * whose 12 fields equal those of `s` (or -1), *differ != 0 if a `same` pair differs — one launch, one pooled pinned copy,
 * one event wait.  scratch: GDR_REUSE_MAX + 1 device words (the library allocates no device memory). */
#define GDR_REUSE_MAX 32
int gdr_view_reuse_probe(const gdr_settings* s, int32_t n, const gdr_settings* candidates, const gdr_same_as* same,
                         uint32_t* scratch, int32_t* match, uint32_t* differ, void* stream);

/* The same for ALL views of one Gaussian set (the forward of the multi-view node, v14): V <= GDR_MAX_NODE_VIEWS views of one
 * image size; K1 in launches of <= GDR_MAX_VIEWS views (inputs read once each), then every view's chain binning -> K6 on
 * streams[v % n_streams] (streams[0] = the caller's: it is made to wait for the others before the call returns), the V
 * duplicate counts read back with ONE pooled pinned copy after everything is enqueued.  ONE allocation for all views
 * (gdr_views_plan_for: V x the per-view state + the packed counters).  loss_mode / targets / w_* / go_scale / losses as in
 * gdr_composite_forward_views.  states: V structs, filled (cov3D of every view points at view 0's copy).  Returns GDR_OK, or
 * GDR_ERR_WORKSPACE with every states[v].D set: plan again with exact_D = the largest of them and call again. */
#define GDR_MAX_NODE_VIEWS 256
typedef struct gdr_views_plan {
    gdr_view_plan view;      /* the per-view plan (capacity, seg_len, deferred, have_binning) */
    uint64_t bytes_view;     /* stride between the views' state in the allocation */
    uint64_t bytes_shared;
    uint64_t bytes;          /* V x bytes_view + bytes_shared */
    int32_t V;
    int32_t reserved;
} gdr_views_plan;
int gdr_views_plan_for(int32_t V, int32_t N, int32_t H, int32_t W, uint64_t exact_D, const gdr_view_opts* opts,
                       gdr_views_plan* plan);
int gdr_forward_views(int32_t V, const gdr_settings* s, const gdr_inputs* in, const gdr_views_plan* plan, void* workspace,
                      const gdr_view_opts* opts, const gdr_outputs* outs, int32_t loss_mode, const float* const* targets,
                      float w_depth, float w_alpha, float go_scale, float* losses, void* const* streams, int32_t n_streams,
                      gdr_view_state* states);
void gdr_view_history_reset(void);
/* the history behind gdr_view_plan_for, per scene shape: the decaying maximum of duplicates per Gaussian (0 = none yet).
 * _set seeds or overrides it — a caller that knows its scene statistics skips the read-back of a shape's first call; the
 * tests force an overflow with a tiny value (the call then answers GDR_ERR_WORKSPACE and is repeated exactly sized). */
/* the launch-size report words of a shape's previous call (row = view of the call, < 64): {tiles in the tile sort's long class,
 * in its medium class, "deep forward applied", busy tiles}, 0xFFFFFFFF = nothing reported yet; set != 0 overwrites them (tests:
 * wrong hints must only cost time) */
int gdr_view_history_report(int32_t N, int32_t H, int32_t W, int32_t surfel, int32_t row, uint32_t* words, int32_t set);
double gdr_view_history_get(int32_t N, int32_t H, int32_t W, int32_t surfel);
void gdr_view_history_set(int32_t N, int32_t H, int32_t W, int32_t surfel, double duplicates_per_gaussian);
/* Which K7 serves a scene shape (v15).  K7 publishes one float-atomic record line per (list entry, 4x4 pixel block) hit and
 * the device retires ~21 G such lines per second whatever they carry: wherever Gaussians span several blocks that rate, not
 * the arithmetic, bounds K7.  A second kernel lets the two rows of an 8x4 area walk the union of their lists and publish
 * ONE line where that saves enough lines per extra iteration — faster for Gaussians of 10+ pixels spread over the image,
 * slower for sub-pixel Gaussians and object-like scenes.  The gradients are the same sums in another order.  Per (device,
 * N bucket, duplicates-per-Gaussian class of the views — gdr_binning.k7_class —, image size, views per launch, entry kind)
 * the library times both kernels ONCE (events around four consecutive launches, rows / pairs / rows / pairs, after the key's
 * first 8 launches; never on a stream under graph capture) and keeps the faster (v16; until v15 the round was repeated every
 * 256 launches, so the variant could change mid-run).  v17: ONE confirmation — launches 64..67 of the key time both kernels
 * once more and the first pick is overturned only if it loses that round by > 5 % (a wrong pick from one noisy timing cost
 * 3 % at C4, 15 % at C2 for the life of the process); after it the choice is final.  A scene that drifts moves to another
 * class, i.e. another key with rounds of its own.
 * kind: 0 gdr_backward / gdr_render_backward(_views), 1 the _loss entries, 2 the mean2D-only entries, 3 the 2DGS K7s
 * (gsr_backward / gsr_render_backward(_views): 16 + 4 totals per pair, two record lines), 4 gdr_render_backward_mean2d_loss.
 * _override: -1 measure and choose (default), 0 rows only, 1 row pairs always (tests, A/B, bit-reproducible runs: the Python
 * loader maps GDR_K7_PAIRS=0/1 onto it).  _get: the choice (-1 = not decided yet: rows serve meanwhile) and the round's times in
 * microseconds of the most recently used key of that (N bucket, image size, V, kind) — error if there is none.
 * gdr_view_history_reset restarts the choices. */
void gdr_k7_tune_override(int32_t mode);
int gdr_k7_tune_get(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind, int32_t* chosen, float* us_rows, float* us_pairs);
/* v17: both rounds of that key — *rounds_done 0 (first round still to come / running), 1 (decided once), 2 (confirmed: final);
 * us4 = {rows, pairs} of the first round, {rows, pairs} of the confirmation round, microseconds, 0 = not measured yet. */
int gdr_k7_tune_get_rounds(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind, int32_t* chosen, int32_t* rounds_done,
                           float* us4);
/* test hook: make `variant` the FIRST pick of that key (as if its first round had chosen it) — the confirmation round must
 * then correct it if it is the wrong one.  Error once the key's choice is final. */
int gdr_k7_tune_force_first(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind, int32_t variant);

/* ---- backward (K7 + K8/K9) --------------------------------------------------------
 * Replaces `_C.rasterize_gaussians_backward`, reached through autograd from the losses on the render outputs
 * (/root/reference/lightning/network.py:746-752, 836, 854, 971) and through `vjp` w.r.t. the (N,4) means2D carrier
 * (/root/reference/lightning/network.py:865-878). */
int gdr_backward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                 const gdr_binning* bin, const gdr_image* img, uint64_t D,
                 const int32_t* radii, const gdr_grad_inputs* gin, const gdr_grad_outputs* gout,
                 void* stream);

/* ---- multi-view entry points (one Gaussian set, V <= GDR_MAX_VIEWS views of one image size) ----
 * The callers render all target views of one Gaussian set back to back
 * (/root/reference/lightning/network.py:826-838, 848-856, 964-972).  These variants do the
 * view-independent work once: the per-Gaussian inputs are read once for all V views, cov3D is
 * written once (to geoms[0].cov3D — every geoms[v].cov3D is ignored), and the backward sums the
 * V per-view partial gradients in registers before writing each output once.
 * s, geoms: arrays of V structs; radii, grad_recs: arrays of V device pointers (host arrays).
 * Requires in->shs, in->scales, in->rotations (no colors_precomp / cov3D_precomp).
 * Sequence: gdr_preprocess_forward_views; read the V values geoms[v].num_rendered (or, for
 *           device-sized calls — gdr_binning.d_dev — only start copying them to the host);
 *           per view gdr_render_forward; (device-sized: compare counts and capacities, repeat
 *           the views that did not fit); ...; per view gdr_render_backward (K7 into its own
 *           N*16-float record); gdr_preprocess_backward_views. */
#define GDR_MAX_VIEWS 8
int gdr_preprocess_forward_views(int32_t V, const gdr_settings* s, const gdr_inputs* in,
                                 const gdr_geom* geoms, int32_t* const* radii, void* stream);
int gdr_render_backward(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                        const gdr_image* img, const gdr_grad_inputs* gin, float* grad_rec,
                        void* stream);
int gdr_preprocess_backward_views(int32_t V, const gdr_settings* s, const gdr_inputs* in,
                                  const gdr_geom* geoms, const int32_t* const* radii,
                                  float* const* grad_recs, const gdr_grad_outputs* gout, void* stream);

/* K7 of V <= GDR_MAX_VIEWS views of one image size in ONE launch (v14).  The views of a node are independent given the
 * Gaussians; one launch over V x (segments + tiles) workgroups keeps the chip full across the views' kernel tails where V
 * launches on side streams leave it to the dispatcher (reference-scale scenes: one view is 1-2.5 k workgroups for 1280
 * resident slots).  Arrays of V structs / V device pointers (host arrays); every view writes its own N*16-float record
 * (cleared here unless bins[v].grad_rec_cleared).  interleave != 0: consecutive workgroups take the same slot of
 * consecutive views (all views' longest work items first); 0: the views follow one another in the grid (a view's
 * records — 128 MB at 2 M Gaussians — are not evicted from the caches by the other views' gathers).
 * _loss_views: g = device array of V upstream scalars (see gdr_render_backward_loss); colors / targets: V x (3,H,W).
 * _mean2d_views: every view ADDS into the one (N,4) buffer (see gdr_render_backward_mean2d). */
int gdr_render_backward_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms, const gdr_binning* bins,
                              const gdr_image* imgs, const gdr_grad_inputs* gins, float* const* grad_recs,
                              int32_t interleave, void* stream);
int gdr_render_backward_loss_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms,
                                   const gdr_binning* bins, const gdr_image* imgs, const float* const* colors,
                                   const float* const* targets, float w_depth, float w_alpha, const float* g,
                                   float* const* grad_recs, int32_t interleave, void* stream);
int gdr_render_backward_mean2d_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms,
                                     const gdr_binning* bins, const gdr_image* imgs, const float* const* dL_dcolors,
                                     float* dL_dmean2D, int32_t interleave, void* stream);

/* ---- screen-space gradient only ----------------------------------------------
 * The densification step differentiates an image loss w.r.t. the (N,4) means2D carrier of several
 * views and uses nothing else (/root/reference/lightning/network.py:865-878).  This K7 variant
 * ADDS one view's dL/dmean2D (x, y signed; |x|, |y| summed per pixel) into dL_dmean2D (N,4), which the
 * caller zeroes once before the first view; no per-view records, no K8/K9. */
int gdr_render_backward_mean2d(const gdr_settings* s, int32_t N, const gdr_geom* geom,
                               const gdr_binning* bin, const gdr_image* img, const float* dL_dcolor,
                               float* dL_dmean2D, void* stream);

/* the same with the image MSE folded into the prologue (see gdr_render_backward_loss): the per-pixel upstream gradient
 * is *g * 2/(3 H W) * (clamp(color) - target) inside [0,1], 0 outside; no dL/dimage tensor */
int gdr_render_backward_mean2d_loss(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                                    const gdr_image* img, const float* color, const float* target, const float* g,
                                    float* dL_dmean2D, void* stream);

/* The abs-grad-only path end to end (replaces the vjp of /root/reference/lightning/network.py:843-878 and the
 * torch.topk of :876-893):
 *   gdr_composite_forward_lossgrad   K6 with the MSE of one view folded in and NO image output: loss += mean_{c,p}
 *                                    (clamp(color) - target)^2 (device float, caller zeroes), and dL_dcolor (3,H,W) :=
 *                                    go_scale * d loss / d color per pixel — the only per-pixel array it writes besides the
 *                                    backward state (final T, contributor count).  go_scale = the upstream scalar known on
 *                                    the host (1 / views for a mean over the views).
 *   gdr_render_backward_mean2d       consumes that dL_dcolor (above).
 *   gdr_topk_absgrad                 selection on score_i = ||dL_dmean2D[i, 2:4]||_2: mask[i] = 1 for the k largest
 *                                    scores among the candidates (candidates == NULL: all N; fewer than k candidates —
 *                                    counted on the device — : every candidate with score >= 0, the reference's
 *                                    `gradient_point >= 0` branch, which drops NaN; otherwise NaN ranks above every
 *                                    number, as in torch.topk); indices (k, may be
 *                                    NULL) receives the selected ids in no particular order (the reference only builds
 *                                    a mask from them).  Radix select, no sort; ties at the threshold are cut arbitrarily,
 *                                    as torch.topk leaves them.  workspace: gdr_topk_workspace_bytes() bytes. */
int gdr_composite_forward_lossgrad(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                                   const float* target, float go_scale, float* loss, float* dL_dcolor, void* stream);
size_t gdr_topk_workspace_bytes(void);
int gdr_topk_absgrad(int32_t N, const float* dL_dmean2D, const uint8_t* candidates, int32_t k, void* workspace,
                     uint8_t* mask, int32_t* indices, void* stream);

/* ---- fused image loss either side of the path ---------------------------------
 * loss += mean_{c,p}(clamp(color,0,1) - target)^2 + w_depth mean(depth) + w_alpha mean(alpha) for ONE view
 * (renderer.py:261 clamp, loss.py:37-38 MSE; depth/alpha terms: the measurement loss of SURVEY §8d).
 * color, target: (3,H,W); depth, alpha: (H,W); loss: device float the caller zeroes (accumulated with one
 * atomic per workgroup).  backward: g = device pointer to the upstream scalar (NULL = 1). */
int gdr_view_loss_forward(const float* color, const float* depth, const float* alpha, const float* target,
                          int32_t H, int32_t W, float w_depth, float w_alpha, float* loss, void* stream);
int gdr_view_loss_backward(const float* color, const float* target, int32_t H, int32_t W, float w_depth,
                           float w_alpha, const float* g, float* dL_dcolor, float* dL_ddepth, float* dL_dalpha,
                           void* stream);

/* ---- host-boundary helper: *flag |= 1 if the n_bytes (a multiple of 4; a, b 16-byte aligned) at a and b differ in any
 * 32-bit word.  Used by the Python boundary to verify that two calls of one render group were handed the same activated
 * tensors (see generativedensification_amd/viewgroup.py); one read of both buffers, no host synchronisation. */
int gdr_words_differ(const void* a, const void* b, uint64_t n_bytes, uint32_t* flag, void* stream);
/* ... for up to GDR_SAME_AS_MAX buffer pairs in one launch (v13; 4 until v15) */
int gdr_words_differ_multi(int32_t n, const void* const* a, const void* const* b, const uint64_t* n_bytes, uint32_t* flag,
                           void* stream);

/* ---- host-boundary helper: n_bytes from device memory to PINNED host memory behind the work queued on `stream`, with an
 * event the library pools (v13).  *ticket identifies the copy; gdr_host_copy_wait blocks until it has landed and releases
 * the ticket (every ticket must be waited for exactly once).  The upstream extension reads `num_rendered` back with a
 * blocking copy inside rasterize_gaussians; the Python boundary here reads the duplicate count of a device-sized call this
 * way after everything else of the call is enqueued — two C calls instead of ~20 us of tensor / event objects. */
int gdr_host_copy_begin(void* dst_pinned, const void* src_dev, uint64_t n_bytes, void* stream, void** ticket);
int gdr_host_copy_wait(void* ticket);

/* Zero-fill n_bytes of device memory behind the work queued on `stream` (v13): what the caller of the K7 entry points does to
 * the gradient records when it sets gdr_binning.grad_rec_cleared — one C call on a raw stream handle instead of a stream
 * context + tensor op per view on the Python side. */
int gdr_clear_async(void* dst, uint64_t n_bytes, void* stream);

/* ---- K10: visibility mask (upstream markVisible; unused by the reference) -------- */
int gdr_mark_visible(int32_t N, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present, void* stream);

/* ---- opt-in per-kernel timing (measurement aid, bench.py `roofline`) -----------------
 * When enabled every kernel launch is bracketed by two HIP events recorded on the SAME
 * stream the kernel runs on; gdr_profile_collect waits for them and returns, per kernel id
 * (0 <= id < gdr_kernel_count(), names from gdr_kernel_name), the summed elapsed ms and the
 * launch count since the last reset.  Process-wide switch; off by default (no events). */
int gdr_profile_enable(int on);
int gdr_profile_collect(double* ms_total, uint64_t* launches, int32_t n, int32_t reset);
int gdr_kernel_count(void);
const char* gdr_kernel_name(int32_t id);

#ifdef __cplusplus
}
#endif
#endif /* GDR_H */
