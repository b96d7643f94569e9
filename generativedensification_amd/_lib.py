"""ctypes binding of libgdr_hip.so (C ABI: include/gdr.h).

The HIP library is the product; there is NO CPU fallback: if the shared object is
missing or a call fails, we raise.  `torch` must be imported before the library is
loaded so that both share one HIP runtime (libamdhip64.so.7, matched by SONAME).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime first — see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GDR_LIB_PATH: developer override (A/B of two builds of the library); the product default is the in-tree library.  A
# library whose gdr_build_tag() is not "release" (a measurement build) is refused unless GDR_ALLOW_EXPERIMENTAL_LIB=1.
LIB_PATH = os.environ.get("GDR_LIB_PATH") or os.path.join(_HERE, "lib", "libgdr_hip.so")

GDR_OK = 0
GDR_IN_RAW_OPACITY, GDR_IN_RAW_SCALES, GDR_IN_RAW_ROTATIONS, GDR_IN_NO_DEPTH_TO_MEAN = 1, 2, 4, 8
GDR_MAX_VIEWS = 8
GDR_MAX_NODE_VIEWS = 256
ABI_VERSION = 17
GDR_SAME_AS_MAX, GDR_REUSE_MAX = 8, 32
GDR_DEFAULT_SEG_LEN = 256
GDR_ERR_WORKSPACE = -4


class GdrSettings(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("bg", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p)]


class GdrInputs(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("means3D", C.c_void_p), ("opacities", C.c_void_p),
                ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class GdrGeom(C.Structure):
    _fields_ = [("depths", C.c_void_p), ("rec", C.c_void_p), ("cov3D", C.c_void_p), ("rect", C.c_void_p),
                ("tiles_touched", C.c_void_p), ("clamped", C.c_void_p), ("block_sums", C.c_void_p),
                ("block_offs", C.c_void_p),
                ("num_rendered", C.c_void_p)]


class GdrBinning(C.Structure):
    _fields_ = [("keys", C.c_void_p * 2), ("values", C.c_void_p * 2), ("hist", C.c_void_p),
                ("sorted", C.c_int32), ("global_sort", C.c_int32), ("scratch32", C.c_void_p),
                ("seg_extra", C.c_void_p), ("seg_count", C.c_void_p), ("seg_state", C.c_void_p),
                ("seg_len", C.c_int32), ("seg_cap", C.c_int32), ("deep_max_busy", C.c_int32), ("deep_min_mean", C.c_int32),
                ("d_dev", C.c_void_p), ("stats_out", C.c_void_p), ("hint_long", C.c_int32), ("hint_medium", C.c_int32),
                ("hint_no_deep", C.c_int32), ("grad_rec_cleared", C.c_int32), ("tile_hist", C.c_void_p),
                ("hist_width", C.c_int32), ("hist_tiles", C.c_int32), ("k7_class", C.c_int32), ("reserved2", C.c_int32)]


class GdrImage(C.Structure):
    _fields_ = [("ranges", C.c_void_p), ("n_contrib", C.c_void_p), ("final_T", C.c_void_p),
                ("tile_order", C.c_void_p), ("seg_base", C.c_void_p)]


class GdrOutputs(C.Structure):
    _fields_ = [("color", C.c_void_p), ("depth", C.c_void_p), ("alpha", C.c_void_p),
                ("radii", C.c_void_p)]


class GdrGradInputs(C.Structure):
    _fields_ = [("dL_dcolor", C.c_void_p), ("dL_ddepth", C.c_void_p), ("dL_dalpha", C.c_void_p)]


class GdrGradOutputs(C.Structure):
    _fields_ = [("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
                ("dL_dcolors", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p), ("scratch", C.c_void_p),
                ("accumulate", C.c_int32), ("reserved", C.c_int32)]


class GdrViewPlan(C.Structure):
    _fields_ = [("capacity", C.c_uint64), ("bytes", C.c_uint64), ("seg_len", C.c_int32), ("deferred", C.c_int32),
                ("have_binning", C.c_int32), ("reserved", C.c_int32)]


class GdrViewOpts(C.Structure):
    _fields_ = [("seg_len", C.c_int32), ("deep_max_busy", C.c_int32), ("deep_min_mean", C.c_int32),
                ("global_sort", C.c_int32), ("radix_partition", C.c_int32), ("no_hints", C.c_int32)]


class GdrSameAs(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("a", C.c_void_p * GDR_SAME_AS_MAX),
                ("b", C.c_void_p * GDR_SAME_AS_MAX), ("n_bytes", C.c_uint64 * GDR_SAME_AS_MAX)]


class GdrViewState(C.Structure):
    _fields_ = [("geom", GdrGeom), ("bin", GdrBinning), ("img", GdrImage), ("D", C.c_uint64), ("differ", C.c_uint32),
                ("reserved", C.c_uint32)]


class GdrViewsPlan(C.Structure):
    _fields_ = [("view", GdrViewPlan), ("bytes_view", C.c_uint64), ("bytes_shared", C.c_uint64), ("bytes", C.c_uint64),
                ("V", C.c_int32), ("reserved", C.c_int32)]


class GsrInputs(C.Structure):   # include/gsr.h
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("means3D", C.c_void_p), ("opacities", C.c_void_p),
                ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("transMat_precomp", C.c_void_p), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class GsrOutputs(C.Structure):
    _fields_ = [("color", C.c_void_p), ("allmap", C.c_void_p), ("radii", C.c_void_p)]


class GsrGradInputs(C.Structure):
    _fields_ = [("dL_dcolor", C.c_void_p), ("dL_dallmap", C.c_void_p)]


class GsrGradOutputs(C.Structure):
    _fields_ = [("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
                ("dL_dcolors", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("dL_drotations", C.c_void_p), ("dL_dtransMat", C.c_void_p), ("scratch", C.c_void_p),
                ("accumulate", C.c_int32), ("reserved", C.c_int32)]


GSR_REC_FLOATS, GSR_GRAD_FLOATS = 24, 32

# every symbol include/gdr.h and include/gsr.h declare, with its prototype
_PROTOS = {
    "gdr_abi_version": (C.c_int, []),
    "gdr_last_error": (C.c_char_p, []),
    "gdr_geom_bytes": (C.c_size_t, [C.c_int32]),
    "gdr_binning_bytes": (C.c_size_t, [C.c_uint64]),
    "gdr_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gdr_geom_carve": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(GdrGeom)]),
    "gdr_binning_carve": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(GdrBinning)]),
    "gdr_binning_bytes_for": (C.c_size_t, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32]),
    "gdr_binning_carve_for": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GdrBinning)]),
    "gdr_build_tag": (C.c_char_p, []),
    "gdr_words_differ": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "gdr_words_differ_multi": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p,
                                         C.c_void_p]),
    "gdr_host_copy_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gdr_host_copy_wait": (C.c_int, [C.c_void_p]),
    "gdr_clear_async": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "gdr_image_carve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(GdrImage)]),
    "gdr_preprocess_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrInputs), C.POINTER(GdrGeom),
                                         C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
    "gdr_render_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrInputs), C.POINTER(GdrGeom),
                                     C.POINTER(GdrBinning), C.POINTER(GdrImage), C.c_uint64,
                                     C.POINTER(GdrOutputs), C.c_void_p]),
    "gdr_binning_forward": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                      C.POINTER(GdrImage), C.c_uint64, C.c_void_p, C.c_void_p]),
    "gdr_composite_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                        C.POINTER(GdrImage), C.POINTER(GdrOutputs), C.c_void_p]),
    "gdr_composite_forward_loss": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                             C.POINTER(GdrImage), C.POINTER(GdrOutputs), C.c_void_p, C.c_float, C.c_float,
                                             C.c_void_p, C.c_void_p]),
    "gdr_composite_forward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                              C.POINTER(GdrImage), C.POINTER(GdrOutputs), C.c_int32, C.POINTER(C.c_void_p),
                                              C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "gdr_render_backward_loss": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                           C.POINTER(GdrImage), C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "gdr_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrInputs), C.POINTER(GdrGeom),
                              C.POINTER(GdrBinning), C.POINTER(GdrImage), C.c_uint64,
                              C.POINTER(GdrOutputs), C.POINTER(C.c_uint32), C.c_void_p]),
    "gdr_view_plan_for": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(GdrViewOpts),
                                    C.POINTER(GdrViewPlan)]),
    "gdr_forward_view": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrInputs), C.POINTER(GdrViewPlan), C.c_void_p,
                                   C.POINTER(GdrViewOpts), C.POINTER(GdrSameAs), C.POINTER(GdrOutputs),
                                   C.POINTER(GdrViewState), C.c_void_p]),
    "gdr_view_reuse_probe": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrSettings), C.POINTER(GdrSameAs), C.c_void_p,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_void_p]),
    "gdr_views_plan_for": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(GdrViewOpts),
                                     C.POINTER(GdrViewsPlan)]),
    "gdr_forward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.POINTER(GdrInputs), C.POINTER(GdrViewsPlan), C.c_void_p,
                                    C.POINTER(GdrViewOpts), C.POINTER(GdrOutputs), C.c_int32, C.POINTER(C.c_void_p), C.c_float,
                                    C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32,
                                    C.POINTER(GdrViewState)]),
    "gdr_view_history_reset": (None, []),
    "gdr_k7_tune_override": (None, [C.c_int32]),
    "gdr_k7_tune_get_rounds": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "gdr_k7_tune_force_first": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gdr_k7_tune_get": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gdr_view_history_report": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_int32]),
    "gdr_view_history_get": (C.c_double, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gdr_view_history_set": (None, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    "gdr_backward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrInputs), C.POINTER(GdrGeom),
                               C.POINTER(GdrBinning), C.POINTER(GdrImage), C.c_uint64, C.c_void_p,
                               C.POINTER(GdrGradInputs), C.POINTER(GdrGradOutputs), C.c_void_p]),
    "gdr_preprocess_forward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.POINTER(GdrInputs),
                                               C.POINTER(GdrGeom), C.POINTER(C.c_void_p), C.c_void_p]),
    "gdr_render_backward": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                      C.POINTER(GdrImage), C.POINTER(GdrGradInputs), C.c_void_p, C.c_void_p]),
    "gdr_render_backward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom),
                                            C.POINTER(GdrBinning), C.POINTER(GdrImage), C.POINTER(GdrGradInputs),
                                            C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "gdr_render_backward_loss_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom),
                                                 C.POINTER(GdrBinning), C.POINTER(GdrImage), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p), C.c_float, C.c_float, C.c_void_p,
                                                 C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "gdr_render_backward_mean2d_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom),
                                                   C.POINTER(GdrBinning), C.POINTER(GdrImage), C.POINTER(C.c_void_p),
                                                   C.c_void_p, C.c_int32, C.c_void_p]),
    "gdr_render_backward_mean2d": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                             C.POINTER(GdrImage), C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdr_render_backward_mean2d_loss": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                                  C.POINTER(GdrImage), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]),
    "gdr_composite_forward_lossgrad": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                                 C.POINTER(GdrImage), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdr_topk_workspace_bytes": (C.c_size_t, []),
    "gdr_topk_absgrad": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "gdr_preprocess_backward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.POINTER(GdrInputs),
                                                C.POINTER(GdrGeom), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(GdrGradOutputs), C.c_void_p]),
    "gdr_view_loss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                        C.c_float, C.c_void_p, C.c_void_p]),
    "gdr_view_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdr_profile_enable": (C.c_int, [C.c_int]),
    "gdr_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int32, C.c_int32]),
    "gdr_kernel_count": (C.c_int, []),
    "gdr_kernel_name": (C.c_char_p, [C.c_int32]),
    "gdr_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    # include/gsr.h — 2DGS surfel path
    "gsr_geom_bytes": (C.c_size_t, [C.c_int32]),
    "gsr_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gsr_geom_carve": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(GdrGeom)]),
    "gsr_image_carve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(GdrImage)]),
    "gsr_preprocess_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrGeom), C.c_void_p,
                                         C.POINTER(C.c_uint32), C.c_void_p]),
    "gsr_render_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrGeom),
                                     C.POINTER(GdrBinning), C.POINTER(GdrImage), C.c_uint64, C.POINTER(GsrOutputs),
                                     C.c_void_p]),
    "gsr_composite_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                        C.POINTER(GdrImage), C.POINTER(GsrOutputs), C.c_void_p]),
    "gsr_forward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                              C.POINTER(GdrImage), C.c_uint64, C.POINTER(GsrOutputs), C.POINTER(C.c_uint32),
                              C.c_void_p]),
    "gsr_maps_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_maps_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_view_loss_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32] + [C.c_float] * 5 + [C.c_void_p, C.c_void_p]),
    "gsr_view_loss_backward": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32] + [C.c_float] * 5 + [C.c_void_p] * 5),
    "gsr_knn_cells": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gsr_knn_mean_dist2": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_preprocess_forward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrGeom),
                                               C.POINTER(C.c_void_p), C.c_void_p]),
    "gsr_render_backward": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                                      C.POINTER(GdrImage), C.POINTER(GsrGradInputs), C.c_void_p, C.c_void_p]),
    "gsr_preprocess_backward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrGeom),
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(GsrGradOutputs),
                                                C.c_void_p]),
    "gsr_forward_view": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrViewPlan), C.c_void_p,
                                   C.POINTER(GdrViewOpts), C.POINTER(GdrSameAs), C.POINTER(GsrOutputs),
                                   C.POINTER(GdrViewState), C.c_void_p]),
    "gsr_render_backward_views": (C.c_int, [C.c_int32, C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom),
                                            C.POINTER(GdrBinning), C.POINTER(GdrImage), C.POINTER(GsrGradInputs),
                                            C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "gsr_means2d_of_view": (C.c_int, [C.POINTER(GdrSettings), C.c_int32, C.POINTER(GdrGeom), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "gsr_backward": (C.c_int, [C.POINTER(GdrSettings), C.POINTER(GsrInputs), C.POINTER(GdrGeom), C.POINTER(GdrBinning),
                               C.POINTER(GdrImage), C.c_uint64, C.c_void_p, C.POINTER(GsrGradInputs),
                               C.POINTER(GsrGradOutputs), C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def load():
    """Load libgdr_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP rasterizer is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "generativedensification_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.gdr_abi_version() != ABI_VERSION:
            raise RuntimeError("libgdr_hip.so ABI version mismatch")
        tag = (lib.gdr_build_tag() or b"").decode()
        if tag != "release" and os.environ.get("GDR_ALLOW_EXPERIMENTAL_LIB") != "1":
            raise RuntimeError(f"{LIB_PATH} is a '{tag}' build of the library (measurement-only, results may be wrong); "
                               "set GDR_ALLOW_EXPERIMENTAL_LIB=1 to load it anyway")
        if os.environ.get("GDR_K7_PAIRS") is not None:     # developer A/B: pin the K7 variant (include/gdr.h gdr_k7_tune_override)
            lib.gdr_k7_tune_override(int(os.environ["GDR_K7_PAIRS"]))
        _lib = lib
    return _lib


BOUNDARY_PATH = os.path.join(_HERE, "lib", "_gdr_boundary.so")
_boundary = None     # the module; False = not available (one warning was given)


def boundary():
    """The compiled host boundary of the per-view render path (csrc/boundary.cpp: provenance key, render-group registry, the
    hub / view autograd nodes, marshalling — the per-call hot path of viewgroup.py / rasterizer.py in C++) or None.  It is
    an accelerator of the HOST code only: every kernel is reached through the same C ABI of the same libgdr_hip.so, which
    the module dlopens by the path given here.  If it was not built, or cannot be imported against this torch (ABI drift
    between the build and the run-time box), the ctypes path of this package serves — correct, ~2x more host time per
    call — and says so once.  GDR_COMPILED_BOUNDARY=0 switches it off (A/B, and the tests that pin the Python path)."""
    global _boundary
    if _boundary is None:
        if os.environ.get("GDR_COMPILED_BOUNDARY", "1") == "0":
            _boundary = False
            return None
        try:
            import importlib.util
            load()        # (the HIP library first: the module resolves its entry points from the same file)
            spec = importlib.util.spec_from_file_location("_gdr_boundary", BOUNDARY_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.ABI_VERSION != ABI_VERSION:
                raise RuntimeError("built against another ABI version")
            mod.init(LIB_PATH)
            _boundary = mod
        except Exception as exc:      # noqa: BLE001 — missing file, torch ABI mismatch, missing symbol: all mean "use ctypes"
            import warnings
            warnings.warn(f"generativedensification_amd: the compiled host boundary ({BOUNDARY_PATH}) is not usable "
                          f"({type(exc).__name__}: {exc}); falling back to the ctypes boundary (same kernels, more host time "
                          "per render call).  Build it with `make -C generativedensification_amd/csrc boundary`.")
            _boundary = False
    return _boundary or None


def check(rc: int, what: str):
    if rc != GDR_OK:
        msg = load().gdr_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def profile_enable(on: bool):
    check(load().gdr_profile_enable(int(on)), "gdr_profile_enable")


def profile_collect(reset: bool = True) -> dict:
    """{kernel name: (total ms, launches)} since the last reset (waits for pending events)."""
    lib = load()
    n = lib.gdr_kernel_count()
    ms = (C.c_double * n)()
    cnt = (C.c_uint64 * n)()
    check(lib.gdr_profile_collect(ms, cnt, n, int(reset)), "gdr_profile_collect")
    return {lib.gdr_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(n)}
