"""oracle/gsr_oracle.py — ctypes/numpy front-end of the 2DGS surfel restatement (oracle/gsr_oracle.c).

*** TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see the header of gsr_oracle.c). ***
Also provides `make_surfel_standin_module()` / `make_simple_knn_stub()`: module objects with the names
/root/reference/lightning/renderer_2dgs.py:7-11 imports (`diff_surfel_rasterization`, `simple_knn._C.distCUDA2`)
so that tests can run the REFERENCE's own 2DGS `Renderer.render_img` on CPU and record golden vectors.
"""
from __future__ import annotations

import ctypes as C
import types
from typing import NamedTuple

import numpy as np

from .gdr_oracle import Oracle, Settings, _p


class SurfelOracle(Oracle):
    """forward()/backward() of the surfel path over numpy arrays; every intermediate is returned."""

    def forward(self, means3D, opacities, s: Settings, shs=None, colors_precomp=None, scales=None,
                rotations=None, transMat_precomp=None, tiles=None) -> dict:
        """tiles: see Oracle.forward (only those tiles are composited / differentiated)."""
        rt, lib = self.rt, self.lib
        tile_sel = tiles   # (`tiles` is reused below for tiles_touched)
        means3D = self._a(means3D)
        N = means3D.shape[0]
        H, W = int(s.image_height), int(s.image_width)
        opac = self._a(opacities, (N,))
        none_if_empty = lambda a: None if a is None or a.size == 0 else a
        shs_a = none_if_empty(self._a(shs))
        M = 0 if shs_a is None else shs_a.shape[1]
        cp = none_if_empty(self._a(colors_precomp))
        sc = none_if_empty(self._a(scales))
        ro = none_if_empty(self._a(rotations))
        tm = none_if_empty(self._a(transMat_precomp))
        if tm is not None:
            tm = tm.reshape(N, 9)
        view, proj = self._a(s.viewmatrix, (16,)), self._a(s.projmatrix, (16,))
        campos, bg = self._a(s.campos, (3,)), self._a(s.bg, (3,))
        ntiles = ((W + 15) // 16) * ((H + 15) // 16)

        radii = np.zeros(N, np.int32)
        xy = np.zeros((N, 2), rt)
        depths = np.zeros(N, rt)
        transMats = np.zeros((N, 9), rt)
        rgb = np.zeros((N, 3), rt)
        normal_opacity = np.zeros((N, 4), rt)
        tiles = np.zeros(N, np.uint32)
        rect = np.zeros((N, 4), np.int32)
        clamped = np.zeros((N, 3), np.uint8)
        lib.oracle_surfel_preprocess_fwd(
            C.c_int(N), C.c_int(int(s.sh_degree)), C.c_int(M), _p(means3D), _p(sc),
            self.creal(float(s.scale_modifier)), _p(ro), _p(opac), _p(shs_a), _p(cp), _p(tm), _p(view), _p(proj),
            _p(campos), C.c_int(W), C.c_int(H), _p(radii), _p(xy), _p(depths), _p(transMats), _p(rgb),
            _p(normal_opacity), _p(tiles), _p(rect), _p(clamped), C.c_int(self.nthreads))
        offsets = np.zeros(N, np.uint32)
        D = int(lib.oracle_scan(C.c_int(N), _p(tiles), _p(offsets))) if N else 0
        keys_u = np.zeros(max(D, 1), np.uint64)
        vals_u = np.zeros(max(D, 1), np.uint32)
        keys = np.zeros(max(D, 1), np.uint64)
        vals = np.zeros(max(D, 1), np.uint32)
        ranges = np.zeros((ntiles, 2), np.uint32)
        lib.oracle_bin(C.c_int(N), C.c_int(W), C.c_int(H), _p(radii), _p(rect), _p(depths), _p(offsets),
                       C.c_uint64(D), _p(keys_u), _p(vals_u), _p(keys), _p(vals), _p(ranges))
        color = np.zeros((3, H, W), rt)
        allmap = np.zeros((7, H, W), rt)
        n_contrib = np.zeros((2, H, W), np.uint32)
        final_T = np.zeros((3, H, W), rt)
        with self._select(tile_sel):
            lib.oracle_surfel_render_fwd(C.c_int(W), C.c_int(H), _p(ranges), _p(vals), _p(xy), _p(rgb), _p(transMats),
                                         _p(normal_opacity), _p(bg), _p(color), _p(allmap), _p(n_contrib), _p(final_T),
                                         C.c_int(self.nthreads))
        return dict(
            _tiles=tile_sel,
            color=color, allmap=allmap, radii=radii, num_rendered=D, xy=xy, depths=depths, transMats=transMats,
            rgb=rgb, normal_opacity=normal_opacity, tiles_touched=tiles, rect=rect, clamped=clamped, offsets=offsets,
            keys_unsorted=keys_u[:D], vals_unsorted=vals_u[:D], keys_sorted=keys[:D], point_list=vals[:D],
            ranges=ranges, n_contrib=n_contrib, final_T=final_T,
            _in=dict(means3D=means3D, opac=opac, shs=shs_a, M=M, cp=cp, sc=sc, ro=ro, tm=tm, view=view, proj=proj,
                     campos=campos, bg=bg, s=s, vals=vals))

    def backward(self, ctx: dict, grad_color, grad_allmap=None) -> dict:
        rt, lib = self.rt, self.lib
        i = ctx["_in"]
        s = i["s"]
        N = i["means3D"].shape[0]
        H, W = int(s.image_height), int(s.image_width)
        M = i["M"]
        gC = self._a(grad_color, (3, H, W))
        gO = self._a(grad_allmap, (7, H, W)) if grad_allmap is not None else np.zeros((7, H, W), rt)
        d_T = np.zeros((N, 9), rt)
        d_m2 = np.zeros((N, 4), rt)
        d_nrm = np.zeros((N, 3), rt)
        d_opac = np.zeros((N, 1), rt)
        d_color = np.zeros((N, 3), rt)
        with self._select(ctx.get("_tiles")):
            lib.oracle_surfel_render_bwd(
                C.c_int(W), C.c_int(H), _p(ctx["ranges"]), _p(i["vals"]), _p(i["bg"]), _p(ctx["xy"]),
                _p(ctx["normal_opacity"]), _p(ctx["transMats"]), _p(ctx["rgb"]), _p(ctx["final_T"]), _p(ctx["n_contrib"]),
                _p(gC), _p(gO), _p(d_T), _p(d_m2), _p(d_nrm), _p(d_opac), _p(d_color), C.c_int(self.nthreads))
        d_means3D = np.zeros((N, 3), rt)
        d_Tout = np.zeros((N, 9), rt)
        d_sh = np.zeros((N, max(M, 1), 3), rt)
        d_scale = np.zeros((N, 2), rt)
        d_rot = np.zeros((N, 4), rt)
        d_m2out = np.zeros((N, 4), rt)
        lib.oracle_surfel_preprocess_bwd(
            C.c_int(N), C.c_int(int(s.sh_degree)), C.c_int(M), _p(i["means3D"]), _p(ctx["radii"]), _p(i["shs"]),
            _p(ctx["clamped"]), _p(i["sc"]), _p(i["ro"]), self.creal(float(s.scale_modifier)), _p(ctx["transMats"]),
            C.c_int(int(i["tm"] is not None)), C.c_int(int(i["cp"] is not None)), _p(i["view"]), _p(i["proj"]),
            _p(i["campos"]), C.c_int(W), C.c_int(H), _p(d_T), _p(d_m2), _p(d_nrm), _p(d_color), _p(d_means3D),
            _p(d_Tout), _p(d_sh), _p(d_scale), _p(d_rot), _p(d_m2out), C.c_int(self.nthreads))
        pre = i["tm"] is not None
        return dict(
            means3D=d_means3D, means2D=d_m2out, shs=d_sh if M else None,
            colors_precomp=d_color if i["cp"] is not None else None, opacities=d_opac,
            scales=None if pre else d_scale, rotations=None if pre else d_rot,
            transMat_precomp=d_Tout if pre else None,
            _partial=dict(transMat=d_T, mean2D=d_m2, normal=d_nrm, color=d_color))


def make_surfel_standin_module(precision: str = "f32", nthreads: int = 1) -> types.ModuleType:
    """Oracle-backed `diff_surfel_rasterization` (GaussianRasterizationSettings, GaussianRasterizer)."""
    import torch

    oracle = SurfelOracle(precision, nthreads=nthreads)
    tdt = torch.float32 if precision == "f32" else torch.float64

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    def _np(t):
        return None if t is None else t.detach().cpu().numpy()

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, transMat_precomp, rs):
            s = Settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, _np(rs.bg), rs.scale_modifier,
                         _np(rs.viewmatrix), _np(rs.projmatrix), rs.sh_degree, _np(rs.campos), rs.prefiltered, rs.debug)
            out = oracle.forward(_np(means3D), _np(opacities), s, shs=_np(sh), colors_precomp=_np(colors_precomp),
                                 scales=_np(scales), rotations=_np(rotations), transMat_precomp=_np(transMat_precomp))
            ctx.oracle_ctx = out
            ctx.shapes = (means2D.shape, sh.shape)
            rec = getattr(_Fn, "record", None)
            if rec is not None:
                rec.append(out)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(tdt)
            return t(out["color"]), torch.from_numpy(out["radii"].copy()), t(out["allmap"])

        @staticmethod
        def backward(ctx, g_color, g_radii, g_allmap):
            g = oracle.backward(ctx.oracle_ctx, _np(g_color), _np(g_allmap))
            t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(tdt)
            m2shape, shshape = ctx.shapes
            gm2 = t(g["means2D"])[:, : m2shape[1]].contiguous()
            if m2shape[1] == 3:
                gm2[:, 2] = 0
            return (t(g["means3D"]), gm2, t(g["shs"]) if shshape[0] else None, t(g["colors_precomp"]),
                    t(g["opacities"]), t(g["scales"]), t(g["rotations"]), t(g["transMat_precomp"]), None)

    def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                            raster_settings):
        return _Fn.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                         raster_settings)

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            if (shs is None) == (colors_precomp is None):
                raise Exception("Please provide excatly one of either SHs or precomputed colors!")
            if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                    (scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
            e = torch.empty(0, dtype=tdt)
            return rasterize_gaussians(
                means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
                opacities, e if scales is None else scales, e if rotations is None else rotations,
                e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)

    mod = types.ModuleType("diff_surfel_rasterization")
    mod.GaussianRasterizationSettings = GaussianRasterizationSettings
    mod.GaussianRasterizer = GaussianRasterizer
    mod.rasterize_gaussians = rasterize_gaussians
    mod._Fn = _Fn
    mod.__oracle_standin__ = True
    return mod


def make_simple_knn_stub():
    """`simple_knn` + `simple_knn._C` with distCUDA2 = the brute-force oracle (CPU tensors, small N) —
    renderer_2dgs.py:11 imports it at module import."""
    import torch

    def distCUDA2(x):
        return torch.from_numpy(knn_mean_dist2(x.detach().cpu().numpy(), "f32")).to(x.dtype)

    pkg = types.ModuleType("simple_knn")
    sub = types.ModuleType("simple_knn._C")
    sub.distCUDA2 = distCUDA2
    pkg._C = sub
    return pkg, sub


def knn_mean_dist2(points, precision: str = "f32", nthreads: int = 1) -> np.ndarray:
    """Brute-force restatement of simple_knn.distCUDA2 (oracle_knn_mean_dist2 in gsr_oracle.c)."""
    o = SurfelOracle(precision, nthreads)
    pts = o._a(points, (-1, 3))
    out = np.zeros(pts.shape[0], o.rt)
    o.lib.oracle_knn_mean_dist2(C.c_int(pts.shape[0]), _p(pts), _p(out), C.c_int(o.nthreads))
    return out
