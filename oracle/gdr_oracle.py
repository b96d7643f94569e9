"""oracle/gdr_oracle.py — ctypes/numpy front-end of the CPU restatement (oracle/gdr_oracle.c).

*** TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see the header of gdr_oracle.c). ***
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg import this
module, and only as the checker / reported CPU baseline.  The product package
(generativedensification_amd, diff_gaussian_rasterization) never imports it.

It also provides `make_standin_module()`: an oracle-backed module object with the
reference boundary's names (GaussianRasterizationSettings, GaussianRasterizer —
imported at /root/reference/lightning/renderer.py:10-13) which tests inject as
`sys.modules['diff_gaussian_rasterization']` to run the REFERENCE's own
`Renderer.render_img` on CPU and record golden vectors (tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import types
from typing import NamedTuple, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force: bool = False) -> None:
    """Compile both precisions of the C restatement with gcc (oracle/Makefile)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f"libgdr_oracle_{p}.so")) for p in ("f32", "f64")
    )
    srcs = [os.path.join(_HERE, f) for f in ("gdr_oracle.c", "gsr_oracle.c", "oracle_common.h")]
    if not need:
        for p in ("f32", "f64"):
            so = os.path.join(_HERE, f"libgdr_oracle_{p}.so")
            if any(os.path.getmtime(so) < os.path.getmtime(src) for src in srcs):
                need = True
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def _lib(precision: str):
    if precision not in _LIBS:
        path = os.path.join(_HERE, f"libgdr_oracle_{precision}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.oracle_scan.restype = C.c_uint64
        lib.oracle_real_bytes.restype = C.c_int
        _LIBS[precision] = lib
    return _LIBS[precision]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Settings(NamedTuple):
    """Plain-python mirror of the 12 settings fields (lightning/renderer.py:111-124)."""

    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    sh_degree: int
    campos: np.ndarray
    prefiltered: bool = False
    debug: bool = False


class Oracle:
    """forward()/backward() over numpy arrays; every intermediate is returned."""

    def __init__(self, precision: str = "f32", nthreads: int = 1):
        self.precision = precision
        self.lib = _lib(precision)
        self.rt = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        assert self.lib.oracle_real_bytes() == np.dtype(self.rt).itemsize
        self.nthreads = int(nthreads)

    def _select(self, tiles):
        """Context manager: restrict the render loops to the tile ids in `tiles` (None = all).  See oracle_common.h."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            if tiles is None:
                yield
                return
            t = np.ascontiguousarray(np.asarray(tiles, dtype=np.int32))
            self.lib.oracle_select_tiles(_p(t), C.c_int(int(t.size)))
            try:
                yield
            finally:
                self.lib.oracle_select_tiles(None, C.c_int(0))
        return cm()

    def _a(self, x, shape=None):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=self.rt))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def forward(self, means3D, opacities, s: Settings, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, tiles=None) -> dict:
        """tiles: optional tile ids (y * grid_x + x); only those tiles are composited (the rest of the images stays
        zero) — preprocess and binning always cover everything.  backward() of the returned ctx uses the same tiles."""
        rt, lib = self.rt, self.lib
        tile_sel = tiles   # (`tiles` is reused below for tiles_touched)
        means3D = self._a(means3D)
        N = means3D.shape[0]
        H, W = int(s.image_height), int(s.image_width)
        opac = self._a(opacities, (N,))
        shs_a = self._a(shs)
        M = 0 if shs_a is None or shs_a.size == 0 else shs_a.shape[1]
        if shs_a is not None and shs_a.size == 0:
            shs_a = None
        cp = self._a(colors_precomp)
        if cp is not None and cp.size == 0:
            cp = None
        sc = self._a(scales)
        ro = self._a(rotations)
        c3 = self._a(cov3D_precomp)
        if c3 is not None and c3.size == 0:
            c3 = None
        view = self._a(s.viewmatrix, (16,))
        proj = self._a(s.projmatrix, (16,))
        campos = self._a(s.campos, (3,))
        bg = self._a(s.bg, (3,))
        gxy = ((W + 15) // 16, (H + 15) // 16)
        T = gxy[0] * gxy[1]

        radii = np.zeros(N, np.int32)
        xy = np.zeros((N, 2), rt)
        depths = np.zeros(N, rt)
        cov3D = np.zeros((N, 6), rt)
        rgb = np.zeros((N, 3), rt)
        conic_opacity = np.zeros((N, 4), rt)
        tiles = np.zeros(N, np.uint32)
        rect = np.zeros((N, 4), np.int32)
        clamped = np.zeros((N, 3), np.uint8)
        lib.oracle_preprocess_fwd(
            C.c_int(N), C.c_int(int(s.sh_degree)), C.c_int(M), _p(means3D), _p(sc),
            self.creal(float(s.scale_modifier)), _p(ro), _p(opac), _p(shs_a), _p(cp), _p(c3),
            _p(view), _p(proj), _p(campos), C.c_int(W), C.c_int(H), self.creal(float(s.tanfovx)),
            self.creal(float(s.tanfovy)), C.c_int(int(bool(s.prefiltered))), _p(radii), _p(xy),
            _p(depths), _p(cov3D), _p(rgb), _p(conic_opacity), _p(tiles), _p(rect), _p(clamped),
            C.c_int(self.nthreads))
        if c3 is not None:
            cov3D = c3
        offsets = np.zeros(N, np.uint32)
        D = int(lib.oracle_scan(C.c_int(N), _p(tiles), _p(offsets))) if N else 0
        keys_u = np.zeros(max(D, 1), np.uint64)
        vals_u = np.zeros(max(D, 1), np.uint32)
        keys = np.zeros(max(D, 1), np.uint64)
        vals = np.zeros(max(D, 1), np.uint32)
        ranges = np.zeros((T, 2), np.uint32)
        lib.oracle_bin(C.c_int(N), C.c_int(W), C.c_int(H), _p(radii), _p(rect), _p(depths),
                       _p(offsets), C.c_uint64(D), _p(keys_u), _p(vals_u), _p(keys), _p(vals),
                       _p(ranges))
        color = np.zeros((3, H, W), rt)
        depth = np.zeros((1, H, W), rt)
        alpha = np.zeros((1, H, W), rt)
        n_contrib = np.zeros((H, W), np.uint32)
        final_T = np.zeros((H, W), rt)
        with self._select(tile_sel):
            lib.oracle_render_fwd(C.c_int(W), C.c_int(H), _p(ranges), _p(vals), _p(xy), _p(rgb),
                                  _p(conic_opacity), _p(depths), _p(bg), _p(color), _p(depth),
                                  _p(alpha), _p(n_contrib), _p(final_T), C.c_int(self.nthreads))
        return dict(
            _tiles=tile_sel,
            color=color, depth=depth, alpha=alpha, radii=radii, num_rendered=D,
            xy=xy, depths=depths, cov3D=cov3D, rgb=rgb, conic_opacity=conic_opacity,
            tiles_touched=tiles, rect=rect, clamped=clamped, offsets=offsets,
            keys_unsorted=keys_u[:D], vals_unsorted=vals_u[:D], keys_sorted=keys[:D],
            point_list=vals[:D], ranges=ranges, n_contrib=n_contrib, final_T=final_T,
            _in=dict(means3D=means3D, opac=opac, shs=shs_a, M=M, cp=cp, sc=sc, ro=ro, c3=c3,
                     view=view, proj=proj, campos=campos, bg=bg, s=s, vals=vals),
        )

    def backward(self, ctx: dict, grad_color, grad_depth=None, grad_alpha=None, depth_to_mean: bool = True) -> dict:
        """depth_to_mean=False: parity-risk variant R1 — the per-Gaussian depth gradient is not sent into means3D."""
        rt, lib = self.rt, self.lib
        i = ctx["_in"]
        s = i["s"]
        N = i["means3D"].shape[0]
        H, W = int(s.image_height), int(s.image_width)
        M = i["M"]
        gC = self._a(grad_color, (3, H, W))
        gD = self._a(grad_depth, (H, W)) if grad_depth is not None else np.zeros((H, W), rt)
        gA = self._a(grad_alpha, (H, W)) if grad_alpha is not None else np.zeros((H, W), rt)
        d_mean2D = np.zeros((N, 4), rt)
        d_conic = np.zeros((N, 4), rt)
        d_opac = np.zeros((N, 1), rt)
        d_color = np.zeros((N, 3), rt)
        d_depth = np.zeros(N, rt)
        with self._select(ctx.get("_tiles")):
            lib.oracle_render_bwd(C.c_int(W), C.c_int(H), _p(ctx["ranges"]), _p(i["vals"]), _p(i["bg"]),
                                  _p(ctx["xy"]), _p(ctx["conic_opacity"]), _p(ctx["rgb"]),
                                  _p(ctx["depths"]), _p(ctx["final_T"]), _p(ctx["n_contrib"]), _p(gC),
                                  _p(gD), _p(gA), _p(d_mean2D), _p(d_conic), _p(d_opac), _p(d_color),
                                  _p(d_depth), C.c_int(self.nthreads))
        d_means3D = np.zeros((N, 3), rt)
        d_cov3D = np.zeros((N, 6), rt)
        d_sh = np.zeros((N, max(M, 1), 3), rt)
        d_scale = np.zeros((N, 3), rt)
        d_rot = np.zeros((N, 4), rt)
        cov3D = np.ascontiguousarray(ctx["cov3D"], dtype=rt)
        lib.oracle_preprocess_bwd(
            C.c_int(N), C.c_int(int(s.sh_degree)), C.c_int(M), _p(i["means3D"]), _p(ctx["radii"]),
            _p(i["shs"]), _p(ctx["clamped"]), _p(i["sc"]), _p(i["ro"]),
            self.creal(float(s.scale_modifier)), _p(cov3D), C.c_int(int(i["c3"] is not None)),
            C.c_int(int(i["cp"] is not None)), _p(i["view"]), _p(i["proj"]), _p(i["campos"]),
            C.c_int(W), C.c_int(H), self.creal(float(s.tanfovx)), self.creal(float(s.tanfovy)),
            _p(d_mean2D), _p(d_conic), _p(d_color), _p(d_depth if depth_to_mean else np.zeros_like(d_depth)),
            _p(d_means3D), _p(d_cov3D),
            _p(d_sh), _p(d_scale), _p(d_rot), C.c_int(self.nthreads))
        return dict(
            means3D=d_means3D, means2D=d_mean2D, shs=d_sh if M else None,
            colors_precomp=d_color if i["cp"] is not None else None, opacities=d_opac,
            scales=d_scale if i["c3"] is None else None, rotations=d_rot if i["c3"] is None else None,
            cov3D_precomp=d_cov3D if i["c3"] is not None else None,
            _partial=dict(conic=d_conic, color=d_color, depth=d_depth, cov3D=d_cov3D),
        )

    def mark_visible(self, means3D, viewmatrix):
        means3D = self._a(means3D)
        N = means3D.shape[0]
        out = np.zeros(N, np.uint8)
        self.lib.oracle_mark_visible(C.c_int(N), _p(means3D), _p(self._a(viewmatrix, (16,))), _p(out))
        return out.astype(bool)


# --------------------------------------------------------------------------------------
# Oracle-backed stand-in for the boundary package (tests / golden generation only).
# --------------------------------------------------------------------------------------
def make_standin_module(precision: str = "f32", nthreads: int = 1) -> types.ModuleType:
    import torch

    oracle = Oracle(precision, nthreads=nthreads)
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
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                    cov3Ds_precomp, rs):
            s = Settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, _np(rs.bg),
                         rs.scale_modifier, _np(rs.viewmatrix), _np(rs.projmatrix), rs.sh_degree,
                         _np(rs.campos), rs.prefiltered, rs.debug)
            out = oracle.forward(
                _np(means3D), _np(opacities), s,
                shs=_np(sh) if sh.numel() else None,
                colors_precomp=_np(colors_precomp) if colors_precomp.numel() else None,
                scales=_np(scales) if scales.numel() else None,
                rotations=_np(rotations) if rotations.numel() else None,
                cov3D_precomp=_np(cov3Ds_precomp) if cov3Ds_precomp.numel() else None)
            ctx.oracle_ctx = out
            ctx.shapes = (means2D.shape, sh.shape)
            ctx.record = getattr(_Fn, "record", None)
            if ctx.record is not None:
                ctx.record.append(out)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(tdt)
            return t(out["color"]), torch.from_numpy(out["radii"].copy()), t(out["depth"]), t(out["alpha"])

        @staticmethod
        def backward(ctx, g_color, g_radii, g_depth, g_alpha):
            g = oracle.backward(ctx.oracle_ctx, _np(g_color), _np(g_depth), _np(g_alpha))
            t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(tdt)
            m2shape, shshape = ctx.shapes
            gm2 = t(g["means2D"])[:, : m2shape[1]].contiguous()
            if m2shape[1] == 3:
                gm2[:, 2] = 0
            gsh = t(g["shs"]) if shshape[0] else None
            return (t(g["means3D"]), gm2, gsh, t(g["colors_precomp"]), t(g["opacities"]),
                    t(g["scales"]), t(g["rotations"]), t(g["cov3D_precomp"]), None)

    def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, raster_settings):
        return _Fn.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                         cov3Ds_precomp, raster_settings)

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                    rotations=None, cov3D_precomp=None):
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

    mod = types.ModuleType("diff_gaussian_rasterization")
    mod.GaussianRasterizationSettings = GaussianRasterizationSettings
    mod.GaussianRasterizer = GaussianRasterizer
    mod.rasterize_gaussians = rasterize_gaussians
    mod._Fn = _Fn
    mod.__oracle_standin__ = True
    return mod
