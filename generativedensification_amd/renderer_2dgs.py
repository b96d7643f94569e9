"""Host-side 2DGS render adaptor — counterpart of the reference's surfel `Renderer`
(/root/reference/lightning/renderer_2dgs.py:98-283) for boxes where /root/reference does not exist.  Same method
names, argument meaning and return values:

    render_img(cam, rays, centers, shs, opacity, scales, rotations, device, cov3D_precomp=None, prex='',
               depth_ratio=0.0, screenspace_points=None)
      rays is None -> the clamped image (3,H,W)                                            (renderer_2dgs.py:236-238)
      else         -> {image (H,W,3), depth (H,W,1), acc_map (H,W), rend_normal (H,W,3) world space,
                       depth_normal (H,W,3) pseudo normal of the depth map x alpha, rend_dist (H,W)}  (:241-278)

Steps mirrored (:190-278): sigmoid(opacity), exp(scales (N,2)), normalize(rotations); (N,4) zero screen-space
carrier; rasterizer -> (image, radii, allmap); clamp; allmap slicing; normal rotated by world_view[:3,:3].T;
expected depth = allmap[0] / alpha with nan -> 0; surf_depth = (1-r) expected + r median; pseudo normals from the
unprojected depth map (central differences, cross product, normalised, zero border) times detached alpha.
fused=True hands the RAW opacity / scale / rotation tensors to the rasterizer (activations inside K1s/K9s).
Pinned against the real class by tests/golden/render2dgs_*.npz.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib as L
from .rasterizer import GaussianRasterizationSettings
from .surfel_rasterizer import (GaussianRasterizer, _RasterizeSurfels, render_surfel_views_loss_raw,
                                render_surfel_views_raw)

RAW_ALL = L.GDR_IN_RAW_OPACITY | L.GDR_IN_RAW_SCALES | L.GDR_IN_RAW_ROTATIONS


def depths_to_points(rays: torch.Tensor, depthmap: torch.Tensor) -> torch.Tensor:
    """rays (H,W,6) = origin | direction; point = origin + depth * direction  (renderer_2dgs.py:75-77)."""
    return rays[..., :3].reshape(-1, 3) + depthmap.reshape(-1, 1) * rays[..., 3:].reshape(-1, 3)


def depth_to_normal(rays: torch.Tensor, depth: torch.Tensor):
    """Pseudo surface normal of a depth map (1,H,W): normalised cross product of the central differences of the
    unprojected points along rows and columns, zero on the 1-pixel border (renderer_2dgs.py:79-90).
    Returns (normal (H,W,3), points (H,W,3))."""
    pts = depths_to_points(rays, depth).reshape(*depth.shape[1:], 3)
    d_rows = pts[2:, 1:-1] - pts[:-2, 1:-1]
    d_cols = pts[1:-1, 2:] - pts[1:-1, :-2]
    inner = torch.nn.functional.normalize(torch.cross(d_rows, d_cols, dim=-1), dim=-1)
    normal = torch.zeros_like(pts)
    normal[1:-1, 1:-1, :] = inner
    return normal, pts


@torch.no_grad()
def _activation_scale(x: torch.Tensor) -> torch.Tensor:
    """Initial isotropic surfel scale (N,2) = distance-scale of the 3 nearest neighbours: sqrt(clamp_min(distCUDA2(x),
    1e-7)) repeated on both axes (renderer_2dgs.py:92-96)."""
    from .knn import dist2

    return torch.sqrt(torch.clamp_min(dist2(x), 0.0000001))[..., None].repeat(1, 2)


class _SurfelMaps(torch.autograd.Function):
    """allmap -> (depth (H,W,1), acc_map (H,W), rend_normal (H,W,3), depth_normal (H,W,3), rend_dist (H,W)) in one HIP
    kernel forward and two backward (include/gsr.h gsr_maps_*): the lines 241-278 of the reference adaptor."""

    @staticmethod
    def forward(ctx, allmap, rays, viewmatrix, depth_ratio):
        import ctypes as C

        lib = L.load()
        dev = allmap.device
        allmap = allmap.contiguous()
        rays = rays.to(device=dev, dtype=torch.float32).contiguous()
        view = viewmatrix.to(device=dev, dtype=torch.float32).contiguous()
        H, W = int(allmap.shape[1]), int(allmap.shape[2])
        f32 = dict(dtype=torch.float32, device=dev)
        depth, acc = torch.empty(H, W, 1, **f32), torch.empty(H, W, **f32)
        rn, dn, dist = torch.empty(H, W, 3, **f32), torch.empty(H, W, 3, **f32), torch.empty(H, W, **f32)
        with torch.cuda.device(dev):
            L.check(lib.gsr_maps_forward(allmap.data_ptr(), rays.data_ptr(), view.data_ptr(), H, W, float(depth_ratio),
                                         depth.data_ptr(), acc.data_ptr(), rn.data_ptr(), dn.data_ptr(), dist.data_ptr(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gsr_maps_forward")
        ctx.save_for_backward(allmap, rays, view)
        ctx.depth_ratio = float(depth_ratio)
        return depth, acc, rn, dn, dist

    @staticmethod
    def backward(ctx, g_depth, g_acc, g_rn, g_dn, g_dist):
        import ctypes as C

        lib = L.load()
        allmap, rays, view = ctx.saved_tensors
        dev = allmap.device
        H, W = int(allmap.shape[1]), int(allmap.shape[2])
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_depth, g_acc, g_rn, g_dn, g_dist)]
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        out = torch.empty_like(allmap)
        scratch = torch.empty(6, H, W, dtype=torch.float32, device=dev) if gs[3] is not None else None
        with torch.cuda.device(dev):
            L.check(lib.gsr_maps_backward(allmap.data_ptr(), rays.data_ptr(), view.data_ptr(), H, W, ctx.depth_ratio,
                                          ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]), ptr(scratch),
                                          out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "gsr_maps_backward")
        return out, None, None, None


class Renderer(nn.Module):
    def __init__(self, sh_degree: int = 3, white_background: bool = True, radius: float = 1, fused: bool = True):
        super().__init__()
        self.fused = fused
        self.sh_degree = sh_degree
        self.white_background = white_background
        self.radius = radius
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.bg_color = torch.tensor([1, 1, 1] if white_background else [0, 0, 0], dtype=torch.float32)

    def set_bg_color(self, bg):
        self.bg_color = bg

    def get_scaling(self, s):
        return self.scaling_activation(s)

    def get_rotation(self, r):
        return self.rotation_activation(r)

    def get_opacity(self, o):
        return self.opacity_activation(o)

    def raster_settings(self, viewpoint_camera, scaling_modifier: float = 1.0, device="cuda"):
        """The 12-field settings record (what set_rasterizer wraps in a GaussianRasterizer module)."""
        return GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=self.bg_color.to(device), scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=self.sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)

    def set_rasterizer(self, viewpoint_camera, scaling_modifier: float = 1.0, device="cuda"):
        return GaussianRasterizer(raster_settings=self.raster_settings(viewpoint_camera, scaling_modifier, device))

    def render_views(self, cams, rays_list, bg_colors, centers, shs, opacity, scales, rotations, device, prex="",
                     depth_ratio=0.0, screenspace_points=None, raw=False):
        """All `cams` of one surfel set in ONE rasterizer node (the per-view loops of network.py:826-838 / 964-972 with
        the 2DGS adaptor): K1s for every view, one read-back of the duplicate counts, gradients summed over the views
        inside K9s.  Returns the list of dicts render_img(cam, rays, ...) returns, or with raw=True per-view dicts in the
        rasterizer's own layout {color (3,H,W) UNclamped, allmap (7,H,W)} for a loss that folds the maps in
        (losses.surfel_view_loss_fused).  bg_colors: None, one tensor, or one per view."""
        sets = []
        for j, cam in enumerate(cams):
            if bg_colors is not None:
                self.set_bg_color(bg_colors[j] if isinstance(bg_colors, (list, tuple)) else bg_colors)
            sets.append(self.raster_settings(cam, device=device))
        if screenspace_points is None:
            screenspace_points = torch.zeros((centers.shape[0], 4), dtype=centers.dtype, requires_grad=True,
                                             device=device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        if self.fused:
            colors, radii, allmaps = render_surfel_views_raw(centers, screenspace_points, shs, opacity, scales, rotations,
                                                             sets, RAW_ALL)
        else:
            colors, radii, allmaps = render_surfel_views_raw(
                centers, screenspace_points, shs, self.get_opacity(opacity), self.get_scaling(scales),
                self.get_rotation(rotations), sets, 0)
        if raw:
            return [{f"color{prex}": c, f"allmap{prex}": a} for c, a in zip(colors, allmaps)]
        outs = []
        for v, cam in enumerate(cams):
            depth, acc, rend_normal, depth_normal, rend_dist = _SurfelMaps.apply(allmaps[v], rays_list[v],
                                                                                 cam.world_view_transform, float(depth_ratio))
            outs.append({f"image{prex}": colors[v].clamp(0, 1).permute(1, 2, 0), f"depth{prex}": depth,
                         f"acc_map{prex}": acc, f"rend_normal{prex}": rend_normal, f"depth_normal{prex}": depth_normal,
                         f"rend_dist{prex}": rend_dist})
        return outs

    def render_views_loss(self, cams, rays_list, bg_colors, targets_chw, centers, shs, opacity, scales, rotations, device,
                          depth_ratio=0.0, w_dist=1000.0, w_normal=0.2, w_depth=0.1, w_alpha=0.1, screenspace_points=None):
        """Per-view losses (V,) = synthetic.surfel_loss of what render_views would return for each view, with the fused
        loss kernels of a view inside the rasterizer node, on the view's side stream (no dL/dimage
        tensors, no per-view autograd nodes).  targets_chw: (V,3,H,W)."""
        sets = []
        for j, cam in enumerate(cams):
            if bg_colors is not None:
                self.set_bg_color(bg_colors[j] if isinstance(bg_colors, (list, tuple)) else bg_colors)
            sets.append(self.raster_settings(cam, device=device))
        if screenspace_points is None:
            screenspace_points = torch.zeros((centers.shape[0], 4), dtype=centers.dtype, requires_grad=True,
                                             device=device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        views = [cam.world_view_transform for cam in cams]
        tg = [targets_chw[j] for j in range(len(sets))]
        if self.fused:
            args = (centers, screenspace_points, shs, opacity, scales, rotations)
            flags = RAW_ALL
        else:
            args = (centers, screenspace_points, shs, self.get_opacity(opacity), self.get_scaling(scales),
                    self.get_rotation(rotations))
            flags = 0
        losses, _ = render_surfel_views_loss_raw(*args, sets, rays_list, views, tg, depth_ratio, w_dist, w_normal, w_depth,
                                                 w_alpha, flags)
        return losses

    def render_img(self, cam, rays, centers, shs, opacity, scales, rotations, device, cov3D_precomp=None, prex="",
                   depth_ratio=0.0, screenspace_points=None):
        rasterizer = self.set_rasterizer(cam, device=device)
        if screenspace_points is None:
            screenspace_points = torch.zeros((centers.shape[0], 4), dtype=centers.dtype, requires_grad=True,
                                             device=device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        if self.fused and cov3D_precomp is None and scales is not None and rotations is not None and centers.is_cuda:
            e = torch.empty(0, dtype=torch.float32, device=centers.device)
            image, radii, allmap = _RasterizeSurfels.apply(centers, screenspace_points, shs, e, opacity, scales,
                                                           rotations, e, rasterizer.raster_settings, RAW_ALL)
        else:
            image, radii, allmap = rasterizer(
                means3D=centers, means2D=screenspace_points, shs=shs, opacities=self.get_opacity(opacity),
                scales=None if scales is None else self.get_scaling(scales),
                rotations=None if rotations is None else self.get_rotation(rotations), cov3D_precomp=cov3D_precomp)
        image = image.clamp(0, 1)
        if rays is None:
            return image
        if self.fused and allmap.is_cuda:   # lines 241-278 of the reference adaptor in one kernel (same dict)
            depth, acc, rend_normal, depth_normal, rend_dist = _SurfelMaps.apply(allmap, rays, cam.world_view_transform,
                                                                                 float(depth_ratio))
            return {f"image{prex}": image.permute(1, 2, 0), f"depth{prex}": depth, f"acc_map{prex}": acc,
                    f"rend_normal{prex}": rend_normal, f"depth_normal{prex}": depth_normal,
                    f"rend_dist{prex}": rend_dist}
        alpha = allmap[1:2]
        normal_world = (allmap[2:5].permute(1, 2, 0) @ cam.world_view_transform[:3, :3].T).permute(2, 0, 1)
        depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
        depth_expected = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
        surf_depth = depth_expected * (1 - depth_ratio) + depth_ratio * depth_median
        surf_normal, _ = depth_to_normal(rays, surf_depth)
        surf_normal = surf_normal.permute(2, 0, 1) * alpha.detach()
        return {
            f"image{prex}": image.permute(1, 2, 0),
            f"depth{prex}": surf_depth.permute(1, 2, 0),
            f"acc_map{prex}": alpha.squeeze(0),
            f"rend_normal{prex}": normal_world.permute(1, 2, 0),
            f"depth_normal{prex}": surf_normal.permute(1, 2, 0),
            f"rend_dist{prex}": allmap[6:7].squeeze(0),
        }
