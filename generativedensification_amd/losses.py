"""Fused image-space loss either side of the render path: clamp (renderer.py:261) + MSE
(loss.py:37-38) + the depth/alpha mean terms of the measurement loss, one HIP reduction kernel
forward and one elementwise kernel backward instead of ~12 torch launches per view.  Same value and
gradients as `synthetic.view_loss` on the dict `Renderer.render_img` returns."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class _ViewLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, depth, alpha, target_chw, w_depth, w_alpha):
        lib = L.load()
        dev = color.device
        color, depth, alpha = color.contiguous(), depth.contiguous(), alpha.contiguous()
        H, W = int(color.shape[-2]), int(color.shape[-1])
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            L.check(lib.gdr_view_loss_forward(color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), target_chw.data_ptr(),
                                              H, W, float(w_depth), float(w_alpha), loss.data_ptr(), st),
                    "gdr_view_loss_forward")
        ctx.save_for_backward(color, target_chw)
        ctx.w = (float(w_depth), float(w_alpha), H, W)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        color, target = ctx.saved_tensors
        w_depth, w_alpha, H, W = ctx.w
        dev = color.device
        g = g.to(torch.float32).contiguous()
        dc = torch.empty_like(color)
        dd = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        da = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            L.check(lib.gdr_view_loss_backward(color.data_ptr(), target.data_ptr(), H, W, w_depth, w_alpha, g.data_ptr(),
                                               dc.data_ptr(), dd.data_ptr(), da.data_ptr(), st), "gdr_view_loss_backward")
        return dc, dd, da, None, None, None


def view_loss_fused(color_chw: torch.Tensor, depth: torch.Tensor, alpha: torch.Tensor, target_chw: torch.Tensor,
                    w_depth: float = 0.1, w_alpha: float = 0.1) -> torch.Tensor:
    """color (3,H,W) UNclamped rasterizer output, depth/alpha (1,H,W), target (3,H,W) contiguous fp32 on the GPU."""
    if not color_chw.is_cuda:
        raise RuntimeError("view_loss_fused runs on ROCm/HIP tensors only")
    return _ViewLoss.apply(color_chw, depth, alpha, target_chw, w_depth, w_alpha)


class _SurfelViewLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, allmap, rays, viewmatrix, target_chw, depth_ratio, w_dist, w_normal, w_depth, w_alpha):
        lib = L.load()
        dev = color.device
        color, allmap = color.contiguous(), allmap.contiguous()
        rays = rays.to(device=dev, dtype=torch.float32).contiguous()
        view = viewmatrix.to(device=dev, dtype=torch.float32).contiguous()
        H, W = int(color.shape[-2]), int(color.shape[-1])
        w = (float(depth_ratio), float(w_dist), float(w_normal), float(w_depth), float(w_alpha))
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            L.check(lib.gsr_view_loss_forward(color.data_ptr(), allmap.data_ptr(), rays.data_ptr(), view.data_ptr(),
                                              target_chw.data_ptr(), H, W, *w, loss.data_ptr(), st), "gsr_view_loss_forward")
        ctx.save_for_backward(color, allmap, rays, view, target_chw)
        ctx.w, ctx.hw = w, (H, W)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        color, allmap, rays, view, target = ctx.saved_tensors
        H, W = ctx.hw
        dev = color.device
        g = g.to(torch.float32).contiguous()
        dc, da = torch.empty_like(color), torch.empty_like(allmap)
        scratch = torch.empty(9, H, W, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            L.check(lib.gsr_view_loss_backward(color.data_ptr(), allmap.data_ptr(), rays.data_ptr(), view.data_ptr(),
                                               target.data_ptr(), H, W, *ctx.w, g.data_ptr(), scratch.data_ptr(),
                                               dc.data_ptr(), da.data_ptr(), st), "gsr_view_loss_backward")
        return dc, da, None, None, None, None, None, None, None, None


def surfel_view_loss_fused(color_chw, allmap, rays, viewmatrix, target_chw, depth_ratio=0.0, w_dist=1000.0, w_normal=0.2,
                           w_depth=0.1, w_alpha=0.1) -> torch.Tensor:
    """`synthetic.surfel_loss` of the dict the 2DGS adaptor would return for (color, allmap), without materialising the
    dict: color (3,H,W) UNclamped, allmap (7,H,W), rays (H,W,6), viewmatrix = cam.world_view_transform, target (3,H,W)."""
    if not color_chw.is_cuda:
        raise RuntimeError("surfel_view_loss_fused runs on ROCm/HIP tensors only")
    return _SurfelViewLoss.apply(color_chw, allmap, rays, viewmatrix, target_chw, depth_ratio, w_dist, w_normal, w_depth,
                                 w_alpha)
