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
