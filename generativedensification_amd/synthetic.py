"""Seeded synthetic Gaussian sets with the statistics of the reference's decoder output
(/root/reference/lightning/network.py:323,372-375,689-693)."""
from __future__ import annotations

import math

import torch


def morton_order(centers: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts points by the 3D Morton code of their position in their bounding box."""
    c = centers.detach().float().cpu()
    lo, hi = c.min(0).values, c.max(0).values
    q = ((c - lo) / (hi - lo).clamp_min(1e-12) * (2 ** bits - 1)).long().clamp_(0, 2 ** bits - 1)
    code = torch.zeros(c.shape[0], dtype=torch.long)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code)


def make_scene(n: int, seed: int, sh_degree: int = 3, sigma0=(0.0052, 0.00065), device="cpu", layout="cube",
               order="random"):
    """Raw (pre-activation) attributes exactly as `Renderer.render_img` receives them
    (lightning/renderer.py:209-230): centers (N,3), shs (N,M,3), opacity logits (N,1),
    log-scales (N,3), raw quaternions (N,4).  `sigma0` may be one value or a tuple that
    is mixed in equal parts (C2: 50/50 coarse-like / densified-like).  `layout`: "cube" = uniform in the
    reference's scene cube (the BASELINE workloads); "shell" = an object-like stand-in, centres on a bumpy
    sphere shell of radius ~0.3 (skewed tile lists: long at the silhouette, empty outside the object)."""
    g = torch.Generator().manual_seed(seed)
    if not isinstance(sigma0, (tuple, list)):
        sigma0 = (sigma0,)
    centers = torch.rand(n, 3, generator=g) - 0.5
    if layout == "shell":
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
        bump = 0.04 * torch.sin(9.0 * d[:, :1]) * torch.cos(7.0 * d[:, 1:2])
        centers = d * (0.3 + bump + 0.004 * torch.randn(n, 1, generator=g))
    elif layout != "cube":
        raise ValueError(f"unknown layout {layout!r}")
    logs = torch.empty(n, 3)
    per = (n + len(sigma0) - 1) // len(sigma0)
    for k, s0 in enumerate(sigma0):
        lo, hi = k * per, min(n, (k + 1) * per)
        if hi > lo:
            logs[lo:hi] = math.log(s0) + 0.3 * torch.randn(hi - lo, 3, generator=g)
    rots = torch.randn(n, 4, generator=g)
    opac = -2.18 + 1.5 * torch.randn(n, 1, generator=g)
    m = (sh_degree + 1) ** 2
    shs = torch.empty(n, m, 3)
    shs[:, 0] = torch.randn(n, 3, generator=g)
    if m > 1:
        shs[:, 1:] = 0.1 * torch.randn(n, m - 1, 3, generator=g)
    perm = torch.randperm(n, generator=g)  # interleave the sigma0 populations
    if order == "morton":   # memory order = spatial order (what a voxel-grid decoder emits), same set of Gaussians
        perm = perm[morton_order(centers[perm])]
    elif order != "random":
        raise ValueError(f"unknown order {order!r}")
    out = dict(centers=centers[perm], shs=shs[perm], opacity=opac[perm], scales=logs[perm],
               rotations=rots[perm])
    return {k: v.contiguous().to(device) for k, v in out.items()}


def make_targets(n_views: int, h: int, w: int, seed: int, device="cpu"):
    g = torch.Generator().manual_seed(seed + 7919)
    return torch.rand(n_views, h, w, 3, generator=g).to(device)


def view_loss(out: dict, target: torch.Tensor, prex: str = "") -> torch.Tensor:
    """SURVEY §8d loss: MSE(clamp(image), target) + 0.1 mean(depth) + 0.1 mean(alpha) —
    exercises the colour, depth and alpha gradient paths (network.py:746-752 keeps all three)."""
    return (((out[f"image{prex}"] - target) ** 2).mean() + 0.1 * out[f"depth{prex}"].mean()
            + 0.1 * out[f"acc_map{prex}"].mean())


def views_loss(out: dict, targets: torch.Tensor, prex: str = "") -> torch.Tensor:
    """Per-view losses (V,) of view-stacked outputs: the same loss as view_loss, evaluated on the stacked
    tensors the way the reference takes its loss on the concatenated views (network.py:974-978, loss.py:37-48)."""
    return (((out[f"image{prex}"] - targets) ** 2).mean(dim=(1, 2, 3)) + 0.1 * out[f"depth{prex}"].mean(dim=(1, 2, 3))
            + 0.1 * out[f"acc_map{prex}"].mean(dim=(1, 2)))


def surfel_loss(out: dict, target: torch.Tensor, prex: str = "") -> torch.Tensor:
    """Measurement loss of the 2DGS path: the terms the reference's loss takes on the surfel adaptor's dict
    (lightning/loss.py:37-38 MSE on the clamped image; :49-61 distortion.mean() * 1000 and the normal-consistency
    term ((1 - <rend_normal, depth_normal>) * acc_map.detach()).mean() * 0.2) plus 0.1 mean(depth) + 0.1 mean(alpha)
    so that every allmap channel receives a gradient (same role as `view_loss` for the 3DGS path)."""
    loss = ((out[f"image{prex}"] - target) ** 2).mean()
    loss = loss + 1000.0 * out[f"rend_dist{prex}"].mean()
    normal_error = ((1 - (out[f"rend_normal{prex}"] * out[f"depth_normal{prex}"]).sum(dim=-1))
                    * out[f"acc_map{prex}"].detach()).mean()
    return loss + 0.2 * normal_error + 0.1 * out[f"depth{prex}"].mean() + 0.1 * out[f"acc_map{prex}"].mean()
