"""oracle/torch_ref.py — second, independent restatement: vectorised PyTorch with AUTOGRAD.

*** TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (same caveat as gdr_oracle.c). ***
Purpose: the C oracle (gdr_oracle.c) carries a hand-derived backward; this file derives
every gradient by torch autograd from a forward written directly from SURVEY.md
Appendix A, so the two cross-check each other (tests/test_oracle_cpu.py).  Small sizes
only (it materialises a pixels x tile-list matrix per 16x16 tile).

Deliberate non-smooth conventions that autograd must reproduce (Appendix A.4/A.5):
  * alpha = min(0.99, o*G) is differentiated straight-through (upstream 3DGS ignores the min);
  * skip tests (power>0, alpha<1/255, T(1-alpha)<1e-4) are masks on detached values;
  * the frustum clamp of t.x/t.z, t.y/t.z uses the clamped value as a constant
    (zero d/dt.x when clamped, no extra d/dt.z term).
"""
from __future__ import annotations

import math

import numpy as np
import torch

C0 = 0.28209479177387814  # lightning/renderer.py:17
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if deg >= 1:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg >= 3:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy),
              C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)  # (N, (deg+1)^2)


def quat_to_R(q: torch.Tensor) -> torch.Tensor:
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def render(means3D, opacities, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
           viewmatrix, projmatrix, sh_degree, campos, shs=None, colors_precomp=None, scales=None,
           rotations=None, cov3D_precomp=None, means2D_probe=None):
    """Returns (color (3,H,W), radii (N,), depth (1,H,W), alpha (1,H,W)); differentiable."""
    dt = means3D.dtype
    N = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    V, P = viewmatrix.to(dt), projmatrix.to(dt)
    ones = torch.ones(N, 1, dtype=dt)
    p1 = torch.cat([means3D, ones], 1)
    p_view = (p1 @ V)[:, :3]
    p_hom = p1 @ P
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    if means2D_probe is not None:
        ndc = ndc + means2D_probe[:, :2]
    if cov3D_precomp is None:
        R = quat_to_R(rotations)
        Mm = R * (scale_modifier * scales)[:, None, :]
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    rx, ry = (p_view[:, 0] / tz).detach(), (p_view[:, 1] / tz).detach()
    tx = torch.where((rx < -limx) | (rx > limx), (rx.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where((ry < -limy) | (ry > limy), (ry.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    Wm = V[:3, :3].T  # world->camera rotation: p_view = Wm p + t
    A = J @ Wm
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_safe = torch.where(det == 0, torch.ones_like(det), det)
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], 1)
    mid = 0.5 * (a + c)
    disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc))).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    pxd, pyd = px.detach(), py.detach()
    rad_i = radius.to(torch.int64)

    def tr(v):  # C int truncation toward zero
        return torch.trunc(v).to(torch.int64)

    rminx = tr((pxd - radius) / 16).clamp(0, gx)
    rminy = tr((pyd - radius) / 16).clamp(0, gy)
    rmaxx = tr((pxd + radius + 15) / 16).clamp(0, gx)
    rmaxy = tr((pyd + radius + 15) / 16).clamp(0, gy)
    visible = (tz.detach() > 0.2) & (det.detach() != 0) & (((rmaxx - rminx) * (rmaxy - rminy)) > 0)
    radii = torch.where(visible, rad_i, torch.zeros_like(rad_i)).to(torch.int32)

    if colors_precomp is None:
        dirs = means3D - campos.to(dt)[None, :]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        nb = (sh_degree + 1) ** 2
        Bk = sh_basis(sh_degree, dirs)
        rgb = (Bk[:, :, None] * shs[:, :nb, :]).sum(1) + 0.5
        rgb = torch.where(rgb.detach() < 0, torch.zeros_like(rgb), rgb)  # clamp mask
    else:
        rgb = colors_precomp
    opac = opacities.reshape(-1)
    depth_key = tz.detach().to(torch.float32).view(torch.int32).to(torch.int64)

    color = torch.zeros(3, H, W, dtype=dt)
    depth = torch.zeros(1, H, W, dtype=dt)
    alpha = torch.zeros(1, H, W, dtype=dt)
    bgv = bg.to(dt)
    idx_all = torch.arange(N)
    for tyi in range(gy):
        for txi in range(gx):
            sel = visible & (rminx <= txi) & (txi < rmaxx) & (rminy <= tyi) & (tyi < rmaxy)
            ids = idx_all[sel]
            y0, x0 = tyi * 16, txi * 16
            y1, x1 = min(y0 + 16, H), min(x0 + 16, W)
            ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
            pxf, pyf = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)
            if ids.numel() == 0:
                color[:, y0:y1, x0:x1] = bgv[:, None, None].expand(3, y1 - y0, x1 - x0)
                continue
            order = torch.argsort(depth_key[ids] * (N + 1) + ids)  # (depth bits, index): unique
            ids = ids[order]
            dx = px[ids][None, :] - pxf[:, None]
            dy = py[ids][None, :] - pyf[:, None]
            con = conic[ids]
            power = -0.5 * (con[:, 0][None] * dx * dx + con[:, 2][None] * dy * dy) - con[:, 1][None] * dx * dy
            G = torch.exp(torch.clamp(power, max=0.0))
            oG = opac[ids][None] * G
            al = oG + (torch.clamp(oG, max=0.99) - oG).detach()
            valid = (power.detach() <= 0) & (al.detach() >= 1.0 / 255.0)
            a_eff = torch.where(valid, al, torch.zeros_like(al))
            one_m = 1.0 - a_eff
            Tincl = torch.cumprod(one_m, dim=1)
            Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
            stop = valid & (Tincl.detach() < 1e-4)
            keep = valid & (torch.cumsum(stop.to(torch.int64), 1) == 0)
            w = torch.where(keep, a_eff * Texcl, torch.zeros_like(a_eff))
            a_kept = torch.where(keep, a_eff, torch.zeros_like(a_eff))
            Tfin = torch.prod(1.0 - a_kept, dim=1)
            Cc = w @ rgb[ids]  # (P,3)
            Dd = w @ p_view[ids, 2]
            Aa = w.sum(1)
            hh, ww = y1 - y0, x1 - x0
            color[:, y0:y1, x0:x1] = (Cc + Tfin[:, None] * bgv[None, :]).T.reshape(3, hh, ww)
            depth[0, y0:y1, x0:x1] = Dd.reshape(hh, ww)
            alpha[0, y0:y1, x0:x1] = Aa.reshape(hh, ww)
    return color, radii, depth, alpha


def settings_kwargs(s) -> dict:
    """Build render(**kwargs) from any 12-field settings record (torch or numpy fields)."""
    t = lambda v: v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
    return dict(image_height=s.image_height, image_width=s.image_width, tanfovx=float(s.tanfovx),
                tanfovy=float(s.tanfovy), bg=t(s.bg), scale_modifier=float(s.scale_modifier),
                viewmatrix=t(s.viewmatrix).reshape(4, 4), projmatrix=t(s.projmatrix).reshape(4, 4),
                sh_degree=int(s.sh_degree), campos=t(s.campos))
