"""oracle/torch_ref_surfel.py — independent vectorised PyTorch restatement of the 2DGS surfel path with AUTOGRAD.

*** TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (same caveat as gsr_oracle.c). ***
The C oracle carries a hand-derived backward; here every gradient comes from torch autograd of a forward written
directly from the algorithm statement in gsr_oracle.c's header, so the two cross-check each other
(tests/test_oracle_surfel_cpu.py).  Small sizes only.

Non-smooth conventions autograd must reproduce: alpha = min(0.99, o G) straight-through; every skip test, the
rho3d <= rho2d branch choice, the dual-visible normal flip and the median-contributor choice are masks / indices
on detached values.
"""
from __future__ import annotations

import torch

from .torch_ref import quat_to_R, sh_basis

NEAR_N, FAR_N, FILTER_SIZE, FILTER_INV_SQUARE = 0.2, 100.0, 0.707106, 2.0


def transmats(means3D, scales, rotations, scale_modifier, projmatrix, W, H):
    """(N,3,3) rows Tu, Tv, Tw and the WORLD-space normal (N,3)."""
    dt = means3D.dtype
    R = quat_to_R(rotations)
    L0 = R[:, :, 0] * (scale_modifier * scales[:, 0:1])
    L1 = R[:, :, 1] * (scale_modifier * scales[:, 1:2])
    Pm = projmatrix.to(dt)
    z = torch.zeros(means3D.shape[0], 1, dtype=dt)
    o = torch.ones(means3D.shape[0], 1, dtype=dt)
    Hm = torch.stack([torch.cat([L0, z], 1), torch.cat([L1, z], 1), torch.cat([means3D, o], 1)], 1)  # (N,3,4)
    clip = Hm @ Pm  # (N,3,4): clip-space images of (L0,0), (L1,0), (p,1)
    Tu = clip[:, :, 0] * (W / 2) + clip[:, :, 3] * ((W - 1) / 2)
    Tv = clip[:, :, 1] * (H / 2) + clip[:, :, 3] * ((H - 1) / 2)
    Tw = clip[:, :, 3]
    return torch.stack([Tu, Tv, Tw], 1), R[:, :, 2]


def aabb(T, cutoff=3.0):
    Tu, Tv, Tw = T[:, 0], T[:, 1], T[:, 2]
    t = torch.tensor([cutoff * cutoff, cutoff * cutoff, -1.0], dtype=T.dtype)
    d = (t * Tw * Tw).sum(1)
    ok = d != 0
    f = t[None] / torch.where(ok, d, torch.ones_like(d))[:, None]
    p = torch.stack([(f * Tu * Tw).sum(1), (f * Tv * Tw).sum(1)], 1)
    h0 = p * p - torch.stack([(f * Tu * Tu).sum(1), (f * Tv * Tv).sum(1)], 1)
    return ok, p, torch.sqrt(torch.clamp(h0, min=1e-4))


def render(means3D, opacities, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
           projmatrix, sh_degree, campos, shs=None, colors_precomp=None, scales=None, rotations=None,
           transMat_precomp=None, probe=None):
    """Returns (color (3,H,W), radii (N,), allmap (7,H,W)); differentiable.  `probe`: optional dict that receives
    the intermediate `T_ray` (retain_grad'ed copy of T used by the ray-splat intersection only: its .grad is the raw
    dL/dT of the render stage, without the centre path) and `depth` (N,)."""
    dt = means3D.dtype
    N = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    V = viewmatrix.to(dt)
    p1 = torch.cat([means3D, torch.ones(N, 1, dtype=dt)], 1)
    p_view = (p1 @ V)[:, :3]
    if transMat_precomp is None:
        T, n_world = transmats(means3D, scales, rotations, scale_modifier, projmatrix, W, H)
        n_view = n_world @ V[:3, :3]
    else:
        T = transMat_precomp.reshape(N, 3, 3)
        n_view = torch.tensor([0.0, 0.0, 1.0], dtype=dt).expand(N, 3)
    cosv = -(p_view * n_view).sum(1).detach()
    n_view = n_view * torch.where(cosv > 0, 1.0, -1.0).to(dt)[:, None]
    T_ray = T + 0
    if probe is not None:
        if not T_ray.requires_grad:
            T_ray.requires_grad_(True)
        T_ray.retain_grad()
        probe["T_ray"] = T_ray
        probe["depth"] = T[:, 2, 2].detach()
    ok, centre, extent = aabb(T)
    ext = extent.detach()
    radius = torch.ceil(torch.clamp(torch.maximum(ext[:, 0], ext[:, 1]), min=3.0 * FILTER_SIZE))
    cd = centre.detach()
    tr = lambda v: torch.trunc(v).to(torch.int64)
    rminx = tr((cd[:, 0] - radius) / 16).clamp(0, gx)
    rminy = tr((cd[:, 1] - radius) / 16).clamp(0, gy)
    rmaxx = tr((cd[:, 0] + radius + 15) / 16).clamp(0, gx)
    rmaxy = tr((cd[:, 1] + radius + 15) / 16).clamp(0, gy)
    tz = p_view[:, 2].detach()
    visible = (tz > 0.2) & (cosv != 0) & ok & (((rmaxx - rminx) * (rmaxy - rminy)) > 0)
    radii = torch.where(visible, radius.to(torch.int64), torch.zeros(N, dtype=torch.int64)).to(torch.int32)

    if colors_precomp is None:
        dirs = means3D - campos.to(dt)[None, :]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        nb = (sh_degree + 1) ** 2
        rgb = (sh_basis(sh_degree, dirs)[:, :, None] * shs[:, :nb, :]).sum(1) + 0.5
        rgb = torch.where(rgb.detach() < 0, torch.zeros_like(rgb), rgb)
    else:
        rgb = colors_precomp
    opac = opacities.reshape(-1)
    depth_key = tz.to(torch.float32).view(torch.int32).to(torch.int64)

    color = torch.zeros(3, H, W, dtype=dt)
    allmap = torch.zeros(7, H, W, dtype=dt)
    bgv = bg.to(dt)
    idx_all = torch.arange(N)
    for tyi in range(gy):
        for txi in range(gx):
            sel = visible & (rminx <= txi) & (txi < rmaxx) & (rminy <= tyi) & (tyi < rmaxy)
            ids = idx_all[sel]
            y0, x0 = tyi * 16, txi * 16
            y1, x1 = min(y0 + 16, H), min(x0 + 16, W)
            hh, ww = y1 - y0, x1 - x0
            ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
            pxf, pyf = xs.reshape(-1).to(dt)[:, None], ys.reshape(-1).to(dt)[:, None]
            if ids.numel() == 0:
                color[:, y0:y1, x0:x1] = bgv[:, None, None].expand(3, hh, ww)
                continue
            ids = ids[torch.argsort(depth_key[ids] * (N + 1) + ids)]
            Tu, Tv, Tw = T_ray[ids, 0][None], T_ray[ids, 1][None], T_ray[ids, 2][None]  # (1,n,3)
            k = pxf[:, :, None] * Tw - Tu
            l = pyf[:, :, None] * Tw - Tv
            pc = torch.cross(k, l, dim=2)
            pz = pc[:, :, 2]
            pz_ok = pz.detach() != 0
            pz_s = torch.where(pz_ok, pz, torch.ones_like(pz))
            sx, sy = pc[:, :, 0] / pz_s, pc[:, :, 1] / pz_s
            rho3d = sx * sx + sy * sy
            dx, dy = centre[ids, 0][None] - pxf, centre[ids, 1][None] - pyf
            rho2d = FILTER_INV_SQUARE * (dx * dx + dy * dy)
            use3d = rho3d.detach() <= rho2d.detach()
            rho = torch.where(use3d, rho3d, rho2d)
            dep = torch.where(use3d, sx * Tw[:, :, 0] + sy * Tw[:, :, 1] + Tw[:, :, 2], Tw[:, :, 2].expand_as(sx))
            power = -0.5 * rho
            G = torch.exp(torch.clamp(power, max=0.0))
            oG = opac[ids][None] * G
            al = oG + (torch.clamp(oG, max=0.99) - oG).detach()
            valid = pz_ok & (dep.detach() >= NEAR_N) & (power.detach() <= 0) & (al.detach() >= 1.0 / 255.0)
            a_eff = torch.where(valid, al, torch.zeros_like(al))
            Tincl = torch.cumprod(1.0 - a_eff, dim=1)
            Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
            stop = valid & (Tincl.detach() < 1e-4)
            keep = valid & (torch.cumsum(stop.to(torch.int64), 1) == 0)
            w = torch.where(keep, a_eff * Texcl, torch.zeros_like(a_eff))
            a_kept = torch.where(keep, a_eff, torch.zeros_like(a_eff))
            Tfin = torch.prod(1.0 - a_kept, dim=1)
            dep_s = torch.where(keep, dep, torch.ones_like(dep))
            m = FAR_N / (FAR_N - NEAR_N) * (1.0 - NEAR_N / dep_s)
            mw, mmw = m * w, m * m * w
            M1 = torch.cumsum(mw, 1) - mw
            M2 = torch.cumsum(mmw, 1) - mmw
            A = 1.0 - Texcl
            dist = (w * (m * m * A + M2 - 2.0 * m * M1)).sum(1)
            Dd = (w * dep_s).sum(1)
            Nn = w @ n_view[ids]
            Cc = w @ rgb[ids]
            # median: the last kept contributor that still saw T > 0.5
            cand = keep & (Texcl.detach() > 0.5)
            pos = torch.arange(ids.numel())[None, :].expand_as(cand)
            last = torch.where(cand, pos, torch.full_like(pos, -1)).max(1).values
            med = torch.where(last >= 0, dep.gather(1, last.clamp(min=0)[:, None])[:, 0], torch.zeros_like(Dd))
            color[:, y0:y1, x0:x1] = (Cc + Tfin[:, None] * bgv[None, :]).T.reshape(3, hh, ww)
            allmap[0, y0:y1, x0:x1] = Dd.reshape(hh, ww)
            allmap[1, y0:y1, x0:x1] = (1.0 - Tfin).reshape(hh, ww)
            allmap[2:5, y0:y1, x0:x1] = Nn.T.reshape(3, hh, ww)
            allmap[5, y0:y1, x0:x1] = med.reshape(hh, ww)
            allmap[6, y0:y1, x0:x1] = dist.reshape(hh, ww)
    return color, radii, allmap
