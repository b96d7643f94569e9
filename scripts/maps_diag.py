import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util as U
from generativedensification_amd.camera import build_rays, look_at_c2w, MiniCam
from generativedensification_amd.renderer_2dgs import _SurfelMaps, depth_to_normal
dev = torch.device("cuda:0")
H, W = 57, 83
c2w = look_at_c2w(torch.tensor([0.9, -1.1, 1.2]))
cam = MiniCam(c2w, W, H, torch.tensor(0.75), torch.tensor(0.75), 1.1, 2.7, dev)
rays = build_rays(c2w, 0.75, 0.75, H, W).to(dev)
g = torch.Generator().manual_seed(5)
ratio = 0.0
am = torch.rand(7, H, W, generator=g); am[0] = am[1] * (1.5 + am[0]); am[5] = 1.5 + am[5]; am[2:5] -= 0.5
hole = torch.rand(H, W, generator=g) < 0.15; am[:, hole] = 0.0; am = am.to(dev)
ups = [torch.randn(s, generator=g).to(dev) for s in ((H, W, 1), (H, W), (H, W, 3), (H, W, 3), (H, W))]
def torch_ops(a, dt):
    a = a.to(dt); r = rays.to(dt)
    alpha = a[1:2]
    nw = (a[2:5].permute(1, 2, 0) @ cam.world_view_transform[:3, :3].T.to(dt)).permute(2, 0, 1)
    med = torch.nan_to_num(a[5:6], 0, 0); exp = torch.nan_to_num(a[0:1] / alpha, 0, 0)
    sd = exp * (1 - ratio) + ratio * med
    sn, _ = depth_to_normal(r, sd); sn = sn.permute(2, 0, 1) * alpha.detach()
    return sd.permute(1, 2, 0), alpha.squeeze(0), nw.permute(1, 2, 0), sn.permute(1, 2, 0), a[6]
res = {}
for dt in (torch.float32, torch.float64):
    a1 = am.clone().to(dt).requires_grad_(True)
    ref = torch_ops(a1, dt)
    (gr,) = torch.autograd.grad(sum((o * u.to(dt)).sum() for o, u in zip(ref, ups)), a1)
    res[dt] = torch.nan_to_num(gr, 0.0, 0.0, 0.0).double().cpu().numpy()
a2 = am.clone().requires_grad_(True)
got = _SurfelMaps.apply(a2, rays, cam.world_view_transform, ratio)
(gg,) = torch.autograd.grad(sum((o * u).sum() for o, u in zip(got, ups)), a2)
gg = gg.double().cpu().numpy()
for ch in range(7):
    sc = np.abs(res[torch.float64][ch]).max()
    print(ch, "scale", sc, "hip-f64", np.abs(gg[ch] - res[torch.float64][ch]).max() / max(sc, 1e-30), "t32-f64", np.abs(res[torch.float32][ch] - res[torch.float64][ch]).max() / max(sc, 1e-30),
          "outl", U.outlier_fraction(gg[ch], res[torch.float64][ch], 1e-3, 1e-5 * sc), U.outlier_fraction(res[torch.float32][ch], res[torch.float64][ch], 1e-3, 1e-5 * sc))
d = np.abs(gg[1] - res[torch.float64][1])
idx = np.argsort(-d.ravel())[:6]
amc = am.double().cpu().numpy()
for i in idx:
    y, x = divmod(int(i), W)
    print("pix", y, x, "got", gg[1, y, x], "ref64", res[torch.float64][1, y, x], "ref32", res[torch.float32][1, y, x], "alpha", amc[1, y, x], "D", amc[0, y, x],
          "g0 got", gg[0, y, x], "g0 ref", res[torch.float64][0, y, x], "upacc", float(ups[1][y, x]))
