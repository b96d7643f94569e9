import math, os, sys, resource
sys.path.insert(0, ".")
import torch
from generativedensification_amd import rasterizer as R, viewgroup as VG
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.synthetic import make_scene
import diff_gaussian_rasterization as D
dev = torch.device("cuda:0")
N, h, w = 30000, 128, 128
sc = {k: v.requires_grad_(True) for k, v in make_scene(N, 1, sh_degree=1, sigma0=(0.0052,), device=dev).items()}
cams = orbit_cameras(4, w, h, device=dev)
sets = [R.GaussianRasterizationSettings(h, w, math.tan(.375), math.tan(.375), torch.ones(3, device=dev), 1.0, c.world_view_transform,
                                        c.full_proj_transform, 1, c.camera_center, False, False) for c in cams]
def step():
    losses = []
    for rs in sets:
        ssp = torch.zeros(N, 4, device=dev, requires_grad=True)
        c, r, d, a = D.GaussianRasterizer(rs)(means3D=sc["centers"], means2D=ssp, shs=sc["shs"], opacities=torch.sigmoid(sc["opacity"]),
                                              scales=torch.exp(sc["scales"]), rotations=torch.nn.functional.normalize(sc["rotations"]))
        losses.append(c.mean())
    sum(losses).backward()
    for t in sc.values(): t.grad = None
for k in range(1201):
    step()
    if k % 300 == 0:
        torch.cuda.synchronize()
        print(k, "cuda MB", round(torch.cuda.memory_allocated() / 2**20, 1), "reserved", round(torch.cuda.memory_reserved() / 2**20, 1),
              "rss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, "groups", len(VG._GROUPS),
              "rb pool", {n: len(v) for n, v in R._CountReadback._pool.items()})
