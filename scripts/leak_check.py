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
def fresh(rs):      # the same 12 fields in new device tensors, as the reference's per-call MiniCam / bg.to(device) produce
    return rs._replace(bg=rs.bg.clone(), viewmatrix=rs.viewmatrix.clone(), projmatrix=rs.projmatrix.clone(), campos=rs.campos.clone())


def call(rs, ssp):
    return D.GaussianRasterizer(rs)(means3D=sc["centers"], means2D=ssp, shs=sc["shs"], opacities=torch.sigmoid(sc["opacity"]),
                                    scales=torch.exp(sc["scales"]), rotations=torch.nn.functional.normalize(sc["rotations"]))


def step_reference_sequence(k):
    """coarse loop, vjp over repeated views (forward reuse + mean2D-only K7), one backward; every third step is abandoned
    without a backward (the group, its cache and its parked K7 results must die with the graph)"""
    from torch.autograd.functional import vjp
    losses = [call(rs, torch.zeros(N, 4, device=dev, requires_grad=True))[0].mean() for rs in sets]

    def fn(ssp):
        return sum(call(fresh(rs), ssp)[0].mean() for rs in sets[:2])
    vjp(fn, torch.zeros(N, 4, device=dev))
    if k % 3 != 2:
        sum(losses).backward()
    for t in sc.values(): t.grad = None


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
    step_reference_sequence(k)
    if k % 300 == 0:
        torch.cuda.synchronize()
        print(k, "cuda MB", round(torch.cuda.memory_allocated() / 2**20, 1), "reserved", round(torch.cuda.memory_reserved() / 2**20, 1),
              "rss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, "groups", len(VG.live_group_views()),
              "rb pool", {n: len(v) for n, v in R._CountReadback._pool.items()}, "reuse", dict(VG._REUSE_STATS), "hist", len(VG._REUSE_HIST))
