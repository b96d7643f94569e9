"""Host-side cost of a grouped rasterizer call, segment by segment (GPU box)."""
import math, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativedensification_amd import viewgroup as G, rasterizer as R, _lib as L
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.synthetic import make_scene
import diff_gaussian_rasterization as D
dev = torch.device("cuda:0")
N, V, h, w = 300_000, 8, 512, 512
sc = make_scene(N, 1, sh_degree=1, sigma0=(0.0052,), device=dev)
base = {k: torch.stack([v, v]).requires_grad_(True) for k, v in sc.items()}
cams = orbit_cameras(V, w, h, device=dev)
sets = [R.GaussianRasterizationSettings(h, w, math.tan(.375), math.tan(.375), torch.ones(3, device=dev), 1.0, c.world_view_transform,
                                        c.full_proj_transform, 1, c.camera_center, False, False) for c in cams]
pc = time.perf_counter
seg = {}
def timed(name, fn):
    def wrap(*a, **k):
        t = pc(); r = fn(*a, **k); seg[name] = seg.get(name, 0) + pc() - t; return r
    return wrap
G._find_group = timed("find_group", G._find_group)
G._same_as_pairs = timed("same_as", G._same_as_pairs)
R.forward_raw = timed("forward_raw", R.forward_raw)
for grouped in (True, False, True):
    G.GROUP_VIEWS = grouped
    seg.clear()
    tf = tb = 0
    for it in range(6):
        for p in base.values(): p.grad = None
        torch.cuda.synchronize()
        t0 = pc()
        i = 1
        centers = base["centers"][i]
        losses = []
        for rs in sets:
            ssp = torch.zeros(N, 4, device=dev, requires_grad=True) + 0
            c, r, d, a = D.GaussianRasterizer(rs)(means3D=centers, means2D=ssp, shs=base["shs"][i], opacities=torch.sigmoid(base["opacity"][i]),
                                                  scales=torch.exp(base["scales"][i]), rotations=torch.nn.functional.normalize(base["rotations"][i]))
            losses.append(c.mean() + d.mean() + a.mean())
        t1 = pc()
        sum(losses).backward()
        t2 = pc()
        torch.cuda.synchronize()
        t3 = pc()
        del c, r, d, a, losses, ssp
        if it >= 2:
            tf += t1 - t0; tb += t2 - t1
    print(f"grouped={grouped}: host fwd {tf/4/V*1e6:.0f} us/view, host bwd {tb/4/V*1e6:.0f} us/view, last step wall {1e3*(t3-t0):.2f} ms;",
          {k: round(v / 6 / V * 1e6) for k, v in seg.items()}, "us/view")
