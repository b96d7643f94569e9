"""Host-side cost of one GaussianRasterizer call (forward) and of its backward: cProfile over a loop on a small scene, where
the GPU is never the bottleneck (GPU box)."""
import cProfile, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativedensification_amd import rasterizer as R
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.synthetic import make_scene
import diff_gaussian_rasterization as D
dev = torch.device("cuda:0")
N, h, w = int(os.environ.get("HP_N", 50_000)), 256, 256
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
for _ in range(5): step()
torch.cuda.synchronize()
import gc; gc.disable()
t = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("per call (fwd+bwd share) us:", (time.perf_counter() - t) / 80 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(40)
st.print_callees("forward_raw")
