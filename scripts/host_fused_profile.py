"""Host time of one FUSED multi-view step (render_views_loss + backward) on a scene small enough that the GPU never limits:
total per step and the cProfile breakdown (forward in this thread; the backward runs in autograd's thread and is timed as a
whole).  usage: python scripts/host_fused_profile.py [views] [size]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativedensification_amd import rasterizer as R
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.renderer import Renderer
from generativedensification_amd.synthetic import make_scene, make_targets
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
N = 20000
scene = make_scene(N, 1, sh_degree=1, sigma0=(0.0052,), device=dev)
for t in scene.values(): t.requires_grad_(True)
cams = orbit_cameras(V, S, S, device=dev)
bgs = [torch.ones(3, device=dev)] * V
targets = make_targets(V, S, S, 1, device=dev).permute(0, 3, 1, 2).contiguous()
ren = Renderer(sh_degree=1)
T = dict(fwd=0.0, bwd=0.0)
def step(prof=None):
    t0 = time.perf_counter()
    if prof: prof.enable()
    losses = ren.render_views_loss(cams, bgs, targets, scene["centers"], scene["shs"], scene["opacity"], scene["scales"], scene["rotations"], dev)
    if prof: prof.disable()
    t1 = time.perf_counter()
    losses.sum().backward()
    T["fwd"] += t1 - t0; T["bwd"] += time.perf_counter() - t1
for _ in range(10): step()
torch.cuda.synchronize()
import gc; gc.disable()
T["fwd"] = T["bwd"] = 0.0
K = 50
t = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
print(f"V={V} {S}x{S}: step {(time.perf_counter() - t) / K * 1e6:.0f} us | forward call {T['fwd'] / K * 1e6:.0f} us, backward() {T['bwd'] / K * 1e6:.0f} us")
pr = cProfile.Profile()
for _ in range(K): step(pr)
torch.cuda.synchronize()
st = pstats.Stats(pr); st.strip_dirs().sort_stats("tottime").print_stats(22)
