"""Where the host time of the unchanged caller's loop goes (small scene: the GPU is never the bottleneck).
Per 4-view step: the caller's own torch ops, our forward per call (with the callee breakdown from cProfile written to
gpurun_out/host_forward.pstats), backward() wall time and the part of it spent inside our backward functions."""
import cProfile, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativedensification_amd import rasterizer as R, viewgroup as VG
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.synthetic import make_scene
import diff_gaussian_rasterization as D
dev = torch.device("cuda:0")
N, h, w = int(os.environ.get("HP_N", 5000)), 128, 128
sc = {k: v.requires_grad_(True) for k, v in make_scene(N, 1, sh_degree=1, sigma0=(0.0052,), device=dev).items()}
cams = orbit_cameras(4, w, h, device=dev)
sets = [R.GaussianRasterizationSettings(h, w, math.tan(.375), math.tan(.375), torch.ones(3, device=dev), 1.0, c.world_view_transform,
                                        c.full_proj_transform, 1, c.camera_center, False, False) for c in cams]
T = dict(act=0.0, call=0.0, loss=0.0, bwd=0.0, our_bwd=0.0)
def wrap(cls):
    f = cls.backward
    def timed(ctx, *g):
        t = time.perf_counter(); out = f(ctx, *g); T["our_bwd"] += time.perf_counter() - t; return out
    cls.backward = staticmethod(timed)
for name in dir(VG):
    c = getattr(VG, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function: wrap(c)
for name in dir(R):
    c = getattr(R, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function: wrap(c)
def step(prof=None):
    losses = []
    for rs in sets:
        t0 = time.perf_counter()
        ssp = torch.zeros(N, 4, device=dev, requires_grad=True)
        op, scl, rot = torch.sigmoid(sc["opacity"]), torch.exp(sc["scales"]), torch.nn.functional.normalize(sc["rotations"])
        t1 = time.perf_counter()
        if prof: prof.enable()
        c, r, d, a = D.GaussianRasterizer(rs)(means3D=sc["centers"], means2D=ssp, shs=sc["shs"], opacities=op, scales=scl, rotations=rot)
        if prof: prof.disable()
        t2 = time.perf_counter()
        losses.append(c.clamp(0, 1).mean())
        t3 = time.perf_counter()
        T["act"] += t1 - t0; T["call"] += t2 - t1; T["loss"] += t3 - t2
    t = time.perf_counter(); sum(losses).backward(); T["bwd"] += time.perf_counter() - t
for _ in range(10): step()
torch.cuda.synchronize()
import gc; gc.disable()
for k in T: T[k] = 0.0
B = VG._B if getattr(VG, "COMPILED", False) else None     # the compiled boundary's nodes time themselves (csrc/boundary.cpp)
if B is not None: B.backward_host_ns(True)
K = 50
t = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
wall = time.perf_counter() - t
if B is not None: T["our_bwd"] += B.backward_host_ns(True)[0] * 1e-9
print("host boundary:", "compiled (csrc/boundary.cpp)" if B is not None else "python + ctypes")
print(f"N={N}: step {wall / K * 1e6:.0f} us | per view: caller activations {T['act'] / K / 4 * 1e6:.0f}, OUR forward call {T['call'] / K / 4 * 1e6:.0f}, "
      f"caller loss {T['loss'] / K / 4 * 1e6:.0f}, backward() {T['bwd'] / K / 4 * 1e6:.0f} of which our backward functions {T['our_bwd'] / K / 4 * 1e6:.0f}")
pr = cProfile.Profile()
for _ in range(K): step(pr)
torch.cuda.synchronize()
pr.dump_stats("gpurun_out/host_forward.pstats")
