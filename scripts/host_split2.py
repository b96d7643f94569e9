"""Fine split of the host time of one grouped GaussianRasterizer call (forward): timing wrappers around the pieces of
viewgroup.grouped_call / rasterizer.forward_raw (small scene, GPU never the bottleneck)."""
import collections, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generativedensification_amd import rasterizer as R, viewgroup as VG, _lib as L
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.synthetic import make_scene
import diff_gaussian_rasterization as D
dev = torch.device("cuda:0")
N, h, w = int(os.environ.get("HP_N", 5000)), 128, 128
sc = {k: v.requires_grad_(True) for k, v in make_scene(N, 1, sh_degree=1, sigma0=(0.0052,), device=dev).items()}
cams = orbit_cameras(4, w, h, device=dev)
sets = [R.GaussianRasterizationSettings(h, w, math.tan(.375), math.tan(.375), torch.ones(3, device=dev), 1.0, c.world_view_transform,
                                        c.full_proj_transform, 1, c.camera_center, False, False) for c in cams]
T = collections.defaultdict(float); CNT = collections.defaultdict(int)
def timed(name, f):
    def g(*a, **k):
        t = time.perf_counter()
        try: return f(*a, **k)
        finally: T[name] += time.perf_counter() - t; CNT[name] += 1
    return g
for mod, names in ((R, ["forward_raw", "_settings_struct", "_inputs_struct", "_carve_binning", "_launch_stats", "_d_capacity", "_d_record", "_f32", "_stream", "shape_key"]),
                   (VG, ["grouped_call", "_find_group", "_signature", "_same_as_pairs"])):
    for n in names:
        if hasattr(mod, n): setattr(mod, n, timed(mod.__name__.split(".")[-1] + "." + n, getattr(mod, n)))
R._CountReadback.__init__ = timed("CountReadback.__init__", R._CountReadback.__init__)
R._CountReadback.wait = timed("CountReadback.wait", R._CountReadback.wait)
R._State._view = timed("State._view", R._State._view)
lib = L.load()
class LibProxy:
    def __init__(s, lib): s._lib = lib; s._c = {}
    def __getattr__(s, n):
        f = s._c.get(n)
        if f is None: f = s._c[n] = timed("lib." + n, getattr(s._lib, n))
        return f
proxy = LibProxy(lib)
L.load = lambda: proxy
_empty = torch.empty
torch.empty = timed("torch.empty", _empty)
for cls in (VG._GroupView, VG._Hub):
    cls.forward = staticmethod(timed(cls.__name__ + ".forward", cls.forward))
def step():
    losses = []
    for rs in sets:
        ssp = torch.zeros(N, 4, device=dev, requires_grad=True)
        op, scl, rot = torch.sigmoid(sc["opacity"]), torch.exp(sc["scales"]), torch.nn.functional.normalize(sc["rotations"])
        t = time.perf_counter()
        c, r, d, a = D.GaussianRasterizer(rs)(means3D=sc["centers"], means2D=ssp, shs=sc["shs"], opacities=op, scales=scl, rotations=rot)
        T["CALL"] += time.perf_counter() - t; CNT["CALL"] += 1
        losses.append(c.clamp(0, 1).mean())
    sum(losses).backward()
for _ in range(10): step()
torch.cuda.synchronize()
import gc; gc.disable()
T.clear(); CNT.clear()
K = 50
for _ in range(K): step()
torch.cuda.synchronize()
calls = CNT["CALL"]
for n in sorted(T, key=lambda n: -T[n]):
    print(f"{n:32s} {T[n] / calls * 1e6:7.1f} us per call  ({CNT[n] / calls:.1f} x {T[n] / max(1, CNT[n]) * 1e6:.1f} us)")
