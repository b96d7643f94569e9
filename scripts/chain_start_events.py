"""When do the views' forward chains start on the GPU relative to the end of K1 — WITHOUT a profiler (HIP events recorded
in front of every chain's first launch and behind K1).  usage: python scripts/chain_start_events.py c3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from generativedensification_amd import _lib as L, rasterizer as R
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.renderer import Renderer
from generativedensification_amd.synthetic import make_scene, make_targets

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda:0")
n, h, w, deg, V = wl["n"], wl["h"], wl["w"], wl["deg"], wl["views_per_gpu"]
scene = make_scene(n, wl["seed"], sh_degree=deg, sigma0=wl["sigma0"] or (0.0052, 0.00065), device=dev)
for t in scene.values(): t.requires_grad_(True)
cams = orbit_cameras(V, h, w, device=dev)
bgs = [torch.ones(3, device=dev)] * V
targets = make_targets(V, h, w, wl["seed"], device=dev).permute(0, 3, 1, 2).contiguous()
ren = Renderer(sh_degree=deg)
lib = L.load()
EV = {"k1": None, "chains": []}
class Proxy:
    def __getattr__(self, name):
        f = getattr(lib, name)
        if name == "gdr_preprocess_forward_views":
            def g(*a):
                r = f(*a); e = torch.cuda.Event(enable_timing=True); e.record(); EV["k1"] = e; return r
            return g
        if name == "gdr_binning_forward":
            def g(*a):
                sp = a[-1].value if hasattr(a[-1], "value") else int(a[-1])
                s = torch.cuda.ExternalStream(sp) if sp else torch.cuda.default_stream()
                e = torch.cuda.Event(enable_timing=True); e.record(s); EV["chains"].append(e); return f(*a)
            return g
        return f
L.load = lambda: Proxy()
def step():
    EV["chains"] = []
    losses = ren.render_views_loss(cams, bgs, targets, scene["centers"], scene["shs"], scene["opacity"], scene["scales"],
                                   scene["rotations"], dev)
    losses.sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
import time
rec = []
t0 = time.perf_counter()
for k in range(8):          # back to back, as in bench.py: the host runs ahead of the GPU
    step(); rec.append((EV["k1"], list(EV["chains"])))
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"8 steps back to back: host {t_host / 8 * 1e6:.0f} us per step, wall {t_all / 8 * 1e6:.0f} us per step")
for k, (k1, ch) in enumerate(rec):
    print(f"step {k}: chain starts after K1 end (us):", [round(k1.elapsed_time(e) * 1e3) for e in ch])
