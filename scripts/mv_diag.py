"""Noise floor of the multi-view gradient comparison (fused node vs per-view sequence) for V views."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.renderer import Renderer
from generativedensification_amd.synthetic import make_scene, make_targets, view_loss
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, h, w = 30_000, 160, 208
sc = make_scene(n, 77, sh_degree=3, sigma0=(0.0052, 0.00065, 0.02))
cams = orbit_cameras(V, w, h, device=dev)
tg = make_targets(V, h, w, 77).to(dev)
three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])
bgs = [torch.tensor(three[j % 3], device=dev) for j in range(V)]
def run(fused, dtype=torch.float32):
    r = Renderer(sh_degree=3, fused=fused)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
    if fused:
        outs = r.render_views(cams, bgs, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev, screenspace_points=ssp)
    else:
        outs = []
        for c, b in zip(cams, bgs):
            r.set_bg_color(b)
            outs.append(r.render_img(c, None, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev, screenspace_points=ssp))
    per_view = [torch.autograd.grad(view_loss(o, tg[j]), leaves["centers"], retain_graph=True)[0].double().cpu().numpy() for j, o in enumerate(outs)] if not fused else None
    loss = sum(view_loss(o, tg[j]) for j, o in enumerate(outs))
    g = torch.autograd.grad(loss, list(leaves.values()))
    return {k: x.cpu().numpy() for k, x in zip(leaves, g)}, per_view
ref1, pv = run(False); ref2, _ = run(False); fus1, _ = run(True); fus2, _ = run(True)
ri = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
sum64 = np.sum(pv, axis=0)
for k in ref1:
    print(k, "ref-ref", ri(ref2[k], ref1[k]), "fus-fus", ri(fus2[k], fus1[k]), "fus-ref", ri(fus1[k], ref1[k]))
print("centers vs f64 sum of per-view f32 grads: ref", ri(ref1["centers"], sum64), "fused", ri(fus1["centers"], sum64))
d = np.abs(fus1["centers"] - ref1["centers"]); i = np.unravel_index(d.argmax(), d.shape)
print("worst", i, fus1["centers"][i], ref1["centers"][i], "max|ref|", np.abs(ref1["centers"]).max(), "per-view", [float(p[i]) for p in pv])
# ---- which grouping / which view?  (GDR_MAX_VIEWS is the host-side group size)
from generativedensification_amd import _lib as L
for gs in (1, 2, 5, 8):
    L.GDR_MAX_VIEWS = gs
    f, _ = run(True)
    print("group size", gs, "fus-ref centers", ri(f["centers"], ref1["centers"]))
L.GDR_MAX_VIEWS = 8
def one(j, fused):
    r = Renderer(sh_degree=3, fused=fused)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    r.set_bg_color(bgs[j])
    if fused:
        o = r.render_views([cams[j]], [bgs[j]], leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev)[0]
    else:
        o = r.render_img(cams[j], None, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev)
    g = torch.autograd.grad(view_loss(o, tg[j]), list(leaves.values()))
    return {k: x.cpu().numpy() for k, x in zip(leaves, g)}
for j in range(V):
    a, b = one(j, True), one(j, False)
    d = np.abs(a["centers"] - b["centers"]); i = np.unravel_index(d.argmax(), d.shape)
    print("view", j, {k: "%.2e" % ri(a[k], b[k]) for k in a}, "worst centers idx", i[0], a["centers"][i], b["centers"][i])
# ---- forward of views 4, 5: fused node vs reference sequence
with torch.no_grad():
    for j in (3, 4, 5):
        lv = {k: v.to(dev) for k, v in sc.items()}
        rf, rr = Renderer(sh_degree=3, fused=True), Renderer(sh_degree=3, fused=False)
        rr.set_bg_color(bgs[j])
        a = rf.render_views([cams[j]], [bgs[j]], lv["centers"], lv["shs"], lv["opacity"], lv["scales"], lv["rotations"], dev)[0]
        b = rr.render_img(cams[j], None, lv["centers"], lv["shs"], lv["opacity"], lv["scales"], lv["rotations"], dev)
        for k in ("image", "depth", "acc_map"):
            d = (a[k] - b[k]).abs()
            print("view", j, k, "max", float(d.max()), "n>1e-6", int((d > 1e-6).sum()), "n>1e-4", int((d > 1e-4).sum()))
