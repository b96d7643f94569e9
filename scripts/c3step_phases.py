"""Where one sample of the reference's train-step render sequence (bench.py --workload c3step, unchanged caller) spends its
time: wall clock per phase with a device synchronisation at every phase boundary (so the phases do not overlap — the sum is
an upper bound of the step), with the forward reuse of repeated views on and off.  python scripts/c3step_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.autograd.functional import vjp
from generativedensification_amd import viewgroup as G, _lib as L
from generativedensification_amd.camera import MiniCam, orbit_cameras
from generativedensification_amd.renderer import Renderer
from generativedensification_amd.synthetic import make_scene, make_targets

dev = torch.device("cuda:0")
n, nf, h, w, deg, V, VS, K = 262_144, 81_600, 512, 512, 1, 8, 4, 12_000
keys = ("centers", "shs", "opacity", "scales", "rotations")
coarse = make_scene(n, 2, sh_degree=deg, sigma0=(0.0052,), device=dev)
fine = make_scene(nf, 3, sh_degree=deg, sigma0=(0.00065,), device=dev)
lc = {k: coarse[k][None].clone().requires_grad_(True) for k in keys}
lf = {k: fine[k][None].clone().requires_grad_(True) for k in keys}
cams0 = orbit_cameras(V, w, h)
tg = make_targets(V, h, w, 2).to(dev)
bgs = [torch.ones(3, device=dev) for _ in range(V)]
r = Renderer(sh_degree=deg, fused=False)
fresh_cams = "--fresh-cams" in sys.argv


def cam(j):
    c = cams0[j]
    return MiniCam(c.c2w.to(dev), w, h, c.FoVy, c.FoVx, c.znear, c.zfar, dev) if fresh_cams else cams_d[j]


cams_d = orbit_cameras(V, w, h, device=dev)
T = {}


def tick(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


def step(sync=True):
    for p in list(lc.values()) + list(lf.values()):
        p.grad = None
    i = 0
    t = time.perf_counter()
    centers = lc["centers"][i]
    oc = []
    for j in range(V):
        r.set_bg_color(bgs[j])
        oc.append(r.render_img(cam(j), None, centers, lc["shs"][i], lc["opacity"][i], lc["scales"][i], lc["rotations"][i], dev))
    if sync: t = tick("1 coarse forward x8", t)
    box = {}

    def fn(ssp):
        fr = []
        for j in range(VS):
            r.set_bg_color(bgs[j])
            fr.append(r.render_img(cam(j), None, centers, lc["shs"][i], lc["opacity"][i], lc["scales"][i], lc["rotations"][i], dev,
                                   screenspace_points=ssp))
        out = ((torch.stack([f["image"] for f in fr]) - tg[:VS]) ** 2).mean()
        if sync: box["t"] = tick("2 vjp forward x4", box["t"])
        return out
    box["t"] = t
    _, grad = vjp(fn, torch.zeros(n, 4, device=dev))
    if sync: t = tick("3 vjp backward x4", box["t"])
    score = torch.norm(grad[:, 2:4], dim=-1)
    sel = torch.zeros(n, dtype=torch.bool, device=dev)
    sel[torch.topk(score, K, dim=0).indices] = True
    fs = [torch.cat([lf[k][i], lc[k][i][~sel]], dim=0) for k in keys]
    if sync: t = tick("4 topk + fine set", t)
    of = []
    for j in range(V):
        r.set_bg_color(bgs[j])
        of.append(r.render_img(cam(j), None, *fs, dev, prex="_fine"))
    if sync: t = tick("5 fine forward x8", t)
    img_c = torch.cat([o["image"] for o in oc], dim=1)
    img_f = torch.cat([o["image_fine"] for o in of], dim=1)
    gt = torch.cat(list(tg), dim=1)
    total = ((img_c - gt) ** 2).mean() + ((img_f - gt) ** 2).mean() \
        + 0.1 * torch.cat([o["depth"] for o in oc], dim=1).mean() + 0.1 * torch.cat([o["acc_map"] for o in oc], dim=1).mean()
    if sync: t = tick("6 losses", t)
    total.backward()
    if sync: t = tick("7 backward", t)


if "--trace" in sys.argv:     # for rocprofv3 --kernel-trace: a few steady-state samples of the unchanged caller, reuse on
    G.REUSE_FORWARD = True
    for _ in range(10):
        step(sync=False)
    torch.cuda.synchronize()
    print("stats", G._REUSE_STATS)
    sys.exit(0)

for reuse in (False, True):
    G.REUSE_FORWARD = reuse
    for _ in range(4):
        step()
    T.clear()
    reps = 10
    for _ in range(reps):
        step()
    print(f"--- reuse={reuse} fresh_cams={fresh_cams}: per sample, phases synchronised")
    for k in sorted(T):
        print(f"  {k:24s} {1e6 * T[k] / reps:8.0f} us")
    print(f"  {'sum':24s} {1e6 * sum(T.values()) / reps:8.0f} us")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step(sync=False)
    torch.cuda.synchronize()
    print(f"  unsynchronised step      {1e6 * (time.perf_counter() - t0) / reps:8.0f} us   stats {dict(G._REUSE_STATS)}")

if "--profile" in sys.argv:
    import cProfile, pstats
    G.REUSE_FORWARD = True
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step(sync=False)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumtime").print_stats(60)
