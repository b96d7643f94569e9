"""Object-like scenes (`--layout shell`): what the tile lists look like and how much of each list the forward needs.

Per view of the C4 / C3 scenes in both layouts: list-length percentiles, the share of the pairs that sits in lists of
each length class, and the WALKED share — list positions in front of the tile's deepest contributor (max n_contrib of the
tile's pixels) over the list length: everything behind it is sorted, gathered and (in K7) skipped for nothing.
"""
import sys, torch
sys.path.insert(0, ".")

from generativedensification_amd import rasterizer as R
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.renderer import Renderer
from generativedensification_amd.synthetic import make_scene

dev = torch.device("cuda:0")
CASES = (("c4", 2_000_000, 800, 4, 3, (0.00065,)), ("c2", 200_000, 800, 4, 3, (0.0052, 0.00065)))
for name, n, hw, views, deg, sig in CASES:
    for layout in ("cube", "shell"):
        scene = make_scene(n, 3, sh_degree=deg, sigma0=sig, device=dev, layout=layout)
        cams = orbit_cameras(views, hw, hw, device=dev)
        sets = [Renderer(sh_degree=deg).set_rasterizer(c, device=dev).raster_settings for c in cams]
        with torch.no_grad():
            states = R._forward_views_impl(scene["centers"], torch.empty(0, 4, device=dev), scene["shs"], scene["opacity"],
                                           scene["scales"], scene["rotations"], tuple(sets), R.RAW_ALL)[4]
        torch.cuda.synchronize()
        for v, st in enumerate(states[:2]):
            t = st.tensors()
            r = t["ranges"].long()
            L = (r[:, 1] - r[:, 0]).clamp_min(0)
            gx = (hw + 15) // 16
            nc = t["n_contrib"].long()
            H, W = nc.shape
            pad = torch.zeros(gx * 16, gx * 16, dtype=torch.long, device=dev)
            pad[:H, :W] = nc
            walked = pad.view(gx, 16, gx, 16).permute(0, 2, 1, 3).reshape(gx * gx, 256).max(dim=1).values
            Ls = L.sort().values.double()
            busy = L[L > 0]
            q = lambda p: int(torch.quantile(busy.double(), p)) if busy.numel() else 0
            cls = [(0, 256), (256, 2048), (2048, 4096), (4096, 16384), (16384, 1 << 30)]
            share = [float(L[(L > a) & (L <= b)].sum()) / max(1, int(L.sum())) for a, b in cls]
            cnt = [int(((L > a) & (L <= b)).sum()) for a, b in cls]
            pix_walk = float(nc.sum()) / max(1.0, float((L.view(gx, 1, gx, 1).expand(gx, 16, gx, 16).reshape(gx * 16, gx * 16)[:H, :W]).sum()))
            print(f"{name} {layout:5s} v{v} D {st.D} tiles>0 {int((L > 0).sum())}/{L.numel()} p50 {q(.5)} p90 {q(.9)} p99 {q(.99)} max {int(L.max())}"
                  f" | tiles by class {cnt} pair share {[round(s, 3) for s in share]}"
                  f" | walked (tile max n_contrib / L) {float(walked.sum()) / max(1, int(L.sum())):.3f}, per pixel {pix_walk:.3f}"
                  f" | longest tile: L {int(L.max())} walked {int(walked[L.argmax()])}")
