"""Tile list length classes of the bench workloads (tile sort classes: <= 2048, <= 4096, <= 16384, beyond)."""
import sys, torch
sys.path.insert(0, ".")

from generativedensification_amd import rasterizer as R
from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.renderer import Renderer
from generativedensification_amd.synthetic import make_scene
dev = torch.device("cuda:0")
for name, n, hw, views, deg, sig, layout in (("c4", 2_000_000, 800, 4, 3, (0.00065,), "cube"), ("c4", 2_000_000, 800, 4, 3, (0.00065,), "shell"),
                                               ("c2", 200_000, 800, 4, 3, (0.0052, 0.00065), "cube"), ("c2", 200_000, 800, 4, 3, (0.0052, 0.00065), "shell")):
    try:
        scene = make_scene(n, 3, sh_degree=deg, sigma0=sig, device=dev, layout=layout)
    except TypeError:
        scene = make_scene(n, 3, sh_degree=deg, sigma0=sig, device=dev)
    cams = orbit_cameras(views, hw, hw, device=dev)
    sets = [Renderer(sh_degree=deg).set_rasterizer(c, device=dev).raster_settings for c in cams]
    with torch.no_grad():
        states = R._forward_views_impl(scene["centers"], torch.empty(0, 4, device=dev), scene["shs"], scene["opacity"],
                                       scene["scales"], scene["rotations"], tuple(sets), R.RAW_ALL)[4]
    torch.cuda.synchronize()
    for v, st in enumerate(states[:2]):
        r = st.tensors()["ranges"].long()
        L = (r[:, 1] - r[:, 0])
        print(name, layout, "view", v, "D", st.D, "max", int(L.max()), "<=2048:", int((L <= 2048).sum()), "2049-4096:", int(((L > 2048) & (L <= 4096)).sum()),
              "4097-16384:", int(((L > 4096) & (L <= 16384)).sum()), ">16384:", int((L > 16384).sum()))
