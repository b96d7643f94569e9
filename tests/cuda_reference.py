"""Consumer of a CUDA-reference dump (scripts/dump_reference_cuda.py): the only path from "parity unpinned" to pinned.

The reference's rasterizer is an un-vendored submodule (/root/reference/.gitmodules:1-3); the conventions the oracle and the
HIP path assume are listed in INTEGRATION.md section 0.  A maintainer with a CUDA box runs the dump script once; this module
re-creates the dump's seeded inputs, runs a backend (the oracle, or the HIP path) under every variant of the open
conventions and reports which variant the data selects:

    R4  depth image      "sum"  sum_i w_i z_i            | "normalized"  sum_i w_i z_i / sum_i w_i
    R1  dL/ddepth -> means3D   "yes" (ashawkey lineage)  | "no"
    R3  carrier columns 2:4    "abs"  sum_pixels |term|  | "signed" (= columns 0:1) | "zero"

and the north_star bars for the selected variant: radii bit-exact, RGB / depth / alpha within 1e-4 relative, every gradient
within 1e-4 relative per element (util.elem_stats, at most util.MAX_OUTSIDE of the elements outside).
"""
from __future__ import annotations

import math

import numpy as np
import torch

import util as U

DEFAULTS = dict(R4="sum", R1="yes", R3="abs")       # what the product ships
GRAD_KEYS = ("means3D", "means2D", "shs", "opacities", "scales", "rotations")
HOW_TO_SWITCH = dict(R4='Renderer(depth_mode="normalized") (generativedensification_amd/renderer.py)',
                     R1="GDR_DEPTH_TO_MEAN=0 / rasterizer.DEPTH_TO_MEAN = False (include/gdr.h GDR_IN_NO_DEPTH_TO_MEAN)",
                     R3="no switch exists: the (N,4) carrier's |.| columns are what network.py:876-878 consumes")


def case_of(dump) -> dict:
    """The dump's inputs, re-created from its seeds exactly as scripts/dump_reference_cuda.py builds them."""
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene
    n, size, deg, seed = (int(dump[k]) for k in ("n", "size", "deg", "seed"))
    sc = make_scene(n, seed, sh_degree=deg)
    cam = orbit_cameras(4, size, size)[1]
    return dict(N=n, H=size, W=size, deg=deg, means3D=sc["centers"].contiguous(), opacities=torch.sigmoid(sc["opacity"]).contiguous(),
                shs=sc["shs"].contiguous(), colors_precomp=None, scales=torch.exp(sc["scales"]).contiguous(),
                rotations=torch.nn.functional.normalize(sc["rotations"]).contiguous(), cov3D_precomp=None,
                view=cam.world_view_transform.contiguous(), proj=cam.full_proj_transform.contiguous(),
                campos=cam.camera_center.contiguous(), bg=torch.ones(3), tanfovx=math.tan(0.375), tanfovy=math.tan(0.375),
                scale_modifier=1.0)


def oracle_backend(case):
    """run(grads, depth_to_mean) -> (outputs, gradients) through the f32 oracle."""
    from oracle.gdr_oracle import Oracle
    o = Oracle("f32", nthreads=4)
    np_ = lambda t: None if t is None else t.numpy()
    ctx = o.forward(np_(case["means3D"]), np_(case["opacities"]), U.settings_np(case), shs=np_(case["shs"]),
                    scales=np_(case["scales"]), rotations=np_(case["rotations"]))

    def run(grads, depth_to_mean):
        return ctx, o.backward(ctx, *[np_(g) for g in grads], depth_to_mean=depth_to_mean)
    return run


def hip_backend(case):
    """The same through the product path (C ABI, libgdr_hip.so)."""
    from generativedensification_amd import rasterizer as R

    def run(grads, depth_to_mean):
        saved = R.DEPTH_TO_MEAN
        R.DEPTH_TO_MEAN = depth_to_mean
        try:
            return U.run_hip(case, grads)
        finally:
            R.DEPTH_TO_MEAN = saved
    return run


def _close(a, ref, what, report, rtol=1e-4):
    out, worst, maxn = U.elem_stats(a, ref, rtol, 1e-6)
    report.append(f"  {what:28s} outside {out:.2e}  worst/tol {worst:8.1f}  max-norm rel {maxn:.2e}")
    return out < U.MAX_OUTSIDE and maxn < 1e-3


def compare(dump, backend) -> tuple[dict, bool, list]:
    """Returns (selected conventions, every bar met under them, report lines)."""
    report = [f"dump: n={int(dump['n'])} {int(dump['size'])}x{int(dump['size'])} deg={int(dump['deg'])} seed={int(dump['seed'])} "
              f"source={str(dump['source']) if 'source' in dump else 'cuda'}"]
    up = [torch.from_numpy(np.asarray(dump[k], np.float32)) for k in ("upstream_color", "upstream_depth", "upstream_alpha")]
    zero = [torch.zeros_like(u) for u in up]
    sel, ok = {}, True
    out, g_all = backend((up[0], up[1], up[2]), True)
    # integer state and images
    same_radii = np.array_equal(np.asarray(out["radii"]), np.asarray(dump["radii"]))
    report.append(f"  radii bit-exact: {same_radii}")
    ok &= same_radii
    ok &= _close(out["color"], dump["color"], "color", report)
    ok &= _close(out["alpha"], dump["alpha"], "alpha", report)
    # R4
    d_sum = np.asarray(out["depth"], np.float64)
    d_norm = d_sum / np.maximum(np.asarray(out["alpha"], np.float64), 1e-10)
    e_sum, e_norm = U.rel_inf(d_sum, dump["depth"]), U.rel_inf(d_norm, dump["depth"])
    sel["R4"] = "sum" if e_sum <= e_norm else "normalized"
    report.append(f"  R4 depth: sum {e_sum:.2e} | normalized {e_norm:.2e} -> {sel['R4']}")
    ok &= min(e_sum, e_norm) < 1e-4
    # R1: the depth-only scalar's means3D gradient with / without the centre path
    _, g_d1 = backend((zero[0], up[1], zero[2]), True)
    _, g_d0 = backend((zero[0], up[1], zero[2]), False)
    ref = np.asarray(dump["grad_depth_only_means3D"])
    if sel["R4"] == "sum" and ref.shape == np.asarray(g_d1["means3D"]).shape:
        e1, e0 = U.elem_stats(g_d1["means3D"], ref)[0], U.elem_stats(g_d0["means3D"], ref)[0]
        sel["R1"] = "yes" if e1 <= e0 else "no"
        report.append(f"  R1 dL/ddepth -> means3D: yes {e1:.2e} | no {e0:.2e} outside -> {sel['R1']}")
    else:
        sel["R1"] = "undecided (depth convention differs: gradients of the depth image are not comparable)"
        report.append("  R1 " + sel["R1"])
    # R3: columns 2:4 of the carrier under the colour-only scalar
    _, g_c = backend((up[0], zero[1], zero[2]), True)
    m_ref, m = np.asarray(dump["grad_color_only_means2D"]), np.asarray(g_c["means2D"])
    if m_ref.ndim == 2 and m_ref.shape[1] == 4:
        cands = {"abs": m[:, 2:4], "signed": m[:, 0:2], "zero": np.zeros_like(m[:, 2:4])}
        errs = {k: U.elem_stats(v, m_ref[:, 2:4])[0] for k, v in cands.items()}
        sel["R3"] = min(errs, key=errs.get)
        report.append("  R3 carrier[:, 2:4]: " + " | ".join(f"{k} {v:.2e}" for k, v in errs.items()) + f" -> {sel['R3']}")
    else:
        sel["R3"] = f"undecided (carrier gradient has shape {m_ref.shape})"
        report.append("  R3 " + sel["R3"])
    # every gradient of the three scalars under the selected R1 (signed columns always; |.| columns if R3 = abs)
    d2m = sel["R1"] != "no"
    runs = {"all": g_all if d2m else backend((up[0], up[1], up[2]), False)[1], "color_only": g_c, "depth_only": g_d1 if d2m else g_d0}
    for tag, g in runs.items():
        if tag != "color_only" and sel["R4"] != "sum":
            continue
        for k in GRAD_KEYS:
            ref = np.asarray(dump[f"grad_{tag}_{k}"])
            mine = np.asarray(g[k]).reshape(-1, *np.asarray(g[k]).shape[1:])
            if ref.size <= 1:
                continue
            if k == "means2D":
                cols = 4 if sel["R3"] == "abs" and ref.shape[1] == 4 else 2
                ok &= _close(mine[:, :cols], ref[:, :cols], f"grad[{tag}] {k}[:, :{cols}]", report)
            else:
                ok &= _close(mine.reshape(ref.shape), ref, f"grad[{tag}] {k}", report)
    differs = {k: v for k, v in sel.items() if DEFAULTS.get(k) != v}
    for k, v in differs.items():
        report.append(f"  !! {k}: the data selects '{v}', the product default is '{DEFAULTS[k]}' — {HOW_TO_SWITCH[k]}")
    return sel, ok, report
