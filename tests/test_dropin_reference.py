"""CPU, authoring container only: the REFERENCE's caller modules import and run unchanged
against this repo's `diff_gaussian_rasterization` package (the drop-in test of SURVEY §0-3).
Skipped where /root/reference does not exist (the GPU box)."""
import importlib
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lightning")), reason="reference tree not present")


def _fresh_import(name):
    for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
        del sys.modules[k]
    return importlib.import_module(name)


def test_reference_renderer_imports_against_product_package_and_fails_loudly_on_cpu():
    sys.modules.pop("diff_gaussian_rasterization", None)
    import diff_gaussian_rasterization as D  # the PRODUCT package (HIP only)

    assert not getattr(D, "__oracle_standin__", False)
    sys.path.insert(0, REF)
    try:
        ref_renderer = _fresh_import("lightning.renderer")
        ref_utils = _fresh_import("lightning.utils")
        assert ref_renderer.GaussianRasterizer is D.GaussianRasterizer
        c2w = torch.eye(4)
        c2w[2, 3] = -2.0
        cam = ref_utils.MiniCam(c2w, 32, 32, torch.tensor(0.75), torch.tensor(0.75), 0.5, 2.5, "cpu")
        r = ref_renderer.Renderer(sh_degree=1)
        n = 16
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            r.render_img(cam, None, torch.zeros(n, 3), torch.zeros(n, 4, 3), torch.zeros(n, 1), torch.zeros(n, 3),
                         torch.ones(n, 4), "cpu")
        rast = r.set_rasterizer(cam, device="cpu")  # settings are built BY KEYWORD with the 12 names
        assert rast.raster_settings.image_height == 32 and rast.raster_settings.sh_degree == 1
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k.startswith("lightning")]:
            del sys.modules[k]


def test_reference_renderer_runs_on_cpu_with_oracle_standin(oracle_built):
    from oracle.gdr_oracle import make_standin_module

    saved = sys.modules.get("diff_gaussian_rasterization")
    sys.modules["diff_gaussian_rasterization"] = make_standin_module("f32")
    sys.path.insert(0, REF)
    try:
        ref_renderer = _fresh_import("lightning.renderer")
        ref_utils = _fresh_import("lightning.utils")
        from generativedensification_amd.synthetic import make_scene

        sc = make_scene(300, 4, sh_degree=3, sigma0=(0.03,))
        c2w = torch.eye(4)
        c2w[2, 3] = -2.0
        cam = ref_utils.MiniCam(c2w, 48, 32, torch.tensor(0.75), torch.tensor(0.75), 0.5, 3.5, "cpu")
        out = ref_renderer.Renderer(sh_degree=3).render_img(cam, None, sc["centers"], sc["shs"], sc["opacity"],
                                                            sc["scales"], sc["rotations"], "cpu", prex="_fine")
        assert out["image_fine"].shape == (32, 48, 3) and out["depth_fine"].shape == (32, 48, 1)
        assert out["acc_map_fine"].shape == (32, 48)
        assert 0.0 <= float(out["image_fine"].detach().min()) and float(out["image_fine"].detach().max()) <= 1.0
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k.startswith("lightning")]:
            del sys.modules[k]
        if saved is not None:
            sys.modules["diff_gaussian_rasterization"] = saved
        else:
            sys.modules.pop("diff_gaussian_rasterization", None)


# ---- 2DGS adaptor (lightning/renderer_2dgs.py) against the product `diff_surfel_rasterization` ---------------
def _import_ref_2dgs():
    from oracle.gsr_oracle import make_simple_knn_stub

    # renderer_2dgs.py:11 imports simple_knn._C at module import; the reference tree does not contain it
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = make_simple_knn_stub()
    return _fresh_import("lightning.renderer_2dgs"), _fresh_import("lightning.utils")


def _cleanup_2dgs(saved):
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k.startswith("lightning") or k.startswith("simple_knn")]:
        del sys.modules[k]
    if saved is not None:
        sys.modules["diff_surfel_rasterization"] = saved
    else:
        sys.modules.pop("diff_surfel_rasterization", None)


def test_reference_2dgs_adaptor_imports_against_product_package_and_fails_loudly_on_cpu():
    """No stand-ins at all: lightning/renderer_2dgs.py imports the PRODUCT `diff_surfel_rasterization` and the PRODUCT
    `simple_knn._C.distCUDA2` (both HIP only)."""
    saved = sys.modules.pop("diff_surfel_rasterization", None)
    for k in [k for k in sys.modules if k.startswith("simple_knn")]:
        del sys.modules[k]
    import diff_surfel_rasterization as D  # the PRODUCT package (HIP only)

    assert not getattr(D, "__oracle_standin__", False)
    sys.path.insert(0, REF)
    try:
        ref_2dgs, ref_utils = _fresh_import("lightning.renderer_2dgs"), _fresh_import("lightning.utils")
        from generativedensification_amd.knn import dist2

        assert ref_2dgs.distCUDA2 is dist2
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ref_2dgs._activation_scale(torch.zeros(8, 3))
        assert ref_2dgs.GaussianRasterizer is D.GaussianRasterizer
        c2w = torch.eye(4)
        c2w[2, 3] = -2.0
        cam = ref_utils.MiniCam(c2w, 32, 32, torch.tensor(0.75), torch.tensor(0.75), 0.5, 2.5, "cpu")
        r = ref_2dgs.Renderer(sh_degree=1)
        n = 16
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            r.render_img(cam, None, torch.zeros(n, 3), torch.zeros(n, 4, 3), torch.zeros(n, 1), torch.zeros(n, 2),
                         torch.ones(n, 4), "cpu")
        rast = r.set_rasterizer(cam, device="cpu")
        assert rast.raster_settings.image_width == 32 and rast.raster_settings.sh_degree == 1
    finally:
        _cleanup_2dgs(saved)


def test_reference_2dgs_adaptor_runs_on_cpu_with_oracle_standin(oracle_built):
    from oracle.gsr_oracle import make_surfel_standin_module

    saved = sys.modules.get("diff_surfel_rasterization")
    sys.modules["diff_surfel_rasterization"] = make_surfel_standin_module("f32")
    sys.path.insert(0, REF)
    try:
        ref_2dgs, ref_utils = _import_ref_2dgs()
        from generativedensification_amd.camera import build_rays
        from generativedensification_amd.synthetic import make_scene

        sc = make_scene(300, 4, sh_degree=3, sigma0=(0.03,))
        c2w = torch.eye(4)
        c2w[2, 3] = -2.0
        cam = ref_utils.MiniCam(c2w, 48, 32, torch.tensor(0.75), torch.tensor(0.75), 0.5, 3.5, "cpu")
        out = ref_2dgs.Renderer(sh_degree=3).render_img(cam, build_rays(c2w, 0.75, 0.75, 32, 48), sc["centers"], sc["shs"],
                                                        sc["opacity"], sc["scales"][:, :2].contiguous(), sc["rotations"],
                                                        "cpu", prex="_fine")
        assert out["image_fine"].shape == (32, 48, 3) and out["depth_fine"].shape == (32, 48, 1)
        assert out["rend_normal_fine"].shape == (32, 48, 3) and out["rend_dist_fine"].shape == (32, 48)
        assert float(out["acc_map_fine"].detach().max()) > 0.1 and torch.isfinite(out["depth_normal_fine"]).all()
    finally:
        _cleanup_2dgs(saved)
