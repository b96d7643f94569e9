"""Drop-in for the reference's `diff_gaussian_rasterization` package (the un-vendored CUDA
extension of /root/reference/.gitmodules:1-3) backed by the MI355X HIP library.

`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(/root/reference/lightning/renderer.py:10-13) works unchanged with this directory on the
path.  Implementation: generativedensification_amd/rasterizer.py -> libgdr_hip.so.
"""
from generativedensification_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
