"""Drop-in for the `diff_surfel_rasterization` package the reference's 2DGS adaptor imports
(/root/reference/lightning/renderer_2dgs.py:7-10; absent from the reference tree and .gitmodules), backed by the
MI355X HIP library.  Implementation: generativedensification_amd/surfel_rasterizer.py -> libgdr_hip.so (include/gsr.h).
"""
from generativedensification_amd.surfel_rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
