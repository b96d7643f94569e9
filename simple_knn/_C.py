"""`simple_knn._C` — distCUDA2(points (N,3)) -> (N,) mean squared distance to the 3 nearest neighbours."""
from generativedensification_amd.knn import dist2 as distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
