"""`simple_knn._C.distCUDA2` for ROCm: mean squared distance to the 3 nearest neighbours of every point
(/root/reference/lightning/renderer_2dgs.py:11,92-96; point_decoder/layers/head.py:115).  HIP kernels in csrc/knn.hip
behind include/gsr.h; torch does the plumbing between the two stages (bounding box, sort by cell, cell prefix)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def dist2(points: torch.Tensor, cells_per_axis: int | None = None) -> torch.Tensor:
    """points (N,3) on a HIP device -> (N,) fp32, (d1 + d2 + d3) / 3 of the three nearest OTHER points (inf terms when
    fewer than 4 points exist, as the lineage's initial `best = FLT_MAX` would leave)."""
    if not points.is_cuda:
        raise RuntimeError("simple_knn.distCUDA2 (MI355X build) runs on ROCm/HIP device tensors only; no CPU fallback")
    lib = L.load()
    dev = points.device
    pts = points.detach().to(torch.float32).reshape(-1, 3).contiguous()
    N = int(pts.shape[0])
    out = torch.empty(N, dtype=torch.float32, device=dev)
    if N == 0:
        return out
    G = int(cells_per_axis) if cells_per_axis else max(1, min(256, int(round((N / 2.0) ** (1.0 / 3.0)))))
    with torch.no_grad(), torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        bbox = torch.cat([pts.min(0).values, pts.max(0).values]).contiguous()
        cell = torch.empty(N, dtype=torch.int32, device=dev)
        L.check(lib.gsr_knn_cells(pts.data_ptr(), N, bbox.data_ptr(), G, cell.data_ptr(), stream), "gsr_knn_cells")
        order = torch.argsort(cell.long(), stable=True)
        pts_sorted = pts.index_select(0, order).contiguous()
        counts = torch.bincount(cell.long(), minlength=G * G * G)
        cell_start = torch.zeros(G * G * G + 1, dtype=torch.int32, device=dev)
        cell_start[1:] = torch.cumsum(counts, 0).to(torch.int32)
        out_sorted = torch.empty(N, dtype=torch.float32, device=dev)
        L.check(lib.gsr_knn_mean_dist2(pts_sorted.data_ptr(), N, bbox.data_ptr(), G, cell_start.data_ptr(),
                                       out_sorted.data_ptr(), stream), "gsr_knn_mean_dist2")
        out[order] = out_sorted
    return out
