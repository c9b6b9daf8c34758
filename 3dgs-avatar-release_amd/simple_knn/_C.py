"""simple_knn._C.distCUDA2: mean squared distance to the three nearest other points, on gfx950."""
import ctypes

import torch

from gsplat_mi355 import _lib


def distCUDA2(points):
    """points (N,3) float32 GPU tensor -> (N,) float32 (call site: scene/gaussian_model.py:186)."""
    L = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("simple_knn.distCUDA2: points must be a GPU tensor (this build has no CPU path)")
    pts = points.detach().contiguous().float()
    P = int(pts.shape[0])
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        ws_bytes = _lib.nbytes(L.knn_workspace_bytes, P)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pts.device)
        sptr = _lib.stream_ptr(pts.device)
        _lib.check(L.knn_dist2(P, _lib.ptr(pts), _lib.ptr(out), ws.data_ptr(), ws_bytes, sptr))
    return out
