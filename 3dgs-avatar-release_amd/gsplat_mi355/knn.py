"""`knn_points` with the call shape of pytorch3d.ops.knn_points as the reference uses it:
utils/loss_utils.py:76-79,92-96 (K-nearest canonical Gaussians of every canonical Gaussian, for the AIAP
losses) and models/deformer/rigid.py:43 (nearest SMPL vertex of every Gaussian).  Exact search on the GPU
(libgsplat_mi355: Morton order + boxes, the machinery of distCUDA2); no CPU path.
"""
import collections
import ctypes

import torch

from . import _lib

KNN = collections.namedtuple("KNN", ["dists", "idx", "knn"])


def knn_points(p1, p2, K=1, return_sorted=True, return_nn=False):
    """p1: (1, N1, 3) or (N1, 3) queries, p2: (1, N2, 3) or (N2, 3) reference points, fp32 on the GPU.
    Returns KNN(dists (1, N1, K) squared distances ascending, idx (1, N1, K) int64 into p2, knn) -- the batch
    dimension is kept when the inputs have one.  Batch size 1 only (all the reference's calls); the distances
    carry no gradient (the reference only uses the indices and recomputes distances with cdist).  Results are
    always sorted (`return_sorted` is accepted for signature compatibility)."""
    batched = p1.dim() == 3
    if batched:
        if p1.shape[0] != 1 or p2.dim() != 3 or p2.shape[0] != 1:
            raise NotImplementedError("knn_points: batch size 1 only")
        q, r = p1[0], p2[0]
    else:
        q, r = p1, p2
    if not (q.is_cuda and r.is_cuda):
        raise RuntimeError("knn_points: tensors must live on the GPU (no CPU fallback)")
    if q.dtype != torch.float32 or r.dtype != torch.float32 or q.shape[-1] != 3 or r.shape[-1] != 3:
        raise TypeError("knn_points: (N, 3) fp32 points expected")
    K = int(K)
    if K < 1 or K > 8:
        raise NotImplementedError("knn_points: 1 <= K <= 8")
    same = q.data_ptr() == r.data_ptr() and q.shape == r.shape
    qd = q.detach().contiguous()
    rd = qd if same else r.detach().contiguous()
    n1, n2 = int(qd.shape[0]), int(rd.shape[0])
    if n2 == 0:
        raise ValueError("knn_points: empty reference set")
    L = _lib.load()
    dists = torch.empty(n1, K, dtype=torch.float32, device=qd.device)
    idx = torch.empty(n1, K, dtype=torch.int64, device=qd.device)
    ws = torch.empty(_lib.nbytes(L.knn_workspace_bytes, n2), dtype=torch.uint8, device=qd.device)
    with _lib.on_device(qd.device):
        sptr = _lib.stream_ptr(qd.device)
        _lib.check(L.knn_points(n1, _lib.ptr(qd), n2, _lib.ptr(rd), K, _lib.ptr(dists), _lib.ptr(idx), ws.data_ptr(),
                                ws.numel(), sptr))
    nn = None
    if return_nn:
        nn = r[idx.clamp_min(0)]
    if batched:
        return KNN(dists[None], idx[None], nn[None] if nn is not None else None)
    return KNN(dists, idx, nn)
