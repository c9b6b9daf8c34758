"""Multi-GPU: frames of a pose / camera sequence are independent (the reference renders them in a
serial loop with no carried state, render.py:51-62), so they are partitioned over one process per
GPU.  The only exchange step is ONE broadcast of the Gaussian state (RCCL over xGMI when the backend
is "nccl"; "gloo" on CPU for tests): tensors 1-6 of GaussianModel.capture()
(scene/gaussian_model.py:98-112) packed into one flat fp32 buffer (236 B per Gaussian at SH degree 3:
47.2 MB at 200k).  No per-step collective; outputs stay on the rank that rendered them.
"""
import torch

from .scenes import GaussianCloud


def frames_of_rank(rank, world, num_frames_per_rank=None, total=None):
    """Round-robin partition: rank r owns frames r, r + world, r + 2*world, ...
    Either `num_frames_per_rank` frames (weak scaling) or all owned frames below `total`."""
    if total is not None:
        return list(range(rank, total, world))
    return [rank + i * world for i in range(num_frames_per_rank)]


def broadcast_cloud(cloud, n, sh_degree, device, src=0, group=None):
    """One collective: rank `src` sends its packed Gaussian state, every rank returns a GaussianCloud
    on `device`.  `cloud` may be None on the receiving ranks."""
    import torch.distributed as dist
    numel = GaussianCloud.packed_numel(n, sh_degree)
    if dist.get_rank(group) == src:
        flat = cloud.pack().detach().to(device=device, dtype=torch.float32).contiguous()
        assert flat.numel() == numel
    else:
        flat = torch.empty(numel, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src, group=group)
    return GaussianCloud.unpack(flat, n, sh_degree)


def render_sequence(cloud, cameras, render_fn, rank, world):
    """Renders this rank's share of `cameras` (list indexed by frame); returns {frame: result}."""
    out = {}
    for f in frames_of_rank(rank, world, total=len(cameras)):
        out[f] = render_fn(cameras[f], cloud)
    return out
