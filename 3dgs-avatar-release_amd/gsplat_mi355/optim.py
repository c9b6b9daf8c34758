"""Training-step bookkeeping after the backward pass, fused.

* `FusedAdam` -- a drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` the reference builds over its six
  parameter groups (scene/gaussian_model.py:201-216): same constructor arguments, same `param_groups`, same state
  keys (`step`, `exp_avg`, `exp_avg_sq`) -- the reference's densification code edits those state tensors directly
  (scene/gaussian_model.py: cat_tensors_to_optimizer / _prune_optimizer) -- but `step()` is ONE kernel launch
  for all parameters.
* `densify_stats` -- train.py:219-220 + scene/gaussian_model.py:464-466 in one kernel, without the boolean-mask
  indexing (and its host sync) of the torch formulation.
GPU fp32 tensors only.
"""
import ctypes

import torch

from . import _lib


def densify_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """In place, for Gaussians with radii > 0: max_radii2D = max(max_radii2D, radii);
    xyz_gradient_accum += |viewspace_grad[:, :2]|; denom += 1."""
    n = int(radii.shape[0])
    for t, name in ((viewspace_grad, "viewspace_grad"), (max_radii2D, "max_radii2D"),
                    (xyz_gradient_accum, "xyz_gradient_accum"), (denom, "denom")):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("densify_stats: %s must be a contiguous fp32 GPU tensor" % name)
    if radii.dtype != torch.int32 or not radii.is_cuda or not radii.is_contiguous():
        raise RuntimeError("densify_stats: radii must be a contiguous int32 GPU tensor")
    if tuple(viewspace_grad.shape) != (n, 3) or max_radii2D.numel() != n or xyz_gradient_accum.numel() != n or denom.numel() != n:
        raise ValueError("densify_stats: shapes do not match N = %d" % n)
    L = _lib.load()
    with _lib.on_device(radii.device):
        sptr = _lib.stream_ptr(radii.device)
        _lib.check(L.gs_densify_stats(n, _lib.ptr(radii), _lib.ptr(viewspace_grad), _lib.ptr(max_radii2D),
                                      _lib.ptr(xyz_gradient_accum), _lib.ptr(denom), sptr))
    _bump_versions(max_radii2D, xyz_gradient_accum, denom)


def _bump_versions(*tensors):
    """The HIP kernels write through raw pointers, which torch cannot see: bump the autograd version counters as any
    in-place torch op would, so that whatever keys on them (saved-tensor checks, the rasterizer's shared-geometry
    matching) sees the write."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (betas, eps, per-group lr; no weight decay, no amsgrad, no maximize) with a single-launch
    `step()`.  State layout identical to torch.optim.Adam's (`step` tensor, `exp_avg`, `exp_avg_sq`)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam: weight_decay / amsgrad are not used by the reference and not implemented")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        # tensors that share (betas, eps, step number, device) go into one launch
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: contiguous fp32 GPU parameters expected")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam: sparse gradients are not supported")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                step = int(st["step"].item()) if st["step"].device.type == "cpu" else int(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (float(b1), float(b2), float(group["eps"]), step, p.device.index)
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (b1, b2, eps, step, dev_index), items in batches.items():
            dev = torch.device("cuda", dev_index)
            for i in range(0, len(items), _lib.GS_ADAM_MAX_TENSORS):
                chunk = items[i:i + _lib.GS_ADAM_MAX_TENSORS]
                arr = (_lib.GsAdamTensor * len(chunk))()
                for k, (p, g, m, v, lr) in enumerate(chunk):
                    arr[k] = _lib.GsAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr)
                with _lib.on_device(dev):
                    sptr = _lib.stream_ptr(dev)
                    _lib.check(L.gs_adam_step(len(chunk), arr, b1, b2, eps, step, sptr))
                for p, _g, m, v, _lr in chunk:
                    _bump_versions(p, m, v)
        return loss
