"""`render()`-shaped harness: reproduces, in order, the call sequence of the reference's
gaussian_renderer/__init__.py:59-153 on top of the drop-in `diff_gaussian_rasterization` package, and
the part of the train step that touches the rasterizer (train.py:114-124,143-153,179,217-220).
The deformer / texture modules upstream of the seam are out of scope: the Gaussian
state arrives post-activation as a `GaussianCloud`.
"""
import ctypes
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

from . import _lib

_DEBUG = __import__("os").environ.get("GSPLAT_DEBUG", "0") == "1"


class Pipe(object):
    """pipeline.* keys the renderer reads (configs/config.yaml:89-92)."""

    def __init__(self, compute_cov3D_python=False, convert_SHs_python=False, debug=False, fuse_opacity=False):
        self.compute_cov3D_python = compute_cov3D_python
        self.convert_SHs_python = convert_SHs_python
        self.debug = debug
        # not a reference key: render the opacity image inside the colour pass (one rasterizer call with
        # with_opacity=True) instead of the reference's second call with colours = 1
        self.fuse_opacity = fuse_opacity


class RenderPackage(object):
    """Result holder with the fields of gaussian_renderer/__init__.py:20-55,146-153.  `visibility_filter` (= radii > 0,
    :149) is formed when it is first read: a training step that only feeds `radii` to the fused densification
    statistics never launches the comparison."""

    def __init__(self, **data):
        self.data = data

    def _get(self, item):
        data = self.__dict__["data"]
        if item == "visibility_filter" and item not in data:
            data[item] = data["radii"] > 0
        return data[item]

    def __getitem__(self, item):
        return self._get(item)

    def __getattr__(self, item):
        try:
            return self._get(item)
        except KeyError:
            raise AttributeError(item)


_zeros = {}  # device -> (shape, dtype, an all-zero tensor nobody writes): ONE per device, replaced when the shape changes


def _zero_leaf(like):
    """A fresh leaf over a cached all-zero storage: it only exists to receive dL/dmeans2D in `.grad` (the rasterizer never
    reads its values), so its VALUES are shared between frames and must be treated as read-only by callers -- the
    reference's own tensor was a private `zeros + 0`.  One cached buffer per device (densification changes N every few
    hundred steps: older shapes are dropped, not kept).  GSPLAT_DEBUG=1: every call checks that the storage is still zero."""
    ent = _zeros.get(like.device)
    if ent is None or ent[0] != tuple(like.shape) or ent[1] != like.dtype:
        ent = _zeros[like.device] = (tuple(like.shape), like.dtype, torch.zeros(like.shape, dtype=like.dtype, device=like.device))
    elif _DEBUG and bool(ent[2].any()):
        raise RuntimeError("gsplat_mi355.render: the shared all-zero storage behind `viewspace_points` was written to "
                           "(its values are read-only; the gradient arrives in `.grad`)")
    return ent[2].detach().requires_grad_(True)


def render(data, pc, pipe, bg_color, scaling_modifier=1.0, colors_precomp=None, return_opacity=False, l1_target=None):
    """data: camera (FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
    camera_center); pc: tensors named as GaussianModel's getters (xyz, opacity, scales, rotations, shs).
    If `colors_precomp` is given it is used (the reference always does: gaussian_renderer/__init__.py:117-124);
    otherwise SHs are handed to the rasterizer for in-kernel SH->RGB (north_star).
    `l1_target` (not a reference argument): the ground-truth image; the package then carries `l1` = mean |render -
    l1_target|, the reference's `l1_loss(image, gt_image)` (train.py:121), computed and differentiated inside the
    rasterizer's own launches (GaussianRasterizer.forward, `l1_target`)."""
    xyz = pc.xyz
    # gaussian_renderer/__init__.py:76-80 builds `zeros_like(xyz, requires_grad=True) + 0` and retains its gradient: a
    # tensor whose only purpose is to receive dL/dmeans2D in `.grad`.  A zero LEAF does that with one fill launch instead
    # of fill + add + the clone AddBackward makes for the retained gradient (the rasterizer never reads its values).
    # The leaf is a fresh tensor object over a cached all-zero storage nobody writes to (the gradient goes to `.grad`):
    # no fill launch per frame.
    screenspace_points = _zero_leaf(xyz)
    tanfovx = math.tan(data.FoVx * 0.5)
    tanfovy = math.tan(data.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(data.image_height), image_width=int(data.image_width), tanfovx=tanfovx, tanfovy=tanfovy,
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=data.world_view_transform,
        projmatrix=data.full_proj_transform, sh_degree=pc.sh_degree, campos=data.camera_center, prefiltered=False,
        debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D = xyz
    means2D = screenspace_points
    opacity = pc.opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.covariance6(scaling_modifier)
    else:
        scales, rotations = pc.scales, pc.rotations
    if colors_precomp is None and pipe.convert_SHs_python:
        # the reference's texture module (models/texture/texture.py:21-38), fused: colours from SHs seen from the
        # camera centre, in the canonical frame when the cloud carries the deformer's forward transform
        from .prepass import sh2rgb
        colors_precomp = sh2rgb(pc.shs, xyz, data.camera_center, pc.sh_degree,
                                fwd_transform=getattr(pc, "fwd_transform", None))
    shs = None if colors_precomp is not None else pc.shs
    opacity_image = None
    l1 = None
    if return_opacity and getattr(pipe, "fuse_opacity", False):
        res = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
                         scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, with_opacity=True,
                         l1_target=l1_target)
        rendered_image, radii, opacity_image = res[:3]
    else:
        res = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
                         scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, l1_target=l1_target)
        rendered_image, radii = res[:2]
    if l1_target is not None:
        l1 = res[-1]
    if return_opacity and opacity_image is None:
        opacity_image, _ = rasterizer(means3D=means3D, means2D=means2D, shs=None,
                                      colors_precomp=torch.ones(opacity.shape[0], 3, device=opacity.device),
                                      opacities=opacity, scales=scales, rotations=rotations,
                                      cov3D_precomp=cov3D_precomp)
        opacity_image = opacity_image[:1]
    return RenderPackage(deformed_gaussian=pc, render=rendered_image, viewspace_points=screenspace_points,
                         radii=radii, opacity_render=opacity_image, l1=l1)


class _L1Loss(torch.autograd.Function):
    """mean |x - y| with d/dx = sign(x - y) / n written by the same HIP pass (gs_l1_loss)."""

    @staticmethod
    def forward(ctx, x, y):
        L = _lib.load()
        n = x.numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        ws = torch.empty(_lib.nbytes(L.gs_l1_loss_workspace_bytes, n), dtype=torch.uint8, device=x.device)
        with _lib.on_device(x.device):
            sptr = _lib.stream_ptr(x.device)
            _lib.check(L.gs_l1_loss(n, x.data_ptr(), y.data_ptr(), loss.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                                    ws.numel(), sptr))
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        gx = grad * g
        return (gx if ctx.needs_input_grad[0] else None), (-gx if ctx.needs_input_grad[1] else None)


class _BceLoss(torch.autograd.Function):
    """mean BCE of clamp(x, 1e-3, 1 - 1e-3) against y with d/dx from the same HIP pass (gs_bce_loss)."""

    @staticmethod
    def forward(ctx, x, y):
        L = _lib.load()
        n = x.numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        ws = torch.empty(_lib.nbytes(L.gs_l1_loss_workspace_bytes, n), dtype=torch.uint8, device=x.device)
        with _lib.on_device(x.device):
            sptr = _lib.stream_ptr(x.device)
            _lib.check(L.gs_bce_loss(n, x.data_ptr(), y.data_ptr(), loss.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                                     ws.numel(), sptr))
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def bce_mask_loss(opacity, gt_mask):
    """train.py:146-148: F.binary_cross_entropy(torch.clamp(opacity, 1e-3, 1 - 1e-3), gt_mask) as one fused HIP pass
    (the 'bce' form of the mask loss; the default 'l1' form is `l1_loss`).  The mask gets no gradient."""
    if not (opacity.is_cuda and gt_mask.is_cuda):
        raise RuntimeError("bce_mask_loss: both tensors must live on the GPU (no CPU fallback)")
    if opacity.dtype != torch.float32 or gt_mask.dtype != torch.float32:
        raise TypeError("bce_mask_loss: fp32 tensors expected")
    if opacity.shape != gt_mask.shape or opacity.numel() == 0:
        raise ValueError("bce_mask_loss: equal, non-empty shapes expected")
    return _BceLoss.apply(opacity.contiguous(), gt_mask.contiguous())


def l1_loss(network_output, gt):
    """torch.abs(network_output - gt).mean() of utils/loss_utils.py:21-22 as ONE fused HIP pass
    (SURVEY.md 8f row N2).  Device fp32 tensors only: there is no CPU path."""
    if not (network_output.is_cuda and gt.is_cuda):
        raise RuntimeError("l1_loss: both tensors must live on the GPU (the fused HIP kernel has no CPU fallback)")
    if network_output.dtype != torch.float32 or gt.dtype != torch.float32:
        raise TypeError("l1_loss: fp32 tensors expected")
    if network_output.numel() == 0:
        raise ValueError("l1_loss: empty input")
    if network_output.shape != gt.shape:
        network_output, gt = torch.broadcast_tensors(network_output, gt)
    return _L1Loss.apply(network_output.contiguous(), gt.contiguous())


class _Ssim(torch.autograd.Function):
    """Mean SSIM of two (C,H,W) images; gradient w.r.t. the first (gs_ssim_forward / gs_ssim_backward)."""

    @staticmethod
    def forward(ctx, img1, img2):
        L = _lib.load()
        C, H, W = (int(v) for v in img1.shape)
        out = torch.empty((), dtype=torch.float32, device=img1.device)
        need = img1.requires_grad
        maps = torch.empty((3, C, H, W), dtype=torch.float32, device=img1.device) if need else None
        ws = torch.empty(_lib.nbytes(L.gs_ssim_workspace_bytes, C, H, W), dtype=torch.uint8, device=img1.device)
        with _lib.on_device(img1.device):
            sptr = _lib.stream_ptr(img1.device)
            _lib.check(L.gs_ssim_forward(C, H, W, img1.data_ptr(), img2.data_ptr(), out.data_ptr(),
                                         _lib.ptr(maps[0]) if need else None, _lib.ptr(maps[1]) if need else None,
                                         _lib.ptr(maps[2]) if need else None, ws.data_ptr(), ws.numel(), sptr))
        if need:
            ctx.save_for_backward(img1, img2, maps)
        return out

    @staticmethod
    def backward(ctx, g):
        img1, img2, maps = ctx.saved_tensors
        L = _lib.load()
        C, H, W = (int(v) for v in img1.shape)
        grad = torch.empty_like(img1)
        g = g.to(torch.float32).contiguous()
        with _lib.on_device(img1.device):
            sptr = _lib.stream_ptr(img1.device)
            _lib.check(L.gs_ssim_backward(C, H, W, img1.data_ptr(), img2.data_ptr(), maps[0].data_ptr(),
                                          maps[1].data_ptr(), maps[2].data_ptr(), g.data_ptr(), grad.data_ptr(), sptr))
        return grad, None


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:37-67 `ssim` (11x11 Gaussian window, sigma 1.5, zero padding) as two fused HIP kernels
    with the gradient w.r.t. `img1` (train.py:123 uses `1.0 - ssim(image, gt_image)`).
    (C,H,W) or (1,C,H,W) fp32 device tensors; the target image gets no gradient (it never needs one in the
    reference).  Only the reference's defaults (window 11, mean over everything) are implemented."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("ssim: only window_size=11, size_average=True (the reference's call sites) are implemented")
    if not (img1.is_cuda and img2.is_cuda):
        raise RuntimeError("ssim: both tensors must live on the GPU (the fused HIP kernels have no CPU fallback)")
    if img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise TypeError("ssim: fp32 tensors expected")
    if img1.shape != img2.shape:
        raise ValueError("ssim: shapes differ")
    if img1.dim() == 4 and img1.shape[0] == 1:
        img1, img2 = img1[0], img2[0]
    if img1.dim() != 3 or img1.numel() == 0:
        raise ValueError("ssim: (C,H,W) images expected")
    if img2.requires_grad:
        raise NotImplementedError("ssim: gradient w.r.t. the second image is not implemented")
    return _Ssim.apply(img1.contiguous(), img2.contiguous())


class DensifyStats(object):
    """max_radii2D / xyz_gradient_accum / denom bookkeeping of train.py:217-220 and
    scene/gaussian_model.py:464-466."""

    def __init__(self, n, device):
        self.max_radii2D = torch.zeros(n, device=device)
        self.xyz_gradient_accum = torch.zeros(n, 1, device=device)
        self.denom = torch.zeros(n, 1, device=device)

    def update(self, pkg):
        from .optim import densify_stats  # one fused kernel, no boolean-mask indexing (row N4)
        densify_stats(pkg.radii, pkg.viewspace_points.grad, self.max_radii2D, self.xyz_gradient_accum, self.denom)


def train_step(data, pc, pipe, bg_color, gt_image, gt_mask=None, lambda_mask=0.0, stats=None, fuse_l1=False):
    """One forward+backward of the rasterizer part of the reference's train step: L1 image loss
    (train.py:121), optional L1 mask loss on the opacity render (train.py:143-153), .backward()
    (train.py:179), densification statistics (train.py:219-220).  fuse_l1: the image loss comes out of the rasterizer
    itself (render(..., l1_target=gt_image)) instead of a loss call on the rendered image."""
    use_mask = lambda_mask > 0.0 and gt_mask is not None
    pkg = render(data, pc, pipe, bg_color, return_opacity=use_mask, l1_target=gt_image if fuse_l1 else None)
    loss = pkg.l1 if fuse_l1 else l1_loss(pkg.render, gt_image)
    if use_mask:
        loss = loss + lambda_mask * l1_loss(pkg.opacity_render, gt_mask)
    loss.backward()
    if stats is not None:
        with torch.no_grad():
            stats.update(pkg)
    return loss, pkg
