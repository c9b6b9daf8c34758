"""The two per-Gaussian torch chains in front of the rasterizer, as fused HIP ops with autograd
(SURVEY.md 8f row N3): same names, argument meaning and results as the reference's functions.

* `build_covariance_from_scaling_rotation` -- scene/gaussian_model.py:28-32 (the covariance_activation of
  GaussianModel, reached through get_covariance() at :154-157 with either `_rotation` quaternions or the
  deformer's `rotation_precomp` matrices).
* `sh2rgb` -- models/texture/texture.py:21-38 (SH2RGB.forward).
Device fp32 tensors only: there is no CPU path.
"""
import ctypes

import torch

from . import _lib


def _dev32(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (the fused HIP kernels have no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s: fp32 tensor expected" % name)
    return t.contiguous()


def _stream(dev):
    return _lib.stream_ptr(dev)


class _BuildCovariance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, scaling_modifier, is_matrix):
        L = _lib.load()
        n = int(scaling.shape[0])
        cov = torch.empty(n, 6, dtype=torch.float32, device=scaling.device)
        with _lib.on_device(scaling.device):
            _lib.check(L.gs_build_covariance(n, _lib.ptr(scaling), float(scaling_modifier), _lib.ptr(rotation),
                                             1 if is_matrix else 0, _lib.ptr(cov), _stream(scaling.device)))
        ctx.save_for_backward(scaling, rotation)
        ctx.mod, ctx.is_matrix = float(scaling_modifier), bool(is_matrix)
        return cov

    @staticmethod
    def backward(ctx, g):
        scaling, rotation = ctx.saved_tensors
        L = _lib.load()
        n = int(scaling.shape[0])
        g = g.to(torch.float32).contiguous()
        ds = torch.empty_like(scaling)
        dr = torch.empty_like(rotation)
        with _lib.on_device(scaling.device):
            _lib.check(L.gs_build_covariance_backward(n, _lib.ptr(scaling), ctx.mod, _lib.ptr(rotation),
                                                      1 if ctx.is_matrix else 0, _lib.ptr(g), _lib.ptr(ds), _lib.ptr(dr),
                                                      _stream(scaling.device)))
        return ds, dr, None, None


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """strip_symmetric(L @ L^T) with L = R @ diag(scaling_modifier * scaling): (N,6) = [xx, xy, xz, yy, yz, zz].
    `rotation` is (N,4) quaternions (w,x,y,z; normalised as utils/general_utils.py:87-108 does) or (N,3,3)
    matrices (`rotation_precomp`), exactly as utils/general_utils.py:194-207 dispatches on the last dimension."""
    scaling = _dev32(scaling, "scaling")
    rotation = _dev32(rotation, "rotation")
    if scaling.dim() != 2 or scaling.shape[1] != 3:
        raise ValueError("scaling: (N,3) expected")
    is_matrix = rotation.shape[-1] != 4
    if is_matrix and tuple(rotation.shape[1:]) != (3, 3):
        raise ValueError("rotation: (N,4) quaternions or (N,3,3) matrices expected")
    if rotation.shape[0] != scaling.shape[0]:
        raise ValueError("scaling / rotation: different N")
    return _BuildCovariance.apply(scaling, rotation, scaling_modifier, is_matrix)


class _Sh2Rgb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, xyz, campos, fwd_rot, noise, deg):
        L = _lib.load()
        n, m = int(features.shape[0]), int(features.shape[1])
        colors = torch.empty(n, 3, dtype=torch.float32, device=xyz.device)
        clamped = torch.empty(n, dtype=torch.uint8, device=xyz.device)
        noise_arr = (ctypes.c_float * 9)(*noise) if noise is not None else None
        with _lib.on_device(xyz.device):
            _lib.check(L.gs_sh2rgb(n, int(deg), m, _lib.ptr(features), _lib.ptr(xyz), _lib.ptr(campos),
                                   _lib.ptr(fwd_rot), noise_arr, _lib.ptr(colors), _lib.ptr(clamped), _stream(xyz.device)))
        ctx.save_for_backward(features, xyz, campos, fwd_rot if fwd_rot is not None else torch.empty(0, device=xyz.device),
                              clamped)
        ctx.noise, ctx.deg, ctx.has_rot = noise, int(deg), fwd_rot is not None
        return colors

    @staticmethod
    def backward(ctx, g):
        features, xyz, campos, fwd_rot, clamped = ctx.saved_tensors
        L = _lib.load()
        n, m = int(features.shape[0]), int(features.shape[1])
        g = g.to(torch.float32).contiguous()
        dsh = torch.empty_like(features)
        dxyz = torch.empty_like(xyz)
        noise_arr = (ctypes.c_float * 9)(*ctx.noise) if ctx.noise is not None else None
        with _lib.on_device(xyz.device):
            _lib.check(L.gs_sh2rgb_backward(n, ctx.deg, m, _lib.ptr(features), _lib.ptr(xyz), _lib.ptr(campos),
                                            _lib.ptr(fwd_rot) if ctx.has_rot else None, noise_arr, _lib.ptr(clamped),
                                            _lib.ptr(g), _lib.ptr(dsh), _lib.ptr(dxyz), _stream(xyz.device)))
        return dsh, dxyz, None, None, None, None


def sh2rgb(features, xyz, camera_center, active_sh_degree, fwd_transform=None, view_noise=None):
    """colors_precomp of models/texture/texture.py:21-38.  `features` is get_features, (N, M, 3); `camera_center`
    (3,); `fwd_transform` (N,4,4) or (N,3,3) switches on cano_view_dir (its rotation part is used, without
    gradient); `view_noise` an optional 3x3 matrix (already transposed as texture.py:30-33 does), applied as
    dir @ view_noise."""
    features = _dev32(features, "features")
    xyz = _dev32(xyz, "xyz")
    campos = _dev32(camera_center.reshape(-1), "camera_center")
    if features.dim() != 3 or features.shape[2] != 3 or features.shape[0] != xyz.shape[0]:
        raise ValueError("features: (N, M, 3) expected, N as xyz")
    deg = int(active_sh_degree)
    if deg < 0 or deg > 3 or features.shape[1] < (deg + 1) ** 2 or features.shape[1] > 16:
        raise ValueError("sh2rgb: degree 0..3 with (deg+1)^2 <= M <= 16 coefficients expected")
    rot = None
    if fwd_transform is not None:
        rot = _dev32(fwd_transform.detach()[:, :3, :3], "fwd_transform")
    noise = None
    if view_noise is not None:
        noise = [float(v) for v in torch.as_tensor(view_noise, dtype=torch.float32).reshape(-1).tolist()]
        if len(noise) != 9:
            raise ValueError("view_noise: 3x3 expected")
    return _Sh2Rgb.apply(features, xyz, campos, rot, noise, deg)
