"""Shared scene builders for the test-suite (CPU side: numpy/torch only)."""
import math

import numpy as np
import torch

from gsplat_mi355.camera import Camera, focal2fov, orbit_camera
from gsplat_mi355.scenes import synthetic_cloud


def cloud_and_camera(n, W, H, sh_degree=3, seed=0, frame=0, dist2_fn=None, heavy_tail=0.0, scale_mul=1.0, layout="box"):
    from oracle import gs_oracle
    fn = dist2_fn or (lambda p: torch.from_numpy(gs_oracle.dist2(p.cpu().numpy())))
    cloud = synthetic_cloud(n, sh_degree=sh_degree, seed=seed, dist2_fn=fn, heavy_tail=heavy_tail, layout=layout)
    if scale_mul != 1.0:
        cloud.scales = cloud.scales * scale_mul
    cam = orbit_camera(frame, W, H)
    return cloud, cam


def needle_cloud_and_camera(n, W, H, seed=0, major=(0.15, 0.7), sh_degree=1):
    """Long thin Gaussians lying in the image plane (sigma_major hundreds of pixels, sigma_minor at the 0.3 px^2
    low-pass floor, random in-plane angles): their 2-D covariance determinant is a difference of nearly equal numbers,
    the stress case for anything derived from the covariance next to the rounded fp32 conic."""
    cloud, cam = cloud_and_camera(n, W, H, sh_degree=sh_degree, seed=seed)
    g = torch.Generator().manual_seed(seed)
    cloud.xyz = torch.cat([torch.empty(n, 2).uniform_(-0.4, 0.4, generator=g), torch.empty(n, 1).uniform_(-0.3, 0.3, generator=g)], 1)
    th = torch.empty(n).uniform_(0, math.pi, generator=g)
    cloud.rotations = torch.stack([torch.cos(th / 2), torch.zeros(n), torch.zeros(n), torch.sin(th / 2)], 1).contiguous()
    cloud.scales = torch.cat([torch.empty(n, 1).uniform_(*major, generator=g), torch.full((n, 2), 1e-4)], 1).contiguous()
    cloud.opacity = torch.empty(n, 1).uniform_(0.05, 0.99, generator=g)
    return cloud, cam


def product_tile_rect():
    """The tile-rectangle mode the product runs with (GsFwdArgs.tile_rect; GSPLAT_TILE_RECT, default 1): the oracle is
    put in the same mode wherever internal buffers (tiles_touched, lists, ranges) are compared."""
    import os
    return int(os.environ.get("GSPLAT_TILE_RECT", "1"))


def oracle_scene(cloud, cam, bg=(0.0, 0.0, 0.0), color_mode="sh", cov_mode="scale_rot", scale_modifier=1.0,
                 colors=None, tile_rect=None):
    """Builds the oracle's Scene from a cloud + camera in one of the four input combinations of the
    rasterizer API."""
    from oracle import gs_oracle
    kw = {}
    if color_mode == "sh":
        kw["shs"] = cloud.shs.numpy()
        kw["sh_degree"] = cloud.sh_degree
    else:
        kw["colors_precomp"] = (colors if colors is not None else precomp_colors(cloud, cam)).numpy()
    if cov_mode == "scale_rot":
        kw["scales"] = cloud.scales.numpy()
        kw["rotations"] = cloud.rotations.numpy()
    else:
        kw["cov3D_precomp"] = covariance6_cpu(cloud, scale_modifier).numpy()
    return gs_oracle.Scene(cam.image_width, cam.image_height, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                           np.asarray(bg, np.float32), cam.world_view_transform.numpy(),
                           cam.full_proj_transform.numpy(), cam.camera_center.numpy(), cloud.xyz.numpy(),
                           cloud.opacity.numpy(), scale_modifier=scale_modifier,
                           tile_rect=product_tile_rect() if tile_rect is None else tile_rect, **kw)


def precomp_colors(cloud, cam):
    """colors_precomp the way the reference's texture module makes it (models/texture/texture.py:17-38,
    cano_view_dir off): clamp_min(eval_sh(dir) + 0.5, 0)."""
    from oracle import dense_ref
    d = cloud.xyz.double() - cam.camera_center.double()[None]
    dirs = d / (d.norm(dim=1, keepdim=True) + 1e-12)
    raw = dense_ref._sh_rgb(cloud.sh_degree, cloud.shs.double(), dirs)
    return torch.clamp_min(raw + 0.5, 0.0).float()


def rel_to_max(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    m = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / m


def ssim_float64(img1, img2):
    """Independent float64 restatement of the SSIM the oracle implements (window 11, sigma 1.5 with fp32 window
    entries, zero padding, C1 = 1e-4, C2 = 9e-4, mean): returns (value, d value / d img1) via torch autograd."""
    import torch
    import torch.nn.functional as F
    g1 = torch.tensor([np.float32(np.exp(-(k - 5) ** 2 / (2 * 1.5 ** 2))) for k in range(11)], dtype=torch.float32)
    g1 = g1 / g1.sum()
    c = img1.shape[0]
    w = (g1[:, None] @ g1[None, :]).double()[None, None].repeat(c, 1, 1, 1)
    a = torch.from_numpy(np.asarray(img1, np.float32)).double()[None].requires_grad_(True)
    b = torch.from_numpy(np.asarray(img2, np.float32)).double()[None]
    conv = lambda t: F.conv2d(t, w, padding=5, groups=c)
    mu1, mu2 = conv(a), conv(b)
    s1, s2, s12 = conv(a * a) - mu1 ** 2, conv(b * b) - mu2 ** 2, conv(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))
    v = m.mean()
    v.backward()
    return float(v.detach()), a.grad[0].numpy()


def covariance_float64(scaling, modifier, rotation, g6):
    """Float64 torch restatement of scene/gaussian_model.py:28-32 (+ general_utils.py:73-108,194-207) with
    autograd: returns (cov6, d/dscaling, d/drotation) for the upstream gradient g6."""
    import torch
    s = torch.from_numpy(np.asarray(scaling, np.float32)).double().requires_grad_(True)
    r = torch.from_numpy(np.asarray(rotation, np.float32)).double().requires_grad_(True)
    if r.shape[-1] == 4:
        q = r / torch.sqrt((r * r).sum(1))[:, None]
        w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
    else:
        R = r
    L = R @ torch.diag_embed(modifier * s)
    S = L @ L.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
    (cov * torch.from_numpy(np.asarray(g6, np.float32)).double()).sum().backward()
    return cov.detach().numpy(), s.grad.numpy(), r.grad.numpy()


def sh2rgb_float64(features, xyz, campos, deg, fwd_rot, noise, g):
    """Float64 torch restatement of models/texture/texture.py:21-38 with autograd (the SH polynomial written out
    from utils/sh_utils.py:58-101): returns (colors, d/dfeatures, d/dxyz)."""
    import torch
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    f = torch.from_numpy(np.asarray(features, np.float32)).double().requires_grad_(True)  # (N, M, 3)
    p = torch.from_numpy(np.asarray(xyz, np.float32)).double().requires_grad_(True)
    d = p - torch.from_numpy(np.asarray(campos, np.float32)).double()[None]
    if fwd_rot is not None:
        Rb = torch.from_numpy(np.asarray(fwd_rot, np.float32)).double().transpose(1, 2)
        d = torch.matmul(Rb, d.unsqueeze(-1)).squeeze(-1)
    if noise is not None:
        d = torch.matmul(d, torch.from_numpy(np.asarray(noise, np.float32)).double())
    u = d / (d.norm(dim=1, keepdim=True) + 1e-12)
    x, y, z = u[:, 0:1], u[:, 1:2], u[:, 2:3]
    sh = f.transpose(1, 2)  # (N, 3, M) as shs_view
    res = C0 * sh[..., 0]
    if deg > 0:
        res = res - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] +
               C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10] +
               C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
               C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14] +
               C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    col = torch.clamp_min(res + 0.5, 0.0)
    (col * torch.from_numpy(np.asarray(g, np.float32)).double()).sum().backward()
    gp = p.grad.numpy() if p.grad is not None else np.zeros(p.shape)  # degree 0 does not depend on the direction
    return col.detach().numpy(), f.grad.numpy(), gp


def covariance6_cpu(cloud, scale_modifier=1.0):
    """Test input builder (any valid covariance will do): strip_symmetric((R S)(R S)^T) of a CPU cloud in plain
    torch.  The product's GaussianCloud.covariance6 is the fused HIP op (GPU only)."""
    import torch
    q = cloud.rotations / cloud.rotations.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
    L = R * (scale_modifier * cloud.scales).unsqueeze(1)
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1).contiguous()
