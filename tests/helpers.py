"""Shared scene builders for the test-suite (CPU side: numpy/torch only)."""
import math

import numpy as np
import torch

from gsplat_mi355.camera import Camera, focal2fov, orbit_camera
from gsplat_mi355.scenes import synthetic_cloud


def cloud_and_camera(n, W, H, sh_degree=3, seed=0, frame=0, dist2_fn=None, heavy_tail=0.0, scale_mul=1.0):
    from oracle import gs_oracle
    fn = dist2_fn or (lambda p: torch.from_numpy(gs_oracle.dist2(p.cpu().numpy())))
    cloud = synthetic_cloud(n, sh_degree=sh_degree, seed=seed, dist2_fn=fn, heavy_tail=heavy_tail)
    if scale_mul != 1.0:
        cloud.scales = cloud.scales * scale_mul
    cam = orbit_camera(frame, W, H)
    return cloud, cam


def oracle_scene(cloud, cam, bg=(0.0, 0.0, 0.0), color_mode="sh", cov_mode="scale_rot", scale_modifier=1.0,
                 colors=None):
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
        kw["cov3D_precomp"] = cloud.covariance6(scale_modifier).numpy()
    return gs_oracle.Scene(cam.image_width, cam.image_height, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                           np.asarray(bg, np.float32), cam.world_view_transform.numpy(),
                           cam.full_proj_transform.numpy(), cam.camera_center.numpy(), cloud.xyz.numpy(),
                           cloud.opacity.numpy(), scale_modifier=scale_modifier, **kw)


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
