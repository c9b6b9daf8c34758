"""Camera matrices in the conventions the rasterizer kernels read.

Restates scene/cameras.py:32-40 and scene/duck_camera.py:59-76 of the reference (which build on
utils/graphics_utils.py:38-71): `world_view_transform` is the world-to-camera matrix TRANSPOSED
(row-vector convention, translation in the last row), `full_proj_transform = world_view @ P^T`,
`camera_center = inverse(world_view)[3, :3]`.  Checked against fixtures generated from the
reference's own graphics_utils in tests/golden/.
"""
import math

import numpy as np
import torch


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def world_to_view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """W2C 4x4 (float32 numpy) from a camera-to-world rotation R and W2C translation t
    (utils/graphics_utils.py:38-49)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).transpose()
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.float32(np.linalg.inv(C2W))


def projection_matrix(znear, zfar, fovX, fovY):
    """OpenGL-style perspective with z_sign = +1 and P[3,2] = 1 (utils/graphics_utils.py:51-71)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera(object):
    """Exposes exactly the attributes `render()` reads from its `data` argument
    (gaussian_renderer/__init__.py:82-95)."""

    def __init__(self, R, T, FoVx, FoVy, width, height, znear=0.01, zfar=100.0, device="cpu"):
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.FoVx, self.FoVy = float(FoVx), float(FoVy)
        self.image_width, self.image_height = int(width), int(height)
        self.znear, self.zfar = znear, zfar
        wv = torch.tensor(world_to_view(self.R, self.T)).transpose(0, 1)
        pm = projection_matrix(znear, zfar, self.FoVx, self.FoVy).transpose(0, 1)
        self.world_view_transform = wv.contiguous().to(device)
        self.projection_matrix = pm.contiguous().to(device)
        self.full_proj_transform = (wv.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0).contiguous().to(device)
        self.camera_center = wv.inverse()[3, :3].contiguous().to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def orbit_camera(frame, width, height, dtheta=0.01, dist=3.0, device="cpu"):
    """Benchmark camera of SURVEY.md 8d: R = I, T = (0,0,dist), f = 500*W/512, orbiting about y
    by `dtheta` rad per frame (1_render_series_recorded.py:46-58 shape)."""
    th = dtheta * frame
    c, s = math.cos(th), math.sin(th)
    R = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    f = 500.0 * width / 512.0
    return Camera(R, np.array([0.0, 0.0, dist]), focal2fov(f, width), focal2fov(f, height), width, height,
                  device=device)
