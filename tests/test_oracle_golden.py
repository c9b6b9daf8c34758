"""The oracle (and the package's camera helper) against fixtures generated from the reference's own
importable Python (tests/golden/make_golden.py): utils/sh_utils.py and utils/graphics_utils.py.
These are the only parts of the hot path the reference tree pins."""
import os

import numpy as np
import torch

from gsplat_mi355.camera import Camera, focal2fov
from gsplat_mi355.scenes import rgb_to_sh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _all_visible_camera():
    """A camera that sees every point of the fixture (points at radius 2.5, pushed +10 in z)."""
    view = np.eye(4, dtype=np.float32)
    view[3, 2] = 10.0
    proj = np.eye(4, dtype=np.float32)
    proj[2, 3] = 1.0
    proj[3, 3] = 0.0
    return view, (view @ proj).astype(np.float32)


def test_sh_polynomial_matches_reference_eval_sh(oracle):
    g = np.load(os.path.join(GOLD, "sh_eval.npz"))
    dirs = g["dirs"]
    n = dirs.shape[0]
    # reference layout [..., C, coeffs] -> rasterizer layout (n, coeffs, C) (gaussian_model.py:145-148)
    shs = np.ascontiguousarray(g["sh"].transpose(0, 2, 1)).astype(np.float32)
    means = (dirs * 2.5).astype(np.float32)  # campos = 0, so the view direction is `dirs`
    view, projm = _all_visible_camera()
    for deg in range(4):
        sc = oracle.Scene(64, 64, 1.0, 1.0, np.zeros(3), view, projm, np.zeros(3), means, np.ones(n, np.float32),
                          shs=shs, sh_degree=deg, scales=np.full((n, 3), 0.05, np.float32),
                          rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)))
        st = oracle.preprocess(sc)
        vis = st["radii"] > 0
        assert vis.sum() > n // 2
        assert np.abs(st["rgb"][vis] - g["color_deg%d" % deg][vis]).max() < 2e-6
        raw = g["eval_deg%d" % deg][vis] + 0.5
        sure = np.abs(raw) > 1e-5
        assert ((st["clamped"][vis] != 0) == (raw < 0))[sure].all()


def test_rgb2sh():
    g = np.load(os.path.join(GOLD, "sh_eval.npz"))
    got = rgb_to_sh(torch.from_numpy(g["rgb"])).numpy()
    assert np.abs(got - g["rgb2sh"]).max() < 1e-12


def test_camera_matrices_match_reference_graphics_utils():
    g = np.load(os.path.join(GOLD, "cameras.npz"))
    for k in range(int(g["count"])):
        W, H = [int(v) for v in g["WH_%d" % k]]
        fovx, fovy = g["fov_%d" % k]
        cam = Camera(g["R_%d" % k], g["T_%d" % k], fovx, fovy, W, H)
        assert np.array_equal(cam.world_view_transform.numpy(), g["world_view_%d" % k])
        assert np.array_equal(cam.projection_matrix.numpy(), g["proj_%d" % k])
        assert np.array_equal(cam.full_proj_transform.numpy(), g["full_proj_%d" % k])
        assert np.array_equal(cam.camera_center.numpy(), g["center_%d" % k])
        assert focal2fov(500.0 * W / 512.0, W) == fovx


def test_image_losses_match_reference_loss_utils(oracle):
    """N2 pinned: the oracle's L1 and SSIM (values and gradients w.r.t. the rendered image) against fixtures produced by
    the reference's own `l1_loss`, `gaussian`, `create_window`, `ssim`, `_ssim` (utils/loss_utils.py:21-67) run on seeded
    CPU images in fp64 and in fp32 (tests/golden/make_golden.py executes exactly those function definitions).  The
    oracle accumulates in double: it must sit on the reference's fp64 result (1e-9) and within fp32 conv2d rounding of
    the reference's fp32 result; the window entries are the reference's fp32 ones bit for bit."""
    g = np.load(os.path.join(GOLD, "losses.npz"))
    # the reference's 11x11 window (fp32): outer product of the normalised 1-D Gaussian, identical for every channel
    w = g["window"]
    assert w.shape == (3, 1, 11, 11) and w.dtype == np.float32
    g1 = np.array([np.float32(np.exp(-(k - 5) ** 2 / (2 * 1.5 ** 2))) for k in range(11)], np.float32)
    assert abs(float(w[0, 0].sum()) - 1.0) < 1e-6 and np.array_equal(w[0], w[2])
    for k in range(int(g["count"])):
        a, b = g["img1_%d" % k], g["img2_%d" % k]
        v, grad = oracle.ssim(a, b)
        assert abs(v - float(g["ssim_f64_%d" % k])) < 1e-9, k
        gs = np.abs(g["ssim_grad_f64_%d" % k]).max()
        if gs > 0:
            assert np.abs(grad - g["ssim_grad_f64_%d" % k]).max() <= 1e-6 * gs, k
        # the reference's own fp32 evaluation differs from fp64 by conv2d rounding (sigma = E[x^2] - mu^2 cancels)
        assert abs(v - float(g["ssim_f32_%d" % k])) < 2e-5, k
        if gs > 0:
            assert np.abs(grad - g["ssim_grad_f32_%d" % k]).max() <= 5e-4 * gs, k
        lv, lgrad = oracle.l1_loss(a, b)
        assert abs(lv - float(g["l1_f64_%d" % k])) <= 1e-12 + 1e-9 * abs(lv), k
        assert abs(lv - float(g["l1_f32_%d" % k])) <= 1e-6 * max(abs(lv), 1e-30) + 1e-9, k
        assert np.array_equal(lgrad, g["l1_grad_f32_%d" % k]), k  # sign(x - y) / n, bit for bit
    assert abs(float(g["ssim_f64_2"]) - 1.0) < 1e-12  # identical images
