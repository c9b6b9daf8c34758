"""Differential fuzz of the shared-geometry second render (gs_forward_shared: the recorded quadrant lists are walked,
nothing is tested or recorded again) against a stand-alone render of the same inputs: image and every gradient must be
the same bits, on random scenes of both cloud shapes and both kernel sets."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import helpers
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from simple_knn._C import distCUDA2

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
trials = 0
ONES = len(sys.argv) > 3 and sys.argv[3] == "ones"  # the reference's opacity pass: colours = 1, image written as 1 - T
FUSED = len(sys.argv) > 3 and sys.argv[3] in ("fused", "ones")  # default: each call its own backward (bit for bit); fused: tolerance
dgr._FUSE_SECOND = FUSED
worst = 0.0
while time.time() < t_end:
    n = int(rng.choice([300, 3000, 12000, 40000]))
    W, H = int(rng.integers(17, 420)), int(rng.integers(17, 420))
    deg = int(rng.integers(0, 4))
    layout = str(rng.choice(["body", "box"]))
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=int(rng.integers(1 << 30)), layout=layout,
                                          scale_mul=float(rng.uniform(0.3, 3.0)), frame=int(rng.integers(0, 300)),
                                          dist2_fn=lambda p: distCUDA2(p.to(dev)).cpu())
    cloud.opacity = (cloud.opacity * float(rng.choice([0.05, 0.3, 1.0]))).clamp(1e-4, 0.999)
    bg = torch.tensor(rng.random(3), dtype=torch.float32, device=dev)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg,
                                      cam.camera_center.to(dev), False, False)
    gimg = torch.randn(3, H, W, device=dev)
    gop = torch.randn(3, H, W, device=dev)
    cols0 = torch.ones(n, 3, device=dev) if ONES else torch.rand(n, 3, device=dev)
    res = {}
    for share in (True, False):
        dgr._SHARE = share
        dgr.release_shared_geometry()
        h0 = dgr._geom_cache.hits
        xyz = cloud.xyz.to(dev).requires_grad_(True)
        m2d = torch.zeros(n, 3, device=dev, requires_grad=True)
        op = cloud.opacity.to(dev).requires_grad_(True)
        sc = cloud.scales.to(dev).requires_grad_(True)
        rot = cloud.rotations.to(dev).requires_grad_(True)
        shs = cloud.shs.to(dev).requires_grad_(True)
        cols = cols0.clone().requires_grad_(not FUSED)  # (the one-pass backward takes constant colours in the second image)
        rast = GaussianRasterizer(s)
        img1, _ = rast(means3D=xyz, means2D=m2d, opacities=op, shs=shs, scales=sc, rotations=rot)
        img2, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=cols, scales=sc, rotations=rot)
        assert (dgr._geom_cache.hits - h0) == (1 if share else 0)
        ((img1 * gimg).sum() + (img2 * gop).sum()).backward()
        res[share] = [img1.detach(), img2.detach()] + [t.grad.clone() for t in (xyz, m2d, op, sc, rot, shs) + (() if FUSED else (cols,))]
    dgr._SHARE = True
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        tag = ("n=%d %dx%d deg %d %s" % (n, W, H, deg, layout), i, float((a - b).abs().max()))
        if ONES and i == 1:
            assert float((a - b).abs().max()) <= 3e-6, tag
        elif FUSED and i >= 2:
            sc = float(b.abs().max())
            if sc == 0:
                assert float(a.abs().max()) == 0, tag
                continue
            d = float((a - b).abs().max()) / sc
            assert d <= 1e-4, tag + (d,)
            worst = max(worst, d)
        else:
            assert torch.equal(a, b), tag
    trials += 1
print("fuzz: %d scenes, shared second render == stand-alone render, bit for bit%s" % (
    trials, "; fused backward of both images within %.2e of the maximum" % worst if FUSED else ""))
