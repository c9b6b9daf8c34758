"""Differential fuzz of the two sets of render kernels: the few-long-lists machinery (four-wave forward on marked tiles,
backward in chunks) against the one-wave-per-quadrant kernels on the same random scenes.  Transmittance, n_contrib, the
recorded lists and the radii must agree bit for bit; colour and gradients to fp32 rounding."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import helpers
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from gsplat_mi355 import _lib, debug
from simple_knn._C import distCUDA2

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
trials = worst_c = worst_g = 0
marked = chunked = 0
while time.time() < t_end:
    n = int(rng.choice([500, 3000, 9000, 20000, 40000]))
    W, H = int(rng.integers(17, 400)), int(rng.integers(17, 400))
    deg = int(rng.integers(0, 4))
    layout = str(rng.choice(["body", "box"]))
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=int(rng.integers(1 << 30)), layout=layout,
                                          scale_mul=float(rng.uniform(0.3, 3.0)), frame=int(rng.integers(0, 300)),
                                          dist2_fn=lambda p: distCUDA2(p.to(dev)).cpu())
    cloud.opacity = (cloud.opacity * float(rng.choice([0.05, 0.3, 1.0]))).clamp(1e-4, 0.999)
    bg = torch.tensor(rng.random(3), dtype=torch.float32, device=dev)
    import math
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg,
                                      cam.camera_center.to(dev), False, False)
    gimg = torch.randn(3, H, W, device=dev)
    gop = torch.randn(1, H, W, device=dev)
    with_op = bool(rng.integers(0, 2))
    out = {}
    for mode in (1, 0):
        _lib.tuning("fwd4", mode)
        _lib.tuning("bwd_chunks", mode)
        leaves = dict(means3D=cloud.xyz.to(dev).requires_grad_(True), means2D=torch.zeros(n, 3, device=dev, requires_grad=True),
                      opacities=cloud.opacity.to(dev).requires_grad_(True), shs=cloud.shs.to(dev).requires_grad_(True),
                      scales=cloud.scales.to(dev).requires_grad_(True), rotations=cloud.rotations.to(dev).requires_grad_(True))
        res = GaussianRasterizer(s)(with_opacity=with_op, **leaves)
        loss = (res[0] * gimg).sum() + ((res[2] * gop).sum() if with_op else 0.0)
        loss.backward()
        st = debug.forward_state(s, cloud.xyz.to(dev), cloud.opacity.to(dev), shs=cloud.shs.to(dev), scales=cloud.scales.to(dev),
                                 rotations=cloud.rotations.to(dev))
        out[mode] = dict(color=res[0].detach().cpu().numpy(), radii=res[1].cpu().numpy(), st=st,
                         grads={k: v.grad.cpu().numpy() for k, v in leaves.items()})
    _lib.tuning("fwd4", 1)
    _lib.tuning("bwd_chunks", 1)
    a, b = out[1], out[0]
    tag = "n=%d %dx%d deg %d %s op %s" % (n, W, H, deg, layout, with_op)
    assert np.array_equal(a["radii"], b["radii"]), tag
    for f in ("final_T", "n_contrib", "qcount"):
        assert np.array_equal(a["st"]["image"][f], b["st"]["image"][f]), (tag, f)
    assert np.array_equal(a["st"]["binning"]["point_list"], b["st"]["binning"]["point_list"]), tag
    dc = float(np.abs(a["color"] - b["color"]).max())
    assert dc <= 5e-6, (tag, dc)
    worst_c = max(worst_c, dc)
    for k in a["grads"]:
        sc = float(np.abs(b["grads"][k]).max())
        if sc == 0:
            assert np.abs(a["grads"][k]).max() == 0, (tag, k)
            continue
        dg = float(np.abs(a["grads"][k] - b["grads"][k]).max()) / sc
        assert dg <= 1e-4, (tag, k, dg)  # (both walks carry Rem = Gtot - (composited so far), rounding ~1e-7 Gtot)
        worst_g = max(worst_g, dg)
    marked += int((a["st"]["image"]["order"] >> 31).sum())
    chunked += int((a["st"]["image"]["qcount"] > 128).sum())
    trials += 1
print("fuzz: %d scenes, %d marked tiles, %d chunked quadrants; worst colour difference %.2e, worst gradient difference %.2e of the maximum"
      % (trials, marked, chunked, worst_c, worst_g))
