"""Crash / sanity sweep over image shapes and cloud sizes the tests do not reach (full HD, 4K, one million
Gaussians): forward + backward through the drop-in API, outputs finite, second run bitwise identical."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3dgs-avatar-release_amd"))
import torch
from gsplat_mi355.camera import orbit_camera
from gsplat_mi355.render import Pipe, l1_loss, render
from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud

dev = torch.device("cuda:0")
for (n, W, H, deg) in [(500000, 1920, 1080, 3), (100000, 3840, 2160, 2), (1000000, 1000, 700, 1), (3000, 17, 2000, 0),
                       (50000, 2000, 33, 3),
                       (60000, 4096, 2304, 1)]:  # 36 864 tiles: the tile-order kernel's re-reading form (> 32 768 tiles)
    cloud = synthetic_cloud(n, sh_degree=deg, seed=1, device=dev)
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).requires_grad_(True)
    cam = orbit_camera(3, W, H, device=dev)
    gt = torch.rand(3, H, W, device=dev)
    outs = []
    for rep in range(2):
        for f in GaussianCloud.FIELDS:
            getattr(cloud, f).grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pkg = render(cam, cloud, Pipe(), torch.zeros(3, device=dev))
        loss = l1_loss(pkg.render, gt)
        loss.backward()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        outs.append((pkg.render.detach().clone(), cloud.xyz.grad.clone(), cloud.shs.grad.clone()))
    ok = all(torch.isfinite(t).all().item() for t in outs[0])
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    print("N=%d %dx%d deg %d: %.2f ms, visible %d, finite %s, deterministic %s" % (
        n, W, H, deg, dt * 1e3, int((pkg.radii > 0).sum()), ok, same), flush=True)
    assert ok and same
