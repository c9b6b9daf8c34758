"""Soak: thousands of full steps (render with fused opacity -> 0.8 L1 + 0.2 D-SSIM + 0.1 mask L1 -> backward ->
densification statistics -> Adam) over a moving camera, checking for non-finite values, stalls and leaks."""
import sys, time
sys.path.insert(0, "3dgs-avatar-release_amd")
import torch
from gsplat_mi355.camera import orbit_camera
from gsplat_mi355.optim import FusedAdam
from gsplat_mi355.render import DensifyStats, Pipe, l1_loss, render, ssim
from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud

dev = torch.device("cuda:0")
N, W, H, STEPS = 100000, 512, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
LAYOUT = sys.argv[2] if len(sys.argv) > 2 else "box"  # "body": a thin-shell cloud (long tile lists: four-wave forward, chunked backward)
cloud = synthetic_cloud(N, sh_degree=3, seed=3, device=dev, layout=LAYOUT)
for f in GaussianCloud.FIELDS:
    getattr(cloud, f).requires_grad_(True)
lrs = dict(xyz=1e-6, scales=1e-6, rotations=1e-5, opacity=1e-4, shs=1e-3)
opt = FusedAdam([{"params": [getattr(cloud, f)], "lr": lrs[f]} for f in GaussianCloud.FIELDS], lr=0.0, eps=1e-15)
stats = DensifyStats(N, dev)
gt = torch.rand(3, H, W, device=dev)
mask = (torch.rand(1, H, W, device=dev) > 0.5).float()
bg = torch.zeros(3, device=dev)
pipe = Pipe(fuse_opacity=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
worst = 0.0
for it in range(STEPS):
    ts = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    pkg = render(orbit_camera(it % 600, W, H, device=dev), cloud, pipe, bg, return_opacity=True)
    loss = 0.8 * l1_loss(pkg.render, gt) + 0.2 * (1.0 - ssim(pkg.render, gt)) + 0.1 * l1_loss(pkg.opacity_render, mask)
    loss.backward()
    with torch.no_grad():
        stats.update(pkg)
        opt.step()
        cloud.opacity.clamp_(1e-4, 0.9999)
        cloud.scales.clamp_(min=1e-5)
    if it % 500 == 499:
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(loss)) and all(bool(torch.isfinite(getattr(cloud, f)).all()) for f in GaussianCloud.FIELDS)
        print("step %d loss %.5f finite %s  %.3f ms/step  mem %.0f MB" % (
            it + 1, float(loss), ok, (time.perf_counter() - t0) / (it + 1) * 1e3, torch.cuda.max_memory_allocated() / 2**20), flush=True)
        assert ok
    worst = max(worst, time.perf_counter() - ts)
print("done; slowest single step (host side) %.1f ms" % (worst * 1e3))
