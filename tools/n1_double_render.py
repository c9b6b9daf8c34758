"""cost of the reference's two-render train step (colour pass + opacity pass with the
same geometry, gaussian_renderer/__init__.py:121-142; L1 + 0.1 * mask L1): two independent rasterizer calls
("separate"), the second call sharing the first one's geometry ("shared", transparent), and ONE call with
with_opacity=True ("fused": the opacity render and its gradient ride along in the colour pass)."""
import sys, time
sys.path.insert(0, "3dgs-avatar-release_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import diff_gaussian_rasterization as dgr
from gsplat_mi355.camera import orbit_camera
from gsplat_mi355.render import Pipe, train_step
from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud
dev = torch.device("cuda:0")
N, W, H = 200000, 1024, 1024
cloud = synthetic_cloud(N, sh_degree=3, seed=0, device=dev)
for f in GaussianCloud.FIELDS:
    getattr(cloud, f).requires_grad_(True)
cams = [orbit_camera(f, W, H, device=dev) for f in range(64)]
gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
mask = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(2)) > 0.5).float().to(dev)
bg = torch.zeros(3, device=dev)
for mode in ("separate", "shared", "fused", "separate", "shared", "fused"):
    share = mode == "shared"
    pipe = Pipe(compute_cov3D_python=False, fuse_opacity=(mode == "fused"))
    dgr._SHARE = share
    dgr.release_shared_geometry()
    for i in range(5):
        for f in GaussianCloud.FIELDS: getattr(cloud, f).grad = None
        train_step(cams[i], cloud, pipe, bg, gt, gt_mask=mask, lambda_mask=0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 50
    for i in range(K):
        for f in GaussianCloud.FIELDS: getattr(cloud, f).grad = None
        train_step(cams[5 + i], cloud, pipe, bg, gt, gt_mask=mask, lambda_mask=0.1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("%-9s colour + opacity train step: %.3f ms (%.1f steps/s)" % (mode, dt * 1e3, 1 / dt))
