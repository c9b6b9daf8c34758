"""Host time of one training step (forward + L1 loss + backward through autograd) when the GPU work is negligible: the
floor the step time cannot go below however fast the kernels are, and where it goes (cProfile)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.render import Pipe, l1_loss, render
    from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud
    N, W, H = 2000, 64, 64
    dev = torch.device("cuda", 0)
    cloud = synthetic_cloud(N, sh_degree=3, seed=0, device=dev)
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).requires_grad_(True)
    cams = [orbit_camera(f, W, H, device=dev) for f in range(64)]
    gt = torch.rand(3, H, W, device=dev)
    bg = torch.zeros(3, device=dev)
    pipe = Pipe()

    def step(i):
        for f in GaussianCloud.FIELDS:
            getattr(cloud, f).grad = None
        pkg = render(cams[i % 64], cloud, pipe, bg)
        l1_loss(pkg.render, gt).backward()

    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    print("host-bound step (2000 Gaussians, 64 x 64): %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
