"""Forward time of frames whose depth distribution sends the depth ranking down its other paths (tests/test_gpu_parity.py:
test_depth_ranking_with_uneven_depth_distributions has the same scenes): evenly spread depths, K depth levels (24000 / K
equal keys per bucket), half the cloud at one depth, everything at one depth."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gsplat_mi355.camera import orbit_camera  # noqa: E402
from gsplat_mi355.render import Pipe, render  # noqa: E402
from gsplat_mi355.scenes import synthetic_cloud  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, W, H = 24000, 256, 192
    g = torch.Generator().manual_seed(7)
    cases = sys.argv[1:] or ("even", "levels-120", "levels-40", "levels-8", "half-one-depth", "one-depth")
    for case in cases:
        cloud = synthetic_cloud(n, sh_degree=0, seed=61, dist2_fn=lambda p: distCUDA2(p.to(dev)).cpu())
        cloud.scales = cloud.scales * 0.6
        cam = orbit_camera(0, W, H)
        if case.startswith("levels-"):
            k = int(case.split("-")[1])
            cloud.xyz[:, 2] = -0.8 + 1.6 * torch.randint(0, k, (n,), generator=g).float() / k
        elif case == "half-one-depth":
            cloud.xyz[::2, 2] = -0.5
        elif case == "one-depth":
            cloud.xyz[:, 2] = 0.25
        pc = cloud.to(dev)
        cam = cam.to(dev)
        bg = torch.zeros(3, device=dev)
        with torch.no_grad():
            for _ in range(5):
                render(cam, pc, Pipe(), bg)
            torch.cuda.synchronize()
            per_call = []
            for _ in range(60):
                t1 = time.perf_counter()
                render(cam, pc, Pipe(), bg)  # (waits for the frame's pair count: the calls pace themselves to the GPU)
                per_call.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
        per_call.sort()
        print("%-16s %8.1f us per forward (median of 60 calls; longest %8.1f us)" % (case, per_call[30] * 1e6, per_call[-1] * 1e6))


if __name__ == "__main__":
    main()
