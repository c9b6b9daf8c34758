"""Times knn_points (self-KNN as the AIAP loss calls it every step) and distCUDA2 on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3dgs-avatar-release_amd"))
import torch
from gsplat_mi355.knn import knn_points
from simple_knn._C import distCUDA2


def bench(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cloud(dist, n):
    if dist == "uniform":
        return torch.rand(n, 3, device="cuda")
    if dist == "normal":
        return torch.randn(n, 3, device="cuda")
    # "surface": a body-like shell -- points on capsules (limbs, torso) with 1 % radial noise, the shape of an avatar's
    # canonical Gaussians (a 2-D manifold in 3-D, not a volume)
    g = torch.Generator(device="cuda").manual_seed(0)
    t = torch.rand(n, device="cuda", generator=g)
    phi = torch.rand(n, device="cuda", generator=g) * 6.2831853
    part = torch.randint(0, 5, (n,), device="cuda", generator=g)
    radius = torch.tensor([0.15, 0.05, 0.05, 0.07, 0.07], device="cuda")[part]
    length = torch.tensor([0.6, 0.6, 0.6, 0.8, 0.8], device="cuda")[part]
    ox = torch.tensor([0.0, -0.25, 0.25, -0.1, 0.1], device="cuda")[part]
    oy = torch.tensor([0.5, 0.5, 0.5, -0.4, -0.4], device="cuda")[part]
    r = radius * (1 + 0.01 * torch.randn(n, device="cuda", generator=g))
    return torch.stack([ox + r * torch.cos(phi), oy + (t - 0.5) * length, r * torch.sin(phi)], 1).contiguous()


for dist in ("uniform", "normal", "surface"):
    for n in (50000, 200000):
        x = cloud(dist, n)
        line = "%s N=%d:" % (dist, n)
        for K in (1, 3, 6):
            line += "  K=%d %.3f ms" % (K, bench(lambda: knn_points(x[None], x[None], K=K)))
        line += "  distCUDA2 %.3f ms" % bench(lambda: distCUDA2(x))
        print(line)
