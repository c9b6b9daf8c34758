"""Host time of the reference's whole step around the rasterizer on a trained-avatar shaped frame -- two rasterizer calls
(colour + opacity), L1 + mask L1, backward, densification statistics, Adam (`bench.py --workload avatar --opacity second-call
--train-step`) -- where the GPU work (~0.56 ms) is short enough for Python to pace the step: where does the host time go?"""
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
    import bench
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.optim import FusedAdam
    from gsplat_mi355.render import DensifyStats, Pipe, l1_loss, render
    from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud
    wl = sys.argv[1] if len(sys.argv) > 1 else "avatar"
    N, W, H, deg, tail, _ = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev, layout=bench.WORKLOAD_LAYOUT.get(wl, "box"))
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).requires_grad_(True)
    cams = [orbit_camera(f, W, H, device=dev) for f in range(64)]
    gt = torch.rand(3, H, W, device=dev)
    gt_mask = (torch.rand(1, H, W, device=dev) > 0.5).float()
    bg = torch.zeros(3, device=dev)
    pipe = Pipe()
    lrs = dict(xyz=1.6e-9, scales=5e-9, rotations=1e-9, opacity=5e-9, shs=2.5e-9)
    opt = FusedAdam([{"params": [getattr(cloud, f)], "lr": lrs[f], "name": f} for f in GaussianCloud.FIELDS], lr=0.0, eps=1e-15)
    stats = DensifyStats(N, dev)

    def step(i):
        for f in GaussianCloud.FIELDS:
            getattr(cloud, f).grad = None
        pkg = render(cams[i % 64], cloud, pipe, bg, return_opacity=True)
        loss = l1_loss(pkg.render, gt) + 0.1 * l1_loss(pkg.opacity_render, gt_mask)
        loss.backward()
        with torch.no_grad():
            stats.update(pkg)
        opt.step()

    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%s two-call train step: %.1f us/step to issue, %.1f us/step complete" % (wl, t_issue / n * 1e6, (time.perf_counter() - t0) / n * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(32)


if __name__ == "__main__":
    main()
