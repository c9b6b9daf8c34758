"""How steady is the benchmark step over time?  Runs the config-3 step back to back and prints, per block of 100 steps, the
wall time per step (synchronised at block ends only) and the host-side enqueue time per step.
Usage: python tools/step_jitter.py [blocks] [distinct_cameras]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from gsplat_mi355.camera import orbit_camera  # noqa: E402
from gsplat_mi355.render import Pipe, l1_loss, render  # noqa: E402
from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ncam = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N, W, H, deg, tail, do_bwd = bench.WORKLOADS["config3"]
dev = torch.device("cuda:0")
cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev)
for f in GaussianCloud.FIELDS:
    getattr(cloud, f).requires_grad_(True)
cams = [orbit_camera(f, W, H, device=dev) for f in range(ncam)]
gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
pipe = Pipe()


def step(i):
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).grad = None
    pkg = render(cams[i % ncam], cloud, pipe, bg)
    l1_loss(pkg.render, gt).backward()


k = 0
for b in range(blocks):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(100):
        h0 = time.perf_counter()
        step(k)
        host += time.perf_counter() - h0
        k += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats(dev)
    print("block %2d: %.4f ms/step wall, %.4f ms/step host enqueue, allocs %d, cudaMalloc retries %d, reserved %.0f MB" % (
        b, dt * 10, host * 10, st["allocation.all.allocated"], st["num_alloc_retries"], st["reserved_bytes.all.current"] / 1e6), flush=True)
    if b == blocks // 2:
        time.sleep(0.5)  # an idle gap: does the next block start slow?
