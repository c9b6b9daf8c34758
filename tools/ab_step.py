"""In-process A/B of module-level switches on the benchmark step (cdna_hip_programming.md rule 24: interleaved rounds
in ONE process, report the distribution).  Usage:
  python tools/ab_step.py diff_gaussian_rasterization._SPECULATE=1,0 [--workload config3] [--rounds 7] [--steps 150]
  python tools/ab_step.py lib:xcd_map=1,0        (a gs_tuning switch of the library)"""
import argparse
import importlib
import os
import statistics
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

ap = argparse.ArgumentParser()
ap.add_argument("switch")
ap.add_argument("--workload", default="config3")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--also", default="", help="library switches held fixed for the whole run, e.g. fwd4=0,xcd_map=1")
args = ap.parse_args()
target, values = args.switch.split("=")
if target.startswith("lib:"):  # a gs_tuning switch of the library, e.g. lib:xcd_map=1,0
    from gsplat_mi355 import _lib
    values = [int(v) for v in values.split(",")]

    def apply(v):
        _lib.tuning(target[4:], v)
else:
    modname, attr = target.rsplit(".", 1)
    mod = importlib.import_module(modname)
    values = [type(getattr(mod, attr))(int(v)) if isinstance(getattr(mod, attr), (bool, int)) else v for v in values.split(",")]

    def apply(v):
        setattr(mod, attr, v)

if args.also:
    from gsplat_mi355 import _lib as _l
    for kv in args.also.split(","):
        k, v = kv.split("=")
        _l.tuning(k, int(v))
N, W, H, deg, tail, do_bwd = bench.WORKLOADS[args.workload]
dev = torch.device("cuda:0")
cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev, layout=bench.WORKLOAD_LAYOUT.get(args.workload, "box"))
for f in GaussianCloud.FIELDS:
    getattr(cloud, f).requires_grad_(do_bwd)
cams = [orbit_camera(f, W, H, device=dev) for f in range(args.steps)]
gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
pipe = Pipe()


def step(i):
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).grad = None
    if do_bwd:
        pkg = render(cams[i], cloud, pipe, bg)
        l1_loss(pkg.render, gt).backward()
    else:
        with torch.no_grad():
            render(cams[i], cloud, pipe, bg)


t_end = time.perf_counter() + 1.0  # settle the clocks
while time.perf_counter() < t_end:
    step(0)
res = {repr(v): [] for v in values}
for r in range(args.rounds):
    for v in values:
        apply(v)
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        res[repr(v)].append((time.perf_counter() - t0) / args.steps * 1e3)
for k, v in res.items():
    print("%s = %-6s ms/step: median %.4f  min %.4f  max %.4f  (%d rounds)  -> %.1f fps" % (
        target, k, statistics.median(v), min(v), max(v), len(v), 1e3 / statistics.median(v)))
