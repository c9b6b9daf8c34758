"""Which torch operators (and which of their GPU kernels / copies) run in one benchmark step besides the library's own
launches: torch.profiler over a few steps of bench.py's step (config 3 by default), grouped by operator.
Usage: python tools/step_trace.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from gsplat_mi355.camera import orbit_camera  # noqa: E402
from gsplat_mi355.render import Pipe, l1_loss, render  # noqa: E402
from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
N, W, H, deg, tail, do_bwd = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev)
for f in GaussianCloud.FIELDS:
    getattr(cloud, f).requires_grad_(True)
cams = [orbit_camera(f, W, H, device=dev) for f in range(16)]
gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
pipe = Pipe()


def step(i):
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).grad = None
    pkg = render(cams[i], cloud, pipe, bg)
    l1_loss(pkg.render, gt).backward()


for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(4, 8):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
names = {}
for e in evs:
    names.setdefault(e.name[:90], []).append(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total)
print("\nGPU activities over 4 steps:")
for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print("%5d x %8.1f us  %s" % (len(v), sum(v) / len(v), k))
