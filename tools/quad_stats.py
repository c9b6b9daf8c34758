"""Distribution of the per-tile list lengths and of the per-quadrant work (compacted entries up to the quadrant's last
contributor: one step of a render wave each) of a benchmark frame: is a render kernel's time its throughput, or the
dependent chain of its longest wave?  Usage: python tools/quad_stats.py [workload]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings  # noqa: E402
from gsplat_mi355.camera import orbit_camera  # noqa: E402
from gsplat_mi355.debug import forward_state  # noqa: E402
from gsplat_mi355.scenes import synthetic_cloud  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
N, W, H, deg, tail, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev, layout=bench.WORKLOAD_LAYOUT.get(wl, "box"))
cam = orbit_camera(0, W, H, device=dev)
s = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                  bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                  projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center, prefiltered=False,
                                  debug=False)
st = forward_state(s, cloud.xyz, cloud.opacity, shs=cloud.shs, scales=cloud.scales, rotations=cloud.rotations)
r = st["image"]["ranges"].astype(np.int64)
lens = r[:, 1] - r[:, 0]
q = st["image"]["qcount"].astype(np.int64).reshape(-1)
pct = [50, 90, 99, 99.9, 100]
print("%s: D = %d, tiles %d" % (wl, st["D"], len(lens)))
print("tile list length : mean %.0f  " % lens.mean() + "  ".join("p%g %d" % (p, np.percentile(lens, p)) for p in pct))
print("quadrant entries : mean %.0f  " % q.mean() + "  ".join("p%g %d" % (p, np.percentile(q, p)) for p in pct) + "  sum %d" % q.sum())
nsimd = 1024
print("sum / %d SIMDs = %.0f entries per SIMD; longest quadrant = %.2f of that" % (nsimd, q.sum() / nsimd, q.max() / (q.sum() / nsimd)))
