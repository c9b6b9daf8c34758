"""List statistics of the trained-avatar shaped scene (bench.py --workload avatar): how long the tile lists are, how
far the forward walks them and how many entries each quadrant wave evaluates."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings
from gsplat_mi355 import debug
from gsplat_mi355.camera import orbit_camera
from gsplat_mi355.scenes import synthetic_cloud

dev = torch.device("cuda:0")
N, W, H = 200000, 512, 512
cloud = synthetic_cloud(N, sh_degree=3, seed=0, device=dev, layout="body")
cam = orbit_camera(0, W, H, device=dev)
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                  cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
st = debug.forward_state(s, cloud.xyz, cloud.opacity, shs=cloud.shs, scales=cloud.scales, rotations=cloud.rotations)
r = st["image"]["ranges"]
ln = (r[:, 1] - r[:, 0]).astype(np.int64)
nz = ln > 0
gx = W // 16
nc = st["image"]["n_contrib"].reshape(H // 16, 16, gx, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
ncmax = nc.max(1)
print("D", st["D"], "tiles with entries", int(nz.sum()), "of", len(ln))
print("list length over those: mean %.0f max %d p50 %d p90 %d p99 %d" % (ln[nz].mean(), ln.max(), np.percentile(ln[nz], 50),
                                                                         np.percentile(ln[nz], 90), np.percentile(ln[nz], 99)))
print("last contributor (max n_contrib per tile): mean %.0f max %d p90 %d; sum / sum len = %.3f" % (
    ncmax[nz].mean(), ncmax.max(), np.percentile(ncmax[nz], 90), ncmax.sum() / ln.sum()))
print("radii mean %.1f max %d; tiles_touched mean %.2f" % (st["radii"].mean(), st["radii"].max(), st["geom"]["tiles_touched"].mean()))
qc = st["image"]["qcount"].astype(np.int64)
print("compacted entries up to last contributor per quadrant: sum %d (%.3f D), mean %.0f max %d p99 %d" % (
    qc.sum(), qc.sum() / st["D"], qc[qc > 0].mean(), qc.max(), np.percentile(qc[qc > 0], 99)))
# how far each quadrant's walk goes in the TILE list: the max n_contrib over its 64 pixels
ncq = st["image"]["n_contrib"].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64).max(1)
print("walk depth in the tile list per quadrant (last contributor position): mean %.0f max %d p99 %d" % (
    ncq[ncq > 0].mean(), ncq.max(), np.percentile(ncq[ncq > 0], 99)))
