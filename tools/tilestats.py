import sys, os, math
sys.path.insert(0, "3dgs-avatar-release_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from gsplat_mi355.scenes import synthetic_cloud
from gsplat_mi355.camera import orbit_camera
from gsplat_mi355 import debug
from diff_gaussian_rasterization import GaussianRasterizationSettings
dev = torch.device("cuda:0")
N, W, H = 200000, 1024, 1024
cloud = synthetic_cloud(N, sh_degree=3, seed=0, device=dev)
cam = orbit_camera(0, W, H, device=dev)
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), torch.zeros(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
st = debug.forward_state(s, cloud.xyz, cloud.opacity, shs=cloud.shs, scales=cloud.scales, rotations=cloud.rotations)
r = st["image"]["ranges"]; ln = (r[:,1]-r[:,0]).astype(np.int64)
nc = st["image"]["n_contrib"].reshape(64,16,64,16).transpose(0,2,1,3).reshape(4096,256)
ncmax = nc.max(1)
print("D", st["D"], "len mean %.0f max %d p50 %d p90 %d p99 %d" % (ln.mean(), ln.max(), np.percentile(ln,50), np.percentile(ln,90), np.percentile(ln,99)))
print("processed (max n_contrib per tile): mean %.0f max %d p90 %d" % (ncmax.mean(), ncmax.max(), np.percentile(ncmax,90)))
print("sum processed / sum len = %.3f" % (ncmax.sum()/ln.sum()))
print("radii mean %.1f max %d; tiles_touched mean %.1f" % (st["radii"].mean(), st["radii"].max(), st["geom"]["tiles_touched"].mean()))
rec = st["geom"]["rec"]; print("opacity mean %.3f median %.3f frac<1/255 %.4f" % (rec[:,5].mean(), np.median(rec[:,5]), (rec[:,5]<1/255).mean()))
qc = st["image"]["qcount"].astype(np.int64)
print("compacted entries up to last contributor: sum %d (%.3f D), per quadrant mean %.0f max %d; steps incl. fill %d" % (qc.sum(), qc.sum()/st["D"], qc.mean(), qc.max(), (qc + 63 * (qc > 0)).sum()))
# list scheduling of the per-quadrant backward work (steps = m + 63) on 1024 SIMDs x K resident waves
steps = (qc + 63 * (qc > 0)).astype(np.int64)
steps = steps[steps > 0]
srt = np.sort(steps)[::-1]
print("quadrants with work %d; steps: max %d p99 %d p90 %d p50 %d; total %d" % (len(srt), srt[0], np.percentile(srt, 99), np.percentile(srt, 90), np.percentile(srt, 50), srt.sum()))
import heapq
for slots in (1024, 1024 * 5):
    h = [0] * slots
    heapq.heapify(h)
    for w in srt:
        t = heapq.heappop(h)
        heapq.heappush(h, t + int(w))
    mk = max(h)
    print("slots %d: makespan %d steps vs ideal %.0f (critical path %d)" % (slots, mk, srt.sum() / slots, srt[0]))
np.save("gpurun_out/qcount_config3.npy", qc)
