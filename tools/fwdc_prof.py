"""Where the chunk-parallel forward of the long tiles spends its time (avatar-shaped frames).  Needs the variant build
  GSPLAT_VARIANT=fwdcprof GSPLAT_EXTRA_HIPCC_FLAGS=-DFWDC_PROF python 3dgs-avatar-release_amd/build.py
and GSPLAT_LIB_PATH pointing at it (GSPLAT_FWD4=2): every chunk wave stores the constant 100 MHz clock at
  item taken / pass A done / look-back done / pass B done / ticket taken (/ quadrant finished)
into a ninth slot of its record; this script renders a few frames through the C ABI, reads the records of the last one
and prints the timeline of the longest tiles and the distribution over all items."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import bench
    from gsplat_mi355 import _lib
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.scenes import synthetic_cloud
    from diff_gaussian_rasterization import GaussianRasterizationSettings, _make_args
    wl = sys.argv[1] if len(sys.argv) > 1 else "avatar"
    N, W, H, deg, tail, _ = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev, layout=bench.WORKLOAD_LAYOUT.get(wl, "box"))
    cam = orbit_camera(0, W, H, device=dev)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=cloud.sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
    L = _lib.load()
    keep = []
    with torch.cuda.device(dev):
        a = _make_args(settings, cloud.xyz.detach(), cloud.shs.detach(), None, cloud.opacity.detach(), cloud.scales.detach(),
                       cloud.rotations.detach(), None, keep)
        stream = torch.cuda.current_stream(dev)
        sptr = ctypes.c_void_p(stream.cuda_stream)
        gb = _lib.nbytes(L.gs_geom_bytes, N)
        ib = _lib.nbytes(L.gs_image_bytes_for, ctypes.byref(a))
        geom = torch.zeros(gb, dtype=torch.uint8, device=dev)
        img = torch.zeros(ib, dtype=torch.uint8, device=dev)
        radii = torch.zeros(N, dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int64).pin_memory()
        color = torch.zeros(3, H, W, device=dev)
        binning = None
        for frame in range(4):
            _lib.check(L.gs_forward_preprocess(ctypes.byref(a), geom.data_ptr(), gb, img.data_ptr(), ib, radii.data_ptr(),
                                               count.data_ptr(), sptr))
            stream.synchronize()
            D = int(count.item())
            bb = _lib.nbytes(L.gs_binning_bytes, D, W, H)
            if binning is None:
                binning = torch.zeros(bb, dtype=torch.uint8, device=dev)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record(stream)
            _lib.check(L.gs_forward_render(ctypes.byref(a), geom.data_ptr(), gb, binning.data_ptr(), bb, img.data_ptr(), ib, D,
                                           color.data_ptr(), sptr))
            t1.record(stream)
            stream.synchronize()
            print("frame %d: D %d, binning + render %.1f us" % (frame, D, 1000.0 * t0.elapsed_time(t1)))

        def field(which, nbytes, dtype):
            out = ctypes.c_void_p(0)
            _lib.check(L.gs_image_field(img.data_ptr(), W, H, which, ctypes.byref(out)))
            off = out.value - img.data_ptr()
            return img[off:off + nbytes].cpu().numpy().view(dtype).copy()
        hdr = field(6, 64, np.uint32)
        nunits, ch = int(hdr[0]), int(hdr[1])
        print("units %d, entries per chunk %d, tiles per XCD %s" % (nunits, ch, hdr[4:12].tolist()))
        if nunits == 0:
            return
        units = field(7, 8 * nunits, np.uint32).reshape(-1, 2)
        flags = field(8, 16 * nunits, np.uint32)
        SL = 9
        rec = field(9, nunits * 4 * SL * 64 * 4, np.uint32).reshape(nunits * 4, SL, 64)
    ts = rec[:, 8, :6].astype(np.int64)          # [item][6]
    meta = rec[:, 8, 6]
    xcc, blk = meta & 0xFF, meta >> 8
    tile = np.repeat(units[:, 0], 4)
    c = np.repeat(units[:, 1] & 0xFFFF, 4)
    nch = np.repeat(units[:, 1] >> 16, 4)
    q = np.tile(np.arange(4), nunits)
    hits = (flags & 0x7FFFFFFF).astype(np.int64) - 1
    dead = flags >> 31
    t_all = ts[:, 0].min()
    us = (ts - t_all) / 100.0
    dur = np.diff(us[:, :5], axis=1)
    print("all %d items: taken at %.1f .. %.1f us; ticket at up to %.1f us; finishers done by %.1f us" % (
        len(us), us[:, 0].min(), us[:, 0].max(), us[:, 4].max(), ((ts[:, 5][ts[:, 5] > 0] - t_all) / 100.0).max()))
    for name, col in (("pass A", 0), ("look-back", 1), ("pass B", 2), ("ticket", 3)):
        d = dur[:, col]
        print("  %-9s median %.2f  p90 %.2f  max %.2f us" % (name, np.median(d), np.percentile(d, 90), d.max()))
    fin = ts[:, 5] > 0
    print("  finish    median %.2f  p90 %.2f  max %.2f us (%d quadrants)" % (
        np.median((ts[fin, 5] - ts[fin, 4]) / 100.0), np.percentile((ts[fin, 5] - ts[fin, 4]) / 100.0, 90),
        ((ts[fin, 5] - ts[fin, 4]) / 100.0).max(), int(fin.sum())))
    print("  items per XCD:", np.bincount(xcc, minlength=8).tolist())
    order = np.argsort(-nch, kind="stable")
    shown = []
    for i in order:
        if tile[i] not in shown:
            shown.append(tile[i])
        if len(shown) == 2:
            break
    for t in shown:
        print("tile %d, quadrant 0 (us: taken / pass A / look-back / pass B / ticket [/ finished])" % t)
        for i in np.nonzero((tile == t) & (q == 0))[0]:
            print("  c %2d/%d xcc %d block %5d hits %3d dead %d : %s" % (
                c[i], nch[i], xcc[i], blk[i], hits[i], dead[i],
                "  ".join("%7.2f" % x for x in (us[i, :5].tolist() + ([us[i, 5]] if ts[i, 5] > 0 else [])))))


if __name__ == "__main__":
    main()
