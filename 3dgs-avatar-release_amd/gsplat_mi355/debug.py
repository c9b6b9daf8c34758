"""Parity-test introspection: runs the two forward phases through the C ABI and copies every
intermediate of the opaque state buffers back as numpy arrays (gs_*_field of include/gsplat_mi355.h)."""
import ctypes

import numpy as np
import torch

from gsplat_mi355 import _lib
from diff_gaussian_rasterization import GaussianRasterizationSettings, _make_args, _f32c


def _view(owner, ptr, nbytes, dtype):
    """numpy copy of `nbytes` at device address `ptr`, which lies inside the uint8 tensor `owner`."""
    off = ptr - owner.data_ptr()
    assert 0 <= off and off + nbytes <= owner.numel()
    return owner[off:off + nbytes].cpu().numpy().view(dtype).copy()


def forward_state(settings, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None):
    """Runs preprocess + render through the C ABI; returns dict(color, radii, geom=..., binning=..., image=...)."""
    L = _lib.load()
    dev = means3D.device
    means3D = _f32c(means3D, "means3D")
    shs, colors_precomp = _f32c(shs, "shs"), _f32c(colors_precomp, "colors_precomp")
    opacities = _f32c(opacities, "opacities")
    scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov")
    P = int(means3D.shape[0])
    W, H = int(settings.image_width), int(settings.image_height)
    keep = []
    with torch.cuda.device(dev):
        a = _make_args(settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, keep)
        stream = torch.cuda.current_stream(dev)
        sptr = ctypes.c_void_p(stream.cuda_stream)
        gb = _lib.nbytes(L.gs_geom_bytes, P)
        ib = _lib.nbytes(L.gs_image_bytes_for, ctypes.byref(a))
        geom = torch.zeros(gb, dtype=torch.uint8, device=dev)
        img = torch.zeros(ib, dtype=torch.uint8, device=dev)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int64).pin_memory()
        _lib.check(L.gs_forward_preprocess(ctypes.byref(a), geom.data_ptr(), gb, img.data_ptr(), ib, radii.data_ptr(),
                                           count.data_ptr(), sptr))
        stream.synchronize()
        D = int(count.item())
        bb = _lib.nbytes(L.gs_binning_bytes, D, W, H)
        binning = torch.zeros(bb, dtype=torch.uint8, device=dev)
        color = torch.zeros(3, H, W, device=dev)
        _lib.check(L.gs_forward_render(ctypes.byref(a), geom.data_ptr(), gb, binning.data_ptr(), bb, img.data_ptr(), ib,
                                       D, color.data_ptr(), sptr))
        counts = torch.zeros(2, dtype=torch.int64, device=dev)
        _lib.check(L.gs_pair_stats(ctypes.byref(a), geom.data_ptr(), gb, binning.data_ptr(), bb, img.data_ptr(), ib, D,
                                   counts.data_ptr(), sptr))
        stream.synchronize()
        pairs_valid, pairs_walked = (int(v) for v in counts.cpu())

        def field(fn, *args):
            out = ctypes.c_void_p(0)
            _lib.check(fn(*args, ctypes.byref(out)))
            return out.value
        gx, gy = (W + 15) // 16, (H + 15) // 16
        g = dict(
            depths=_view(geom, field(L.gs_geom_field, geom.data_ptr(), P, 0), 4 * P, np.float32),
            tiles_touched=_view(geom, field(L.gs_geom_field, geom.data_ptr(), P, 1), 4 * P, np.uint32),
            rec=_view(geom, field(L.gs_geom_field, geom.data_ptr(), P, 2), 48 * P, np.float32).reshape(P, 12),
            clamped=_view(geom, field(L.gs_geom_field, geom.data_ptr(), P, 3), 4 * P, np.uint32),
            sorted_idx=_view(geom, field(L.gs_geom_field, geom.data_ptr(), P, 4), 4 * P, np.uint32),
        ) if P > 0 else {}
        b = dict(
            point_list=_view(binning, field(L.gs_binning_field, binning.data_ptr(), D, W, H, 0), 4 * D, np.uint32),
        ) if D > 0 else dict(point_list=np.zeros(0, np.uint32))
        im = dict(
            ranges=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 0), 8 * gx * gy, np.uint32).reshape(-1, 2),
            n_contrib=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 1), 4 * W * H, np.uint32).reshape(H, W),
            final_T=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 2), 4 * W * H, np.float32).reshape(H, W),
            qcount=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 3), 16 * gx * gy, np.uint32).reshape(-1, 4),
            # launch order of the tiles (heaviest first); bit 31: rendered by four waves per quadrant (small images)
            order=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 5), 4 * gx * gy, np.uint32),
        )
        # the chunk-parallel forward's work list and hand-off words (gs_tuning "fwd4" = 2 on a small image; else absent)
        probe = ctypes.c_void_p(0)
        if L.gs_image_field(img.data_ptr(), W, H, 6, ctypes.byref(probe)) == 0:  # GS_OK
            hdr = _view(img, field(L.gs_image_field, img.data_ptr(), W, H, 6), 64, np.uint32)
            nu = int(hdr[0])
            im["chunks"] = dict(
                hdr=hdr,
                units=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 7), 8 * nu, np.uint32).reshape(-1, 2),
                flags=_view(img, field(L.gs_image_field, img.data_ptr(), W, H, 8), 16 * nu, np.uint32))
    # the tile of every list entry: the lists are stored tile after tile, so the ranges say it (upstream keeps the tile
    # id in the high half of its sort keys)
    r = im["ranges"].astype(np.int64)
    lens = np.maximum(r[:, 1] - r[:, 0], 0)
    nz = lens > 0
    starts = r[nz, 0]
    ok = int(lens.sum()) == D and (starts.size == 0 or (starts[0] == 0 and np.array_equal(starts[1:], r[nz, 1][:-1])))
    # (consistent only if the non-empty ranges tile [0, D) in tile order; otherwise a pattern no comparison accepts)
    b["tile_ids"] = np.repeat(np.arange(r.shape[0], dtype=np.uint32), lens) if ok else np.full(D, 0xFFFFFFFF, np.uint32)
    return dict(color=color.cpu().numpy(), radii=radii.cpu().numpy(), D=D, geom=g, binning=b, image=im,
                pairs_valid=pairs_valid, pairs_walked=pairs_walked)


def frame_stats(cam, cloud, pipe, bg):
    """(num_rendered D, mean n_contrib per pixel, quadrant hits, pairs composited) of one frame: reported beside every
    benchmark number.  Quadrant hits = (8x8 quadrant, Gaussian) entries up to each quadrant's last contributor: what the
    backward's wave per quadrant iterates over, 64 pixels per entry; pairs composited = the (pixel, Gaussian) pairs with
    alpha >= 1/255 before the pixel is done (gs_pair_stats): the useful part of 64 x quadrant hits."""
    import math
    settings = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=cloud.sh_degree, campos=cam.camera_center, prefiltered=False,
        debug=False)
    with torch.no_grad():
        st = forward_state(settings, cloud.xyz.detach(), cloud.opacity.detach(), shs=cloud.shs.detach(),
                           scales=cloud.scales.detach(), rotations=cloud.rotations.detach())
    return (int(st["D"]), float(st["image"]["n_contrib"].mean()), int(st["image"]["qcount"].astype("int64").sum()),
            int(st["pairs_valid"]))
