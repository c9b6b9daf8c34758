"""GPU parity: the HIP path (through the C ABI / the drop-in Python packages) against the CPU oracle on
the same seeded inputs.  Bar (BASELINE.json north_star): integer / index outputs bit-exact;
image and gradients within 1e-5 relative to the per-tensor maximum, fp32.

A note on the float bar: alpha is compared against 1/255 and T against 1e-4 per (pixel, Gaussian)
pair; the device's v_exp_f32 and glibc's expf differ in the last bits, so out of ~1e6-1e9 pairs a few
land on opposite sides of a threshold (the real CUDA kernel has the same property against any CPU
restatement).  Such a flip moves one pixel by up to T*alpha*|c| ~ 4e-3.

Round 4: that explanation is CHECKED, not assumed, in the full-size tests (`_attribute`).  Every pixel where the
device and the oracle disagree (last contributor, final T beyond 1e-4 relative, colour beyond 1e-5 of the maximum) is
handed to the oracle's search (or_explain_pixels): it must find decisions, each within EXPLAIN_EPS of its threshold
(|power| <= 1e-5 of its terms, |255 alpha - 1| <= 2e-5, |1e4 T (1 - alpha) - 1| <= 2e-4), whose flipping makes the
oracle's own walk reproduce the device's pixel -- unattributed pixels = 0 is asserted.  Image, final_T, n_contrib and
ALL gradients are then compared with the oracle CONDITIONED on those decisions (the same arithmetic, the named
near-threshold decisions taken the device's way), with no exempt fraction for the image and n_contrib.  The smaller
tests keep the older form: a bulk bound (<= 1e-5 of the tensor maximum for all but a small fraction of the elements)
plus a cap on the outliers (<= 1e-2 absolute for the image)."""
import math
import os

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _settings(cam, cloud, bg, dev, scale_modifier=1.0, debug=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.tensor(bg, dtype=torch.float32, device=dev),
        scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), sh_degree=cloud.sh_degree, campos=cam.camera_center.to(dev),
        prefiltered=False, debug=debug)


def _inputs(cloud, cam, color_mode, cov_mode, dev, scale_modifier=1.0):
    kw = {}
    if color_mode == "sh":
        kw["shs"] = cloud.shs.to(dev)
    else:
        kw["colors_precomp"] = helpers.precomp_colors(cloud, cam).to(dev)
    if cov_mode == "scale_rot":
        kw["scales"] = cloud.scales.to(dev)
        kw["rotations"] = cloud.rotations.to(dev)
    else:
        kw["cov3D_precomp"] = helpers.covariance6_cpu(cloud, scale_modifier).to(dev)
    return kw


def _bulk_close(got, want, tol=TOL, frac=2e-5, name="", cap=None):
    """|got - want| / max|want| <= tol for all but `frac` of the elements, and <= `cap` for ALL of them (the outliers
    are threshold flips of single (pixel, Gaussian) pairs: bounded, see the module docstring)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    m = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want) / m
    bad = (err > tol).mean()
    assert bad <= frac, "%s: %.3g of elements beyond %g (max rel err %.3g)" % (name, bad, tol, err.max())
    if cap is not None:
        assert err.max() <= cap, "%s: outlier %.3g of the tensor maximum (cap %g)" % (name, err.max(), cap)
    return err.max()


GRAD_CAP = 2e-3  # no gradient element may be off by more than this fraction of its tensor's maximum

_REPORT = {}


def _report(case, tensor, got, want, extra=None):
    """Evidence for the float bar (profiles/parity_report.json): per tensor, the error relative to the tensor maximum --
    maximum, 99.99th percentile, fraction of elements beyond 1e-5 -- written to gpurun_out/parity_report.json (or
    $GSPLAT_PARITY_REPORT) by the full-size tests."""
    import json
    import os
    got = np.asarray(got, np.float64).reshape(-1)
    want = np.asarray(want, np.float64).reshape(-1)
    m = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want) / m
    rec = {"elements": int(err.size), "max_abs_of_reference": float(m), "max_rel_err": float(err.max()),
           "p9999_rel_err": float(np.quantile(err, 0.9999)), "median_rel_err": float(np.median(err)),
           "frac_beyond_1e-5": float((err > 1e-5).mean()), "frac_beyond_2e-5": float((err > 2e-5).mean()),
           "count_beyond_1e-4": int((err > 1e-4).sum())}
    if extra:
        rec.update(extra)
    _REPORT.setdefault(case, {})[tensor] = rec
    path = os.environ.get("GSPLAT_PARITY_REPORT")
    if path is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "gpurun_out", "parity_report.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        doc = {}
        if os.path.exists(path):
            try:
                doc = json.load(open(path))
            except Exception:
                doc = {}
        doc.setdefault(case, {})[tensor] = rec
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return rec


def _attribute(oracle, sc, fw, dev_color, dev_final_T, dev_n_contrib, tag):
    """Attribution of every device/oracle difference in the forward to threshold decisions (see the module docstring).
    Returns (the forward state of the oracle CONDITIONED on the decisions found, the override table).  Asserts that no
    differing pixel is left unexplained, and that the conditioned oracle then agrees with the device on every pixel:
    n_contrib exactly, final_T and colour within 1e-5 of the maximum, no exempt fraction."""
    oc, oT, on = fw["color"], fw["image"]["final_T"], fw["image"]["n_contrib"]
    m = max(float(np.abs(oc).max()), 1e-30)
    dev_color = np.asarray(dev_color, np.float32).reshape(oc.shape)
    dT = np.asarray(dev_final_T, np.float32).reshape(oT.shape)
    dn = np.asarray(dev_n_contrib, np.uint32).reshape(on.shape)
    differs = (dn != on) | (np.abs(dT - oT) > 1e-4 * np.maximum(dT, oT)) | (np.abs(dev_color - oc).max(0) > 1e-5 * m)
    pids = np.flatnonzero(differs.reshape(-1))
    status, ov = oracle.explain_pixels(sc, fw, pids, dev_color, dT, dn)
    im = oracle.render_forward(sc, fw["geom"], fw["binning"], ov)
    fwc = dict(fw, image=im, color=im["color"])
    err = np.abs(dev_color - im["color"]) / m
    kinds = {1: "skip", 2: "keep", 4: "stop", 8: "go"}
    rec = {"pixels": int(differs.size), "pixels_differing_before": int(pids.size), "unattributed_pixels": int((status < 0).sum()),
           "flipped_decisions": len(ov), "flips_per_pixel_max": int(status.max()) if status.size else 0,
           "largest_margin_of_a_flipped_decision": float(ov.margin.max()) if len(ov) else 0.0,
           "margins_allowed_power_alpha_T": list(oracle.EXPLAIN_EPS),
           "flips_by_outcome": {kinds.get(int(a), str(int(a))): int((ov.act == a).sum()) for a in np.unique(ov.act)},
           "image_max_rel_err_after": float(err.max()), "n_contrib_mismatches_after": int((dn != im["n_contrib"]).sum()),
           "final_T_max_abs_err_after": float(np.abs(dT - im["final_T"]).max())}
    _REPORT.setdefault(tag, {})["attribution"] = rec
    _report(tag, "color_vs_conditioned_oracle", dev_color, im["color"], extra=rec)
    assert rec["unattributed_pixels"] == 0, "%s: %d of %d differing pixels are not explained by decisions within %r of a threshold" % (
        tag, rec["unattributed_pixels"], pids.size, oracle.EXPLAIN_EPS)
    assert rec["n_contrib_mismatches_after"] == 0, tag
    assert err.max() <= TOL, "%s: image off by %.3g of the maximum from the conditioned oracle" % (tag, err.max())
    assert rec["final_T_max_abs_err_after"] <= TOL, tag
    return fwc, ov


def _conditioned_oracle(oracle, sc, fw, settings, xyz, opacity, kw, tag):
    """The device's per-pixel records of this frame (debug.forward_state: the same kernels as the wrapper's path, same
    bits) -> _attribute: the oracle conditioned on the attributed threshold decisions, and the override table."""
    from gsplat_mi355 import debug
    st = debug.forward_state(settings, xyz.detach(), opacity.detach(), **{k: v.detach() for k, v in kw.items()})
    fwc, ov = _attribute(oracle, sc, fw, st["color"], st["image"]["final_T"], st["image"]["n_contrib"], tag)
    return st, fwc, ov


# Small and stress scenes against the conditioned oracle: tensors of a few hundred to a few 10^4 elements whose maxima can be
# tiny sums of a handful of pairs -- the same 1e-5 of the tensor maximum, every element (no exempt fraction)
SMALL_TOL = 1e-5


# Gradients against the CONDITIONED oracle: the north star's 1e-5 of the tensor maximum.  Measured on all 25 full-size
# cases (profiles/r04_parity_report.json): NO element beyond 1e-5 (largest 8.1e-6: dL/dopacity on the avatar frame, sums
# over lists of thousands of entries in another order; typically 1-3e-6), with at most 50 flipped decisions per frame in
# the conditioning (config 5; 16 at config 3), each within 1.6e-5 of its threshold.  The allowance below -- one element
# in a million up to 3e-5 -- is head-room for rounding changes between builds, not something a recorded run uses.
# (Until round 3, against the plain oracle: 2e-4 of the elements exempt -- 1e-3 at config 1 -- up to 2e-3 of the maximum.)
COND_GRAD_FRAC = 1e-6
COND_GRAD_CAP = 3e-5


COMBOS = [("sh", "scale_rot", 3), ("precomp", "cov", 3), ("sh", "cov", 1), ("precomp", "scale_rot", 0), ("sh", "scale_rot", 2)]


@pytest.fixture(params=[0, 1], ids=["upstream-squares", "alpha-bounding-boxes"])
def tile_rect(request):
    """Runs a test in both tile-rectangle modes (GsFwdArgs.tile_rect) of the product; the oracle is put in the same
    mode through helpers.oracle_scene(..., tile_rect=...)."""
    import diff_gaussian_rasterization as dgr
    saved = dgr._TILE_RECT
    dgr._TILE_RECT = request.param
    yield request.param
    dgr._TILE_RECT = saved


@pytest.mark.parametrize("color_mode,cov_mode,deg", COMBOS)
def test_preprocess_and_binning_bit_exact(oracle, tile_rect, color_mode, cov_mode, deg):
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    n, W, H = 3000, 200, 136  # ragged: not a multiple of 16
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=21, scale_mul=1.3)
    cloud.xyz[:50, 2] = -3.5   # behind the near plane
    cloud.xyz[50:80, 0] *= 3.0  # beyond the 1.3*tanfov clamp
    cloud.shs[:, 0] -= 1.2 * (torch.arange(n) % 5 == 0).float()[:, None]
    bg = (0.1, 0.3, 0.2)
    sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode=color_mode, cov_mode=cov_mode, tile_rect=tile_rect)
    fw = oracle.forward(sc)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             **_inputs(cloud, cam, color_mode, cov_mode, dev))
    og, ob, oi = fw["geom"], fw["binning"], fw["image"]
    # --- integers: bit-exact
    assert np.array_equal(st["radii"], og["radii"])
    assert np.array_equal(st["geom"]["tiles_touched"], og["tiles_touched"])
    assert st["D"] == ob["D"]
    vis = og["radii"] > 0
    rec = st["geom"]["rec"]
    rmin = rec[:, 10].view(np.uint32)
    rsz = rec[:, 11].view(np.uint32)
    assert np.array_equal((rmin & 0xFFFF)[vis], og["rect"][vis, 0]) and np.array_equal((rmin >> 16)[vis], og["rect"][vis, 1])
    assert np.array_equal((rsz & 0xFFFF)[vis], (og["rect"][:, 2] - og["rect"][:, 0])[vis])
    assert np.array_equal((rsz >> 16)[vis], (og["rect"][:, 3] - og["rect"][:, 1])[vis])
    assert np.array_equal(st["geom"]["depths"].view(np.uint32), og["depths"].view(np.uint32))  # the sort key
    # --- per-Gaussian floats feeding the integers: also bit-exact (no-contraction build)
    assert np.array_equal(rec[vis, 0:2].view(np.uint32), og["xy"][vis].view(np.uint32))
    assert np.array_equal(rec[vis, 2:5].view(np.uint32), og["conic_opacity"][vis, :3].view(np.uint32))
    assert np.array_equal(rec[vis, 5], og["conic_opacity"][vis, 3])
    assert np.abs(rec[vis, 6:9] - og["rgb"][vis]).max() <= 1e-6
    if color_mode == "sh":
        cl = st["geom"]["clamped"]
        bits = np.stack([(cl >> c) & 1 for c in range(3)], 1)
        assert np.array_equal(bits[vis], og["clamped"][vis])
    # --- the sorted (tile, depth) list and the tile ranges: bit-exact
    assert np.array_equal(st["binning"]["point_list"], ob["point_list"])
    assert np.array_equal(st["binning"]["tile_ids"], (ob["keys"] >> np.uint64(32)).astype(np.uint32))
    nz = ob["ranges"][:, 1] > ob["ranges"][:, 0]
    assert np.array_equal(st["image"]["ranges"][nz], ob["ranges"][nz])
    assert (st["image"]["ranges"][~nz, 1] == st["image"]["ranges"][~nz, 0]).all()
    # --- image
    _bulk_close(st["color"], fw["color"], name="color")
    assert np.abs(st["color"] - fw["color"]).max() < 1e-2
    _bulk_close(st["image"]["final_T"], oi["final_T"], name="final_T")
    assert (st["image"]["n_contrib"] != oi["n_contrib"]).mean() < 1e-4


@pytest.mark.parametrize("color_mode,cov_mode,deg", COMBOS)
def test_backward_matches_oracle(oracle, tile_rect, color_mode, cov_mode, deg):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 4000, 160, 120
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=31, scale_mul=1.3)
    cloud.xyz[:50, 2] = -3.5
    cloud.xyz[50:80, 0] *= 3.0
    cloud.shs[:, 0] -= 1.2 * (torch.arange(n) % 5 == 0).float()[:, None]
    bg = (0.25, 0.5, 0.75)
    sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode=color_mode, cov_mode=cov_mode, tile_rect=tile_rect)
    fw = oracle.forward(sc)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5))
    from gsplat_mi355 import debug
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             **_inputs(cloud, cam, color_mode, cov_mode, dev))
    tag = "small backward %s/%s deg %d tile_rect=%d" % (color_mode, cov_mode, deg, tile_rect)
    fwc, ov = _attribute(oracle, sc, fw, st["color"], st["image"]["final_T"], st["image"]["n_contrib"], tag)
    want = oracle.backward(sc, fwc, gimg.numpy(), ov)

    kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, color_mode, cov_mode, dev).items()}
    means3D = cloud.xyz.to(dev).requires_grad_(True)
    means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
    opac = cloud.opacity.to(dev).requires_grad_(True)
    rast = GaussianRasterizer(_settings(cam, cloud, bg, dev))
    color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw)
    (color * gimg.to(dev)).sum().backward()
    assert np.array_equal(radii.cpu().numpy(), fw["radii"])
    assert np.array_equal(color.detach().cpu().numpy(), st["color"])  # (the wrapper and the two-phase C calls: same bits)
    got = dict(means3D=means3D.grad, means2D=means2D.grad, opacities=opac.grad)
    names = dict(shs="sh", colors_precomp="colors_precomp", scales="scales", rotations="rotations",
                 cov3D_precomp="cov3D_precomp")
    for k, v in kw.items():
        got[names[k]] = v.grad
    for name, gt in got.items():
        assert gt is not None, name
        w = want[name].reshape(gt.shape)
        # every element within 1e-5 of the tensor maximum of the oracle conditioned on the attributed decisions
        _bulk_close(gt.cpu().numpy(), w, tol=1e-5, frac=0.0, name=name)
        assert np.abs(w).max() > 0
    culled = fw["radii"] == 0
    for name, gt in got.items():
        assert (gt.cpu().numpy()[culled] == 0).all(), name


def test_pair_stats_counter_against_the_oracle(oracle):
    """gs_pair_stats (bench.py's `pairs_valid`): the pairs actually composited, counted on the device from a finished
    forward state, against the oracle's own count of blended pairs (n_blended; a handful of threshold decisions may fall
    the other way: 1e-4 relative), and the walked pairs = the sum of the device's n_contrib, exactly."""
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    n, W, H = 12000, 272, 200
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=9, scale_mul=1.5)
    bg = (0.0, 0.0, 0.0)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    fw = oracle.forward(helpers.oracle_scene(cloud, cam, bg=bg), margin=True)
    want = int(fw["image"]["n_blended"].astype(np.int64).sum())
    assert st["pairs_walked"] == int(st["image"]["n_contrib"].astype(np.int64).sum())
    assert want > 100000 and abs(st["pairs_valid"] - want) <= 1e-4 * want
    assert st["pairs_valid"] <= 64 * int(st["image"]["qcount"].astype(np.int64).sum())  # composited pairs are among the evaluated ones


def test_empty_and_degenerate_inputs(oracle):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    cloud, cam = helpers.cloud_and_camera(64, 48, 48, sh_degree=0, seed=1)
    cloud.xyz[:, 2] = -10.0  # everything culled: D == 0
    bg = (0.3, 0.2, 0.1)
    rast = GaussianRasterizer(_settings(cam, cloud, bg, dev))
    means3D = cloud.xyz.to(dev).requires_grad_(True)
    means2D = torch.zeros(64, 3, device=dev, requires_grad=True)
    color, radii = rast(means3D=means3D, means2D=means2D, opacities=cloud.opacity.to(dev), shs=cloud.shs.to(dev),
                        scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    assert (radii == 0).all()
    assert torch.allclose(color, torch.tensor(bg, device=dev)[:, None, None].expand_as(color))
    color.sum().backward()
    assert (means3D.grad == 0).all() and (means2D.grad == 0).all()


def test_argument_validation_matches_upstream_messages():
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    cloud, cam = helpers.cloud_and_camera(16, 32, 32, sh_degree=0, seed=1)
    rast = GaussianRasterizer(_settings(cam, cloud, (0, 0, 0), dev))
    x, o = cloud.xyz.to(dev), cloud.opacity.to(dev)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=x, means2D=x, opacities=o, scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=x, means2D=x, opacities=o, shs=cloud.shs.to(dev))
    with pytest.raises(RuntimeError, match="GPU tensor"):
        rast(means3D=x.cpu(), means2D=x, opacities=o, shs=cloud.shs.to(dev), scales=cloud.scales.to(dev),
             rotations=cloud.rotations.to(dev))


def test_debug_mode_dumps_the_arguments_when_the_native_call_fails(tmp_path, monkeypatch):
    """pipe.debug (configs/config.yaml:92, forwarded at gaussian_renderer/__init__.py:97): upstream's wrapper keeps the
    call's arguments aside and writes snapshot_fw.dump when the native forward raises.  Same here."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    cloud, cam = helpers.cloud_and_camera(16, 32, 32, sh_degree=3, seed=1)
    monkeypatch.chdir(tmp_path)
    rast = GaussianRasterizer(_settings(cam, cloud, (0, 0, 0), dev, debug=True))
    bad_shs = torch.zeros(16, 17, 3, device=dev)  # 17 coefficients per channel: rejected by the library (M <= 16)
    with pytest.raises(RuntimeError, match="bad argument"):
        rast(means3D=cloud.xyz.to(dev), means2D=cloud.xyz.to(dev), opacities=cloud.opacity.to(dev), shs=bad_shs,
             scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    dump = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert dump["settings"]["image_width"] == 32 and dump["settings"]["debug"] is True
    assert dump["tensors"][2].shape == (16, 17, 3) and dump["tensors"][0].device.type == "cpu"
    # without debug: same error, no dump
    (tmp_path / "snapshot_fw.dump").unlink()
    rast = GaussianRasterizer(_settings(cam, cloud, (0, 0, 0), dev, debug=False))
    with pytest.raises(RuntimeError):
        rast(means3D=cloud.xyz.to(dev), means2D=cloud.xyz.to(dev), opacities=cloud.opacity.to(dev), shs=bad_shs,
             scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    assert not (tmp_path / "snapshot_fw.dump").exists()


def test_mark_visible(oracle):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    cloud, cam = helpers.cloud_and_camera(5000, 64, 64, sh_degree=0, seed=2)
    cloud.xyz[:, 2] = torch.linspace(-4.0, 1.0, 5000)
    rast = GaussianRasterizer(_settings(cam, cloud, (0, 0, 0), dev))
    got = rast.markVisible(cloud.xyz.to(dev)).cpu().numpy()
    want = oracle.mark_visible(cloud.xyz.numpy(), cam.world_view_transform.numpy())
    assert got.dtype == np.bool_ and np.array_equal(got, want) and 0 < want.sum() < 5000


def test_distCUDA2_exact_vs_brute_force(oracle):
    from simple_knn._C import distCUDA2
    dev = torch.device("cuda:0")
    for n, seed in [(4096, 0), (5000, 1), (777, 2), (4, 3)]:
        pts = torch.rand(n, 3, generator=torch.Generator().manual_seed(seed)) * 2 - 1
        if n == 5000:
            pts[:, 2] *= 1e-3  # nearly planar: degenerate Morton axis
        got = distCUDA2(pts.to(dev)).cpu().numpy()
        want = oracle.dist2(pts.numpy())
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), n
    # duplicates: distance 0, the caller clamps to 1e-7 (scene/gaussian_model.py:186)
    pts = torch.rand(100, 3).repeat(4, 1)
    got = distCUDA2(pts.to(dev))
    assert (got == 0).all()
    assert (torch.clamp_min(got, 0.0000001) == 1e-7).all()


def test_knn_at_the_reference_sizes_every_search_variant(oracle):
    """distCUDA2 on 50 000 points -- the size the reference calls it with (dataset/zjumocap.py:412 ->
    scene/gaussian_model.py:186) -- bit-exact against the O(N^2) brute force, and distCUDA2 / knn_points K = 6 at 140k
    and 270k points so that the 32- and 64-queries-per-wave variants of the search (knn.hip picks 16 / 32 / 64 from
    the query count) are exercised too.  At the two large sizes the brute force runs on a random 20 000-query subset
    (every GPU row of that subset must match bit for bit)."""
    from gsplat_mi355.knn import knn_points
    from simple_knn._C import distCUDA2
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(50)
    # 50k: a body-like point set (points near a few capsule surfaces) as well as a uniform cube
    u = rng.random((50000, 3)).astype(np.float32) * 2 - 1
    t = rng.random(50000).astype(np.float32)
    ang = (rng.random(50000) * 2 * np.pi).astype(np.float32)
    shell = np.stack([0.12 * np.cos(ang), t * 1.6 - 0.8, 0.12 * np.sin(ang)], 1) + 0.004 * rng.normal(size=(50000, 3))
    for name, pts in (("uniform", u), ("shell", shell.astype(np.float32))):
        got = distCUDA2(torch.from_numpy(pts).to(dev)).cpu().numpy()
        want = oracle.dist2(pts)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    for n in (140000, 270000):
        pts = (rng.random((n, 3)).astype(np.float32) * 2 - 1)
        pts[:, 1] *= 0.3  # anisotropic cloud: boxes of unequal extent
        x = torch.from_numpy(pts).to(dev)
        sub = np.sort(rng.choice(n, 20000, replace=False))
        want_d, want_i = oracle.knn_points(pts[sub], pts, 6)
        res = knn_points(x[None], x[None], K=6, return_sorted=True)
        assert np.array_equal(res.dists[0].cpu().numpy()[sub], want_d), n
        assert np.array_equal(res.idx[0].cpu().numpy()[sub], want_i), n
        # distCUDA2 = mean of the three nearest OTHER points: entries 1..3 of the K = 6 list (entry 0 is the point itself)
        assert np.array_equal(want_i[:, 0], sub)
        want3 = ((want_d[:, 1] + want_d[:, 2]) + want_d[:, 3]) / np.float32(3.0)
        got3 = distCUDA2(x).cpu().numpy()[sub]
        assert np.array_equal(got3.view(np.uint32), want3.astype(np.float32).view(np.uint32)), n


@pytest.mark.parametrize("with_opacity", [False, True])
def test_long_lists_on_a_small_image_backward_in_chunks(oracle, with_opacity):
    """A trained-avatar shaped frame: 30000 Gaussians on a thin shell covering a fraction of a 160 x 160 image, i.e. few
    quadrants with lists of thousands of entries.  On images of up to 2048 tiles the forward checkpoints the compositing
    state every 256 compacted entries and the backward runs one wave per (quadrant, chunk) from those checkpoints
    (common.h, BWD_CH): gradients against the oracle, and against the one-wave-per-quadrant walk of the same frame
    (gs_tuning "bwd_chunks" = 0), with and without the fused opacity channel."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsplat_mi355 import _lib, debug
    dev = torch.device("cuda:0")
    n, W, H = 30000, 160, 160
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=5, layout="body")
    cloud.opacity = cloud.opacity * 0.25  # translucent: the walks go deep into the lists
    bg = (0.1, 0.2, 0.3)
    sc = helpers.oracle_scene(cloud, cam, bg=bg)
    fw = oracle.forward(sc)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3))
    gopa = torch.randn(1, H, W, generator=torch.Generator().manual_seed(4))

    def run(chunks):
        _lib.tuning("bwd_chunks", chunks)
        try:
            kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
            means3D = cloud.xyz.to(dev).requires_grad_(True)
            means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
            opac = cloud.opacity.to(dev).requires_grad_(True)
            rast = GaussianRasterizer(_settings(cam, cloud, bg, dev))
            if with_opacity:
                color, radii, opa = rast(means3D=means3D, means2D=means2D, opacities=opac, with_opacity=True, **kw)
                ((color * gimg.to(dev)).sum() + (opa * gopa.to(dev)).sum()).backward()
            else:
                color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw)
                (color * gimg.to(dev)).sum().backward()
            out = dict(color=color.detach(), means3D=means3D.grad, means2D=means2D.grad, opacities=opac.grad)
            out.update({k: v.grad for k, v in kw.items()})
            return {k: v.cpu().numpy() for k, v in out.items()}
        finally:
            _lib.tuning("bwd_chunks", 1)

    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             **_inputs(cloud, cam, "sh", "scale_rot", dev))
    assert int(st["image"]["qcount"].max()) > 3 * 256  # quadrants of more than three chunks are among them
    a, b = run(1), run(0)
    assert np.array_equal(a["color"], b["color"])  # (the forward is the same kernel: the checkpoints are a side output)
    for k in a:
        scale = np.abs(b[k]).max()
        assert scale > 0 and np.abs(a[k] - b[k]).max() <= 2e-6 * scale, (k, np.abs(a[k] - b[k]).max() / scale)
    if not with_opacity:
        # against the oracle conditioned on the attributed threshold decisions (a translucent thin shell: walks of hundreds of
        # entries per pixel, some ending at the 1e-4 threshold): every gradient element within 1e-5 of its tensor's maximum
        assert np.array_equal(a["color"], st["color"])
        fwc, ov = _attribute(oracle, sc, fw, st["color"], st["image"]["final_T"], st["image"]["n_contrib"], "long lists, small image, chunked backward")
        want = oracle.backward(sc, fwc, gimg.numpy(), ov)
        names = dict(shs="sh", scales="scales", rotations="rotations", means3D="means3D", means2D="means2D", opacities="opacities")
        for k, ok in names.items():
            _bulk_close(a[k], want[ok].reshape(a[k].shape), tol=SMALL_TOL, frac=0.0, name="chunked " + k)


def test_long_lists_on_a_large_image_follow_the_previous_frames_statistics(oracle, monkeypatch):
    """Above 2048 tiles the few-long-lists machinery (four-wave forward, backward in chunks) is the caller's choice
    (GsFwdArgs.long_lists).  With the OPT-IN GSPLAT_LONG_LISTS=auto the wrapper takes it from the statistics the forward
    of the PREVIOUS frame of the same shape left in a pinned word.  A trained-avatar shaped frame on 800 x 800 (2500
    tiles): the first frame runs without it and reports its long lists, the second runs with it; both match the oracle,
    and each other to fp32 rounding.  (The default is a fixed setting: test_large_images_do_not_depend_on_call_history.)"""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 80000, 800, 800
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=6, layout="body")
    cloud.opacity = cloud.opacity * 0.3
    bg = (0.3, 0.2, 0.1)
    sc = helpers.oracle_scene(cloud, cam, bg=bg)
    fw = oracle.forward(sc)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9))
    monkeypatch.setattr(dgr, "_LONG_LISTS", "auto")
    dgr._frame_stats.pop((0, W, H), None)

    def frame():
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        color, radii = GaussianRasterizer(_settings(cam, cloud, bg, dev))(means3D=means3D, means2D=means2D, opacities=opac, **kw)
        mode = color.grad_fn.long_lists
        (color * gimg.to(dev)).sum().backward()
        torch.cuda.synchronize()
        out = dict(color=color.detach(), means3D=means3D.grad, means2D=means2D.grad, opacities=opac.grad)
        out.update({k: v.grad for k, v in kw.items()})
        return mode, {k: v.cpu().numpy() for k, v in out.items()}

    m0, a = frame()
    st = dgr._frame_stats[(0, W, H)]
    assert m0 == 0 and int(st[0]) > 0 and int(st[1]) > 1000  # long lists seen and reported
    m1, b = frame()
    assert m1 == 1
    assert np.abs(a["color"] - b["color"]).max() <= 3e-6
    # (the second frame's kernels -- four-wave forward, chunked backward -- against the oracle conditioned on the attributed
    # decisions; the device records come from the same kernel set: the statistics word now says "long lists")
    from gsplat_mi355 import debug
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             **_inputs(cloud, cam, "sh", "scale_rot", dev))
    assert np.array_equal(st["color"], b["color"])
    fwc, ov = _attribute(oracle, sc, fw, st["color"], st["image"]["final_T"], st["image"]["n_contrib"], "long lists, large image (auto)")
    want = oracle.backward(sc, fwc, gimg.numpy(), ov)
    names = dict(shs="sh", scales="scales", rotations="rotations", means3D="means3D", means2D="means2D", opacities="opacities")
    for k in names:
        scale = np.abs(a[k]).max()
        assert np.abs(a[k] - b[k]).max() <= 3e-6 * scale, (k, np.abs(a[k] - b[k]).max() / scale)
        _bulk_close(b[k], want[names[k]].reshape(b[k].shape), tol=SMALL_TOL, frac=0.0, name="long lists " + k)
    monkeypatch.setattr(dgr, "_LONG_LISTS", "0")
    m2, c = frame()
    assert m2 == 0 and all(np.array_equal(a[k], c[k]) for k in a)  # forced off: the first frame's bits again


@pytest.mark.parametrize("pin", ["default", "0", "1"])
def test_large_images_do_not_depend_on_call_history(pin, monkeypatch):
    """Above 2048 tiles GsFwdArgs.long_lists is a setting fixed for the process (GSPLAT_LONG_LISTS = 0 by default, or 1),
    never a function of earlier frames: the same scene rendered from a cold wrapper state (no statistics word, no pair
    count estimate, nothing on offer), rendered again, and rendered again after an unrelated frame of the same shape has
    been through the wrapper, gives the same bits every time -- image, radii and every gradient -- under either setting."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 60000, 800, 800  # 2500 tiles; a trained-avatar shaped frame: its long lists would flip "auto" after a frame
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=6, layout="body")
    other, _ = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=7)
    bg = (0.3, 0.2, 0.1)
    if pin != "default":
        monkeypatch.setattr(dgr, "_LONG_LISTS", pin)
    assert dgr._LONG_LISTS == ("0" if pin == "default" else pin)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9)).to(dev)

    def cold():
        dgr._frame_stats.clear()
        dgr._last_count.clear()
        dgr.release_shared_geometry()

    def frame(c):
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(c, cam, "sh", "scale_rot", dev).items()}
        means3D = c.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        opac = c.opacity.to(dev).requires_grad_(True)
        color, radii = GaussianRasterizer(_settings(cam, c, bg, dev))(means3D=means3D, means2D=means2D, opacities=opac, **kw)
        mode = color.grad_fn.long_lists
        (color * gimg).sum().backward()
        out = [color.detach(), radii, means3D.grad, means2D.grad, opac.grad] + [v.grad for v in kw.values()]
        return mode, out

    cold()
    m0, a = frame(cloud)
    m1, b = frame(cloud)
    frame(other)
    m2, c = frame(cloud)
    cold()
    m3, d = frame(cloud)
    want = 1 if pin == "1" else 0
    assert (m0, m1, m2, m3) == (want,) * 4
    for x, y, z, w in zip(a, b, c, d):
        assert torch.equal(x, y) and torch.equal(x, z) and torch.equal(x, w)


@pytest.mark.parametrize("W,H", [(1500, 90), (90, 1500), (1100, 1090), (2070, 40)])
def test_tile_lists_across_tile_block_boundaries_with_screen_filling_gaussians(oracle, tile_rect, W, H):
    """The tile lists are built per block of 64 x 4 tiles and per segment of the depth ranking (binning.hip): image shapes
    of many blocks in one direction and one (partial) block in the other, Gaussians whose rectangle is the whole grid
    next to sub-tile ones -- sorted list, ranges and image against the oracle, bit for bit where integer."""
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    n = 900
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=77, scale_mul=1.0)
    cloud.scales[::9] *= 40.0     # every ninth Gaussian covers (nearly) every tile
    cloud.scales[1::9] *= 0.05    # ... and its neighbour not even one pixel
    cloud.opacity[::9] *= 0.05    # (so that the lists behind the big ones still contribute)
    bg = (0.2, 0.1, 0.4)
    sc = helpers.oracle_scene(cloud, cam, bg=bg, tile_rect=tile_rect)
    fw = oracle.forward(sc)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    ob = fw["binning"]
    assert np.array_equal(st["radii"], fw["geom"]["radii"])
    assert np.array_equal(st["geom"]["tiles_touched"], fw["geom"]["tiles_touched"])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert int(fw["geom"]["tiles_touched"].max()) >= 0.9 * gx * gy  # rectangles that are (nearly) the whole grid are among them
    assert st["D"] == ob["D"]
    assert np.array_equal(st["binning"]["point_list"], ob["point_list"])
    assert np.array_equal(st["binning"]["tile_ids"], (ob["keys"] >> np.uint64(32)).astype(np.uint32))
    nz = ob["ranges"][:, 1] > ob["ranges"][:, 0]
    assert np.array_equal(st["image"]["ranges"][nz], ob["ranges"][nz])
    _bulk_close(st["color"], fw["color"], frac=1e-4, name="color %dx%d" % (W, H))


@pytest.mark.parametrize("case", ["one-depth", "half-one-depth", "far-outliers", "two-depths", "mostly-culled", "levels-240",
                                  "levels-120", "levels-40", "levels-8", "clusters-120", "clusters-40"])
def test_depth_ranking_with_uneven_depth_distributions(oracle, case):
    """The depth ranking is a bucket sort over the frame's key range (depth_sort.hip): ~64 keys per bucket when the
    depths are evenly spread, ranked by counting (one, two or four keys per lane up to 256 keys).  Uneven spreads must give
    the same bits through its other paths: a wave's sorting network in LDS (up to 1024 keys), a bucket beyond that (second
    launch, 128 KB of LDS), beyond 16384 (global-memory network), equal keys (ties in ascending index order), a key range
    stretched by far outliers, most Gaussians culled (the bucket of those that touch no tile).  levels-K: K distinct depths
    (24000 / K equal keys per bucket); clusters-K: K tight clusters of distinct keys.  Checked: the ranking itself ((depth
    bits, index) ascending over the Gaussians that touch a tile) and the tile lists against the oracle, bit for bit."""
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    n, W, H = 24000, 256, 192
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=0, seed=61, scale_mul=0.6)
    g = torch.Generator().manual_seed(7)
    if case == "one-depth":
        cloud.xyz[:, 2] = 0.25           # every view-space depth is exactly 3.25: one bucket of 24000 equal keys
    elif case == "half-one-depth":
        cloud.xyz[::2, 2] = -0.5         # 12000 equal keys (a bucket beyond the small LDS size) among spread ones
    elif case == "far-outliers":
        cloud.xyz[:40, 2] = torch.empty(40).uniform_(1e4, 1e6, generator=g)  # the key range spans 18 binades
        cloud.xyz[40:, 2] = cloud.xyz[40:, 2] * 0.05                         # ... and the rest sits in 2 % of it
    elif case == "mostly-culled":
        cloud.xyz[torch.arange(n) % 10 < 7, 2] = -5.0  # 70 % behind the camera: the no-tile bucket, shared by many waves
    elif case.startswith("levels-") or case.startswith("clusters-"):
        k = int(case.split("-")[1])
        level = torch.randint(0, k, (n,), generator=g).float()
        cloud.xyz[:, 2] = -0.8 + 1.6 * level / k
        if case.startswith("clusters-"):
            cloud.xyz[:, 2] += torch.empty(n).uniform_(0.0, 1.6 / k * 0.05, generator=g)
    else:
        cloud.xyz[:, 2] = torch.where(torch.arange(n) % 3 == 0, torch.tensor(0.5), torch.tensor(-0.25))
    bg = (0.0, 0.0, 0.0)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    fw = oracle.forward(helpers.oracle_scene(cloud, cam, bg=bg))
    assert np.array_equal(st["radii"], fw["radii"]) and st["D"] == fw["binning"]["D"] and st["D"] > 0
    tt = st["geom"]["tiles_touched"]
    order = st["geom"]["sorted_idx"].astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(n))  # a permutation
    nvis = int((tt > 0).sum())
    head = order[:nvis]
    assert (tt[head] > 0).all() and (tt[order[nvis:]] == 0).all()  # Gaussians that touch no tile come last
    dbits = st["geom"]["depths"].view(np.uint32)[head].astype(np.int64)
    assert (np.diff(dbits) >= 0).all()
    tie = np.diff(dbits) == 0
    assert (np.diff(head)[tie] > 0).all()
    assert np.array_equal(st["binning"]["point_list"], fw["binning"]["point_list"])
    nz = fw["binning"]["ranges"][:, 1] > fw["binning"]["ranges"][:, 0]
    assert np.array_equal(st["image"]["ranges"][nz], fw["binning"]["ranges"][nz])
    _bulk_close(st["color"], fw["color"], frac=1e-4, name="color " + case)


def test_render_harness_train_step_config_shapes(oracle):
    """The render()-shaped harness on the reference's default input combination
    (colors_precomp + cov3D_precomp, two rasterizer calls per step) against the oracle."""
    from gsplat_mi355.render import DensifyStats, Pipe, train_step
    dev = torch.device("cuda:0")
    n, W, H = 3000, 128, 128
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=8, scale_mul=1.2)
    pc = cloud.to(dev)
    for f in ("xyz", "scales", "rotations", "opacity", "shs"):
        getattr(pc, f).requires_grad_(True)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    mask = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(2)) > 0.5).float().to(dev)
    stats = DensifyStats(n, dev)
    bg = torch.zeros(3, device=dev)
    loss, pkg = train_step(cam.to(dev), pc, Pipe(compute_cov3D_python=True), bg, gt, gt_mask=mask, lambda_mask=0.1,
                           stats=stats)
    # oracle: same two renders
    cam.to("cpu")
    sc = helpers.oracle_scene(cloud, cam, cov_mode="cov")
    fw = oracle.forward(sc)
    ones = torch.ones(n, 3)
    sc1 = helpers.oracle_scene(cloud, cam, color_mode="precomp", colors=ones, cov_mode="cov")
    fw1 = oracle.forward(sc1)
    _bulk_close(pkg.render.detach().cpu().numpy(), fw["color"], name="render")
    _bulk_close(pkg.opacity_render.detach().cpu().numpy()[0], fw1["color"][0], name="opacity_render")
    g0 = (np.sign(fw["color"] - gt.cpu().numpy()) / gt.numel()).astype(np.float32)
    g1 = np.zeros((3, H, W), np.float32)
    g1[0] = 0.1 * np.sign(fw1["color"][0] - mask.cpu().numpy()[0]) / mask.numel()
    b0, b1 = oracle.backward(sc, fw, g0), oracle.backward(sc1, fw1, g1)
    want_m2 = b0["means2D"] + b1["means2D"]
    _bulk_close(pkg.viewspace_points.grad.cpu().numpy(), want_m2, tol=5e-5, frac=1e-3, name="viewspace grad")
    assert np.array_equal(pkg.radii.cpu().numpy(), fw["radii"])
    vf = fw["radii"] > 0
    assert np.array_equal(pkg.visibility_filter.cpu().numpy(), vf)
    assert np.array_equal(stats.denom.cpu().numpy()[:, 0], vf.astype(np.float32))
    assert pc.xyz.grad is not None and torch.isfinite(pc.xyz.grad).all()


def test_full_size_properties_config2_50k_512(oracle):
    """BASELINE config 2 shape (50k Gaussians, 512x512, SH3) through size-independent properties."""
    from gsplat_mi355 import debug
    from simple_knn._C import distCUDA2
    dev = torch.device("cuda:0")
    n, W, H = 50000, 512, 512
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=0, dist2_fn=lambda p: distCUDA2(p.to(dev)).cpu())
    bg = (0.0, 0.0, 0.0)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    D = st["D"]
    assert D == int(st["geom"]["tiles_touched"].astype(np.int64).sum()) == len(st["binning"]["point_list"])
    tiles = st["binning"]["tile_ids"].astype(np.int64)
    pl = st["binning"]["point_list"]
    assert (np.diff(tiles) >= 0).all()
    dbits = st["geom"]["depths"].view(np.uint32)[pl].astype(np.int64)
    same = tiles[1:] == tiles[:-1]
    assert (dbits[1:][same] >= dbits[:-1][same]).all()
    tie = same & (dbits[1:] == dbits[:-1])
    assert (pl[1:][tie] > pl[:-1][tie]).all()
    r = st["image"]["ranges"]
    nz = r[:, 1] > r[:, 0]
    assert int((r[nz, 1] - r[nz, 0]).sum()) == D
    counts = np.bincount(tiles, minlength=r.shape[0])
    assert np.array_equal(counts[nz], (r[nz, 1] - r[nz, 0]))
    assert np.isfinite(st["color"]).all() and (st["color"] >= 0).all()
    assert (st["image"]["final_T"] <= 1).all() and (st["image"]["final_T"] > 0).all()
    # the same scene against the oracle (a few seconds of CPU)
    sc = helpers.oracle_scene(cloud, cam, bg=bg)
    fw = oracle.forward(sc)
    assert np.array_equal(st["radii"], fw["radii"]) and np.array_equal(pl, fw["binning"]["point_list"])
    _bulk_close(st["color"], fw["color"], name="color 50k/512")


def _sorted_list_properties(st):
    """Size-independent properties of the binning state (no oracle): sum of tiles_touched = D = list length; tile ids
    non-decreasing; inside a tile depths non-decreasing and ties in ascending Gaussian index (the stable order of the
    reference's 64-bit key sort); the ranges partition [0, D) and equal the tile histogram."""
    D = st["D"]
    assert D == int(st["geom"]["tiles_touched"].astype(np.int64).sum()) == len(st["binning"]["point_list"])
    tiles = st["binning"]["tile_ids"].astype(np.int64)
    pl = st["binning"]["point_list"]
    assert (np.diff(tiles) >= 0).all()
    dbits = st["geom"]["depths"].view(np.uint32)[pl].astype(np.int64)
    same = tiles[1:] == tiles[:-1]
    assert (dbits[1:][same] >= dbits[:-1][same]).all()
    tie = same & (dbits[1:] == dbits[:-1])
    assert (pl[1:][tie] > pl[:-1][tie]).all()
    r = st["image"]["ranges"]
    nz = r[:, 1] > r[:, 0]
    assert int((r[nz, 1].astype(np.int64) - r[nz, 0]).sum()) == D
    counts = np.bincount(tiles, minlength=r.shape[0])
    assert np.array_equal(counts[nz], (r[nz, 1] - r[nz, 0]))
    order = np.argsort(r[nz, 0], kind="stable")
    rs = r[nz][order]
    assert rs[0, 0] == 0 and rs[-1, 1] == D and np.array_equal(rs[1:, 0], rs[:-1, 1])
    assert np.isfinite(st["color"]).all() and (st["color"] >= 0).all()
    assert (st["image"]["final_T"] <= 1).all() and (st["image"]["final_T"] > 0).all()


def _full_size_case(oracle, case, n, W, H, heavy_tail, tile_rect, frame=0, min_pairs=0, extra_properties=False,
                    layout="box", sh_degree=3, grad_frac=None):
    """One BASELINE configuration at FULL size through the HIP path against the oracle in the same binning mode:
    integers (radii, tiles_touched, num_rendered, the sorted (tile, depth) list, the tile ranges) bit-exact; image,
    final_T and all six gradient tensors inside the float bar with bounded outliers; errors recorded in the parity
    report.  `tile_rect` = 0 is the reference's own binning."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsplat_mi355 import debug
    from simple_knn._C import distCUDA2
    dev = torch.device("cuda:0")
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=sh_degree, seed=0, frame=frame, heavy_tail=heavy_tail,
                                          layout=layout, dist2_fn=lambda p: distCUDA2(p.to(dev)).cpu())
    bg = (0.0, 0.0, 0.0)
    sc = helpers.oracle_scene(cloud, cam, bg=bg, tile_rect=tile_rect)
    fw = oracle.forward(sc)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    assert st["D"] == fw["binning"]["D"] and st["D"] > min_pairs
    assert np.array_equal(st["radii"], fw["radii"])
    assert np.array_equal(st["geom"]["tiles_touched"], fw["geom"]["tiles_touched"])
    assert np.array_equal(st["binning"]["point_list"], fw["binning"]["point_list"])
    assert np.array_equal(st["binning"]["tile_ids"], (fw["binning"]["keys"] >> np.uint64(32)).astype(np.uint32))
    nzr = fw["binning"]["ranges"][:, 1] > fw["binning"]["ranges"][:, 0]
    assert np.array_equal(st["image"]["ranges"][nzr], fw["binning"]["ranges"][nzr])
    assert (st["image"]["ranges"][~nzr, 1] == st["image"]["ranges"][~nzr, 0]).all()
    _sorted_list_properties(st)
    tag = "%s tile_rect=%d frame=%d" % (case, tile_rect, frame)
    flips = int((st["image"]["n_contrib"] != fw["image"]["n_contrib"]).sum())
    _report(tag, "color", st["color"], fw["color"],
            extra={"num_rendered": int(st["D"]), "pixels_with_a_different_last_contributor": flips,
                   "max_abs_err": float(np.abs(st["color"] - fw["color"]).max())})
    _report(tag, "final_T", st["image"]["final_T"], fw["image"]["final_T"])
    # (the plain oracle: bulk bound + cap, as until round 3 -- kept as a second, independent statement)
    _bulk_close(st["color"], fw["color"], name="color " + tag)
    assert np.abs(st["color"] - fw["color"]).max() < 1e-2
    _bulk_close(st["image"]["final_T"], fw["image"]["final_T"], name="final_T " + tag)
    assert flips < 1e-4 * W * H
    # every difference attributed to decisions at a threshold; from here on the oracle conditioned on them
    fwc, ov = _attribute(oracle, sc, fw, st["color"], st["image"]["final_T"], st["image"]["n_contrib"], tag)

    g1 = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5))
    want = oracle.backward(sc, fwc, g1.numpy(), ov)

    def grads(gimg):
        leaves = dict(means3D=cloud.xyz.to(dev).requires_grad_(True),
                      means2D=torch.zeros(n, 3, device=dev, requires_grad=True),
                      opacities=cloud.opacity.to(dev).requires_grad_(True), shs=cloud.shs.to(dev).requires_grad_(True),
                      scales=cloud.scales.to(dev).requires_grad_(True),
                      rotations=cloud.rotations.to(dev).requires_grad_(True))
        color, _ = GaussianRasterizer(_settings(cam, cloud, bg, dev))(**leaves)
        (color * gimg.to(dev)).sum().backward()
        return {k: v.grad.cpu().numpy().astype(np.float64) for k, v in leaves.items()}

    got = grads(g1)
    names = dict(means3D="means3D", means2D="means2D", opacities="opacities", shs="sh", scales="scales",
                 rotations="rotations")
    for k, v in got.items():
        w = want[names[k]].reshape(v.shape)
        _report(tag, "dL_d" + k, v, w)
        _bulk_close(v, w, tol=1e-5, frac=COND_GRAD_FRAC if grad_frac is None else grad_frac, name=k + " " + tag, cap=COND_GRAD_CAP)
    culled = fw["radii"] == 0
    for k, v in got.items():
        assert (v[culled] == 0).all(), k
    if extra_properties:
        g2 = torch.randn(3, H, W, generator=torch.Generator().manual_seed(6))
        again = grads(g1)
        for k in got:
            assert np.array_equal(got[k], again[k]), k  # no atomics anywhere: the same bits
        both, second = grads(g1 + g2), grads(g2)
        for k in got:
            _bulk_close(both[k], got[k] + second[k], tol=2e-5, frac=2e-4, name="linearity " + k)


def test_full_size_config1_shape_10k_256_sh0_on_the_hip_path(oracle, tile_rect):
    """BASELINE config 1 (dummy_dataset: 10k random Gaussians, 256 x 256, SH degree 0) is by definition the no-GPU
    plumbing configuration and runs on the CPU oracle (tests/test_oracle.py); this is the same shape through the HIP
    path, forward + backward against the oracle in both binning modes, so that no BASELINE shape is left unexercised."""
    # (10 000 Gaussians: a tensor has 10-40 k elements.  Until round 3 the handful of threshold flips of a frame needed 1e-3
    # of them exempt -- dL_dopacities passed at exactly 10 of 10 000; with the flips attributed and conditioned on, the
    # same bar as every other shape)
    _full_size_case(oracle, "config1 10k/256x256 SH0", 10000, 256, 256, 0.0, tile_rect, min_pairs=10000, sh_degree=0)


def test_full_size_config3_200k_1024_forward_backward(oracle, tile_rect):
    """The bench workload itself (BASELINE config 3: 200k Gaussians, 1024x1024, SH3, forward + backward) against the
    oracle in BOTH binning modes -- tile_rect = 0 is the reference's own: its 5.7 M-pair list, its ranges and every
    gradient are checked at full size -- plus two properties that need no oracle: the backward is linear in dL/dimage,
    and a second run gives the same bits."""
    _full_size_case(oracle, "config3 200k/1024x1024", 200000, 1024, 1024, 0.0, tile_rect,
                    min_pairs=5000000 if tile_rect == 0 else 3000000, extra_properties=True)


@pytest.mark.parametrize("frame", [0, 150, 299])
def test_full_size_config4_200k_512_pose_sequence_frames(oracle, frame):
    """BASELINE config 4 (200k Gaussians, 512x512 -- the ZJU-MoCap `img_hw` --, a 300-frame sequence sharded over the
    GPUs): three frames of the orbit sequence bench.py renders (first, middle, last), forward + backward, against the
    oracle; frame 0 also in the reference's own binning mode."""
    import diff_gaussian_rasterization as dgr
    saved = dgr._TILE_RECT
    try:
        for mode in ((1, 0) if frame == 0 else (1,)):
            dgr._TILE_RECT = mode
            _full_size_case(oracle, "config4 200k/512x512", 200000, 512, 512, 0.0, mode, frame=frame, min_pairs=1000000)
    finally:
        dgr._TILE_RECT = saved


def test_full_size_config5_500k_2048_heavy_tail(oracle, tile_rect):
    """BASELINE config 5 (500k Gaussians, 2048x2048, 5 % of them with 4x scales: tens of millions of pairs, per-tile
    lists of thousands of entries -- the tile-overflow / sort-stress case) at FULL size in both binning modes:
    point list, tile ids and ranges bit-exact, image and gradients inside the float bar, list properties."""
    _full_size_case(oracle, "config5 500k/2048x2048 heavy tail", 500000, 2048, 2048, 0.05, tile_rect,
                    min_pairs=35000000 if tile_rect == 0 else 20000000)


def test_full_size_trained_avatar_shaped_frame_200k_512(oracle, tile_rect):
    """Not a BASELINE configuration: the shape of a TRAINED avatar at full size (bench.py --workload avatar: 200k
    Gaussians on a thin shell covering a sixth of 512 x 512, mostly opaque -- 150 tiles with lists of 2000-7000 entries),
    the frame the four-wave forward and the chunked backward exist for, through the same checks as the configurations:
    lists and ranges bit-exact, image and gradients inside the float bar, bitwise repeatable, linear."""
    _full_size_case(oracle, "avatar 200k/512x512", 200000, 512, 512, 0.0, tile_rect, min_pairs=300000,
                    extra_properties=True, layout="body")


class _CallCounter(object):
    """Stands in for the loaded C library and counts the calls per entry point (which backward did the step take?)."""

    def __init__(self, lib):
        import collections
        self._lib, self.calls = lib, collections.Counter()

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def counted(*a):
            self.calls[name] += 1
            return fn(*a)
        return counted


class _ConvertedCloud(object):
    """What the reference's render() holds after `scene.convert_gaussians` with its default configuration
    (gaussian_renderer/__init__.py:73,107-119; configs/config.yaml:89-91): positions, opacities, precomputed 3-D
    covariances and precomputed colours -- here all four as autograd LEAVES, so that every gradient the rasterizer
    returns on this input combination can be read off and compared."""

    def __init__(self, cloud, cam, dev):
        self.xyz = cloud.xyz.to(dev).requires_grad_(True)
        self.opacity = cloud.opacity.to(dev).requires_grad_(True)
        self.cov6 = helpers.covariance6_cpu(cloud).to(dev).requires_grad_(True)
        self.colors = helpers.precomp_colors(cloud, cam).to(dev).requires_grad_(True)
        self.sh_degree, self.shs, self.scales, self.rotations = cloud.sh_degree, None, None, None

    def covariance6(self, scaling_modifier=1.0):
        return self.cov6


_TWO_CALL_SCENES = {  # name -> (n, W, H, layout, the HIP kernels the frame runs)
    "config4 200k/512x512": (200000, 512, 512, "box"),     # <= 2048 tiles, no long list: four-wave kernels' one-wave path
    "avatar 200k/512x512": (200000, 512, 512, "body"),     # lists of thousands: four waves per quadrant, backward in chunks
    "config3 200k/1024x1024": (200000, 1024, 1024, "box"),  # > 2048 tiles: one wave per quadrant, single walk
}
_two_call_oracle = {}


def _two_call_reference(oracle, scene, bg):
    """The oracle's two renders of the reference's step -- colour pass and colours = 1 pass on the same geometry,
    precomputed colours and covariances -- for one scene and background (kept for the other modes of the same case)."""
    from simple_knn._C import distCUDA2
    key = (scene, bg)
    if key not in _two_call_oracle:
        dev = torch.device("cuda:0")
        n, W, H, layout = _TWO_CALL_SCENES[scene]
        cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=0, layout=layout,
                                              dist2_fn=lambda p: distCUDA2(p.to(dev)).cpu())
        sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode="precomp", cov_mode="cov")
        sc1 = helpers.oracle_scene(cloud, cam, bg=bg, color_mode="precomp", colors=torch.ones(n, 3), cov_mode="cov")
        _two_call_oracle.clear()  # (one scene's state at a time: a forward state is hundreds of megabytes)
        _two_call_oracle[key] = (cloud, cam, sc, oracle.forward(sc), sc1, oracle.forward(sc1))
    return _two_call_oracle[key]


def _two_call_step(oracle, monkeypatch, scene, bg, mode, use):
    """The reference's real step at full size against the oracle: `render(cam, pc, Pipe(compute_cov3D_python=True), bg,
    return_opacity=True)` -- colors_precomp + cov3D_precomp, the colour image and the opacity image
    (gaussian_renderer/__init__.py:107-142) -- with every switch of the wrapper at its default, a loss on `use` = both
    images / the opacity image only, and ALL gradient tensors of that input combination compared with
    oracle.backward(colour pass) + oracle.backward(colours = 1 pass) at the full-size bar.

    mode "second-call": the reference's unmodified two rasterizer calls.  The second one is served from the first one's
    geometry (gs_forward_shared; colours = 1: the recolouring launch's other workgroups write 1 - T speculatively and the
    render launch behind it leaves the image alone) and both images are differentiated by ONE backward pass
    (gs_backward_with_second; colours all ones: render_bwd_kernel<3>, MODE 0's loop with the closed-form second term;
    arbitrary second colours: render_bwd_kernel<2>).  mode "with-opacity": one call with with_opacity=True
    (gs_opacity_image, gs_backward_with_opacity: render_bwd_kernel<1>)."""
    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import _lib, debug
    from gsplat_mi355.render import Pipe, render
    assert dgr._SHARE and dgr._FUSE_SECOND and dgr._SPECULATE and dgr._LONG_LISTS == "0"  # the defaults are under test
    dev = torch.device("cuda:0")
    n, W, H, layout = _TWO_CALL_SCENES[scene]
    cloud, cam, sc, fw, sc1, fw1 = _two_call_reference(oracle, scene, bg)
    gen = torch.Generator().manual_seed(11)
    g0 = torch.randn(3, H, W, generator=gen)
    g1 = torch.randn(1, H, W, generator=gen)
    counter = _CallCounter(_lib.load())
    monkeypatch.setattr(_lib, "_lib", counter)
    dgr.release_shared_geometry()
    hits0 = dgr._geom_cache.hits
    pc = _ConvertedCloud(cloud, cam, dev)
    pipe = Pipe(compute_cov3D_python=True, fuse_opacity=(mode == "with-opacity"))
    pkg = render(cam.to(dev), pc, pipe, torch.tensor(bg, dtype=torch.float32, device=dev), colors_precomp=pc.colors,
                 return_opacity=True)
    cam.to("cpu")
    loss = (pkg.opacity_render * g1.to(dev)).sum()
    if use == "both":
        loss = loss + (pkg.render * g0.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    calls = counter.calls
    if mode == "second-call":
        assert dgr._geom_cache.hits - hits0 == 1 and calls["gs_forward"] == 1 and calls["gs_forward_shared"] == 1
        assert calls["gs_backward_with_second"] == 1 and calls["gs_backward"] == 0 and calls["gs_backward_with_opacity"] == 0
    else:
        assert calls["gs_forward"] == 1 and calls["gs_forward_shared"] == 0 and calls["gs_opacity_image"] == 1
        assert calls["gs_backward_with_opacity"] == 1 and calls["gs_backward"] == 0 and calls["gs_backward_with_second"] == 0

    tag = "two-call %s bg=%s %s loss on %s" % (scene, "0" if not any(bg) else "%g,%g,%g" % bg, mode, use)
    assert np.array_equal(pkg.radii.cpu().numpy(), fw["radii"])
    color = pkg.render.detach().cpu().numpy()
    opa = pkg.opacity_render.detach().cpu().numpy()
    assert opa.shape == (1, H, W)
    _report(tag, "color", color, fw["color"], extra={"num_rendered": int(fw["binning"]["D"])})
    _report(tag, "opacity_image", opa[0], fw1["color"][0])
    _bulk_close(color, fw["color"], name="color " + tag)
    _bulk_close(opa[0], fw1["color"][0], name="opacity image " + tag)
    assert np.abs(color - fw["color"]).max() < 1e-2 and np.abs(opa[0] - fw1["color"][0]).max() < 1e-2
    # Attribution (module docstring): the device's per-pixel records of this frame (the same inputs through the two-phase
    # C calls: same kernels, same bits as the wrapper's speculative path -- test_speculative_capacity_*), every
    # difference explained by decisions at a threshold, then BOTH images and all gradients against the oracle
    # conditioned on them.  The two passes share geometry, hence decisions: one override table serves both.
    st = debug.forward_state(_settings(cam, cloud, bg, dev), pc.xyz.detach(), pc.opacity.detach(),
                             colors_precomp=pc.colors.detach(), cov3D_precomp=pc.cov6.detach())
    assert np.array_equal(st["color"], color)
    fwc, ov = _attribute(oracle, sc, fw, color, st["image"]["final_T"], st["image"]["n_contrib"], tag)
    im1 = oracle.render_forward(sc1, fw1["geom"], fw1["binning"], ov)
    fw1c = dict(fw1, image=im1, color=im1["color"])
    m1 = max(float(np.abs(im1["color"]).max()), 1e-30)
    assert np.abs(opa[0] - im1["color"][0]).max() <= TOL * m1, "opacity image vs the conditioned oracle: " + tag
    _report(tag, "opacity_image_vs_conditioned_oracle", opa[0], im1["color"][0])

    gop = np.zeros((3, H, W), np.float32)
    gop[0] = g1.numpy()[0]  # the reference keeps channel 0 of the second call's image (`[:1]`)
    b1 = oracle.backward(sc1, fw1c, gop, ov)
    b0 = oracle.backward(sc, fwc, g0.numpy(), ov) if use == "both" else None
    got = dict(means3D=pc.xyz.grad, means2D=pkg.viewspace_points.grad, opacities=pc.opacity.grad,
               colors_precomp=pc.colors.grad, cov3D_precomp=pc.cov6.grad)
    culled = fw["radii"] == 0
    for k, t in got.items():
        if k == "colors_precomp":  # (the second call's colours are constants: only the colour pass reaches them)
            w = b0[k] if b0 is not None else np.zeros((n, 3), np.float32)
            if t is None:
                assert b0 is None
                continue
        else:
            w = b1[k].astype(np.float64) + (b0[k] if b0 is not None else 0.0)
        v = t.cpu().numpy().astype(np.float64)
        w = np.asarray(w, np.float64).reshape(v.shape)
        _report(tag, "dL_d" + k, v, w)
        if np.abs(w).max() == 0:
            assert (v == 0).all(), k
            continue
        _bulk_close(v, w, tol=1e-5, frac=COND_GRAD_FRAC, name=k + " " + tag, cap=COND_GRAD_CAP)
        assert (v[culled] == 0).all(), k


@pytest.mark.parametrize("scene,bg,mode,use", [
    ("config4 200k/512x512", (0.0, 0.0, 0.0), "second-call", "both"),
    ("config4 200k/512x512", (0.0, 0.0, 0.0), "second-call", "opacity"),
    ("config4 200k/512x512", (0.0, 0.0, 0.0), "with-opacity", "both"),
    ("config4 200k/512x512", (0.3, 0.6, 0.1), "second-call", "both"),
    ("config4 200k/512x512", (0.3, 0.6, 0.1), "with-opacity", "both"),
    ("config4 200k/512x512", (0.3, 0.6, 0.1), "with-opacity", "opacity"),
    ("avatar 200k/512x512", (0.0, 0.0, 0.0), "second-call", "both"),
    ("avatar 200k/512x512", (0.0, 0.0, 0.0), "with-opacity", "both"),
    ("avatar 200k/512x512", (0.3, 0.6, 0.1), "second-call", "both"),
    ("avatar 200k/512x512", (0.3, 0.6, 0.1), "second-call", "opacity"),
    ("avatar 200k/512x512", (0.3, 0.6, 0.1), "with-opacity", "both"),
    ("config3 200k/1024x1024", (0.3, 0.6, 0.1), "second-call", "both"),
    ("config3 200k/1024x1024", (0.3, 0.6, 0.1), "with-opacity", "both"),
])
def test_full_size_reference_step_two_images_default_path(oracle, monkeypatch, scene, bg, mode, use):
    """The step the reference actually takes (gaussian_renderer/__init__.py:107-142 under
    configs/config.yaml:73,89-91): precomputed colours + covariances, colour image and opacity image, a loss on both --
    at config 4's size, on a trained-avatar shaped frame and at config 3's size, black and non-black background, through
    the wrapper's DEFAULT path (shared geometry, 1 - T image, one backward for both images) and through the fused
    `with_opacity=True` form; images and all five gradient tensors against the oracle's two passes."""
    _two_call_step(oracle, monkeypatch, scene, bg, mode, use)


@pytest.mark.parametrize("layout,n,W,H", [("box", 6000, 208, 160), ("body", 30000, 256, 256)])
def test_second_render_with_arbitrary_constant_colours_one_backward_for_both_images(oracle, monkeypatch, layout, n, W, H):
    """The one-pass backward of two images of the same geometry (gs_backward_with_second) in its GENERAL form: the second
    call's colours are arbitrary constants, not the reference's all-ones (for which the backward runs a cheaper kernel
    that needs no second colours at all: the full-size two-call tests).  Both images and every gradient against the oracle's
    two passes; on the body layout the lists are long enough for the chunked backward and its second set of checkpoints."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsplat_mi355 import _lib
    dev = torch.device("cuda:0")
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=21, layout=layout, scale_mul=1.0 if layout == "body" else 1.3)
    bg = (0.2, 0.1, 0.4)
    gen = torch.Generator().manual_seed(5)
    cols2 = torch.rand(n, 3, generator=gen)
    g0, g1 = torch.randn(3, H, W, generator=gen), torch.randn(3, H, W, generator=gen)
    counter = _CallCounter(_lib.load())
    monkeypatch.setattr(_lib, "_lib", counter)
    dgr.release_shared_geometry()
    xyz = cloud.xyz.to(dev).requires_grad_(True)
    m2d = torch.zeros(n, 3, device=dev, requires_grad=True)
    op = cloud.opacity.to(dev).requires_grad_(True)
    cov = helpers.covariance6_cpu(cloud).to(dev).requires_grad_(True)
    cols = helpers.precomp_colors(cloud, cam).to(dev).requires_grad_(True)
    rast = GaussianRasterizer(_settings(cam, cloud, bg, dev))
    img1, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=cols, cov3D_precomp=cov)
    img2, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=cols2.to(dev), cov3D_precomp=cov)
    ((img1 * g0.to(dev)).sum() + (img2 * g1.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    assert counter.calls["gs_forward_shared"] == 1 and counter.calls["gs_backward_with_second"] == 1 and counter.calls["gs_backward"] == 0
    sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode="precomp", cov_mode="cov")
    sc2 = helpers.oracle_scene(cloud, cam, bg=bg, color_mode="precomp", colors=cols2, cov_mode="cov")
    fw, fw2 = oracle.forward(sc), oracle.forward(sc2)
    # both images share geometry, hence decisions: attributed on the first, the same override table conditions both passes
    st, fwc, ov = _conditioned_oracle(oracle, sc, fw, _settings(cam, cloud, bg, dev), xyz, op,
                                      dict(colors_precomp=cols, cov3D_precomp=cov), "second render, arbitrary colours, " + layout)
    assert np.array_equal(img1.detach().cpu().numpy(), st["color"])
    im2 = oracle.render_forward(sc2, fw2["geom"], fw2["binning"], ov)
    fw2c = dict(fw2, image=im2, color=im2["color"])
    m2 = max(float(np.abs(im2["color"]).max()), 1e-30)
    assert np.abs(img2.detach().cpu().numpy() - im2["color"]).max() <= TOL * m2
    b0, b1 = oracle.backward(sc, fwc, g0.numpy(), ov), oracle.backward(sc2, fw2c, g1.numpy(), ov)
    for k, t in (("means3D", xyz), ("means2D", m2d), ("opacities", op), ("cov3D_precomp", cov)):
        w = b0[k].astype(np.float64) + b1[k]
        _bulk_close(t.grad.cpu().numpy(), w.reshape(t.shape), tol=SMALL_TOL, frac=0.0, name="two images: " + k)
    _bulk_close(cols.grad.cpu().numpy(), b0["colors_precomp"], tol=SMALL_TOL, frac=0.0, name="first image's colours")


def test_heavy_tail_stress_config5_shape(oracle):
    """BASELINE config 5 in miniature: 5 % of the Gaussians with 4x scales (long per-tile lists, many
    64-entry chunks per quadrant), non-zero background, precomputed covariances."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    n, W, H = 12000, 320, 208
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=77, heavy_tail=0.05, scale_mul=1.6)
    bg = (0.4, 0.1, 0.7)
    sc = helpers.oracle_scene(cloud, cam, bg=bg, cov_mode="cov")
    fw = oracle.forward(sc)
    st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), cov3D_precomp=helpers.covariance6_cpu(cloud).to(dev))
    assert st["D"] == fw["binning"]["D"] and st["D"] > 20 * n
    assert np.array_equal(st["binning"]["point_list"], fw["binning"]["point_list"])
    r = fw["binning"]["ranges"]
    assert (r[:, 1] - r[:, 0]).max() > 1000  # tiles with long lists
    fwc, ov = _attribute(oracle, sc, fw, st["color"], st["image"]["final_T"], st["image"]["n_contrib"], "heavy tail stress")
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9))
    want = oracle.backward(sc, fwc, gimg.numpy(), ov)
    means3D = cloud.xyz.to(dev).requires_grad_(True)
    means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
    opac = cloud.opacity.to(dev).requires_grad_(True)
    shs = cloud.shs.to(dev).requires_grad_(True)
    cov = helpers.covariance6_cpu(cloud).to(dev).requires_grad_(True)
    color, radii = GaussianRasterizer(_settings(cam, cloud, bg, dev))(means3D=means3D, means2D=means2D, opacities=opac,
                                                                     shs=shs, cov3D_precomp=cov)
    (color * gimg.to(dev)).sum().backward()
    for name, t in (("means3D", means3D), ("means2D", means2D), ("opacities", opac), ("sh", shs), ("cov3D_precomp", cov)):
        _bulk_close(t.grad.cpu().numpy(), want[name].reshape(t.shape), tol=SMALL_TOL, frac=0.0, name=name)


def test_bitwise_determinism_and_debug_mode():
    """No atomics anywhere in the path: two runs give bit-identical images and gradients; debug=True
    (synchronise + check after every kernel) gives the same bits."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 6000, 256, 192
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=13, scale_mul=1.4)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = []
    for dbg in (False, False, True):
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        shs = cloud.shs.to(dev).requires_grad_(True)
        scales = cloud.scales.to(dev).requires_grad_(True)
        rot = cloud.rotations.to(dev).requires_grad_(True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        rast = GaussianRasterizer(_settings(cam, cloud, (0.1, 0.2, 0.3), dev, debug=dbg))
        color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rot)
        (color * gimg).sum().backward()
        outs.append([color.detach().clone(), radii.clone(), means3D.grad, means2D.grad, shs.grad, scales.grad, rot.grad,
                     opac.grad])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_speculative_binning_capacity_overflow_and_slack_give_the_same_bits():
    """gs_forward enqueues phase 2 against a binning state sized from the PREVIOUS frame's pair count, before the host
    knows this frame's count (no GPU idle stretch for it).  Whatever the estimate -- none, far too small (overflow: the
    speculative phase renders an empty frame and phase 2 is run again on a state of the right size), exact, or far too
    large -- image, radii and every gradient must be the same bits."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 9000, 272, 200
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=19, scale_mul=1.5)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
    key = (dev.index, n, W, H)

    def run(estimate):
        dgr.release_shared_geometry()
        if estimate is None:
            dgr._last_count.pop(key, None)
        else:
            dgr._last_count[key] = estimate
        leaves = dict(means3D=cloud.xyz.to(dev).requires_grad_(True), means2D=torch.zeros(n, 3, device=dev, requires_grad=True),
                      opacities=cloud.opacity.to(dev).requires_grad_(True), shs=cloud.shs.to(dev).requires_grad_(True),
                      scales=cloud.scales.to(dev).requires_grad_(True), rotations=cloud.rotations.to(dev).requires_grad_(True))
        color, radii = GaussianRasterizer(_settings(cam, cloud, (0.2, 0.1, 0.3), dev))(**leaves)
        (color * gimg).sum().backward()
        torch.cuda.synchronize()
        return [color.detach().clone(), radii.clone()] + [v.grad.clone() for v in leaves.values()], dgr._last_count[key]

    ref, D = run(None)
    assert D > 50000
    for estimate in (1000, D // 2, D - 1, D, int(D / 1.125) + 1, 3 * D, 40 * D):
        got, D2 = run(estimate)
        assert D2 == D
        for a, b in zip(got, ref):
            assert torch.equal(a, b), estimate


@pytest.mark.parametrize("fused_backward", [False, True])
def test_shared_geometry_second_render_is_bitwise_identical(oracle, fused_backward, monkeypatch):
    """the opacity pass of render() (same geometry, colours = 1) reuses the first
    call's preprocess / sort / binning.  Its image must equal the stand-alone call's bit for bit, and so must the
    gradients when each call runs its own backward (fused_backward = False); in-place changes of the geometry must
    invalidate the reuse.  By default (fused_backward = True) the two images are differentiated in ONE pass
    (gs_backward_with_second: the second call hands its gradient image to the first call's backward): the gradients then
    equal the sum of the two separate passes to fp32 rounding."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsplat_mi355 import _lib
    monkeypatch.setattr(dgr, "_FUSE_SECOND", fused_backward)
    # (the default also renders an all-ones second image as 1 - T from the first render's transmittance instead of
    # compositing it: equal to fp32 rounding, so it is switched off with the fused backward for the bit-for-bit run)
    _lib.tuning("ones_fast", 1 if fused_backward else 0)
    dev = torch.device("cuda:0")
    n, W, H = 5000, 192, 160
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=17, scale_mul=1.3)
    settings = _settings(cam, cloud, (0.0, 0.0, 0.0), dev)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4)).to(dev)

    def run(share):
        dgr._SHARE = share
        dgr.release_shared_geometry()
        hits0 = dgr._geom_cache.hits
        xyz = cloud.xyz.to(dev).requires_grad_(True)
        m2d = torch.zeros(n, 3, device=dev, requires_grad=True)
        op = cloud.opacity.to(dev).requires_grad_(True)
        cov = helpers.covariance6_cpu(cloud).to(dev).requires_grad_(True)
        cols = helpers.precomp_colors(cloud, cam).to(dev).requires_grad_(True)
        ones = torch.ones(n, 3, device=dev)
        rast = GaussianRasterizer(settings)
        img1, r1 = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=cols, cov3D_precomp=cov)
        img2, r2 = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        hits_before = dgr._geom_cache.hits - hits0 == 1
        img3, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        assert dgr._geom_cache.hits - hits0 == (1 if share else 0)  # single use: a third call renders in full
        if fused_backward and share:
            assert float((img3 - img2).detach().abs().max()) <= 2e-6  # (1 - T against the composited sum of alpha T)
        else:
            assert torch.equal(img3, img2)
        ((img1 * gimg).sum() + (img2[:1] * gimg[:1]).sum()).backward()
        return [img1.detach(), img2.detach(), r1, r2, xyz.grad, m2d.grad, op.grad, cov.grad, cols.grad], hits_before

    try:
        shared, hit = run(True)
        alone, _ = run(False)
        assert hit
        for i, (a, b) in enumerate(zip(shared, alone)):
            if fused_backward and i == 1:  # the all-ones image: 1 - T
                assert float((a - b).abs().max()) <= 2e-6
            elif fused_backward and i >= 4:  # gradients: one pass over both images against the sum of two passes
                scale = float(b.abs().max())
                assert float((a - b).abs().max()) <= 3e-6 * scale, (i, float((a - b).abs().max()) / scale)
                assert torch.equal(a == 0, b == 0)
            else:
                assert torch.equal(a, b)
        # opacity render = 1 - T for a black background (gaussian_renderer/__init__.py:131-142)
        sc1 = helpers.oracle_scene(cloud, cam, color_mode="precomp", colors=torch.ones(n, 3), cov_mode="cov")
        _bulk_close(shared[1].cpu().numpy(), oracle.forward(sc1)["color"], name="opacity pass")
        # An in-place update of the positions between two calls must not be served from the first call's geometry -- under
        # GRAD mode, where sharing is live (under no_grad nothing is ever offered): the offer is keyed by the tensors'
        # autograd version counters.  `alias` is a detached view of the same storage: writing through it bumps the
        # counter the leaf shares with it.
        dgr._SHARE = True
        dgr.release_shared_geometry()
        xyz = cloud.xyz.to(dev).requires_grad_(True)
        m2d = torch.zeros(n, 3, device=dev, requires_grad=True)
        op, cov = cloud.opacity.to(dev), helpers.covariance6_cpu(cloud).to(dev)
        ones = torch.ones(n, 3, device=dev)
        rast = GaussianRasterizer(settings)
        h0 = dgr._geom_cache.hits
        a1, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        alias = xyz.detach()
        alias.add_(0.05)
        assert alias.data_ptr() == xyz.data_ptr()
        a2, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        assert dgr._geom_cache.hits == h0 and not torch.equal(a1, a2)  # a miss: rendered in full from the new positions
        a3, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        assert dgr._geom_cache.hits == h0 + 1  # (nothing changed since a2: served from its geometry)
        # `.data` writes are invisible to the version counter -- to the cache as to autograd's own saved-tensor checks:
        # unsupported between two calls without a backward in between; release_shared_geometry() is the way out.
        dgr.release_shared_geometry()
        b1, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        xyz.data.add_(0.05)
        dgr.release_shared_geometry()
        b2, _ = rast(means3D=xyz, means2D=m2d, opacities=op, colors_precomp=ones, cov3D_precomp=cov)
        assert dgr._geom_cache.hits == h0 + 1 and not torch.equal(b1, b2)
    finally:
        dgr._SHARE = True
        dgr.release_shared_geometry()
        _lib.tuning("ones_fast", 1)


def test_parameter_updates_through_raw_pointers_never_meet_stale_shared_geometry(monkeypatch):
    """render -> backward -> optimiser step -> render with the SAME camera and the SAME parameter objects: the second
    render must see the updated parameters.  FusedAdam writes through raw pointers (no torch in-place op), and so does
    an external writer simulated here with a DLPack alias, whose version counter is not the parameters'.  (Each call
    runs its own backward here, so that the shared and the stand-alone runs take bit-identical optimiser steps.)"""
    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import _lib as _l
    monkeypatch.setattr(dgr, "_FUSE_SECOND", False)
    _l.tuning("ones_fast", 0)  # (and composites the opacity image in both, instead of 1 - T in the shared run)
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.optim import FusedAdam
    from gsplat_mi355.render import Pipe, l1_loss, render
    from gsplat_mi355.scenes import GaussianCloud
    dev = torch.device("cuda:0")
    n, W, H = 3000, 128, 96
    cloud, _ = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=5, scale_mul=1.3)
    cam = orbit_camera(0, W, H, device=dev)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)

    def fresh():
        return GaussianCloud(*[getattr(cloud, f).to(dev).clone().requires_grad_(True) for f in GaussianCloud.FIELDS],
                             cloud.sh_degree)

    def images(share, writer):
        dgr._SHARE = share
        dgr.release_shared_geometry()
        c = fresh()
        opt = FusedAdam([{"params": [getattr(c, f)], "lr": 1e-2, "name": f} for f in ("xyz", "shs")], lr=0.0, eps=1e-15)
        out = []
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            pkg = render(cam, c, Pipe(), bg, return_opacity=True)  # two rasterizer calls: the second one shares
            out.append(pkg.render.detach().clone())
            (l1_loss(pkg.render, gt) + 0.1 * pkg.opacity_render.mean()).backward()
            if writer == "adam":
                opt.step()
            else:
                alias = torch.from_dlpack(torch.utils.dlpack.to_dlpack(c.xyz.detach()))
                assert alias.data_ptr() == c.xyz.data_ptr()
                alias.add_(0.01 * (it + 1))
        return out

    try:
        for writer in ("adam", "alias"):
            h0 = dgr._geom_cache.hits
            shared = images(True, writer)
            assert dgr._geom_cache.hits - h0 == 3  # the opacity pass of every step was served from the colour pass
            alone = images(False, writer)
            for a, b in zip(shared, alone):
                assert torch.equal(a, b), writer
            assert not torch.equal(shared[0], shared[1]) and not torch.equal(shared[1], shared[2])
    finally:
        dgr._SHARE = True
        dgr.release_shared_geometry()
        _l.tuning("ones_fast", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 64), (3, 333, 517), (1, 7), (3, 1024, 1024)])
def test_fused_l1_loss_matches_oracle_and_torch(oracle, shape):
    """N2: gsplat_mi355.render.l1_loss == torch.abs(a - b).mean() (utils/loss_utils.py:21-22); the gradient is
    bit-exact (sign(x - y) / n), the value within fp32 summation-order tolerance (1e-6 relative)."""
    from gsplat_mi355.render import l1_loss
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.rand(shape, generator=g)
    b = torch.rand(shape, generator=g)
    a.view(-1)[:3] = b.view(-1)[:3]  # sign(0) = 0
    want, want_grad = oracle.l1_loss(a.numpy(), b.numpy())
    x = a.cuda().requires_grad_(True)
    loss = l1_loss(x, b.cuda())
    (loss * 2.0).backward()  # a non-unit upstream gradient must be honoured
    assert float(loss.detach()) == pytest.approx(want, rel=1e-6)
    assert np.array_equal(x.grad.cpu().numpy(), 2.0 * want_grad)
    ref = torch.abs(a.cuda() - b.cuda()).mean()
    assert float(loss.detach()) == pytest.approx(float(ref), rel=1e-6)
    # bitwise reproducible, and gradient w.r.t. the target is the negation
    y = b.cuda().requires_grad_(True)
    loss2 = l1_loss(a.cuda(), y)
    loss2.backward()
    assert float(loss2) == float(loss)
    assert np.array_equal(y.grad.cpu().numpy(), -want_grad)
    with pytest.raises(RuntimeError):
        l1_loss(a, b)  # CPU tensors: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["small-image kernels", "large-image kernels", "precomputed inputs + opacity image",
                                   "second call + mask loss"])
def test_l1_fused_into_the_rasterizer_gives_the_bits_of_the_separate_l1_loss(oracle, scene):
    """Row N2, "producing dL/dimage directly in the layout K7 reads": render(..., l1_target=gt) returns mean |image - gt|
    (train.py:121 / utils/loss_utils.py:21-22) as a by-product of the render launch, and the backward forms that loss's
    gradient per pixel in its own prologue -- no gradient image.  Against the separate path (l1_loss on the rendered image
    -> a gradient image -> the rasterizer's backward): the loss value to summation order, EVERY gradient bit for bit -- also
    with a non-unit weight on the loss, with another consumer of the image beside it, with the opacity image in the
    same pass and with the reference's second call; the value also against the oracle's L1 of the same image."""
    from gsplat_mi355.render import Pipe, l1_loss, render
    dev = torch.device("cuda:0")
    n, W, H = (9000, 208, 160) if scene != "large-image kernels" else (30000, 1040, 800)  # (800 x 1040: 3250 tiles > 2048)
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=17, scale_mul=1.3 if n < 10000 else 2.0)
    gen = torch.Generator().manual_seed(3)
    gt = torch.rand(3, H, W, generator=gen).to(dev)
    gimg = (torch.randn(3, H, W, generator=gen) * 1e-6).to(dev)  # a second consumer of the image, as SSIM would be
    gmask = (torch.rand(1, H, W, generator=gen) > 0.5).float().to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    precomp = scene in ("precomputed inputs + opacity image", "second call + mask loss")
    pipe = Pipe(compute_cov3D_python=precomp, fuse_opacity=(scene == "precomputed inputs + opacity image"))
    want_opacity = scene in ("precomputed inputs + opacity image", "second call + mask loss")

    def run(fused):
        pc = _ConvertedCloud(cloud, cam, dev) if precomp else cloud.to(dev)
        leaves = ([pc.xyz, pc.opacity, pc.cov6, pc.colors] if precomp else
                  [pc.xyz, pc.opacity, pc.scales, pc.rotations, pc.shs])
        for t in leaves:
            t.requires_grad_(True)
        pkg = render(cam.to(dev), pc, pipe, bg, colors_precomp=pc.colors if precomp else None, return_opacity=want_opacity,
                     l1_target=gt if fused else None)
        l1 = pkg.l1 if fused else l1_loss(pkg.render, gt)
        loss = 0.8 * l1 + (pkg.render * gimg).sum()
        if want_opacity:
            loss = loss + 0.1 * l1_loss(pkg.opacity_render, gmask)
        loss.backward()
        torch.cuda.synchronize()
        cam.to("cpu")
        return (float(l1.detach()), pkg.render.detach().cpu().numpy(),
                [t.grad.clone() for t in leaves] + [pkg.viewspace_points.grad.clone()])

    v_sep, img_sep, g_sep = run(False)
    v_fus, img_fus, g_fus = run(True)
    assert np.array_equal(img_sep, img_fus)
    assert v_fus == pytest.approx(v_sep, rel=2e-6) and v_fus > 0
    assert v_fus == pytest.approx(oracle.l1_loss(img_fus, gt.cpu().numpy())[0], rel=2e-6)
    for k, (a, b) in enumerate(zip(g_sep, g_fus)):
        assert float(a.abs().max()) > 0 and torch.equal(a, b), k
    # the loss alone (no other consumer: the backward is given no gradient image at all), and its value under no_grad
    pc = cloud.to(dev)
    pc.xyz.requires_grad_(True)
    pkg = render(cam.to(dev), pc, Pipe(), bg, l1_target=gt)
    pkg.l1.backward()
    pc2 = cloud.to(dev)
    pc2.xyz.requires_grad_(True)
    l1_loss(render(cam, pc2, Pipe(), bg).render, gt).backward()
    assert torch.equal(pc.xyz.grad, pc2.xyz.grad)
    with torch.no_grad():
        assert float(render(cam, pc2, Pipe(), bg, l1_target=gt).l1) == float(pkg.l1.detach())
    cam.to("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 64), (3, 45, 77), (1, 16, 16), (3, 512, 512)])
def test_fused_ssim_matches_oracle(oracle, shape):
    """N2: gsplat_mi355.render.ssim == utils/loss_utils.py:37-67 `ssim` (fp32 there).  Against the float64 oracle:
    value within 1e-5 absolute; gradient within 1e-4 of its maximum (fp32 sigma = E[x^2] - mu^2 cancels ~3 digits;
    the reference's own fp32 conv2d chain deviates from float64 by the same amount -- checked below)."""
    from gsplat_mi355.render import ssim
    g = torch.Generator(device="cpu").manual_seed(9)
    a = torch.rand(shape, generator=g)
    b = (a + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
    want, want_grad = oracle.ssim(a.numpy(), b.numpy())
    x = a.cuda().requires_grad_(True)
    val = ssim(x, b.cuda())
    loss = 0.2 * (1.0 - val)  # D-SSIM term of train.py:123-124
    loss.backward()
    assert float(val) == pytest.approx(want, abs=1e-5)
    got = x.grad.cpu().numpy() / -0.2
    scale = np.abs(want_grad).max()
    assert np.abs(got - want_grad).max() <= 1e-4 * scale
    # the torch fp32 formula the reference runs, on the same device, for scale
    import torch.nn.functional as F
    g1 = torch.tensor([math.exp(-(k - 5) ** 2 / float(2 * 1.5 ** 2)) for k in range(11)])
    g1 = (g1 / g1.sum()).unsqueeze(1)
    w = g1.mm(g1.t()).float()[None, None].expand(shape[0], 1, 11, 11).contiguous().cuda()
    xa = a.cuda()[None].requires_grad_(True)
    xb = b.cuda()[None]
    conv = lambda t: F.conv2d(t, w, padding=5, groups=shape[0])
    mu1, mu2 = conv(xa), conv(xb)
    s1, s2, s12 = conv(xa * xa) - mu1.pow(2), conv(xb * xb) - mu2.pow(2), conv(xa * xb) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1.pow(2) + mu2.pow(2) + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))
    ref = m.mean()
    ref.backward()
    assert float(val) == pytest.approx(float(ref), abs=2e-5)
    assert np.abs(got - xa.grad[0].cpu().numpy()).max() <= 2e-4 * scale
    # identical images: exactly 1 up to rounding, no NaN; bitwise reproducible
    assert float(ssim(a.cuda(), a.cuda())) == pytest.approx(1.0, abs=1e-6)
    assert float(ssim(a.cuda(), b.cuda())) == float(val)
    with pytest.raises(NotImplementedError):
        ssim(a.cuda(), b.cuda(), window_size=7)


@pytest.mark.gpu
@pytest.mark.parametrize("matrix", [False, True])
def test_build_covariance_matches_oracle(oracle, matrix):
    """N3: gsplat_mi355.prepass.build_covariance_from_scaling_rotation == scene/gaussian_model.py:28-32 (value and
    autograd gradients) within 1e-5 of each tensor's maximum; feeds the rasterizer as cov3D_precomp."""
    from gsplat_mi355.prepass import build_covariance_from_scaling_rotation
    from test_oracle import _n3_inputs
    scaling, rot, g6 = _n3_inputs(5000, 5, matrix)
    want_cov, want_ds, want_dr = oracle.build_covariance(scaling, 1.3, rot, g6)
    s = torch.from_numpy(scaling).cuda().requires_grad_(True)
    r = torch.from_numpy(rot).cuda().requires_grad_(True)
    cov = build_covariance_from_scaling_rotation(s, 1.3, r)
    (cov * torch.from_numpy(g6).cuda()).sum().backward()
    for got, want in [(cov.detach(), want_cov), (s.grad, want_ds), (r.grad, want_dr)]:
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    with pytest.raises(RuntimeError):
        build_covariance_from_scaling_rotation(torch.from_numpy(scaling), 1.0, torch.from_numpy(rot))


@pytest.mark.gpu
@pytest.mark.parametrize("deg,use_rot,use_noise", [(3, True, True), (3, False, False), (2, True, False), (1, False, True),
                                                   (0, False, False)])
def test_sh2rgb_matches_oracle(oracle, deg, use_rot, use_noise):
    """N3: gsplat_mi355.prepass.sh2rgb == models/texture/texture.py:21-38: colours within 2e-6, clamp decisions
    identical except within rounding of zero, gradients within 1e-5 of each tensor's maximum."""
    from gsplat_mi355.prepass import sh2rgb
    from test_oracle import _n3_inputs
    rng = np.random.default_rng(33)
    n = 4000
    feats = (0.5 * rng.normal(size=(n, 16, 3))).astype(np.float32)
    xyz = rng.normal(size=(n, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, 4.0], np.float32)
    _, R, _ = _n3_inputs(n, 8, True)
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    T[:, :3, :3] = R
    T[:, :3, 3] = rng.normal(size=(n, 3))
    th = 0.4
    noise = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    gcol = rng.normal(size=(n, 3)).astype(np.float32)
    want_col, want_cl, want_dsh, want_dp = oracle.sh2rgb(feats, xyz, campos, deg, R if use_rot else None,
                                                          noise if use_noise else None, gcol)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    p = torch.from_numpy(xyz).cuda().requires_grad_(True)
    col = sh2rgb(f, p, torch.from_numpy(campos).cuda(), deg, fwd_transform=torch.from_numpy(T).cuda() if use_rot else None,
                 view_noise=noise if use_noise else None)
    (col * torch.from_numpy(gcol).cuda()).sum().backward()
    got = col.detach().cpu().numpy()
    assert np.abs(got - want_col).max() <= 2e-6
    # a colour within rounding of the clamp may fall on the other side: exclude those Gaussians from the gradient
    # check.  The pre-clamp value r is recovered from a second oracle run with +1 added to every channel (DC term).
    shifted = feats.copy()
    shifted[:, 0, :] += np.float32(1.0 / 0.28209479177387814)
    col1, _ = oracle.sh2rgb(shifted, xyz, campos, deg, R if use_rot else None, noise if use_noise else None)
    r = col1.astype(np.float64) - 1.0  # exact where r > -1; anything below is safely clamped
    keep = (np.abs(r) > 1e-5).all(1)
    assert keep.mean() > 0.99
    gd, gp = f.grad.cpu().numpy(), p.grad.cpu().numpy()
    assert np.abs(gd[keep] - want_dsh[keep]).max() <= 1e-5 * np.abs(want_dsh).max()
    assert np.abs(gp[keep] - want_dp[keep]).max() <= 1e-5 * np.abs(want_dp).max()
    if deg < 3:
        nb = (deg + 1) ** 2
        assert np.all(gd[:, nb:, :] == 0)  # coefficients above the active degree get exactly zero


@pytest.mark.gpu
def test_render_harness_prepass_path_matches_in_kernel_path(oracle):
    """The reference's call pattern (cov3D_precomp from get_covariance, colors_precomp from the texture module) through
    the fused N3 ops gives the image of the in-kernel path (scales / quaternions / SHs handed to the rasterizer) and
    the same gradients for SHs, positions and scalings."""
    from gsplat_mi355.render import Pipe, render
    from gsplat_mi355.scenes import GaussianCloud
    from gsplat_mi355.camera import orbit_camera
    cloud, _ = helpers.cloud_and_camera(3000, 160, 128, sh_degree=3, seed=4)
    dev = torch.device("cuda:0")
    cam = orbit_camera(0, 160, 128, device=dev)
    gimg = torch.rand(3, 128, 160, generator=torch.Generator().manual_seed(3)).to(dev)
    out = {}
    for name, pipe in [("kernel", Pipe()), ("prepass", Pipe(compute_cov3D_python=True, convert_SHs_python=True))]:
        c = GaussianCloud(*[getattr(cloud, f).to(dev).clone().requires_grad_(True) for f in GaussianCloud.FIELDS], cloud.sh_degree)
        pkg = render(cam, c, pipe, torch.zeros(3, device=dev))
        (pkg.render * gimg).sum().backward()
        out[name] = (pkg.render.detach().cpu().numpy(), c.shs.grad.cpu().numpy(), c.xyz.grad.cpu().numpy(),
                     c.scales.grad.cpu().numpy())
    for a, b, nm in zip(out["kernel"], out["prepass"], ["image", "d/dshs", "d/dxyz", "d/dscales"]):
        _bulk_close(b, a, tol=2e-5, frac=1e-3, name=nm)


@pytest.mark.gpu
def test_knn_points_exact_vs_oracle(oracle):
    """N4: gsplat_mi355.knn.knn_points (pytorch3d.ops.knn_points call shape) == brute force, bit-exact distances and
    identical indices: self-KNN K = 6 (utils/loss_utils.py:92-96), K = 5 (:76-79), query vs a small vertex set
    K = 1 (models/deformer/rigid.py:43), duplicated points, fewer reference points than K."""
    from gsplat_mi355.knn import knn_points
    rng = np.random.default_rng(6)
    pts = np.concatenate([rng.normal(size=(20000, 3)), 0.01 * rng.normal(size=(3000, 3)) + 2.0]).astype(np.float32)
    pts[100:110] = pts[0:10]  # duplicates
    x = torch.from_numpy(pts).cuda()
    for K in (6, 5, 1):
        want_d, want_i = oracle.knn_points(pts, pts, K)
        got = knn_points(x[None], x[None], K=K, return_sorted=True)
        assert got.idx.shape == (1, len(pts), K) and got.idx.dtype == torch.int64
        assert np.array_equal(got.dists[0].cpu().numpy(), want_d)
        assert np.array_equal(got.idx[0].cpu().numpy(), want_i)
    verts = rng.normal(size=(6890, 3)).astype(np.float32)
    want_d, want_i = oracle.knn_points(pts, verts, 1)
    got = knn_points(x.unsqueeze(0), torch.from_numpy(verts).cuda().unsqueeze(0))
    assert np.array_equal(got.idx[0, :, 0].cpu().numpy(), want_i[:, 0])
    assert np.array_equal(got.dists[0].cpu().numpy(), want_d)
    few = torch.from_numpy(verts[:3]).cuda()
    got = knn_points(x[:10], few, K=5)
    assert np.all(got.idx[:, 3:].cpu().numpy() == -1) and np.all(got.idx[:, :3].cpu().numpy() >= 0)
    with pytest.raises(RuntimeError):
        knn_points(torch.from_numpy(pts), torch.from_numpy(pts), K=3)


@pytest.mark.gpu
def test_fused_adam_and_densify_stats_match_oracle_and_torch(oracle):
    """N4: gsplat_mi355.optim.FusedAdam follows torch.optim.Adam(l, lr=0.0, eps=1e-15) with the reference's six
    parameter groups (scene/gaussian_model.py:201-216) step for step (fp32: 1e-6 of the tensor's scale), keeps
    torch's state layout, and densify_stats reproduces train.py:219-220 + gaussian_model.py:464-466 exactly."""
    from gsplat_mi355.optim import FusedAdam, densify_stats
    rng = np.random.default_rng(12)
    n = 5000
    shapes = [(n, 3), (n, 1, 3), (n, 15, 3), (n, 1), (n, 3), (n, 4)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3]
    init = [rng.normal(size=s).astype(np.float32) for s in shapes]
    dev = torch.device("cuda:0")
    mine = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in init]
    ref = [torch.nn.Parameter(torch.from_numpy(a).to(dev)) for a in init]
    o_mine = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(mine, lrs))], lr=0.0, eps=1e-15)
    o_ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], lr=0.0, eps=1e-15)
    orc = [(a.astype(np.float64), np.zeros(a.shape), np.zeros(a.shape)) for a in init]
    for step in range(1, 5):
        grads = [(rng.normal(size=s) * 10.0 ** rng.integers(-4, 1)).astype(np.float32) for s in shapes]
        grads[0][::2] = 0.0
        for pm, pr, g in zip(mine, ref, grads):
            pm.grad = torch.from_numpy(g).to(dev)
            pr.grad = torch.from_numpy(g).to(dev)
        o_mine.step()
        o_ref.step()
        orc = [oracle.adam_step(p, g, m, v, lr, 0.9, 0.999, 1e-15, step) for (p, m, v), g, lr in zip(orc, grads, lrs)]
        for pm, pr, (q, m, v) in zip(mine, ref, orc):
            got = pm.detach().cpu().numpy().astype(np.float64)
            scale = np.abs(q).max()
            assert np.abs(got - q).max() <= 2e-6 * scale
            assert np.abs(got - pr.detach().cpu().numpy()).max() <= 2e-6 * scale
            st = o_mine.state[pm]
            assert set(st.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == step
            assert np.abs(st["exp_avg"].cpu().numpy() - m).max() <= 2e-6 * max(np.abs(m).max(), 1e-30)
            assert np.abs(st["exp_avg_sq"].cpu().numpy() - v).max() <= 2e-6 * max(np.abs(v).max(), 1e-30)
    radii = rng.integers(-1, 40, size=n).astype(np.int32)
    radii[radii < 0] = 0
    vg = rng.normal(size=(n, 3)).astype(np.float32)
    mr, acc, dn = (rng.random(n).astype(np.float32) * 20 for _ in range(3))
    want = oracle.densify_stats(radii, vg, mr, acc, dn)
    t = [torch.from_numpy(a).to(dev) for a in (mr, acc.reshape(n, 1), dn.reshape(n, 1))]
    densify_stats(torch.from_numpy(radii).to(dev), torch.from_numpy(vg).to(dev), *t)
    for got, w in zip(t, want):
        assert np.array_equal(got.cpu().numpy().reshape(-1), w)


@pytest.mark.gpu
def test_end_to_end_optimisation_reduces_the_loss():
    """Everything together, as the reference's training step strings it: render -> 0.8 L1 + 0.2 D-SSIM -> backward ->
    densification statistics -> Adam (all through the HIP library).  Fitting a perturbed cloud to the image of the
    unperturbed one must reduce the loss substantially in 40 steps, with finite parameters throughout."""
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.optim import FusedAdam
    from gsplat_mi355.render import DensifyStats, Pipe, l1_loss, render, ssim
    from gsplat_mi355.scenes import GaussianCloud
    dev = torch.device("cuda:0")
    cloud, _ = helpers.cloud_and_camera(4000, 128, 128, sh_degree=1, seed=7)
    cam = orbit_camera(0, 128, 128, device=dev)
    bg = torch.zeros(3, device=dev)
    target_cloud = GaussianCloud(*[getattr(cloud, f).to(dev) for f in GaussianCloud.FIELDS], cloud.sh_degree)
    with torch.no_grad():
        gt = render(cam, target_cloud, Pipe(), bg).render.clone()
    g = torch.Generator().manual_seed(0)
    fields = {f: getattr(cloud, f).clone() for f in GaussianCloud.FIELDS}
    fields["shs"] = fields["shs"] + 0.3 * torch.randn(fields["shs"].shape, generator=g)
    fields["opacity"] = (fields["opacity"] * 0.6).clamp(0.01, 0.99)
    c = GaussianCloud(*[fields[f].to(dev).requires_grad_(True) for f in GaussianCloud.FIELDS], cloud.sh_degree)
    lrs = dict(xyz=1e-4, scales=1e-3, rotations=1e-3, opacity=2e-2, shs=2e-2)
    opt = FusedAdam([{"params": [getattr(c, f)], "lr": lrs[f], "name": f} for f in GaussianCloud.FIELDS], lr=0.0, eps=1e-15)
    stats = DensifyStats(4000, dev)
    losses = []
    for it in range(40):
        opt.zero_grad(set_to_none=True)
        pkg = render(cam, c, Pipe(), bg)
        loss = 0.8 * l1_loss(pkg.render, gt) + 0.2 * (1.0 - ssim(pkg.render, gt))
        loss.backward()
        with torch.no_grad():
            stats.update(pkg)
            opt.step()
            c.opacity.clamp_(1e-4, 0.9999)  # the reference keeps opacity inside (0, 1) through its sigmoid activation
        losses.append(float(loss.detach()))
    assert all(np.isfinite(l) for l in losses)
    assert losses[-1] < 0.6 * losses[0], losses[::8]
    for f in GaussianCloud.FIELDS:
        assert torch.isfinite(getattr(c, f)).all()
    assert float(stats.denom.max()) == 40.0 and float(stats.max_radii2D.max()) > 0


@pytest.mark.gpu
def test_random_small_scenes_against_oracle(oracle):
    """Fuzz: 24 random small scenes (Gaussian count 1..2500, ragged image sizes down to a single tile, every SH degree,
    all four input combinations, list lengths around the 16-entry round boundaries of the backward) -- radii exact,
    image and all gradients within the float bar."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2024)
    for trial in range(24):
        n = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 65, 200, 700, 2500]))
        W, H = int(rng.integers(9, 150)), int(rng.integers(9, 150))
        deg = int(rng.integers(0, 4))
        color_mode, cov_mode = COMBOS[trial % len(COMBOS)][:2]
        cloud, cam = helpers.cloud_and_camera(max(n, 4), W, H, sh_degree=deg, seed=100 + trial, scale_mul=float(rng.uniform(0.5, 3.0)))
        if n < 4:
            for f in ("xyz", "scales", "rotations", "opacity", "shs"):
                setattr(cloud, f, getattr(cloud, f)[:n].clone())
        bg = tuple(float(v) for v in rng.random(3))
        sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode=color_mode, cov_mode=cov_mode)
        fw = oracle.forward(sc)
        gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(trial))
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, color_mode, cov_mode, dev).items()}
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(cloud.xyz.shape[0], 3, device=dev, requires_grad=True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        color, radii = GaussianRasterizer(_settings(cam, cloud, bg, dev))(means3D=means3D, means2D=means2D, opacities=opac, **kw)
        (color * gimg.to(dev)).sum().backward()
        tag = "trial %d (n=%d %dx%d deg %d %s/%s)" % (trial, n, W, H, deg, color_mode, cov_mode)
        assert np.array_equal(radii.cpu().numpy(), fw["radii"]), tag
        # every difference in the forward attributed to decisions at a threshold; gradients against the oracle conditioned on them
        st, fwc, ov = _conditioned_oracle(oracle, sc, fw, _settings(cam, cloud, bg, dev), means3D, opac, kw, "fuzz " + tag)
        assert np.array_equal(color.detach().cpu().numpy(), st["color"]), tag
        want = oracle.backward(sc, fwc, gimg.numpy(), ov)
        names = dict(shs="sh", colors_precomp="colors_precomp", scales="scales", rotations="rotations",
                     cov3D_precomp="cov3D_precomp")
        got = dict(means3D=means3D.grad, means2D=means2D.grad, opacities=opac.grad)
        for k, v in kw.items():
            got[names[k]] = v.grad
        for name, gt in got.items():
            w = want[name].reshape(gt.shape)
            if np.abs(w).max() == 0:
                assert np.abs(gt.cpu().numpy()).max() == 0, tag + " " + name
                continue
            _bulk_close(gt.cpu().numpy(), w, tol=SMALL_TOL, frac=0.0, name=tag + " " + name)


@pytest.mark.gpu
def test_backward_with_saved_tensors_relocated_by_hooks():
    """torch.autograd.graph.save_on_cpu moves every saved tensor to the host and back into NEW device storage for the
    backward: the wrapper must not use the device addresses it took in the forward (its argument block is rebuilt when
    the saved tensors have moved).  Same gradients, bit for bit, as without the hook."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 4000, 96, 80
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=12, scale_mul=1.5)
    bg = (0.1, 0.1, 0.1)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)

    def run(offload):
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        rast = GaussianRasterizer(_settings(cam, cloud, bg, dev))
        if offload:
            with torch.autograd.graph.save_on_cpu():
                color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw)
        else:
            color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw)
        (color * gimg).sum().backward()
        return [t.grad.clone() for t in (means3D, means2D, opac, kw["shs"], kw["scales"], kw["rotations"])]

    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_random_dense_small_scenes_against_oracle(oracle):
    """Fuzz of the few-long-lists paths (four-wave forward on marked tiles, backward in chunks): ten random DENSE small
    scenes -- thousands of entries per tile on ragged images of a few dozen tiles, thin-shell and box clouds, opaque to
    translucent, every SH degree -- radii exact, image and all gradients within the float bar, and the tile marks /
    chunk checkpoints actually in play."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    marked = chunked = 0
    for trial in range(10):
        n = int(rng.choice([6000, 12000, 25000]))
        W, H = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        deg = int(rng.integers(0, 4))
        layout = "body" if trial % 2 == 0 else "box"
        cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=300 + trial, layout=layout,
                                              scale_mul=float(rng.uniform(0.6, 1.5)))
        cloud.opacity = cloud.opacity * float(rng.choice([0.1, 0.4, 1.0]))
        bg = tuple(float(v) for v in rng.random(3))
        sc = helpers.oracle_scene(cloud, cam, bg=bg)
        fw = oracle.forward(sc)
        gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(trial))
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        settings = _settings(cam, cloud, bg, dev)
        color, radii = GaussianRasterizer(settings)(means3D=means3D, means2D=means2D, opacities=opac, **kw)
        (color * gimg.to(dev)).sum().backward()
        tag = "trial %d (n=%d %dx%d deg %d %s)" % (trial, n, W, H, deg, layout)
        # (dense scenes: hundreds of pairs per pixel, opaque ones stop at the 1e-4 threshold -- dozens of flipped decisions
        # per frame, every one attributed; then the conditioned oracle)
        st, fwc, ov = _conditioned_oracle(oracle, sc, fw, settings, means3D, opac, kw, "dense fuzz " + tag)
        want = oracle.backward(sc, fwc, gimg.numpy(), ov)
        marked += int((st["image"]["order"] >> 31).sum())
        chunked += int((st["image"]["qcount"] > 256).sum())
        assert np.array_equal(radii.cpu().numpy(), fw["geom"]["radii"]), tag
        assert np.array_equal(st["binning"]["point_list"], fw["binning"]["point_list"]), tag
        assert np.array_equal(color.detach().cpu().numpy(), st["color"]), tag
        got = dict(means3D=means3D.grad, means2D=means2D.grad, opacities=opac.grad, sh=kw["shs"].grad,
                   scales=kw["scales"].grad, rotations=kw["rotations"].grad)
        for name, gt in got.items():
            w = want[name].reshape(gt.shape)
            _bulk_close(gt.cpu().numpy(), w, tol=SMALL_TOL, frac=0.0, name=tag + " " + name)
    assert marked > 20 and chunked > 20  # the paths under test did run


@pytest.mark.gpu
@pytest.mark.parametrize("bgval", [(0.0, 0.0, 0.0), (0.3, 0.6, 0.1)])
def test_fused_opacity_render_matches_the_second_rasterizer_call(oracle, bgval):
    """N1, second form: rasterizer(..., with_opacity=True) returns the image the reference gets from its second call
    with colours = 1 (`[:1]`, gaussian_renderer/__init__.py:132-142) and its backward returns the SUM of the two calls'
    gradients -- compared here with the two-call path of this library and with the oracle's opacity render.  (This is
    a self-comparison of the gradients; BOTH forms are compared with the oracle's two backward passes, every gradient
    tensor, in test_full_size_reference_step_two_images_default_path.)"""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 3000, 150, 110
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=13, scale_mul=1.2)
    gen = torch.Generator().manual_seed(8)
    gimg = torch.randn(3, H, W, generator=gen).to(dev)
    gop = torch.randn(1, H, W, generator=gen).to(dev)
    rast = GaussianRasterizer(_settings(cam, cloud, bgval, dev))

    def leaves():
        return dict(means3D=cloud.xyz.to(dev).clone().requires_grad_(True),
                    means2D=torch.zeros(n, 3, device=dev, requires_grad=True),
                    opacities=cloud.opacity.to(dev).clone().requires_grad_(True),
                    shs=cloud.shs.to(dev).clone().requires_grad_(True),
                    scales=cloud.scales.to(dev).clone().requires_grad_(True),
                    rotations=cloud.rotations.to(dev).clone().requires_grad_(True))

    a = leaves()
    color, radii, opa = rast(with_opacity=True, **a)
    ((color * gimg).sum() + (opa * gop).sum()).backward()
    b = leaves()
    color2, radii2 = rast(**b)
    opa2, _ = rast(means3D=b["means3D"], means2D=b["means2D"], opacities=b["opacities"], shs=None,
                   colors_precomp=torch.ones(n, 3, device=dev), scales=b["scales"], rotations=b["rotations"])
    ((color2 * gimg).sum() + (opa2[:1] * gop).sum()).backward()
    assert torch.equal(color, color2) and torch.equal(radii, radii2)
    assert np.abs((opa - opa2[:1]).detach().cpu().numpy()).max() <= 2e-6
    sc = helpers.oracle_scene(cloud, cam, bg=bgval, color_mode="precomp", cov_mode="scale_rot", colors=torch.ones(n, 3))
    assert np.abs(opa[0].detach().cpu().numpy() - oracle.forward(sc)["color"][0]).max() <= 1e-2
    _bulk_close(opa[0].detach().cpu().numpy(), oracle.forward(sc)["color"][0], name="opacity render")
    for k in a:
        _bulk_close(a[k].grad.cpu().numpy(), b[k].grad.cpu().numpy(), tol=2e-5, frac=1e-4, name="fused vs two calls: " + k)
    # only the opacity render used downstream
    c = leaves()
    _, _, opa3 = rast(with_opacity=True, **c)
    (opa3 * gop).sum().backward()
    d = leaves()
    opa4, _ = rast(means3D=d["means3D"], means2D=d["means2D"], opacities=d["opacities"], shs=None,
                   colors_precomp=torch.ones(n, 3, device=dev), scales=d["scales"], rotations=d["rotations"])
    (opa4[:1] * gop).sum().backward()
    for k in ("means3D", "opacities", "scales", "rotations"):
        _bulk_close(c[k].grad.cpu().numpy(), d[k].grad.cpu().numpy(), tol=2e-5, frac=1e-4, name="opacity only: " + k)


@pytest.mark.gpu
def test_fused_bce_mask_loss_matches_torch():
    """N2: gsplat_mi355.render.bce_mask_loss == F.binary_cross_entropy(torch.clamp(opacity, 1e-3, 1 - 1e-3), mask)
    (train.py:146-148), pinned against torch's own float64 evaluation on the CPU: value 1e-6 relative, gradient 1e-5
    of its maximum, zero gradient where the clamp is active."""
    import torch.nn.functional as F
    from gsplat_mi355.render import bce_mask_loss
    g = torch.Generator().manual_seed(17)
    x = torch.rand(1, 97, 131, generator=g)
    x.view(-1)[:50] = 0.0      # below the clamp
    x.view(-1)[50:100] = 1.0   # above the clamp
    y = (torch.rand(1, 97, 131, generator=g) > 0.4).float()
    xd = x.double().requires_grad_(True)
    ref = F.binary_cross_entropy(torch.clamp(xd, 1.0e-3, 1.0 - 1.0e-3), y.double())
    (3.0 * ref).backward()
    xg = x.cuda().requires_grad_(True)
    loss = bce_mask_loss(xg, y.cuda())
    (3.0 * loss).backward()
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=2e-6)
    want = xd.grad.numpy()
    got = xg.grad.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    assert np.all(got.reshape(-1)[:100] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("color_mode,cov_mode,deg,bg", [("sh", "scale_rot", 3, (0.0, 0.0, 0.0)),
                                                         ("precomp", "cov", 0, (0.3, 0.6, 0.1))])
def test_hip_path_against_dense_float64_autograd_directly(oracle, color_mode, cov_mode, deg, bg):
    """A second, independent check of the product path: the HIP rasterizer against the dense float64 PyTorch
    restatement with autograd (oracle/dense_ref.py) -- not through the C oracle -- on a scene small enough for it:
    image within 1e-5, every gradient within 2e-4 of its maximum (the bar the C oracle itself is held to)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from test_oracle import _dense_grads
    dev = torch.device("cuda:0")
    n, W, H = 300, 48, 32
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=11, scale_mul=1.5)
    cloud.shs[:, 0] -= 1.2 * (torch.arange(n) % 7 == 0).float()[:, None]
    cloud.xyz[::13, 0] *= 2.4
    sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode=color_mode, cov_mode=cov_mode)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2))
    color64, radii64, grads64, _ = _dense_grads(sc, cloud, cam, gimg.numpy().astype(np.float32), color_mode, cov_mode, bg)
    kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, color_mode, cov_mode, dev).items()}
    means3D = cloud.xyz.to(dev).requires_grad_(True)
    means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
    opac = cloud.opacity.to(dev).requires_grad_(True)
    color, radii = GaussianRasterizer(_settings(cam, cloud, bg, dev))(means3D=means3D, means2D=means2D, opacities=opac, **kw)
    (color * gimg.to(dev)).sum().backward()
    assert np.array_equal(radii.cpu().numpy(), radii64)
    assert np.abs(color.detach().cpu().numpy() - color64).max() < 1e-5
    names = dict(shs="sh", colors_precomp="colors_precomp", scales="scales", rotations="rotations", cov3D_precomp="cov3D_precomp")
    got = dict(means3D=means3D.grad, means2D=means2D.grad, opacities=opac.grad)
    for k, v in kw.items():
        got[names[k]] = v.grad
    for name, ref in grads64.items():
        err = helpers.rel_to_max(got[name].cpu().numpy().reshape(ref.shape), ref)
        assert err < 2e-4, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("four_wave_forward", [0, 1])
def test_snug_tile_rectangles_give_bitwise_the_same_outputs_as_upstream_squares(oracle, four_wave_forward):
    """GsFwdArgs.tile_rect: 1 (default, bounding box of the alpha >= 1/255 region) against 0 (upstream's 3-sigma
    square) through the product: colour and radii bitwise identical, every gradient equal to fp32 rounding,
    num_rendered much smaller; and each mode matches the oracle's binning in the same mode bit for bit.
    On small images the forward renders the tiles with long lists four entries per step (render_fwd.hip), which adds a
    pixel's colour up in four partial sums: WHICH tiles depends on the list lengths, i.e. on the mode, so with that
    kernel in play (the default, four_wave_forward = 1) the colours agree to fp32 rounding; with the one-entry-per-step
    kernel everywhere (gs_tuning "fwd4" = 0) bit for bit."""
    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import _lib, debug
    _lib.tuning("fwd4", four_wave_forward)
    try:
        _snug_vs_squares(oracle, dgr, debug, bitwise=not four_wave_forward)
    finally:
        _lib.tuning("fwd4", int(os.environ.get("GSPLAT_FWD4", "1")))  # (the process's own setting)


def _snug_vs_squares(oracle, dgr, debug, bitwise):
    dev = torch.device("cuda:0")
    n, W, H = 6000, 200, 150
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=41, scale_mul=1.4)
    cloud.opacity[::5] *= 0.05
    cloud.opacity[::37] = 0.003
    bg = (0.2, 0.4, 0.6)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    saved = dgr._TILE_RECT
    res = {}
    try:
        for mode in (0, 1):
            dgr._TILE_RECT = mode
            kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
            means3D = cloud.xyz.to(dev).requires_grad_(True)
            means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
            opac = cloud.opacity.to(dev).requires_grad_(True)
            settings = _settings(cam, cloud, bg, dev)
            color, radii = dgr.GaussianRasterizer(settings)(means3D=means3D, means2D=means2D, opacities=opac, **kw)
            (color * gimg).sum().backward()
            st = debug.forward_state(settings, means3D.detach(), opac.detach(), **{k: v.detach() for k, v in kw.items()})
            res[mode] = dict(color=color.detach().clone(), radii=radii.clone(), D=st["D"], st=st,
                             grads=[t.grad.clone() for t in (means3D, means2D, opac, kw["shs"], kw["scales"], kw["rotations"])])
    finally:
        dgr._TILE_RECT = saved
    assert torch.equal(res[0]["radii"], res[1]["radii"])
    if bitwise:
        assert torch.equal(res[0]["color"], res[1]["color"])
    else:
        assert (res[0]["color"] - res[1]["color"]).abs().max().item() <= 3e-6  # (sums of hundreds of terms, values up to 1)
        wide = [int((res[m]["st"]["image"]["order"] >> 31).sum()) for m in (0, 1)]
        assert wide[0] > 0  # (the four-wave kernel did render tiles of this frame)
    # the gradient rows are the same in both modes; the per-Gaussian reduction adds them in an order that depends on
    # the pair numbering, so the sums agree to fp32 rounding rather than bit for bit (the oracle, which accumulates in
    # double, is bitwise identical: tests/test_oracle.py)
    for a, b in zip(res[0]["grads"], res[1]["grads"]):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
        assert np.array_equal(a == 0, b == 0)
    assert res[1]["D"] < 0.8 * res[0]["D"]
    for mode in (0, 1):  # each mode against the oracle in the same mode, bit for bit
        fw = oracle.forward(helpers.oracle_scene(cloud, cam, bg=bg, tile_rect=mode))
        assert res[mode]["D"] == fw["binning"]["D"]
        assert np.array_equal(res[mode]["st"]["geom"]["tiles_touched"], fw["geom"]["tiles_touched"])
        assert np.array_equal(res[mode]["st"]["binning"]["point_list"], fw["binning"]["point_list"])


@pytest.mark.gpu
def test_snug_tile_rectangles_stay_conservative_for_needle_gaussians(oracle):
    """Ill-conditioned 2-D covariances (needles hundreds of pixels long, thin axis at the low-pass floor): the alpha >=
    1/255 region of the ROUNDED fp32 conic reaches beyond the exact ellipse's bounding box; tile_rect = 1 widens its box
    by the conditioning bound (gs_math.h: snug_half_widths), so the product's image stays bitwise that of upstream's
    squares, and each mode's binning matches the oracle in the same mode."""
    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import debug
    dev = torch.device("cuda:0")
    n, W, H = 200, 2048, 2048
    saved = dgr._TILE_RECT
    try:
        for seed in (0, 1, 2):
            cloud, cam = helpers.needle_cloud_and_camera(n, W, H, seed=seed)
            bg = (0.1, 0.2, 0.3)
            res = {}
            for mode in (0, 1):
                dgr._TILE_RECT = mode
                st = debug.forward_state(_settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                                         shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
                fw = oracle.forward(helpers.oracle_scene(cloud, cam, bg=bg, tile_rect=mode))
                assert st["D"] == fw["binning"]["D"]
                assert np.array_equal(st["geom"]["tiles_touched"], fw["geom"]["tiles_touched"])
                assert np.array_equal(st["binning"]["point_list"], fw["binning"]["point_list"])
                # NOT the 1e-5 bar: the quadratic form of a needle is a sum of terms ~ (a c / det) times larger than
                # the result, so any two fp32 evaluation orders (the oracle's, this kernel's log2-domain FMAs, upstream's
                # nvcc contraction) differ by ~2^-24 a c / det in the exponent -- percent-level alphas for these Gaussians
                assert np.abs(st["color"] - fw["color"]).max() < 0.05
                res[mode] = st
            assert res[1]["D"] < 0.6 * res[0]["D"]
            assert np.array_equal(res[0]["color"], res[1]["color"]), seed
            assert np.array_equal(res[0]["image"]["final_T"], res[1]["image"]["final_T"]), seed
    finally:
        dgr._TILE_RECT = saved


def test_two_host_threads_on_two_streams_render_the_single_stream_bits():
    """The library keeps no state between calls and works on the caller's stream: frames rendered
    concurrently from two host threads, each on its own stream, equal the single-stream frames bit for bit
    (forward and gradients)."""
    import threading
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    n, W, H = 8000, 256, 208
    cloud, _ = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=21, scale_mul=1.3)
    from gsplat_mi355.camera import orbit_camera
    cams = [orbit_camera(f * 5, W, H) for f in range(6)]
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(8)).to(dev)
    base = {k: getattr(cloud, k).to(dev) for k in ("xyz", "opacity", "shs", "scales", "rotations")}

    def frame(i):
        leaves = dict(means3D=base["xyz"].clone().requires_grad_(True), means2D=torch.zeros(n, 3, device=dev, requires_grad=True),
                      opacities=base["opacity"].clone().requires_grad_(True), shs=base["shs"].clone().requires_grad_(True),
                      scales=base["scales"].clone().requires_grad_(True), rotations=base["rotations"].clone().requires_grad_(True))
        color, radii = GaussianRasterizer(_settings(cams[i], cloud, (0.1, 0.2, 0.3), dev))(**leaves)
        (color * gimg).sum().backward()
        return [color.detach(), radii] + [v.grad for v in leaves.values()]

    want = [frame(i) for i in range(len(cams))]
    torch.cuda.synchronize()
    got, errors = {}, []

    def worker(tid):
        try:
            st = torch.cuda.Stream(device=dev)
            st.wait_stream(torch.cuda.default_stream(dev))
            with torch.cuda.stream(st):
                for rep in range(3):
                    for i in range(tid, len(cams), 2):
                        got[i] = frame(i)
                st.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for i in range(len(cams)):
        for a, b in zip(got[i], want[i]):
            assert torch.equal(a, b), i


@pytest.mark.gpu
def test_gradient_row_marks_set_by_the_forward_and_by_a_repeated_backward():
    """The backward starts from ROW_UNWRITTEN in every mark word of its gradient rows.  The forward's render launch sets
    them on the side (binning state; a state word says they are untouched), a backward that finds them used sets them
    itself: the first backward of a forward, a second one of the same forward (retain_graph) and a run with the forward's
    side job switched off (gs_tuning "fwd_marks" = 0) all give the same bits.  The two-render step, whose second render
    shares the first one's binning state, is covered by the same rule (its tests run with the default switches)."""
    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import _lib
    dev = torch.device("cuda:0")
    n, W, H = 8000, 320, 240
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=77, scale_mul=1.3)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)

    def run(backwards):
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        leaves = (means3D, means2D, opac, kw["shs"], kw["scales"], kw["rotations"])
        color, _ = dgr.GaussianRasterizer(_settings(cam, cloud, (0.1, 0.2, 0.3), dev))(means3D=means3D, means2D=means2D,
                                                                                     opacities=opac, **kw)
        out = []
        for k in range(backwards):
            for t in leaves:
                t.grad = None
            (color * gimg).sum().backward(retain_graph=k + 1 < backwards)
            out.append([t.grad.clone() for t in leaves])
        return color.detach().clone(), out

    color, (first, second) = run(2)
    assert all(float(g.abs().max()) > 0 for g in first[:3])
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    _lib.tuning("fwd_marks", 0)
    try:
        color0, (plain,) = run(1)
    finally:
        _lib.tuning("fwd_marks", 1)
    assert torch.equal(color, color0)
    for a, b in zip(first, plain):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("with_opacity", [False, True])
def test_call_under_no_grad_takes_the_path_without_an_autograd_node_and_gives_the_same_bits(with_opacity):
    """Under torch.no_grad() (the reference's render loop, render.py:51-62) rasterize_gaussians calls the forward directly
    instead of through autograd.Function.apply: same image, radii and opacity image bit for bit, no graph attached -- also
    when the inputs require gradients, as the reference's parameters do while it renders for evaluation."""
    import diff_gaussian_rasterization as dgr
    dev = torch.device("cuda:0")
    n, W, H = 6000, 200, 150
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=5, scale_mul=1.3)
    kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
    means3D = cloud.xyz.to(dev).requires_grad_(True)
    means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
    opac = cloud.opacity.to(dev).requires_grad_(True)
    rast = dgr.GaussianRasterizer(_settings(cam, cloud, (0.3, 0.1, 0.2), dev))
    extra = {"with_opacity": True} if with_opacity else {}
    out_grad = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw, **extra)
    assert out_grad[0].requires_grad and out_grad[0].grad_fn is not None
    dgr.release_shared_geometry()
    calls = []
    real_apply = dgr._RasterizeGaussians.apply
    try:
        dgr._RasterizeGaussians.apply = lambda *a: (calls.append(1), real_apply(*a))[1]
        with torch.no_grad():
            out = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw, **extra)
    finally:
        dgr._RasterizeGaussians.apply = real_apply
    assert not calls  # (no autograd.Function.apply on this path)
    assert len(out) == len(out_grad) == (3 if with_opacity else 2)
    for a, b in zip(out, out_grad):
        assert not a.requires_grad and a.grad_fn is None
        assert torch.equal(a, b.detach())


@pytest.mark.gpu
def test_depth_ranking_of_a_million_equal_depths():
    """1.2 M Gaussians at ONE view-space depth: one bucket of the depth ranking holds them all, is cut into 1024 sub-buckets
    by sampling (depth_sort.hip: ds_giant_bucket) and every sub-bucket (~1200 composites) is still too large for a wave,
    so the whole workgroup sorts it: the last of the ranking's paths.  Equal keys rank in ascending index order -- the
    stable order -- so the ranking of the Gaussians that touch a tile must be their indices in ascending order."""
    from gsplat_mi355 import debug
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.scenes import GaussianCloud
    dev = torch.device("cuda:0")
    n, W, H = 1_200_000, 256, 256
    g = torch.Generator().manual_seed(5)
    xyz = torch.empty(n, 3).uniform_(-0.9, 0.9, generator=g)
    xyz[:, 2] = 0.125  # every view-space depth is exactly 3.125
    cloud = GaussianCloud(xyz, torch.full((n, 3), 0.004), torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1),
                                  torch.full((n, 1), 0.5), torch.rand(n, 1, 3, generator=g), 0)
    cam = orbit_camera(0, W, H)
    st = debug.forward_state(_settings(cam, cloud, (0.0, 0.0, 0.0), dev), cloud.xyz.to(dev), cloud.opacity.to(dev),
                             shs=cloud.shs.to(dev), scales=cloud.scales.to(dev), rotations=cloud.rotations.to(dev))
    tt = st["geom"]["tiles_touched"]
    order = st["geom"]["sorted_idx"].astype(np.int64)
    nvis = int((tt > 0).sum())
    assert nvis > 1_000_000 and len(np.unique(st["geom"]["depths"][tt > 0])) == 1
    assert np.array_equal(order[:nvis], np.nonzero(tt > 0)[0])
    assert np.array_equal(np.sort(order[nvis:]), np.nonzero(tt == 0)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["depth_sort", "xcd_map", "nt_stores", "bwd_order", "fwd_marks"])
def test_library_switches_without_an_effect_on_the_results(switch):
    """gs_tuning switches documented as "without effect on the results" (include/gsplat_mi355.h): the LSD radix sort in
    place of the bucket sort of the depth ranking, the tile -> XCD mapping of the render launches, streaming stores for
    the row marks, the backward's own tile order, the forward's side job -- image, radii and every gradient bit for bit
    the default's (the per-pixel and per-Gaussian sums do not depend on which wave runs when)."""
    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import _lib
    dev = torch.device("cuda:0")
    n, W, H = 9000, 272, 208
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=23, scale_mul=1.2)
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9)).to(dev)

    def run():
        kw = {k: v.clone().requires_grad_(True) for k, v in _inputs(cloud, cam, "sh", "scale_rot", dev).items()}
        means3D = cloud.xyz.to(dev).requires_grad_(True)
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        opac = cloud.opacity.to(dev).requires_grad_(True)
        color, radii = dgr.GaussianRasterizer(_settings(cam, cloud, (0.2, 0.3, 0.1), dev))(means3D=means3D, means2D=means2D,
                                                                                            opacities=opac, **kw)
        (color * gimg).sum().backward()
        return [color.detach().clone(), radii.clone()] + [t.grad.clone() for t in (means3D, means2D, opac, kw["shs"], kw["scales"],
                                                                                   kw["rotations"])]

    ref = run()
    _lib.tuning(switch, 0)
    try:
        alt = run()
    finally:
        _lib.tuning(switch, 1)
    for a, b in zip(ref, alt):
        assert torch.equal(a, b)
