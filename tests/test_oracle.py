"""Self-checks of the CPU oracle (oracle/gs_oracle.c): closed forms, structural invariants, and every
analytic backward formula against float64 autograd of an independent dense restatement
(oracle/dense_ref.py).  The reference holds no tests or vectors for this path, so
these are what pins the oracle; see DESIGN.md "parity unpinned"."""
import math

import numpy as np
import pytest
import torch

import helpers
from gsplat_mi355.camera import Camera, focal2fov


def _single(oracle, W=64, H=64, pos=(0.0, 0.0, 0.0), sigma=0.05, opacity=0.8, rgb=(0.2, 0.5, 0.9), bg=(0.1, 0.2, 0.3),
            z=3.0):
    f = 500.0 * W / 512.0
    cam = Camera(np.eye(3), np.array([0.0, 0.0, z]), focal2fov(f, W), focal2fov(f, H), W, H)
    sc = oracle.Scene(W, H, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), np.array(bg, np.float32),
                      cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy(),
                      np.array([pos], np.float32), np.array([opacity], np.float32),
                      colors_precomp=np.array([rgb], np.float32), scales=np.full((1, 3), sigma, np.float32),
                      rotations=np.array([[1, 0, 0, 0]], np.float32))
    return sc, cam, f


def test_closed_form_single_isotropic_gaussian(oracle):
    W = H = 64
    # centre the Gaussian exactly on pixel (32, 20): ndc = (2*px + 1)/W - 1
    z = 3.0
    f = 500.0 * W / 512.0
    px, py = 32, 20
    x = ((2 * px + 1) / W - 1) * (W / (2 * f)) * z
    y = ((2 * py + 1) / H - 1) * (H / (2 * f)) * z
    sc, cam, f = _single(oracle, W, H, pos=(x, y, 0.0), z=z)
    fw = oracle.forward(sc)
    st = fw["geom"]
    assert abs(st["xy"][0, 0] - px) < 1e-3 and abs(st["xy"][0, 1] - py) < 1e-3
    s2 = (f * 0.05 / z) ** 2 + 0.3  # cov2D = (fx sigma / z)^2 I + 0.3 I near the optical axis
    lam = 1.0 / st["conic_opacity"][0, 0]
    assert abs(lam - s2) / s2 < 2e-2  # off-axis Jacobian terms are second order
    assert st["radii"][0] == math.ceil(3 * math.sqrt(max(1.0 / st["conic_opacity"][0, 0], 1.0 / st["conic_opacity"][0, 2])) - 1e-4) or \
        st["radii"][0] == math.ceil(3 * math.sqrt(s2))
    a = min(0.99, 0.8)
    got = fw["color"][:, py, px]
    want = a * np.array([0.2, 0.5, 0.9]) + (1 - a) * np.array([0.1, 0.2, 0.3])
    assert np.abs(got - want).max() < 2e-5
    assert abs(fw["image"]["final_T"][py, px] - (1 - a)) < 1e-6
    assert fw["image"]["n_contrib"][py, px] == 1
    # far corner: only background
    assert np.abs(fw["color"][:, 0, 0] - np.array([0.1, 0.2, 0.3])).max() < 1e-7


def test_near_plane_cull_at_0p2(oracle):
    for zc, expect in [(0.2, False), (0.2001, True), (0.1, False), (-1.0, False)]:
        sc, cam, f = _single(oracle, pos=(0, 0, zc), sigma=0.001, z=0.0)  # z_view == zc exactly
        st = oracle.preprocess(sc)
        assert (st["radii"][0] > 0) == expect, zc
        assert oracle.mark_visible(sc.means3D, sc.viewmatrix)[0] == expect
        if not expect:
            assert st["tiles_touched"][0] == 0


def test_two_gaussians_blend_is_depth_ordered(oracle):
    W = H = 32
    f = 500.0 * W / 512.0
    cam = Camera(np.eye(3), np.array([0.0, 0.0, 3.0]), focal2fov(f, W), focal2fov(f, H), W, H)
    x = ((2 * 16 + 1) / W - 1) * (W / (2 * f))
    cols = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    res = []
    for zs in [(0.0, 0.5), (0.5, 0.0)]:
        means = np.array([[x * (3 + zs[0]), x * (3 + zs[0]), zs[0]], [x * (3 + zs[1]), x * (3 + zs[1]), zs[1]]], np.float32)
        sc = oracle.Scene(W, H, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), np.zeros(3),
                          cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy(),
                          means, np.array([0.6, 0.6], np.float32), colors_precomp=cols,
                          scales=np.full((2, 3), 0.3, np.float32), rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (2, 1)))
        res.append(oracle.forward(sc)["color"][:, 16, 16])
    # front Gaussian dominates: red in the first arrangement, green in the second
    assert res[0][0] > res[0][1] and res[1][1] > res[1][0]
    assert abs(res[0][0] - 0.6) < 1e-2 and abs(res[0][1] - 0.4 * 0.6) < 1e-2


@pytest.mark.parametrize("n,W,H", [(1500, 96, 64), (400, 40, 40)])
def test_structural_invariants(oracle, n, W, H):
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=3, seed=3)
    sc = helpers.oracle_scene(cloud, cam, bg=(0.2, 0.1, 0.0))
    fw = oracle.forward(sc)
    st, bn, im = fw["geom"], fw["binning"], fw["image"]
    D = bn["D"]
    assert D == int(st["tiles_touched"].astype(np.int64).sum()) == len(bn["point_list"])
    keys = bn["keys"]
    assert (np.diff(keys.astype(np.uint64)) >= 0).all() if D > 1 else True
    # ranges partition [0, D) over the non-empty tiles
    r = bn["ranges"]
    nz = r[:, 1] > r[:, 0]
    assert int((r[nz, 1] - r[nz, 0]).sum()) == D
    starts = np.sort(r[nz, 0])
    ends = np.sort(r[nz, 1])
    assert starts[0] == 0 and ends[-1] == D and (starts[1:] == ends[:-1]).all()
    # within a tile: depth non-decreasing, ties in ascending Gaussian index
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    depth_bits = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    same = tiles[1:] == tiles[:-1]
    tie = same & (depth_bits[1:] == depth_bits[:-1])
    assert (bn["point_list"][1:][tie] > bn["point_list"][:-1][tie]).all()
    assert (st["depths"][bn["point_list"]].view(np.uint32) == depth_bits).all()
    # culled => radii 0, tiles 0
    culled = st["radii"] == 0
    assert (st["tiles_touched"][culled] == 0).all()
    # image - T*bg >= 0
    bgv = np.array([0.2, 0.1, 0.0], np.float32)[:, None, None]
    assert (fw["color"] - im["final_T"][None] * bgv >= -1e-6).all()
    # n_contrib never exceeds the tile's list length
    gx = (W + 15) // 16
    for ty in range((H + 15) // 16):
        for tx in range(gx):
            t = ty * gx + tx
            blk = im["n_contrib"][ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
            assert blk.max() <= r[t, 1] - r[t, 0]


def test_opacity_render_is_one_minus_T_for_black_bg(oracle):
    cloud, cam = helpers.cloud_and_camera(800, 64, 64, seed=5)
    ones = torch.ones(cloud.num, 3)
    sc = helpers.oracle_scene(cloud, cam, color_mode="precomp", colors=ones)
    fw = oracle.forward(sc)
    assert np.abs(fw["color"][0] - (1 - fw["image"]["final_T"])).max() < 2e-6


def test_permutation_invariance(oracle):
    cloud, cam = helpers.cloud_and_camera(700, 64, 48, seed=7)
    sc = helpers.oracle_scene(cloud, cam)
    a = oracle.forward(sc)["color"]
    perm = torch.randperm(cloud.num, generator=torch.Generator().manual_seed(1))
    from gsplat_mi355.scenes import GaussianCloud
    c2 = GaussianCloud(cloud.xyz[perm], cloud.scales[perm], cloud.rotations[perm], cloud.opacity[perm], cloud.shs[perm],
                       cloud.sh_degree)
    b = oracle.forward(helpers.oracle_scene(c2, cam))["color"]
    assert np.abs(a - b).max() < 1e-6  # random depths: no ties, order fully determined


def _dense_grads(sc, cloud, cam, gimg, color_mode, cov_mode, bg, scale_modifier=1.0, colors=None):
    from oracle import dense_ref
    dt = torch.float64
    t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dt)
    leaves = {}

    def leaf(name, arr):
        v = t(arr).clone().requires_grad_(True)
        leaves[name] = v
        return v
    means3D = leaf("means3D", sc.means3D)
    means2D = leaf("means2D", np.zeros((sc.P, 3)))
    opac = leaf("opacities", sc.opacities.reshape(-1, 1))
    kw = {}
    if color_mode == "sh":
        kw["shs"] = leaf("sh", sc.shs)
        kw["sh_degree"] = sc.sh_degree
    else:
        kw["colors_precomp"] = leaf("colors_precomp", sc.colors_precomp)
    if cov_mode == "scale_rot":
        kw["scales"] = leaf("scales", sc.scales)
        kw["rotations"] = leaf("rotations", sc.rotations)
    else:
        kw["cov3D_precomp"] = leaf("cov3D_precomp", sc.cov3D_precomp)
    color, radii, aux = dense_ref.render(sc.W, sc.H, sc.tanfovx, sc.tanfovy, t(sc.bg), t(sc.viewmatrix), t(sc.projmatrix),
                                         t(sc.campos), means3D, means2D, opac, scale_modifier=scale_modifier, **kw)
    (color * t(gimg)).sum().backward()
    return color.detach().numpy(), radii.numpy(), {k: v.grad.numpy() for k, v in leaves.items()}, aux


@pytest.mark.parametrize("color_mode,cov_mode,deg,bg", [
    ("sh", "scale_rot", 3, (0.0, 0.0, 0.0)),
    ("sh", "scale_rot", 1, (0.3, 0.6, 0.1)),
    ("precomp", "cov", 3, (1.0, 1.0, 1.0)),
    ("sh", "cov", 2, (0.0, 0.0, 0.0)),
    ("precomp", "scale_rot", 0, (0.2, 0.2, 0.2)),
])
def test_backward_matches_float64_autograd(oracle, color_mode, cov_mode, deg, bg):
    n, W, H = 300, 48, 32
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=11, scale_mul=1.5)
    # push SH so that some colours clamp at 0 and move a few points off-axis past the 1.3*tanfov clamp
    cloud.shs[:, 0] -= 1.2 * (torch.arange(n) % 7 == 0).float()[:, None]
    cloud.xyz[::13, 0] *= 2.4
    sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode=color_mode, cov_mode=cov_mode)
    fw = oracle.forward(sc)
    g = torch.Generator().manual_seed(2)
    gimg = torch.randn(3, H, W, generator=g).numpy().astype(np.float32)
    gr = oracle.backward(sc, fw, gimg)
    color, radii, dg, aux = _dense_grads(sc, cloud, cam, gimg, color_mode, cov_mode, bg)
    assert np.array_equal(radii, fw["radii"])
    assert np.abs(color - fw["color"]).max() < 5e-6
    assert (fw["geom"]["clamped"].sum() > 0) or color_mode != "sh"
    for name in dg:
        ref = dg[name]
        got = gr[name].reshape(ref.shape)
        err = helpers.rel_to_max(got, ref)
        assert err < 2e-4, (name, err)
        assert np.abs(ref).max() > 0, name


def test_backward_scale_modifier_semantics(oracle):
    """dL/dscale is taken w.r.t. mod*scale (the reference kernel drops the factor; SURVEY A8 vi)."""
    n, W, H = 150, 32, 32
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=0, seed=4, scale_mul=1.2)
    sc = helpers.oracle_scene(cloud, cam, scale_modifier=0.7)
    fw = oracle.forward(sc)
    gimg = np.ones((3, H, W), np.float32)
    gr = oracle.backward(sc, fw, gimg)
    _, _, dg, _ = _dense_grads(sc, cloud, cam, gimg, "sh", "scale_rot", (0, 0, 0), scale_modifier=0.7)
    assert helpers.rel_to_max(gr["scales"], dg["scales"]) < 2e-4
    assert helpers.rel_to_max(gr["rotations"], dg["rotations"]) < 2e-4


def test_culled_gaussians_get_exact_zero_grads(oracle):
    n, W, H = 200, 32, 32
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=9)
    cloud.xyz[:40, 2] = -3.5  # behind the camera plane z_view <= 0.2
    sc = helpers.oracle_scene(cloud, cam)
    fw = oracle.forward(sc)
    assert (fw["radii"][:40] == 0).all()
    gr = oracle.backward(sc, fw, np.ones((3, H, W), np.float32))
    for k in ("means3D", "means2D", "sh", "opacities", "scales", "rotations", "cov3D_precomp", "colors_precomp"):
        assert (gr[k][:40] == 0).all(), k


def test_config0_plumbing_10k_256_sh0(oracle):
    """BASELINE.json configs[0]: 10k Gaussians, 256x256, SH degree 0, CPU only."""
    from scipy.spatial import cKDTree

    def d2(p):
        pts = p.numpy().astype(np.float64)
        d, _ = cKDTree(pts).query(pts, k=4)
        return torch.from_numpy((d[:, 1:] ** 2).mean(1).astype(np.float32))
    cloud, cam = helpers.cloud_and_camera(10000, 256, 256, sh_degree=0, seed=0, dist2_fn=d2)
    sc = helpers.oracle_scene(cloud, cam)
    fw = oracle.forward(sc)
    assert fw["color"].shape == (3, 256, 256) and np.isfinite(fw["color"]).all()
    assert (fw["radii"] > 0).sum() > 9000
    gt = torch.rand(3, 256, 256, generator=torch.Generator().manual_seed(1)).numpy()
    gimg = (np.sign(fw["color"] - gt) / gt.size).astype(np.float32)  # d l1_loss / d image (utils/loss_utils.py:21-22)
    gr = oracle.backward(sc, fw, gimg)
    assert all(np.isfinite(v).all() for v in gr.values() if v is not None)
    assert np.abs(gr["means2D"][:, :2]).max() > 0


def test_dist2_brute_force_vs_kdtree(oracle):
    from scipy.spatial import cKDTree
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(3000, 3, generator=g) * 2 - 1).numpy().astype(np.float32)
    got = oracle.dist2(pts)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(1)
    assert np.abs(got - want).max() / want.max() < 1e-5
    # duplicated points: the twin is at distance 0 (self excluded by index only)
    pts2 = np.concatenate([pts[:10], pts[:10], pts[:10], pts[:10]])
    assert (oracle.dist2(pts2) == 0).all()


def test_l1_loss_closed_forms(oracle):
    """N2 oracle (utils/loss_utils.py:21-22): mean |x - y| and its autograd gradient sign(x - y) / n."""
    x = np.array([[0.5, 0.25, 1.0], [0.0, 0.75, 0.125]], np.float32)
    y = np.array([[0.25, 0.25, 0.0], [0.5, 0.5, 0.125]], np.float32)
    v, g = oracle.l1_loss(x, y)
    assert v == pytest.approx((0.25 + 0 + 1.0 + 0.5 + 0.25 + 0) / 6, rel=1e-12)
    inv = np.float32(1.0) / np.float32(6)
    assert np.array_equal(g, np.array([[inv, 0, inv], [-inv, inv, 0]], np.float32))
    # against torch's own formula and autograd on the CPU
    import torch
    rng = np.random.default_rng(3)
    a = rng.random((3, 37, 41), dtype=np.float32)
    b = rng.random((3, 37, 41), dtype=np.float32)
    b[0, :5] = a[0, :5]  # exact zeros exercise sign(0) = 0
    ta = torch.from_numpy(a).requires_grad_(True)
    loss = torch.abs(ta - torch.from_numpy(b)).mean()
    loss.backward()
    v, g = oracle.l1_loss(a, b)
    assert v == pytest.approx(float(loss), rel=1e-6)
    assert np.array_equal(g, ta.grad.numpy())


def test_ssim_oracle_vs_float64_autograd_and_closed_forms(oracle):
    """N2 oracle (utils/loss_utils.py:27-67): identical images give exactly 1 with zero gradient; value and
    gradient agree with an independent float64 conv2d + autograd restatement; ragged sizes (zero padding)."""
    rng = np.random.default_rng(11)
    a = rng.random((3, 23, 31), dtype=np.float32)
    v, g = oracle.ssim(a, a)
    assert v == pytest.approx(1.0, abs=1e-12)
    assert np.abs(g).max() < 1e-9
    for shape in [(3, 40, 37), (1, 11, 64), (3, 5, 7)]:
        a = rng.random(shape, dtype=np.float32)
        b = np.clip(a + 0.15 * rng.standard_normal(shape).astype(np.float32), 0, 1)
        v, g = oracle.ssim(a, b)
        v64, g64 = helpers.ssim_float64(a, b)
        assert v == pytest.approx(v64, abs=2e-7)
        assert np.abs(g - g64).max() <= 5e-6 * np.abs(g64).max()
    # a constant offset lowers the luminance term only: ssim < 1 and finite gradients
    v, g = oracle.ssim(np.full((1, 16, 16), 0.25, np.float32), np.full((1, 16, 16), 0.75, np.float32))
    assert 0.0 < v < 1.0 and np.isfinite(g).all()


def _n3_inputs(n, seed, matrix):
    rng = np.random.default_rng(seed)
    scaling = np.exp(rng.normal(-3.0, 0.7, (n, 3))).astype(np.float32)
    q = rng.normal(size=(n, 4)).astype(np.float32) * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)  # not unit
    if matrix:
        # a rotation composed with a bone rotation, as rigid.py:229-230 builds rotation_precomp
        import torch
        qq = q / np.linalg.norm(q, axis=1, keepdims=True)
        w, x, y, z = qq.T
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                      1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                      1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3).astype(np.float32)
        rot = R
    else:
        rot = q
    g6 = rng.normal(size=(n, 6)).astype(np.float32)
    return scaling, rot, g6


@pytest.mark.parametrize("matrix", [False, True])
def test_build_covariance_oracle_vs_float64_autograd(oracle, matrix):
    """N3 oracle: strip_symmetric(L L^T) and the gradients autograd derives (upper-triangle read), for quaternion
    and rotation_precomp inputs; closed form for the identity rotation."""
    scaling, rot, g6 = _n3_inputs(300, 5, matrix)
    cov, ds, dr = oracle.build_covariance(scaling, 1.7, rot, g6)
    cov64, ds64, dr64 = helpers.covariance_float64(scaling, 1.7, rot, g6)
    assert np.abs(cov - cov64).max() <= 1e-6 * np.abs(cov64).max()
    assert np.abs(ds - ds64).max() <= 1e-6 * np.abs(ds64).max()
    assert np.abs(dr - dr64).max() <= 1e-6 * np.abs(dr64).max()
    ident = np.tile(np.array([[2.0, 0, 0, 0]], np.float32), (4, 1))  # unnormalised identity quaternion
    s = np.array([[1, 2, 3]] * 4, np.float32)
    c = oracle.build_covariance(s, 0.5, ident)
    assert np.allclose(c, [[0.25, 0, 0, 1.0, 0, 2.25]] * 4)


def test_sh2rgb_oracle_vs_reference_golden_and_float64_autograd(oracle):
    """N3 oracle: pinned by tests/golden/sh_eval.npz (the reference's own eval_sh + 0.5 + clamp, degrees 0..3) for
    unit directions; the canonical-frame rotation, the view noise, the +1e-12 normalisation and all gradients
    against a float64 autograd restatement."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sh_eval.npz"))
    feats = np.ascontiguousarray(g["sh"].transpose(0, 2, 1)).astype(np.float32)  # reference [N, C, M] -> get_features [N, M, 3]
    dirs = g["dirs"].astype(np.float32)
    for deg in range(4):
        col, cl = oracle.sh2rgb(feats, 3.0 * dirs, np.zeros(3, np.float32), deg)
        want = g["color_deg%d" % deg]
        assert np.abs(col - want).max() <= 2e-6
        for c in range(3):
            assert np.array_equal((cl >> c) & 1, ((g["eval_deg%d" % deg][:, c] + 0.5) < 0).astype(np.uint8))
    rng = np.random.default_rng(21)
    n = 200
    feats = (0.5 * rng.normal(size=(n, 16, 3))).astype(np.float32)
    xyz = rng.normal(size=(n, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, 4.0], np.float32)
    _, R, _ = _n3_inputs(n, 8, True)
    th = 0.3
    noise = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
    gcol = rng.normal(size=(n, 3)).astype(np.float32)
    for deg, rot, nz in [(3, R, noise), (2, R, None), (1, None, None), (0, None, noise)]:
        col, cl, dsh, dp = oracle.sh2rgb(feats, xyz, campos, deg, rot, nz, gcol)
        col64, dsh64, dp64 = helpers.sh2rgb_float64(feats, xyz, campos, deg, rot, nz, gcol)
        assert np.abs(col - col64).max() <= 1e-6
        assert np.abs(dsh - dsh64).max() <= 1e-6 * max(np.abs(dsh64).max(), 1e-30)
        assert np.abs(dp - dp64).max() <= 1e-6 * max(np.abs(dp64).max(), 1e-30)


def test_knn_points_oracle_vs_scipy_kdtree(oracle):
    """N4 oracle: K nearest (self included) against scipy's cKDTree; ties by index on duplicated points."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(700, 3)).astype(np.float32)
    d, ix = oracle.knn_points(pts, pts, 6)
    dd, ii = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=6)
    assert np.array_equal(ix, ii)
    assert np.allclose(d, dd ** 2, rtol=1e-5, atol=1e-7)
    assert np.all(ix[:, 0] == np.arange(700)) and np.all(d[:, 0] == 0)
    dup = np.concatenate([pts[:5], pts[:5], pts[5:50]])
    d, ix = oracle.knn_points(dup, dup, 2)
    assert np.array_equal(ix[:5], np.stack([np.arange(5), np.arange(5) + 5], 1))  # equal distance 0: smaller index first
    q = rng.normal(size=(40, 3)).astype(np.float32)
    d1, i1 = oracle.knn_points(q, pts, 1)
    _, j1 = cKDTree(pts.astype(np.float64)).query(q.astype(np.float64), k=1)
    assert np.array_equal(i1[:, 0], j1)


def test_adam_oracle_vs_torch_optim_adam(oracle):
    """N4 oracle: the float64 Adam restatement reproduces torch.optim.Adam(l, lr=0.0, eps=1e-15) as
    scene/gaussian_model.py:201-216 builds it (per-group learning rates), over several steps, on the CPU."""
    import torch
    rng = np.random.default_rng(4)
    shapes, lrs = [(50, 3), (50, 1, 3), (50, 15, 3), (50, 1)], [1.6e-4, 2.5e-3, 1.25e-4, 5e-2]
    params = [torch.nn.Parameter(torch.from_numpy(rng.normal(size=s)).double()) for s in shapes]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], lr=0.0, eps=1e-15)
    mine = [(p.detach().numpy().copy(), np.zeros(s), np.zeros(s)) for p, s in zip(params, shapes)]
    for step in range(1, 6):
        grads = [rng.normal(size=s) * (step % 2 + 0.1) for s in shapes]
        grads[0][::3] = 0.0  # culled Gaussians have zero gradients: Adam still moves them with the old moments
        for p, g in zip(params, grads):
            p.grad = torch.from_numpy(g)
        opt.step()
        mine = [oracle.adam_step(p, g, m, v, lr, 0.9, 0.999, 1e-15, step) for (p, m, v), g, lr in zip(mine, grads, lrs)]
        for p, (q, m, v) in zip(params, mine):
            assert np.allclose(p.detach().numpy(), q, rtol=1e-12, atol=1e-15)
            assert np.allclose(opt.state[p]["exp_avg"].numpy(), m, rtol=1e-12, atol=1e-18)
            assert np.allclose(opt.state[p]["exp_avg_sq"].numpy(), v, rtol=1e-12, atol=1e-18)


def test_snug_tile_rectangles_change_no_output(oracle):
    """GsFwdArgs.tile_rect = 1 (bounding box of the alpha >= 1/255 region instead of upstream's 3-sigma square): in the
    oracle the rectangle is a subset of the square, far fewer pairs are emitted, and colour, radii, final_T and EVERY
    gradient are BITWISE those of the square -- the tiles left out contribute exactly nothing."""
    n, W, H = 6000, 200, 150
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=2, seed=41, scale_mul=1.4)
    cloud.opacity[::5] *= 0.05   # low opacities shrink the box further
    cloud.opacity[::37] = 0.003  # below 1/255: visible (radius > 0) but binned nowhere
    out = {}
    for mode in (0, 1):
        sc = helpers.oracle_scene(cloud, cam, bg=(0.2, 0.4, 0.6), tile_rect=mode)
        fw = oracle.forward(sc)
        g = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).numpy()
        out[mode] = (fw, oracle.backward(sc, fw, g))
    (f0, b0), (f1, b1) = out[0], out[1]
    r0, r1 = f0["geom"]["rect"], f1["geom"]["rect"]
    vis = f0["radii"] > 0
    assert np.array_equal(f0["radii"], f1["radii"])
    assert (r1[vis, 0] >= r0[vis, 0]).all() and (r1[vis, 1] >= r0[vis, 1]).all()
    assert (r1[vis, 2] <= r0[vis, 2]).all() and (r1[vis, 3] <= r0[vis, 3]).all()
    assert f1["binning"]["D"] < 0.8 * f0["binning"]["D"]
    assert (f1["geom"]["tiles_touched"][(cloud.opacity.numpy().reshape(-1) < 1 / 255) & vis] == 0).all()
    assert np.array_equal(f0["color"], f1["color"])
    assert np.array_equal(f0["image"]["final_T"], f1["image"]["final_T"])
    for k in b0:
        if b0[k] is not None and isinstance(b0[k], np.ndarray):
            assert np.array_equal(b0[k], b1[k]), k


def test_snug_tile_rectangles_stay_conservative_for_needle_gaussians(oracle):
    """Long thin Gaussians (sigma_major ~100-470 px at 2048^2, thin axis at the low-pass floor): the per-pixel alpha
    test runs on the rounded fp32 conic, whose determinant cancels, so its alpha >= 1/255 region reaches beyond the
    exact ellipse's bounding box (by ~15 px at 450 px).  The box is widened by the conditioning bound
    (gs_math.h: snug_half_widths), so colour and final_T stay BITWISE those of upstream's squares."""
    n, W, H = 200, 2048, 2048
    for seed in (0, 1):
        cloud, cam = helpers.needle_cloud_and_camera(n, W, H, seed=seed)
        out = {}
        for mode in (0, 1):
            out[mode] = oracle.forward(helpers.oracle_scene(cloud, cam, bg=(0.1, 0.2, 0.3), tile_rect=mode))
        assert out[0]["radii"].max() > 1000
        assert out[1]["binning"]["D"] < 0.6 * out[0]["binning"]["D"]
        assert np.array_equal(out[0]["color"], out[1]["color"])
        assert np.array_equal(out[0]["image"]["final_T"], out[1]["image"]["final_T"])
