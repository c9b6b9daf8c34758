"""ctypes front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker / reported baseline.  The shipped package never imports this module.
PARITY UNPINNED (see the header of gs_oracle.c): the reference tree holds neither source nor
golden vectors for the rasterizer; the pieces it does pin (SH polynomial, camera matrices) are
checked against fixtures in tests/golden/.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GS_ORACLE_SO: another build of the same source (the sanitizer build of `make SAN=1`: tests/test_oracle_sanitized.py)
_SO = os.environ.get("GS_ORACLE_SO") or os.path.join(_HERE, "_build", "libgs_oracle.so")


class OrArgs(ctypes.Structure):
    _fields_ = [
        ("P", c_int), ("deg", c_int), ("M", c_int), ("W", c_int), ("H", c_int),
        ("bg", c_void_p), ("means3D", c_void_p), ("shs", c_void_p), ("colors_precomp", c_void_p),
        ("opacities", c_void_p), ("scales", c_void_p), ("rotations", c_void_p),
        ("cov3D_precomp", c_void_p), ("viewmatrix", c_void_p), ("projmatrix", c_void_p),
        ("campos", c_void_p),
        ("scale_modifier", c_float), ("tanfovx", c_float), ("tanfovy", c_float),
        ("prefiltered", c_int), ("tile_rect", c_int),
    ]


def build(force=False):
    """Compile oracle/gs_oracle.c with gcc (strict fp32, OpenMP)."""
    src = os.path.join(_HERE, "gs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.or_scan.restype = c_int64
    return _lib


def _f32(x):
    return None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class Scene(object):
    """Plain container of the rasterizer inputs as float32 numpy arrays (None = absent, mirroring
    the reference wrapper's empty-tensor convention, gaussian_renderer/__init__.py:107-129)."""

    def __init__(self, W, H, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos, means3D, opacities,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 sh_degree=0, scale_modifier=1.0, prefiltered=False, tile_rect=0):
        self.W, self.H = int(W), int(H)
        self.tanfovx, self.tanfovy = float(tanfovx), float(tanfovy)
        self.bg = _f32(bg).reshape(3)
        self.viewmatrix = _f32(viewmatrix).reshape(16)
        self.projmatrix = _f32(projmatrix).reshape(16)
        self.campos = _f32(campos).reshape(3)
        self.means3D = _f32(means3D).reshape(-1, 3)
        self.P = self.means3D.shape[0]
        self.opacities = _f32(opacities).reshape(self.P)
        self.shs = _f32(shs)
        self.colors_precomp = _f32(colors_precomp)
        self.scales = _f32(scales)
        self.rotations = _f32(rotations)
        self.cov3D_precomp = _f32(cov3D_precomp)
        self.sh_degree = int(sh_degree)
        self.M = 0 if self.shs is None else int(self.shs.shape[1])
        self.scale_modifier = float(scale_modifier)
        self.prefiltered = bool(prefiltered)
        self.tile_rect = int(tile_rect)  # 0 = upstream's 3-sigma square, 1 = bounding box of the alpha >= 1/255 region
        if (self.shs is None) == (self.colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((self.scales is None or self.rotations is None) and self.cov3D_precomp is None) or \
                ((self.scales is not None or self.rotations is not None) and self.cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")

    def cargs(self):
        a = OrArgs()
        a.P, a.deg, a.M, a.W, a.H = self.P, self.sh_degree, self.M, self.W, self.H
        a.bg, a.means3D, a.shs = _ptr(self.bg), _ptr(self.means3D), _ptr(self.shs)
        a.colors_precomp, a.opacities = _ptr(self.colors_precomp), _ptr(self.opacities)
        a.scales, a.rotations, a.cov3D_precomp = _ptr(self.scales), _ptr(self.rotations), _ptr(self.cov3D_precomp)
        a.viewmatrix, a.projmatrix, a.campos = _ptr(self.viewmatrix), _ptr(self.projmatrix), _ptr(self.campos)
        a.scale_modifier, a.tanfovx, a.tanfovy = self.scale_modifier, self.tanfovx, self.tanfovy
        a.prefiltered = int(self.prefiltered)
        a.tile_rect = int(self.tile_rect)
        return a


def preprocess(sc):
    P = sc.P
    st = dict(
        depths=np.zeros(P, np.float32), radii=np.zeros(P, np.int32), xy=np.zeros((P, 2), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
        clamped=np.zeros((P, 3), np.uint8), cov3D=np.zeros((P, 6), np.float32),
        tiles_touched=np.zeros(P, np.uint32), rect=np.zeros((P, 4), np.int32))
    a = sc.cargs()
    rc = lib().or_preprocess(ctypes.byref(a), _ptr(st["depths"]), _ptr(st["radii"]), _ptr(st["xy"]),
                             _ptr(st["conic_opacity"]), _ptr(st["rgb"]), _ptr(st["clamped"]),
                             _ptr(st["cov3D"]), _ptr(st["tiles_touched"]), _ptr(st["rect"]))
    assert rc == 0
    return st


def binning(sc, st):
    """A5: inclusive scan, duplicateWithKeys, stable sort, identifyTileRanges."""
    P = sc.P
    L = lib()
    offsets = np.zeros(P, np.uint32)
    D = int(L.or_scan(c_int(P), _ptr(st["tiles_touched"]), _ptr(offsets)))
    keys_u = np.zeros(max(D, 1), np.uint64)
    vals_u = np.zeros(max(D, 1), np.uint32)
    L.or_duplicate_with_keys(c_int(P), c_int(sc.W), _ptr(st["depths"]), _ptr(st["radii"]), _ptr(st["rect"]),
                             _ptr(offsets), _ptr(keys_u), _ptr(vals_u))
    keys = np.zeros(max(D, 1), np.uint64)
    vals = np.zeros(max(D, 1), np.uint32)
    rc = L.or_sort_pairs(c_int64(D), _ptr(keys_u), _ptr(vals_u), _ptr(keys), _ptr(vals))
    assert rc == 0
    gx, gy = (sc.W + 15) // 16, (sc.H + 15) // 16
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.or_tile_ranges(c_int64(D), _ptr(keys), c_int(gx * gy), _ptr(ranges))
    return dict(offsets=offsets, D=D, keys_unsorted=keys_u[:D], vals_unsorted=vals_u[:D],
                keys=keys[:D], point_list=vals[:D], ranges=ranges)


class Overrides(object):
    """Decisions of the compositing loop forced to a given outcome (gs_oracle.c, "threshold decisions"): `key` =
    pixel << 32 | index into the point list (uint64, ascending), `act` = OV_SKIP 1 / OV_KEEP 2 / OV_STOP 4 / OV_GO 8,
    `margin` = the relative distance of each overridden decision from its threshold."""

    def __init__(self, key, act, margin):
        order = np.argsort(np.asarray(key, np.uint64), kind="stable")
        self.key = np.ascontiguousarray(np.asarray(key, np.uint64)[order])
        self.act = np.ascontiguousarray(np.asarray(act, np.uint8)[order])
        self.margin = np.ascontiguousarray(np.asarray(margin, np.float32)[order])
        assert self.key.size == self.act.size and (np.diff(self.key.astype(np.int64)) > 0).all()

    def __len__(self):
        return int(self.key.size)

    def cargs(self):
        n = len(self)
        return c_int(n), (_ptr(self.key) if n else None), (_ptr(self.act) if n else None)


_NO_OVERRIDES = (c_int(0), None, None)


def render_forward(sc, st, bn, overrides=None, margin=False):
    """A6.  `overrides`: an Overrides table (or None); `margin`: also return, per pixel, the smallest relative margin
    of any threshold decision the walk took."""
    W, H = sc.W, sc.H
    out = np.zeros((3, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    mg = np.zeros((H, W), np.float32) if margin else None
    nb = np.zeros((H, W), np.uint32) if margin else None
    pl = np.ascontiguousarray(bn["point_list"]) if bn["D"] > 0 else np.zeros(1, np.uint32)
    if overrides is None and not margin:
        rc = lib().or_render_forward(c_int(W), c_int(H), _ptr(bn["ranges"]), _ptr(pl), _ptr(st["xy"]),
                                     _ptr(st["conic_opacity"]), _ptr(st["rgb"]), _ptr(sc.bg),
                                     _ptr(out), _ptr(final_T), _ptr(n_contrib))
    else:
        ov = overrides.cargs() if overrides is not None else _NO_OVERRIDES
        rc = lib().or_render_forward_ex(c_int(W), c_int(H), _ptr(bn["ranges"]), _ptr(pl), _ptr(st["xy"]),
                                        _ptr(st["conic_opacity"]), _ptr(st["rgb"]), _ptr(sc.bg), ov[0], ov[1], ov[2],
                                        _ptr(out), _ptr(final_T), _ptr(n_contrib), _ptr(mg), _ptr(nb))
    assert rc == 0
    im = dict(color=out, final_T=final_T, n_contrib=n_contrib)
    if margin:
        im["margin"] = mg
        im["n_blended"] = nb  # pairs composited per pixel (alpha >= 1/255, before the pixel was done)
    return im


def forward(sc, overrides=None, margin=False):
    """Full forward: returns (color[3,H,W], radii[P]) plus every intermediate."""
    st = preprocess(sc)
    bn = binning(sc, st)
    im = render_forward(sc, st, bn, overrides, margin)
    return dict(geom=st, binning=bn, image=im, color=im["color"], radii=st["radii"])


# relative margins within which a decision may be flipped to explain a device result: |power| / (size of its terms),
# |255 alpha - 1|, |1e4 T (1 - alpha) - 1|.  The device evaluates alpha in the log2 domain with FMA contraction and
# v_exp_f32 (a few 1e-6 relative); T is a product of up to hundreds of factors (1 - alpha), each off by that much.
EXPLAIN_EPS = (1e-5, 2e-5, 2e-4)


def explain_pixels(sc, fw, pids, dev_color, dev_final_T, dev_n_contrib, eps=EXPLAIN_EPS, tol_c=None, tol_T=1e-4, max_flips=3):
    """Attribution of device/oracle differences to threshold decisions (or_explain_pixels).  `pids`: flat pixel
    indices; dev_*: the device's full images.  Returns (status[len(pids)] -- flips used, -1 = NOT explained --, Overrides)."""
    W, H = sc.W, sc.H
    st, bn = fw["geom"], fw["binning"]
    pids = np.ascontiguousarray(np.asarray(pids, np.uint32).reshape(-1))
    n = int(pids.size)
    dc = np.ascontiguousarray(np.asarray(dev_color, np.float32).reshape(3, -1)[:, pids])
    dT = np.ascontiguousarray(np.asarray(dev_final_T, np.float32).reshape(-1)[pids])
    dl = np.ascontiguousarray(np.asarray(dev_n_contrib, np.uint32).reshape(-1)[pids])
    if tol_c is None:
        tol_c = 1e-5 * max(float(np.abs(fw["color"]).max()), 1e-30)
    cap = max(8 * n, 8)
    key, act, mg = np.zeros(cap, np.uint64), np.zeros(cap, np.uint8), np.zeros(cap, np.float32)
    status = np.zeros(max(n, 1), np.int32)
    e = np.ascontiguousarray(np.asarray(eps, np.float32))
    pl = np.ascontiguousarray(bn["point_list"]) if bn["D"] > 0 else np.zeros(1, np.uint32)
    m = lib().or_explain_pixels(c_int(W), c_int(H), _ptr(bn["ranges"]), _ptr(pl), _ptr(st["xy"]), _ptr(st["conic_opacity"]),
                                _ptr(st["rgb"]), _ptr(sc.bg), c_int(n), _ptr(pids), _ptr(dc), _ptr(dT), _ptr(dl), _ptr(e),
                                c_float(tol_c), c_float(tol_T), c_int(max_flips), c_int(cap), _ptr(key), _ptr(act), _ptr(mg),
                                _ptr(status))
    assert m >= 0
    return status[:n], Overrides(key[:m], act[:m], mg[:m])


def backward(sc, fw, dL_dpix, overrides=None):
    """Full backward for dL/dcolor = dL_dpix[3,H,W]; returns the eight gradient tensors of the
    reference wrapper (A3) plus the per-Gaussian 2-D intermediates.  `overrides`: the table the forward `fw` was
    rendered with (its final_T / n_contrib already carry the stop decisions)."""
    P, W, H = sc.P, sc.W, sc.H
    st, bn, im = fw["geom"], fw["binning"], fw["image"]
    g = np.ascontiguousarray(np.asarray(dL_dpix, np.float32).reshape(3, H, W))
    d_mean2D = np.zeros((P, 2), np.float64)
    d_conic = np.zeros((P, 3), np.float64)
    d_op = np.zeros(P, np.float64)
    d_col = np.zeros((P, 3), np.float64)
    pl = np.ascontiguousarray(bn["point_list"]) if bn["D"] > 0 else np.zeros(1, np.uint32)
    ov = overrides.cargs() if overrides is not None else _NO_OVERRIDES
    rc = lib().or_render_backward_ex(c_int(P), c_int(W), c_int(H), c_int64(bn["D"]), _ptr(bn["ranges"]), _ptr(pl),
                                     _ptr(st["xy"]), _ptr(st["conic_opacity"]), _ptr(st["rgb"]), _ptr(sc.bg),
                                     _ptr(im["final_T"]), _ptr(im["n_contrib"]), _ptr(g), ov[0], ov[1], ov[2],
                                     _ptr(d_mean2D), _ptr(d_conic), _ptr(d_op), _ptr(d_col))
    assert rc == 0
    m2 = d_mean2D.astype(np.float32)
    cn = d_conic.astype(np.float32)
    cl = d_col.astype(np.float32)
    d_means3D = np.zeros((P, 3), np.float32)
    d_cov3D = np.zeros((P, 6), np.float32)
    d_sh = np.zeros((P, sc.M, 3), np.float32) if sc.shs is not None else None
    d_scale = np.zeros((P, 3), np.float32) if sc.scales is not None else None
    d_rot = np.zeros((P, 4), np.float32) if sc.scales is not None else None
    a = sc.cargs()
    rc = lib().or_preprocess_backward(ctypes.byref(a), _ptr(st["radii"]), _ptr(st["cov3D"]), _ptr(st["clamped"]),
                                      _ptr(m2), _ptr(cn), _ptr(cl), _ptr(d_means3D), _ptr(d_sh),
                                      _ptr(d_cov3D), _ptr(d_scale), _ptr(d_rot))
    assert rc == 0
    d_means2D = np.zeros((P, 3), np.float32)
    d_means2D[:, :2] = m2
    return dict(means3D=d_means3D, means2D=d_means2D, sh=d_sh, colors_precomp=cl,
                opacities=d_op.astype(np.float32).reshape(P, 1), scales=d_scale, rotations=d_rot,
                cov3D_precomp=d_cov3D, conic=cn)


def mark_visible(means3D, viewmatrix):
    m = _f32(means3D).reshape(-1, 3)
    v = _f32(viewmatrix).reshape(16)
    out = np.zeros(m.shape[0], np.uint8)
    lib().or_mark_visible(c_int(m.shape[0]), _ptr(m), _ptr(v), _ptr(out))
    return out.astype(bool)


def dist2(points):
    """A9 restatement: exact brute-force mean squared distance to the 3 nearest other points."""
    p = _f32(points).reshape(-1, 3)
    out = np.zeros(p.shape[0], np.float32)
    lib().or_dist2(c_int(p.shape[0]), _ptr(p), _ptr(out))
    return out


def l1_loss(x, y):
    """N2 restatement: (mean |x - y| as float64, d/dx = sign(x - y) / n as fp32)."""
    a, b = _f32(x).reshape(-1), _f32(y).reshape(-1)
    assert a.size == b.size and a.size > 0
    g = np.zeros(a.size, np.float32)
    fn = lib().or_l1_loss
    fn.restype = ctypes.c_double
    v = fn(ctypes.c_longlong(a.size), _ptr(a), _ptr(b), _ptr(g))
    return float(v), g.reshape(np.shape(x))


def ssim(img1, img2):
    """N2 restatement: (mean SSIM of two (C,H,W) images as float64, its gradient w.r.t. img1 as fp32)."""
    a, b = _f32(img1), _f32(img2)
    assert a.shape == b.shape and a.ndim == 3
    g = np.zeros(a.shape, np.float32)
    fn = lib().or_ssim
    fn.restype = ctypes.c_double
    v = fn(c_int(a.shape[0]), c_int(a.shape[1]), c_int(a.shape[2]), _ptr(a), _ptr(b), _ptr(g))
    return float(v), g


def build_covariance(scaling, modifier, rotation, dL_dcov6=None):
    """N3 restatement (scene/gaussian_model.py:28-32): cov6, and with dL_dcov6 also (d/dscaling, d/drotation)."""
    sc = _f32(scaling).reshape(-1, 3)
    rot = _f32(rotation)
    is_matrix = rot.shape[-1] != 4
    n = sc.shape[0]
    cov = np.zeros((n, 6), np.float32)
    if dL_dcov6 is None:
        lib().or_build_covariance(c_int(n), _ptr(sc), c_float(modifier), _ptr(rot), c_int(int(is_matrix)), _ptr(cov), None, None, None)
        return cov
    g = _f32(dL_dcov6).reshape(n, 6)
    ds, dr = np.zeros_like(sc), np.zeros_like(rot)
    lib().or_build_covariance(c_int(n), _ptr(sc), c_float(modifier), _ptr(rot), c_int(int(is_matrix)), _ptr(cov), _ptr(g), _ptr(ds), _ptr(dr))
    return cov, ds, dr


def sh2rgb(features, xyz, campos, deg, fwd_rot=None, noise=None, dL_dcolors=None):
    """N3 restatement (models/texture/texture.py:21-38): colours (+ clamped mask), and with dL_dcolors also
    (d/dfeatures, d/dxyz).  features is (N, M, 3)."""
    sh = _f32(features)
    n, m = sh.shape[0], sh.shape[1]
    p = _f32(xyz).reshape(n, 3)
    cp = _f32(campos).reshape(3)
    R = _f32(fwd_rot).reshape(n, 9) if fwd_rot is not None else None
    nz = _f32(noise).reshape(9) if noise is not None else None
    col = np.zeros((n, 3), np.float32)
    cl = np.zeros(n, np.uint8)
    if dL_dcolors is None:
        lib().or_sh2rgb(c_int(n), c_int(deg), c_int(m), _ptr(sh), _ptr(p), _ptr(cp), _ptr(R) if R is not None else None,
                        _ptr(nz) if nz is not None else None, _ptr(col), _ptr(cl), None, None, None)
        return col, cl
    g = _f32(dL_dcolors).reshape(n, 3)
    dsh, dp = np.zeros_like(sh), np.zeros_like(p)
    lib().or_sh2rgb(c_int(n), c_int(deg), c_int(m), _ptr(sh), _ptr(p), _ptr(cp), _ptr(R) if R is not None else None,
                    _ptr(nz) if nz is not None else None, _ptr(col), _ptr(cl), _ptr(g), _ptr(dsh), _ptr(dp))
    return col, cl, dsh, dp


def knn_points(queries, ref, K):
    """N4 restatement: (squared distances (Nq,K) ascending, indices (Nq,K) int64), brute force."""
    q, r = _f32(queries).reshape(-1, 3), _f32(ref).reshape(-1, 3)
    d = np.zeros((q.shape[0], K), np.float32)
    ix = np.zeros((q.shape[0], K), np.int64)
    lib().or_knn_points(c_int(q.shape[0]), _ptr(q), c_int(r.shape[0]), _ptr(r), c_int(K), _ptr(d), _ptr(ix))
    return d, ix


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """N4 restatement of one torch.optim.Adam step (no weight decay / amsgrad) in float64 numpy; returns the new
    (param, exp_avg, exp_avg_sq)."""
    p, g, m, v = (np.asarray(a, np.float64) for a in (param, grad, exp_avg, exp_avg_sq))
    m = m + (g - m) * (1.0 - beta1)
    v = v * beta2 + (1.0 - beta2) * g * g
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    p = p - (lr / bc1) * (m / (np.sqrt(v) / np.sqrt(bc2) + eps))
    return p, m, v


def densify_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """N4 restatement of train.py:219-220 + scene/gaussian_model.py:464-466; returns the three updated arrays."""
    vis = np.asarray(radii) > 0
    mr, acc, dn = (np.array(a, np.float32).reshape(-1) for a in (max_radii2D, xyz_gradient_accum, denom))
    g = np.asarray(viewspace_grad, np.float32)
    mr[vis] = np.maximum(mr[vis], np.asarray(radii)[vis].astype(np.float32))
    acc[vis] += np.sqrt(g[vis, 0] * g[vis, 0] + g[vis, 1] * g[vis, 1])
    dn[vis] += 1
    return mr, acc, dn


def set_num_threads(n):
    lib().or_set_num_threads(c_int(int(n)))


def num_threads():
    return int(lib().or_num_threads())
