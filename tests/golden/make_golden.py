"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (only possible in the build
container, where /root/reference exists; the GPU box and the test-suite only read the .npz files).

  - sh_eval.npz     : utils/sh_utils.py eval_sh / RGB2SH on seeded inputs, degrees 0..3, plus the
                      "+0.5, clamp_min(0)" colour of models/texture/texture.py:35-37
  - cameras.npz     : utils/graphics_utils.py getWorld2View2 / getProjectionMatrix / focal2fov assembled
                      by the recipe of scene/cameras.py:35-40 for the benchmark cameras

  - losses.npz      : utils/loss_utils.py l1_loss / ssim (with gaussian, create_window, _ssim) on seeded CPU images, in
                      fp32 (the precision the reference runs them in) and fp64, values and autograd gradients w.r.t.
                      the first image.  The module itself does not import here (cv2 / pytorch3d at module level, which
                      those five functions do not use), so ONLY those function definitions are taken from the file's
                      syntax tree and executed, with the names they need (torch, F, Variable, exp) in scope; nothing of
                      the reference's text is stored.

  - prepass.npz     : (round 4) the per-Gaussian chains either side of the rasterizer, executed from the reference's own
                      definitions on seeded CPU inputs: build_rotation / build_scaling_rotation / strip_symmetric
                      (utils/general_utils.py:73-108,194-207) and build_covariance_from_scaling_rotation
                      (scene/gaussian_model.py:28-32) for quaternion AND 3x3 `rotation_precomp` inputs
                      (models/deformer/rigid.py:225-232), values and autograd gradients; SH2RGB.forward
                      (models/texture/texture.py:21-38) with and without cano_view_dir / view noise, values and
                      gradients; add_densification_stats (scene/gaussian_model.py:464-466) and the max_radii2D
                      statement of train.py:219; the scale rule of create_from_pcd (scene/gaussian_model.py:186-187)
                      on brute-force neighbour distances.  Those definitions allocate on "cuda": they are executed
                      with a `torch` name in scope that maps the device to the CPU (nothing else is changed); only
                      inputs and outputs are stored.

Run:  python tests/golden/make_golden.py
"""
import ast
import importlib.util
import math
import os

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_functions(rel, names, scope):
    """Executes only the named top-level function definitions of a reference file (its module-level imports may need
    packages that are absent here and that these functions do not use)."""
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), [n.name for n in picked]
    ns = dict(scope)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


class _CpuTorch(object):
    """`torch` as the executed reference functions see it: the factory functions they call with device="cuda" allocate
    on the CPU instead; every other attribute is torch's own."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def _cpu(kw):
        if "device" in kw:
            kw = dict(kw, device="cpu")
        return kw

    def zeros(self, *a, **kw):
        return torch.zeros(*a, **self._cpu(kw))

    def ones(self, *a, **kw):
        return torch.ones(*a, **self._cpu(kw))

    def tensor(self, *a, **kw):
        return torch.tensor(*a, **self._cpu(kw))


def _method(rel, cls, name):
    """The FunctionDef of method `name` of class `cls` in a reference file (to be executed as a plain function)."""
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and m.name == name:
                    return m, path
    raise KeyError((rel, cls, name))


def _exec_nodes(nodes, path, scope):
    ns = dict(scope)
    exec(compile(ast.Module(body=list(nodes), type_ignores=[]), path, "exec"), ns)
    return ns


class _Stub(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_prepass():
    T = _CpuTorch()
    out = {}
    g = torch.Generator().manual_seed(777)
    # ---- covariance chain
    ns = _load_functions("utils/general_utils.py", ["strip_lowerdiag", "strip_symmetric", "build_rotation", "build_scaling_rotation"],
                         dict(torch=T))
    ns2 = _load_functions("scene/gaussian_model.py", ["build_covariance_from_scaling_rotation"],
                          dict(torch=T, build_scaling_rotation=ns["build_scaling_rotation"], strip_symmetric=ns["strip_symmetric"]))
    build_cov = ns2["build_covariance_from_scaling_rotation"]
    n = 96
    scaling = torch.exp(torch.randn(n, 3, generator=g) * 0.7 - 3.0)
    quat = torch.randn(n, 4, generator=g)            # NOT normalised: build_rotation normalises
    quat[:8] *= torch.logspace(-3, 3, 8)[:, None]    # ... whatever the norm
    quat_unit = quat / quat.norm(dim=1, keepdim=True)
    # a rotation_precomp as the rigid deformer makes it (rigid.py:229-231): T_fwd[:, :3, :3] @ build_rotation(q): a general
    # 3x3 (the blended bone transform is not orthonormal)
    A = torch.eye(3)[None] + 0.2 * torch.randn(n, 3, 3, generator=g)
    rot_precomp = torch.matmul(A, ns["build_rotation"](quat)).contiguous()
    g6 = torch.randn(n, 6, generator=g)
    out.update(cov_scaling=scaling.numpy(), cov_quat=quat.numpy(), cov_quat_unit=quat_unit.numpy(),
               cov_rot_precomp=rot_precomp.numpy(), cov_g6=g6.numpy(), build_rotation=ns["build_rotation"](quat).numpy())
    for tag, rot in (("quat", quat), ("quat_unit", quat_unit), ("matrix", rot_precomp)):
        for mod in (1.0, 0.6):
            for dt_tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
                if dt_tag == "f64":
                    continue  # (the reference's functions allocate float32: fp32 is the only precision they run in)
                s_ = scaling.to(dt).clone().requires_grad_(True)
                r_ = rot.to(dt).clone().requires_grad_(True)
                cov = build_cov(s_, mod, r_)
                (cov * g6.to(dt)).sum().backward()
                k = "cov_%s_mod%g" % (tag, mod)
                out[k] = cov.detach().numpy()
                out[k + "_dscaling"] = s_.grad.numpy()
                out[k + "_drotation"] = r_.grad.numpy()
    # ---- SH2RGB.forward
    sh_utils = _load("ref_sh_utils_p", "utils/sh_utils.py")
    fwd_node, path = _method("models/texture/texture.py", "SH2RGB", "forward")
    fwd = _exec_nodes([fwd_node], path, dict(torch=T, eval_sh=sh_utils.eval_sh, augm_rots=sh_utils.augm_rots))["forward"]
    n = 80
    feats = torch.randn(n, 16, 3, generator=g) * 0.4
    feats[:, 0] += 0.5
    feats[::7, 0] -= 2.5  # some colours below zero: the clamp
    xyz = torch.randn(n, 3, generator=g)
    campos = torch.tensor([0.3, -0.2, 3.0])
    Rf = ns["build_rotation"](torch.randn(n, 4, generator=g))
    Tf = torch.eye(4)[None].repeat(n, 1, 1)
    Tf[:, :3, :3] = torch.matmul(torch.eye(3)[None] + 0.1 * torch.randn(n, 3, 3, generator=g), Rf)
    Tf[:, :3, 3] = torch.randn(n, 3, generator=g)
    gcol = torch.randn(n, 3, generator=g)
    out.update(sh_features=feats.numpy(), sh_xyz=xyz.numpy(), sh_campos=campos.numpy(), sh_fwd_transform=Tf.numpy(), sh_gcol=gcol.numpy())

    class Cfg(dict):
        __getattr__ = dict.get

    for deg in range(4):
        for mode in ("plain", "cano", "cano_noise"):
            M = (deg + 1) ** 2 if deg < 3 else 16
            f_ = feats[:, :M].clone().requires_grad_(True)
            x_ = xyz.clone().requires_grad_(True)
            gs_ = _Stub(get_features=f_, get_xyz=x_, max_sh_degree=int(round(M ** 0.5)) - 1, active_sh_degree=deg, fwd_transform=Tf)
            cfg = Cfg(cano_view_dir=mode != "plain", view_noise=15.0 if mode == "cano_noise" else 0.0)
            me = _Stub(cfg=cfg, training=mode == "cano_noise")
            k = "sh2rgb_deg%d_%s" % (deg, mode)
            if mode == "cano_noise":
                # the noise matrix the forward will draw: the same generator state, drawn once beforehand
                np.random.seed(100 + deg)
                out[k + "_noise"] = np.asarray(sh_utils.augm_rots(15.0, 15.0, 15.0), np.float32).T.copy()  # (`.transpose(0, 1)`)
                np.random.seed(100 + deg)
            col = fwd(me, gs_, _Stub(camera_center=campos))
            (col * gcol).sum().backward()
            out[k] = col.detach().numpy()
            out[k + "_dfeatures"] = f_.grad.numpy()
            out[k + "_dxyz"] = x_.grad.numpy() if x_.grad is not None else np.zeros((n, 3), np.float32)
    # ---- densification statistics: scene/gaussian_model.py:464-466 + the statement of train.py:219
    add_node, path = _method("scene/gaussian_model.py", "GaussianModel", "add_densification_stats")
    add_stats = _exec_nodes([add_node], path, dict(torch=T))["add_densification_stats"]
    tpath = os.path.join(REF, "train.py")
    stmt = [nd for nd in ast.walk(ast.parse(open(tpath).read(), filename=tpath))
            if isinstance(nd, ast.Assign) and isinstance(nd.targets[0], ast.Subscript)
            and isinstance(nd.targets[0].value, ast.Attribute) and nd.targets[0].value.attr == "max_radii2D"]
    assert len(stmt) == 1 and stmt[0].lineno == 219, [x.lineno for x in stmt]
    n = 500
    radii = (torch.rand(n, generator=g) * 40).int() * (torch.rand(n, generator=g) > 0.3).int()
    vgrad = torch.randn(n, 3, generator=g) * 1e-3
    mr0 = (torch.rand(n, generator=g) * 30).floor()
    acc0, den0 = torch.rand(n, 1, generator=g), (torch.rand(n, 1, generator=g) * 5).floor()
    model = _Stub(max_radii2D=mr0.clone(), xyz_gradient_accum=acc0.clone(), denom=den0.clone())
    vis = radii > 0
    _exec_nodes(stmt, tpath, dict(torch=T, gaussians=model, visibility_filter=vis, radii=radii))
    add_stats(model, _Stub(grad=vgrad), vis)
    out.update(st_radii=radii.numpy(), st_vgrad=vgrad.numpy(), st_max_radii2D_in=mr0.numpy(), st_accum_in=acc0.numpy(),
               st_denom_in=den0.numpy(), st_max_radii2D=model.max_radii2D.numpy(), st_accum=model.xyz_gradient_accum.numpy(),
               st_denom=model.denom.numpy())
    # ---- the scale rule of create_from_pcd (scene/gaussian_model.py:186-187) on brute-force neighbour distances
    cnode, path = _method("scene/gaussian_model.py", "GaussianModel", "create_from_pcd")
    rule = [nd for nd in cnode.body if isinstance(nd, ast.Assign) and isinstance(nd.targets[0], ast.Name)
            and nd.targets[0].id in ("dist2", "scales")]
    assert [nd.targets[0].id for nd in rule] == ["dist2", "scales"] and rule[0].lineno == 186
    n = 700
    pts = (torch.rand(n, 3, generator=g) * 2 - 1).numpy().astype(np.float32)
    pts[-60:] = np.tile(pts[:20], (3, 1))  # four copies of 20 points: their three nearest are at distance 0 -> the 1e-7 clamp
    d = ((pts[:, None, :] - pts[None, :, :]) ** 2)
    d2 = (d[..., 0] + d[..., 1]) + d[..., 2]
    np.fill_diagonal(d2, np.inf)
    near = np.sort(d2, axis=1)[:, :3].astype(np.float32)
    mean3 = ((near[:, 0] + near[:, 1]) + near[:, 2]) / np.float32(3.0)
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # (`....float().cuda()` inside the executed statements)
    try:
        nsr = _exec_nodes(rule, path, dict(torch=T, np=np, pcd=_Stub(points=pts), distCUDA2=lambda t: torch.from_numpy(mean3)))
    finally:
        torch.Tensor.cuda = saved_cuda
    out.update(pcd_points=pts, pcd_mean_d2=mean3, pcd_dist2=nsr["dist2"].numpy(), pcd_scales=nsr["scales"].numpy())
    np.savez_compressed(os.path.join(HERE, "prepass.npz"), **out)
    print("wrote prepass.npz (%d arrays)" % len(out))


def make_losses():
    import torch.nn.functional as F
    from math import exp
    from torch.autograd import Variable
    ns = _load_functions("utils/loss_utils.py", ["l1_loss", "gaussian", "create_window", "ssim", "_ssim"],
                         dict(torch=torch, F=F, Variable=Variable, exp=exp))
    out = {}
    g = torch.Generator().manual_seed(4321)
    k = 0
    for shape in [(3, 48, 64), (3, 37, 53), (1, 16, 16)]:
        a = torch.rand(shape, generator=g)
        b = (a + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
        if k == 2:
            b = a.clone()  # identical images: SSIM = 1
        out["img1_%d" % k], out["img2_%d" % k] = a.numpy(), b.numpy()
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            x = a.to(dt).clone().requires_grad_(True)
            v = ns["ssim"](x, b.to(dt))
            v.backward()
            out["ssim_%s_%d" % (tag, k)] = v.detach().numpy()
            out["ssim_grad_%s_%d" % (tag, k)] = x.grad.numpy()
            x = a.to(dt).clone().requires_grad_(True)
            v = ns["l1_loss"](x, b.to(dt))
            v.backward()
            out["l1_%s_%d" % (tag, k)] = v.detach().numpy()
            out["l1_grad_%s_%d" % (tag, k)] = x.grad.numpy()
        k += 1
    out["count"] = np.array(k)
    w = ns["create_window"](11, 3)
    out["window"] = w.numpy()
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("wrote losses.npz")


def main():
    make_losses()
    make_prepass()
    sh_utils = _load("ref_sh_utils", "utils/sh_utils.py")
    gu = _load("ref_graphics_utils", "utils/graphics_utils.py")

    g = torch.Generator().manual_seed(1234)
    n = 64
    sh = torch.randn(n, 3, 16, generator=g, dtype=torch.float64) * 0.5  # reference layout [..., C, coeffs]
    d = torch.randn(n, 3, generator=g, dtype=torch.float64)
    dirs = d / d.norm(dim=1, keepdim=True)
    out = {"sh": sh.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        res = sh_utils.eval_sh(deg, sh, dirs)
        out["eval_deg%d" % deg] = res.numpy()
        out["color_deg%d" % deg] = torch.clamp_min(res + 0.5, 0.0).numpy()
    rgb = torch.rand(n, 3, generator=g, dtype=torch.float64)
    out["rgb"] = rgb.numpy()
    out["rgb2sh"] = sh_utils.RGB2SH(rgb).numpy()
    np.savez(os.path.join(HERE, "sh_eval.npz"), **out)

    cams = {}
    k = 0
    for (W, H) in [(256, 256), (512, 512), (1024, 1024), (640, 360)]:
        for th in [0.0, 0.01, 0.7]:
            c, s = math.cos(th), math.sin(th)
            R = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
            T = np.array([0.1 * k, -0.05 * k, 3.0])
            f = 500.0 * W / 512.0
            fovx, fovy = gu.focal2fov(f, W), gu.focal2fov(f, H)
            wv = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
            pm = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
            full = (wv.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
            center = wv.inverse()[3, :3]
            cams["R_%d" % k] = R
            cams["T_%d" % k] = T
            cams["WH_%d" % k] = np.array([W, H])
            cams["fov_%d" % k] = np.array([fovx, fovy])
            cams["world_view_%d" % k] = wv.numpy()
            cams["proj_%d" % k] = pm.numpy()
            cams["full_proj_%d" % k] = full.numpy()
            cams["center_%d" % k] = center.numpy()
            k += 1
    cams["count"] = np.array(k)
    np.savez(os.path.join(HERE, "cameras.npz"), **cams)
    print("wrote sh_eval.npz, cameras.npz")


if __name__ == "__main__":
    main()
