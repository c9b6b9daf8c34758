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
