"""Self-oracle dumps: every intermediate of A4-A8 of oracle/gs_oracle.c for small scenes, one per
input combination of the rasterizer API, COMMITTED -- so that an edit of the oracle that moves any number it produces is
noticed (tests/test_oracle_golden.py::test_oracle_still_produces_its_committed_dumps).  These are outputs of THIS
repository's oracle, not of the reference (which holds no rasterizer source): they pin the checker against drift, not
against upstream.  Inputs are stored with the outputs, so the test does not depend on any random-number generator.

Run:  python tests/golden/make_self_dumps.py   (writes tests/golden/self_oracle.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "3dgs-avatar-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = [  # (name, colour inputs, covariance inputs, SH degree, tile_rect, background)
    ("sh_scale_rot_deg3", "sh", "scale_rot", 3, 0, (0.0, 0.0, 0.0)),
    ("precomp_cov", "precomp", "cov", 3, 0, (0.25, 0.5, 0.75)),
    ("sh_cov_deg1", "sh", "cov", 1, 0, (0.1, 0.3, 0.2)),
    ("precomp_scale_rot", "precomp", "scale_rot", 0, 0, (0.0, 0.0, 0.0)),
    ("sh_scale_rot_deg2_snug_rect", "sh", "scale_rot", 2, 1, (0.2, 0.1, 0.4)),
]
SCENE_FIELDS = ("bg", "viewmatrix", "projmatrix", "campos", "means3D", "opacities", "shs", "colors_precomp", "scales",
                "rotations", "cov3D_precomp")


def scene_of(d, name):
    from oracle import gs_oracle
    g = lambda k: d["%s/in/%s" % (name, k)] if ("%s/in/%s" % (name, k)) in d else None
    meta = d[name + "/in/meta"]  # W, H, sh_degree, tile_rect
    tan = d[name + "/in/tan"]
    return gs_oracle.Scene(int(meta[0]), int(meta[1]), float(tan[0]), float(tan[1]), g("bg"), g("viewmatrix"), g("projmatrix"),
                           g("campos"), g("means3D"), g("opacities"), shs=g("shs"), colors_precomp=g("colors_precomp"),
                           scales=g("scales"), rotations=g("rotations"), cov3D_precomp=g("cov3D_precomp"),
                           sh_degree=int(meta[2]), tile_rect=int(meta[3]))


def outputs_of(oracle, sc, gimg):
    fw = oracle.forward(sc, margin=True)
    bw = oracle.backward(sc, fw, gimg)
    out = {}
    for k in ("depths", "radii", "xy", "conic_opacity", "rgb", "clamped", "cov3D", "tiles_touched", "rect"):
        out["geom/" + k] = fw["geom"][k]
    for k in ("offsets", "keys", "point_list", "ranges"):
        out["binning/" + k] = fw["binning"][k]
    out["binning/D"] = np.array(fw["binning"]["D"])
    for k in ("color", "final_T", "n_contrib", "n_blended"):
        out["image/" + k] = fw["image"][k]
    for k, v in bw.items():
        if v is not None:
            out["grad/" + k] = v
    return out


def main():
    import helpers
    from oracle import gs_oracle
    gs_oracle.build()
    d = {}
    for ci, (name, color_mode, cov_mode, deg, tile_rect, bg) in enumerate(CASES):
        n, W, H = 900, 64, 48
        cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=deg, seed=100 + ci, scale_mul=1.4)
        cloud.xyz[:20, 2] = -3.5    # behind the near plane: culled
        cloud.xyz[20:30, 0] *= 3.0  # beyond the 1.3 tan(fov) clamp
        cloud.shs[:, 0] -= 1.2 * (torch.arange(n) % 5 == 0).float()[:, None]  # some colours below zero: the clamp flags
        sc = helpers.oracle_scene(cloud, cam, bg=bg, color_mode=color_mode, cov_mode=cov_mode, tile_rect=tile_rect)
        for k in SCENE_FIELDS:
            v = getattr(sc, k)
            if v is not None:
                d["%s/in/%s" % (name, k)] = v
        d[name + "/in/meta"] = np.array([W, H, deg, tile_rect])
        d[name + "/in/tan"] = np.array([sc.tanfovx, sc.tanfovy], np.float64)
        gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(7 + ci)).numpy()
        d[name + "/in/dL_dpix"] = gimg
        sc2 = scene_of(d, name)  # (what the test will build: from the stored arrays)
        for k, v in outputs_of(gs_oracle, sc2, gimg).items():
            d["%s/out/%s" % (name, k)] = v
    np.savez_compressed(os.path.join(HERE, "self_oracle.npz"), **d)
    print("wrote self_oracle.npz: %d arrays, %d cases" % (len(d), len(CASES)))


if __name__ == "__main__":
    main()
