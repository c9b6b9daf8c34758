"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle SAN=1`): every entry point of
oracle/gs_oracle.c runs once on a small seeded scene in a child process that preloads the sanitizer runtime; any
out-of-bounds access, use of uninitialised heap state the sanitizers can see, signed overflow or misaligned access ends
the child with a non-zero exit code.  (GPU sanitizers are not available on this pool: the HIP kernels are covered by the
parity tests against this oracle, the oracle by the sanitizers.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import math, os, sys
import numpy as np
sys.path.insert(0, os.environ["GS_ROOT"])
from oracle import gs_oracle as o
assert o._SO.endswith("libgs_oracle_san.so")
rng = np.random.default_rng(3)
for (P, W, H, deg, use_cov, use_col, rect) in [(700, 80, 56, 3, False, False, 0), (300, 33, 47, 1, True, True, 1), (5, 16, 16, 0, False, True, 1)]:
    M = (deg + 1) ** 2
    xyz = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    scales = (0.05 * np.exp(rng.normal(0, 0.3, (P, 3)))).astype(np.float32)
    q = rng.normal(size=(P, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = rng.uniform(0.02, 0.9, (P, 1)).astype(np.float32)
    shs = (rng.normal(0, 0.3, (P, M, 3))).astype(np.float32)
    f = 60.0
    tanx, tany = W / (2 * f), H / (2 * f)
    view = np.eye(4, dtype=np.float32); view[3, 2] = 3.0  # row-vector convention: translation in the last row
    zn, zf = 0.01, 100.0
    Pm = np.zeros((4, 4), np.float32)
    Pm[0, 0], Pm[1, 1], Pm[3, 2], Pm[2, 2], Pm[2, 3] = 1 / tanx, 1 / tany, 1.0, zf / (zf - zn), -(zf * zn) / (zf - zn)
    proj = view @ Pm.T
    kw = {}
    if use_col: kw["colors_precomp"] = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    else: kw.update(shs=shs, sh_degree=deg)
    if use_cov: kw["cov3D_precomp"] = o.build_covariance(scales, 1.0, q)
    else: kw.update(scales=scales, rotations=q)
    sc = o.Scene(W, H, tanx, tany, np.array([0.1, 0.2, 0.3], np.float32), view, proj, np.array([0, 0, -3], np.float32), xyz, op,
                 tile_rect=rect, **kw)
    fw = o.forward(sc)
    bw = o.backward(sc, fw, rng.normal(size=(3, H, W)).astype(np.float32))
    assert np.isfinite(fw["color"]).all() and all(np.isfinite(v).all() for v in bw.values() if v is not None)
    assert fw["binning"]["D"] == int(fw["geom"]["tiles_touched"].sum())
    o.mark_visible(xyz, view)
    o.dist2(xyz)
    o.knn_points(xyz[: P // 2], xyz, min(4, P - 1))
    img = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    o.l1_loss(fw["color"], img)
    o.ssim(fw["color"], img)
    o.build_covariance(scales, 1.3, q, rng.normal(size=(P, 6)).astype(np.float32))
    o.sh2rgb(shs, xyz, np.array([0, 0, -3], np.float32), deg, dL_dcolors=rng.normal(size=(P, 3)).astype(np.float32))
print("sanitized oracle ok")
'''


def test_oracle_entry_points_under_asan_and_ubsan(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "SAN=1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    so = os.path.join(ROOT, "oracle", "_build", "libgs_oracle_san.so")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "gcc's libasan.so not found: %r" % asan
    env = dict(os.environ, GS_ORACLE_SO=so, GS_ROOT=ROOT, LD_PRELOAD=asan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
