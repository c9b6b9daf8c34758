"""The oracle's threshold-decision machinery (oracle/gs_oracle.c, "threshold decisions"): per-pixel margins, override
tables and the search that attributes a device result to flipped decisions -- checked here on the CPU against a
SIMULATED device: the oracle itself with a known set of decisions flipped.  tests/test_gpu_parity.py uses the same
machinery against the HIP path to turn "outliers are threshold flips" into a checked statement."""
import numpy as np
import pytest

import helpers


def _pixel_decisions(fw, sc, pid):
    """Plain-numpy walk of one pixel (A6), returning per decision (list index j, kind, relative margin, natural outcome).
    Only used to PICK decisions to flip; the oracle's own C walk is what is under test."""
    W = sc.W
    st, bn = fw["geom"], fw["binning"]
    px, py = np.float32(pid % W), np.float32(pid // W)
    tile = (pid // W // 16) * ((W + 15) // 16) + (pid % W) // 16
    r0, r1 = (int(v) for v in bn["ranges"][tile])
    f = np.float32
    T = f(1.0)
    out = []
    for j in range(r0, r1):
        g = int(bn["point_list"][j])
        dx, dy = st["xy"][g, 0] - px, st["xy"][g, 1] - py
        A, B, C, o = (f(v) for v in st["conic_opacity"][g])
        power = f(-0.5) * (A * dx * dx + C * dy * dy) - B * dx * dy
        terms = f(0.5) * (abs(A) * dx * dx + abs(C) * dy * dy) + abs(B * dx * dy)
        if terms > 0:
            out.append((j, "power", float(abs(power) / terms), bool(power > 0)))
        if power > 0:
            continue
        alpha = min(f(0.99), o * f(np.exp(power)))
        out.append((j, "alpha", float(abs(alpha * f(255.0) - f(1.0))), bool(alpha < f(1.0) / f(255.0))))
        if alpha < f(1.0) / f(255.0):
            continue
        test_T = T * (f(1.0) - alpha)
        out.append((j, "T", float(abs(test_T * f(10000.0) - f(1.0))), bool(test_T < f(0.0001))))
        if test_T < f(0.0001):
            break
        T = test_T
    return out


@pytest.fixture(scope="module")
def scene(oracle):
    cloud, cam = helpers.cloud_and_camera(6000, 192, 144, sh_degree=2, seed=3, scale_mul=1.6)
    cloud.opacity = (cloud.opacity * 3.0).clamp(max=0.97)  # opaque enough for pixels to reach the T threshold
    sc = helpers.oracle_scene(cloud, cam, bg=(0.2, 0.3, 0.1))
    return sc, oracle.forward(sc, margin=True)


def test_margins_and_plain_forward(oracle, scene):
    sc, fw = scene
    plain = oracle.forward(sc)
    for k in ("color", "final_T", "n_contrib"):  # asking for margins changes nothing
        assert np.array_equal(plain["image"][k], fw["image"][k])
    mg = fw["image"]["margin"]
    assert mg.shape == (sc.H, sc.W) and (mg >= 0).all() and np.isfinite(mg).mean() > 0.9
    # spot check against the numpy walk: the smallest margin of the pixel's decisions
    rng = np.random.default_rng(0)
    for pid in rng.choice(sc.W * sc.H, 12, replace=False):
        dec = _pixel_decisions(fw, sc, int(pid))
        if dec:
            assert mg.reshape(-1)[pid] == pytest.approx(min(d[2] for d in dec), rel=1e-3, abs=1e-7)


def test_flipped_decisions_are_found_and_the_conditioned_oracle_reproduces_the_device(oracle, scene):
    sc, fw = scene
    eps = (2e-3, 2e-3, 2e-3)
    mg = fw["image"]["margin"].reshape(-1)
    picks = np.argsort(mg)[:40]
    keys, acts, kinds = [], [], []
    for pid in picks:
        dec = [d for d in _pixel_decisions(fw, sc, int(pid)) if d[2] <= eps[0] * 0.5]
        if not dec:
            continue
        j, kind, margin, outcome = min(dec, key=lambda d: d[2])
        act = {"power": 2 if outcome else 1, "alpha": 2 if outcome else 1, "T": 8 if outcome else 4}[kind]
        keys.append((int(pid) << 32) | j)
        acts.append(act)
        kinds.append(kind)
    assert len(keys) >= 10 and {"alpha", "T"} <= set(kinds)
    truth = oracle.Overrides(keys, acts, np.zeros(len(keys), np.float32))
    dev = oracle.forward(sc, overrides=truth)["image"]  # the simulated device
    changed = np.flatnonzero((dev["n_contrib"] != fw["image"]["n_contrib"]).reshape(-1) |
                             (np.abs(dev["final_T"] - fw["image"]["final_T"]).reshape(-1) > 1e-4 * fw["image"]["final_T"].reshape(-1)))
    assert set(changed) == set(int(k) >> 32 for k in keys)  # every flip is visible in final_T or n_contrib, nothing else moved
    status, found = oracle.explain_pixels(sc, fw, changed, dev["color"], dev["final_T"], dev["n_contrib"], eps=eps)
    assert (status == 1).all()
    assert np.array_equal(found.key, truth.key) and np.array_equal(found.act, truth.act)
    assert (found.margin <= eps[0]).all() and (found.margin > 0).any()
    cond = oracle.forward(sc, overrides=found)["image"]
    for k in ("color", "final_T", "n_contrib"):
        assert np.array_equal(cond[k], dev[k]), k
    # a pixel that matches without flips: status 0, no override
    same = np.setdiff1d(np.arange(sc.W * sc.H), changed)[:5]
    status0, none = oracle.explain_pixels(sc, fw, same, dev["color"], dev["final_T"], dev["n_contrib"], eps=eps)
    assert (status0 == 0).all() and len(none) == 0
    # a difference that is NOT a threshold flip is reported as unexplained
    bad = dev["color"].copy()
    bad.reshape(3, -1)[1, changed[0]] += 0.01
    status_bad, _ = oracle.explain_pixels(sc, fw, changed[:1], bad, dev["final_T"], dev["n_contrib"], eps=eps)
    assert status_bad[0] == -1
    # ... and so is a real flip when the margins allowed are too tight to reach it
    tight = tuple(float(found.margin.min()) * 0.5 for _ in range(3))
    status_tight, _ = oracle.explain_pixels(sc, fw, changed, dev["color"], dev["final_T"], dev["n_contrib"], eps=tight)
    assert (status_tight == -1).all()

    # backward: the overridden decisions move exactly the Gaussians of the overridden pixels' lists
    g = np.random.default_rng(1).standard_normal((3, sc.H, sc.W)).astype(np.float32)
    fwc = dict(fw, image=cond, color=cond["color"])
    b0, b1 = oracle.backward(sc, fw, g), oracle.backward(sc, fwc, g, found)
    touched = np.zeros(sc.P, bool)
    gx = (sc.W + 15) // 16
    for pid in changed:
        t = (pid // sc.W // 16) * gx + (pid % sc.W) // 16
        r0, r1 = fw["binning"]["ranges"][t]
        touched[fw["binning"]["point_list"][r0:r1]] = True
    moved = np.abs(b0["opacities"][:, 0] - b1["opacities"][:, 0]) > 0
    assert moved.any() and not (moved & ~touched).any()
    assert np.isfinite(b1["means3D"]).all()
