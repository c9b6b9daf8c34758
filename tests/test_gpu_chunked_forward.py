"""The chunk-parallel forward of long tile lists (gs_tuning "fwd4" = 2; render_fwd.hip: render_chunk) through the C ABI.
Opt-in in round 4 (the default stays the four-wave kernel), so the default `-m gpu` run covers it here: the premise its
work hand-out is numbered on, its state as the debug fields show it, and the parity tests of the long-list frames, of the
shared-geometry second render and of a hipGraph replay once more with the switch on -- the same assertions (integers exact,
every image / gradient difference against the oracle attributed to threshold decisions, bitwise repeatability, the second
render's bits equal to a stand-alone render's)."""
import numpy as np
import pytest
import torch

import helpers
import test_gpu_capture as tc
import test_gpu_parity as tp

pytestmark = pytest.mark.gpu


@pytest.fixture
def chunked():
    import os
    from gsplat_mi355 import _lib
    _lib.tuning("fwd4", 2)
    try:
        yield
    finally:
        _lib.tuning("fwd4", int(os.environ.get("GSPLAT_FWD4", "1")))  # (the process's own setting: a whole-suite run with the switch on stays on)


def test_workgroups_are_dealt_round_robin_over_the_xcds():
    """Worker b of the chunk-parallel forward takes item b >> 3 of the list of the XCD it runs on: for that numbering to hand
    out every item exactly once, the eight workgroups 8 r .. 8 r + 7 of a launch must run on eight different XCDs.  On an
    MI355X in SPX mode the dispatcher deals workgroups round-robin, starting wherever the previous launch stopped:
    xcc[b] = (b + first) % 8.  The kernel reads the XCD and claims its item, so a different deal would cost speed and leave
    items to the sweep, never a wrong image -- but it is what the numbering is built on, so it is checked, over a few
    launches with different starting points."""
    from gsplat_mi355 import _lib
    firsts = set()
    for n in (4096, 13, 4096, 1027, 4096):
        x = _lib.xcc_probe(torch.device("cuda:0"), n).numpy()
        assert np.array_equal(x, (np.arange(n) + int(x[0])) % 8), x[:16]
        firsts.add(int(x[0]))
    assert set(_lib.xcc_probe(torch.device("cuda:0"), 64).numpy().tolist()) == set(range(8))


def test_the_long_tiles_are_cut_into_chunks_and_every_chunk_reports(oracle, chunked):
    """The work list as the debug fields show it on a long-list frame: units = sum over the marked tiles of
    ceil(list / entries per chunk), every marked tile of at least two chunks, its units consecutive and in chunk order; every
    (unit, quadrant) has published its hits; the hits of a quadrant's live chunks add up to at least its count up to the last
    contributor; the items per XCD add up to 4 x units; and two renders of the frame are bitwise identical."""
    import ctypes
    from gsplat_mi355 import _lib, debug
    dev = torch.device("cuda:0")
    n, W, H = 30000, 160, 160
    cloud, cam = helpers.cloud_and_camera(n, W, H, sh_degree=1, seed=5, layout="body")
    cloud.opacity = cloud.opacity * 0.25
    bg = (0.1, 0.2, 0.3)
    args = (tp._settings(cam, cloud, bg, dev), cloud.xyz.to(dev), cloud.opacity.to(dev))
    kw = tp._inputs(cloud, cam, "sh", "scale_rot", dev)
    st = debug.forward_state(*args, **kw)
    st2 = debug.forward_state(*args, **kw)
    for k in ("n_contrib", "final_T", "qcount"):
        assert np.array_equal(st["image"][k], st2["image"][k]), k
    assert np.array_equal(st["color"].view(np.uint32), st2["color"].view(np.uint32))
    cw = st["image"]["chunks"]
    units, ch = int(cw["hdr"][0]), int(cw["hdr"][1])
    assert units > 0 and ch >= 256 and ch & (ch - 1) == 0
    assert int(cw["hdr"][4:12].sum()) == 4 * units
    r = st["image"]["ranges"].astype(np.int64)
    lens = r[:, 1] - r[:, 0]
    order = st["image"]["order"]
    marked = np.sort(order[order >> 31 == 1] & 0x7FFFFFFF)
    assert marked.size > 0 and units == int(np.ceil(lens[marked] / ch).sum())
    tile, c, nch = cw["units"][:, 0], cw["units"][:, 1] & 0xFFFF, cw["units"][:, 1] >> 16
    assert np.array_equal(np.sort(np.unique(tile)), marked) and int(nch.min()) >= 2
    first = np.nonzero(c == 0)[0]
    for u0 in first:
        k = int(nch[u0])
        assert np.array_equal(tile[u0:u0 + k], np.full(k, tile[u0])) and np.array_equal(c[u0:u0 + k], np.arange(k))
        assert np.array_equal(nch[u0:u0 + k], np.full(k, k))
    flags = cw["flags"].reshape(units, 4)
    assert (flags != 0).all()  # hits + 1
    hits = (flags & 0x7FFFFFFF).astype(np.int64) - 1
    dead = flags >> 31
    qc = st["image"]["qcount"].astype(np.int64)
    for u0 in first:
        k = int(nch[u0])
        live = np.where(dead[u0:u0 + k] == 0, hits[u0:u0 + k], 0).sum(axis=0)
        assert (live >= qc[tile[u0]]).all(), (int(tile[u0]), live, qc[tile[u0]])
        assert (np.diff(dead[u0:u0 + k].astype(np.int64), axis=0) >= 0).all()  # dead chunks come last: T only falls


@pytest.mark.parametrize("with_opacity", [False, True])
def test_long_lists_small_image_chunked_forward_and_backward(oracle, chunked, with_opacity):
    tp.test_long_lists_on_a_small_image_backward_in_chunks(oracle, with_opacity)


@pytest.mark.parametrize("fused_backward", [False, True])
def test_shared_geometry_second_render_chunked(oracle, chunked, fused_backward, monkeypatch):
    tp.test_shared_geometry_second_render_is_bitwise_identical(oracle, fused_backward, monkeypatch)


def test_second_render_with_arbitrary_colours_chunked(oracle, chunked, monkeypatch):
    tp.test_second_render_with_arbitrary_constant_colours_one_backward_for_both_images(oracle, monkeypatch, "body", 30000, 256, 256)


def test_full_size_avatar_frame_chunked(oracle, chunked):
    tp.test_full_size_trained_avatar_shaped_frame_200k_512(oracle, 1)


def test_graph_replay_chunked(chunked, monkeypatch):
    tc.test_render_and_training_step_replayed_from_a_graph_are_bit_identical_to_the_eager_runs(monkeypatch)
