"""Multi-process path on CPU with the gloo backend, world_size 2: one broadcast of the
packed Gaussian state, frames partitioned round-robin, no per-step collective.  The renderer used
here is the CPU oracle (test infrastructure); on GPUs the same functions run over RCCL (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from gsplat_mi355.scenes import GaussianCloud
from gsplat_mi355.sharding import broadcast_cloud, frames_of_rank, render_sequence


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, frames, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gs_oracle
        gs_oracle.set_num_threads(2)
        cloud = None
        if rank == 0:
            cloud, _ = helpers.cloud_and_camera(n, 64, 48, sh_degree=2, seed=5)
        cloud = broadcast_cloud(cloud, n, 2, torch.device("cpu"), src=0)
        assert isinstance(cloud, GaussianCloud) and cloud.num == n
        cams = [helpers.orbit_camera(f, 64, 48, dtheta=0.05) for f in range(frames)]

        def render_fn(cam, pc):
            return gs_oracle.forward(helpers.oracle_scene(pc, cam))["color"]
        out = render_sequence(cloud, cams, render_fn, rank, world)
        assert sorted(out) == frames_of_rank(rank, world, total=frames)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{"f%d" % k: v for k, v in out.items()},
                 packed=cloud.pack().numpy())
    finally:
        dist.destroy_process_group()


def test_frames_partition():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += frames_of_rank(r, world, total=300)
        assert sorted(seen) == list(range(300))
        sizes = [len(frames_of_rank(r, world, total=300)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert frames_of_rank(3, 8, num_frames_per_rank=4) == [3, 11, 19, 27]


def test_pack_unpack_roundtrip():
    cloud, _ = helpers.cloud_and_camera(100, 32, 32, sh_degree=3, seed=1)
    flat = cloud.pack()
    assert flat.numel() == GaussianCloud.packed_numel(100, 3) == 100 * 59  # 236 B per Gaussian
    c2 = GaussianCloud.unpack(flat, 100, 3)
    for f in GaussianCloud.FIELDS:
        assert torch.equal(getattr(cloud, f), getattr(c2, f))


def test_two_rank_gloo_broadcast_and_sharded_render(tmp_path, oracle):
    n, frames, world = 300, 5, 2
    mp.spawn(_worker, args=(world, _free_port(), n, frames, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["packed"], r1["packed"])  # identical Gaussian state on both ranks
    cloud, _ = helpers.cloud_and_camera(n, 64, 48, sh_degree=2, seed=5)
    for f in range(frames):
        src = r0 if f % 2 == 0 else r1
        assert ("f%d" % f) in src.files
        want = oracle.forward(helpers.oracle_scene(cloud, helpers.orbit_camera(f, 64, 48, dtheta=0.05)))["color"]
        assert np.array_equal(src["f%d" % f], want)
