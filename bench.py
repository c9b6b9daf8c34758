"""bench.py -- headline benchmark of the MI355X-native Gaussian-splat rasterizer.

Metric (BASELINE.json): render fps (fwd+bwd) @ 200k Gaussians, 1024x1024, SH degree 3.
A "step" = one pass of the hot path over one frame: rasterizer forward (SH + scale/rotation inputs,
in-kernel SH->RGB and cov3D), L1 loss against a fixed target (train.py:121 / utils/loss_utils.py:21-22),
rasterizer backward.  All inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: frames of an orbit sequence are sharded over the ranks (rank r renders frames r, r+N, ...), the
reference's independent frame loop (render.py:51-62); the Gaussian state is built on rank 0 and sent with ONE
RCCL broadcast before the timed region; no data-path collective per step.  Weak scaling: every
rank renders K frames.  Without a launcher (`WORLD_SIZE` unset) `--gpus N` starts the N ranks itself, before
anything in this process has touched a GPU.  Rank 0 prints one JSON line.

The line carries two timed legs of the same workload: `value` / `roofline` for the default binning
(tile_rect = 1: a Gaussian is binned into the bounding box of its alpha >= 1/255 region) and `upstream_rect`
for the reference's own binning (tile_rect = 0: the 3-sigma square, whose tile lists, ranges and num_rendered
are upstream's bit for bit).  At N = 1 it also holds `roofline.traffic` (HBM bytes of the dominant kernel) and
`roofline.valu` (issued VALU instructions) from rocprofv3 --pmc passes over a few frames of the same workload,
run as child processes of this command before it touches the GPU itself (--no-pmc skips them).
"""
import argparse
import csv
import gc
import glob
import json
import math
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (spec)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; a wave64 VALU instruction holds its SIMD-32 for 2 cycles

WORKLOADS = {
    # name: (N, W, H, sh_degree, heavy_tail, backward)
    "config2": (50000, 512, 512, 3, 0.0, False),
    "config3": (200000, 1024, 1024, 3, 0.0, True),
    "config4": (200000, 512, 512, 3, 0.0, True),
    "config5": (500000, 2048, 2048, 3, 0.05, True),
    "tiny": (20000, 256, 256, 3, 0.0, True),
    # not a BASELINE configuration: the shape of a TRAINED avatar (200k Gaussians on a thin shell covering a fraction of
    # the 512 x 512 image, mostly opaque) next to config 4's initial-state cloud -- few, long tile lists
    "avatar": (200000, 512, 512, 3, 0.0, True),
    "avatar1k": (200000, 1024, 1024, 3, 0.0, True),  # the same cloud at ZJU-MoCap's native resolution
    "avatar50k": (50000, 512, 512, 3, 0.0, True),    # the reference's initial point count (dataset/zjumocap.py:412)
}
WORKLOAD_LAYOUT = {"avatar": "body", "avatar1k": "body", "avatar50k": "body"}

# kernel (rocprofv3 name) -> bench stage it belongs to
KERNEL_STAGE = {"preprocess_kernel": "preprocess", "render_fwd_kernel": "render_fwd", "render_bwd_kernel": "render_bwd",
                "segment_reduce_kernel": "gaussian_bwd", "gaussian_bwd_kernel": "gaussian_bwd"}


def stage_bytes(N, D, px, sh=True):
    """ALGORITHMIC (compulsory) bytes per launch group, SURVEY.md 8(d): B_in = 236 (SH3 + scale/rot),
    B_state = 79, B_grad = 248."""
    B_in, B_state, B_grad = (236 if sh else 52), 79, 248
    return {
        "preprocess": N * (B_in + 4 + B_state),
        "binning": D * (12 + 12),
        "render_fwd": D * 40 + px * (12 + 8),
        "render_bwd": px * (12 + 8) + D * (4 + 40) + D * 36,
        "gaussian_bwd": N * (B_in + B_state + 40 + B_grad),
    }


def stage_flops(pairs):
    """SURVEY.md 8(d) secondary accounting: ~25 flop + 1 exp per (pixel, Gaussian) pair forward, ~70 + 1 exp
    backward."""
    return {"render_fwd": float(pairs) * 26, "render_bwd": float(pairs) * 71}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--loss", default="l1", choices=["l1", "l1+dssim"],
                    help="l1 = the BASELINE metric's loss; l1+dssim = 0.8 L1 + 0.2 (1 - SSIM), the reference's full image loss")
    ap.add_argument("--l1", default="fused", choices=["fused", "separate"],
                    help="where the L1 image loss is computed: fused = inside the rasterizer's own launches (render(..., "
                         "l1_target=gt): value from the render launch, gradient formed in the backward's pixel prologue, no "
                         "gradient image); separate = gsplat_mi355.render.l1_loss on the rendered image (its own two "
                         "launches, a gradient image written and read), the reference's call pattern at train.py:121")
    ap.add_argument("--prepass", action="store_true",
                    help="the reference's avatar call pattern: covariance from scaling + rotation_precomp (3x3) and colours "
                         "from SHs in the canonical frame computed BEFORE the rasterizer (fused N3 ops), passed as "
                         "cov3D_precomp / colors_precomp")
    ap.add_argument("--opacity", default="none", choices=["none", "second-call", "fused"],
                    help="also render the opacity image and add the reference's mask loss (0.1 * L1, train.py:143-153): "
                         "second-call = the reference's second rasterizer call (served by the shared-geometry path), "
                         "fused = rasterizer(..., with_opacity=True)")
    ap.add_argument("--train-step", action="store_true",
                    help="also run what follows the backward in the reference's step: densification statistics and the Adam "
                         "update of all parameters (fused N4 ops)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="STRONG scaling: a sequence of this many frames partitioned round-robin over the ranks (rank r "
                         "renders frames r, r + N, ...: render.py:51-62; BASELINE config 4 is --workload config4 "
                         "--total-frames 300 --gpus 8); value = total frames / the slowest rank's time.  Default 0: weak "
                         "scaling, every rank renders --steps frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-upstream-leg", action="store_true", help="skip the second timed leg (tile_rect = 0)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (traffic / valu become null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the run rocprofv3 wraps: frames only
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks (this process has not touched a GPU yet)
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: run `python -m torch.distributed.run --nproc-per-node N`
    on this same file as a CHILD process (never an exec), relay its output, return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# ------------------------------------------------------------------------------------------------
# rocprofv3 --pmc child passes (N = 1 only; before this process touches the GPU)
# ------------------------------------------------------------------------------------------------
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]]


def _short_kernel(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"<.*$", "", name)
    return name.strip()


def pmc_passes(args, argv):
    """Mean counter value per launch and kernel over a few frames of this workload: one rocprofv3 child per
    counter group (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc alone, no trace
    options).  Returns ({kernel: {counter: mean}}, note) -- ({}, reason) when rocprofv3 is missing or a pass fails."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    child_args = [a for a in argv if a not in ("--no-cpu-baseline",)]
    acc = {}
    tmp = tempfile.mkdtemp(prefix="gsplat_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=tmp)
    try:
        for counters in PMC_PASSES:
            out = os.path.join(tmp, "_".join(counters)[:40])
            cmd = [exe, "--pmc"] + counters + ["-d", out, "--output-format", "csv", "--", sys.executable,
                                               os.path.abspath(__file__)] + child_args + ["--pmc-child"]
            try:
                r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            except subprocess.TimeoutExpired:
                return {}, "rocprofv3 pass timed out (%s)" % ",".join(counters)
            if r.returncode != 0:
                return {}, "rocprofv3 pass failed (%s): rc %d" % (",".join(counters), r.returncode)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {}, "rocprofv3 pass wrote no counter file (%s)" % ",".join(counters)
            for f in files:
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        k = _short_kernel(row["Kernel_Name"])
                        a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                        a[0] += float(row["Counter_Value"])
                        a[1] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
    return res, "rocprofv3 --pmc child passes of this command (%s), mean per launch" % " | ".join(",".join(c) for c in PMC_PASSES)


def pin_rank_to_cores(local_rank, local_world):
    """One rank = one Python loop issuing a frame every 0.2-0.6 ms + autograd's backward thread: a rank that migrates or
    shares its cores with seven others stalls its GPU (the pair-count wait lets the host run one frame ahead at most:
    DESIGN.md 1).  Gives rank r of the node the r-th contiguous slice of the cores this process may run on -- before
    anything touches the GPU -- and returns "first-last (count)", or None where affinity cannot be set or
    GSPLAT_BENCH_NO_AFFINITY=1 says not to."""
    if os.environ.get("GSPLAT_BENCH_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity") or local_world < 1:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 2:  # fewer than two cores per rank: leave the scheduler alone
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        os.environ.setdefault("OMP_NUM_THREADS", str(min(per, 8)))
        return "%d-%d (%d cores)" % (mine[0], mine[-1], len(mine))
    except OSError:
        return None


def frame_plan(rank, world, steps, warmup, total_frames=0):
    """(frames this rank renders, in order; K = how many of them the timed region covers).  The first 8 + warmup frames are
    rendered before the timed region (first touches, settle, warm-up), the last K inside it.
    Weak scaling (total_frames = 0): rank r owns frames r, r + world, ... and times `steps` of them.
    Strong scaling: the reference's frame loop over a sequence of fixed length (render.py:51-62), partitioned -- rank r
    times every frame r, r + world, ... below total_frames exactly once (300 frames over 8 ranks: 38 or 37 each)."""
    from gsplat_mi355.sharding import frames_of_rank
    if total_frames > 0:
        own = frames_of_rank(rank, world, total=total_frames)
        if not own:
            return [], 0
        return [own[i % len(own)] for i in range(8 + warmup)] + own, len(own)
    return frames_of_rank(rank, world, steps + warmup + 8), steps


def traffic_bytes(c):
    """HBM-side bytes of one launch from FETCH_SIZE / WRITE_SIZE (rocprofv3 reports KiB): 2 x FETCH_SIZE + WRITE_SIZE.
    On gfx950 FETCH_SIZE counts every read request of the L2 -- a 128-byte line -- as 64 bytes
    (MI355X_MICROARCH.md, HBM section, states it for wide streaming reads).  Calibrated for THIS library's access
    shapes with tools/fetch_calib.hip (profiles/r02_fetch_calibration.md): dword and 16-byte streams, packed 48-byte
    records in order and at random, records in 128-byte slots and random 32-byte rows all read FETCH_SIZE = 1/2 of the
    128-byte lines they touch; WRITE_SIZE is exact, also for scattered 32-byte rows."""
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    f, w = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
    return {"bytes": int(2 * f + w), "fetch_size_raw": int(f), "write_size": int(w)}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args, argv))
    world = int(env_world or "1")
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    affinity = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else None

    pmc, pmc_note = {}, "skipped (--no-pmc)"
    if world == 1 and not args.pmc_child and not args.no_pmc:
        pmc, pmc_note = pmc_passes(args, argv)  # children; this process has not initialised the GPU yet

    rehearsal = False
    dist = None
    # GSPLAT_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, broadcast, checksums, barriers, max over
    # ranks) also with ONE rank -- tests/test_gpu_bench.py runs the RCCL backend that way on a one-GPU box
    use_dist = world > 1 or os.environ.get("GSPLAT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal of the multi-rank code path on a one-GPU box (never for a reported number): all ranks on cuda:0,
        # gloo instead of RCCL -- GSPLAT_BENCH_REHEARSAL=1
        rehearsal = os.environ.get("GSPLAT_BENCH_REHEARSAL") == "1"
        if rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import diff_gaussian_rasterization as dgr
    from gsplat_mi355 import _lib
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.debug import frame_stats
    from gsplat_mi355.render import Pipe, l1_loss, render, ssim
    from gsplat_mi355.scenes import GaussianCloud, synthetic_cloud
    from gsplat_mi355.sharding import broadcast_cloud, frames_of_rank

    _lib.load()
    N, W, H, deg, tail, do_bwd = WORKLOADS[args.workload]

    # ---- Gaussian state: built on rank 0, one RCCL broadcast
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev,
                            layout=WORKLOAD_LAYOUT.get(args.workload, "box")) if rank == 0 else None
    t_bcast = 0.0
    if use_dist:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cloud = broadcast_cloud(cloud, N, deg, dev, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    state_sums = None
    if use_dist:
        # evidence that every rank renders the SAME Gaussian state: a checksum of the packed buffer's bit patterns per rank
        cs = cloud.pack().view(torch.int32).to(torch.int64).sum().reshape(1)
        if rehearsal:
            cs = cs.cpu()
        got = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(got, cs)
        state_sums = [int(g.item()) for g in got]
    for f in GaussianCloud.FIELDS:
        getattr(cloud, f).requires_grad_(do_bwd)

    Wm = args.warmup
    strong = args.total_frames > 0
    frames, K = frame_plan(rank, world, args.steps, Wm, args.total_frames)
    if K == 0:
        print("bench.py: --total-frames %d leaves rank %d of %d without a frame" % (args.total_frames, rank, world), file=sys.stderr)
        sys.exit(2)
    cams = [orbit_camera(f, W, H, device=dev) for f in frames]
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    bg = torch.zeros(3, device=dev)
    pipe = Pipe(compute_cov3D_python=args.prepass, convert_SHs_python=args.prepass, fuse_opacity=(args.opacity == "fused"))
    gt_mask = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(2)) > 0.5).float().to(dev)
    if args.prepass:
        # what models/deformer/rigid.py:222-231 attaches: a (detached) forward bone transform per Gaussian and the
        # rotation matrix composed with it
        gq = torch.Generator().manual_seed(2)
        bq = torch.nn.functional.normalize(torch.randn(N, 4, generator=gq), dim=1).to(dev)
        w, x, y, z = bq.unbind(1)
        bone = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                            1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                            1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
        cloud.fwd_transform = bone
        q = torch.nn.functional.normalize(cloud.rotations.detach(), dim=1)
        w, x, y, z = q.unbind(1)
        Rg = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                          1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                          1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
        cloud.rotation_precomp = Rg.contiguous().requires_grad_(do_bwd)

    opt = stats = None
    if args.train_step and do_bwd:
        from gsplat_mi355.optim import FusedAdam
        from gsplat_mi355.render import DensifyStats
        # the cloud holds POST-activation values (no log-scale / logit parametrisation), so the reference's learning
        # rates would walk it out of the valid range within tens of steps; the cost of the update does not depend
        # on the rate, so it is kept tiny and the scene -- hence every other stage's work -- stays what it is
        lrs = dict(xyz=1.6e-9, scales=5e-9, rotations=1e-9, opacity=5e-9, shs=2.5e-9)
        opt = FusedAdam([{"params": [getattr(cloud, f)], "lr": lrs.get(f, 1e-3), "name": f} for f in GaussianCloud.FIELDS],
                        lr=0.0, eps=1e-15)
        stats = DensifyStats(N, dev)

    def step(i):
        for f in GaussianCloud.FIELDS:
            getattr(cloud, f).grad = None
        if args.prepass:
            cloud.rotation_precomp.grad = None
        if do_bwd:
            fused = args.l1 == "fused"
            pkg = render(cams[i], cloud, pipe, bg, return_opacity=(args.opacity != "none"), l1_target=gt if fused else None)
            loss = pkg.l1 if fused else l1_loss(pkg.render, gt)
            if args.opacity != "none":
                loss = loss + 0.1 * l1_loss(pkg.opacity_render, gt_mask)
            if args.loss == "l1+dssim":  # train.py:120-124 with lambda_l1 = 0.8, lambda_dssim = 0.2
                loss = 0.8 * loss + 0.2 * (1.0 - ssim(pkg.render, gt))
            loss.backward()
            if opt is not None:
                with torch.no_grad():
                    stats.update(pkg)
                opt.step()
        else:
            with torch.no_grad():
                pkg = render(cams[i], cloud, pipe, bg)
        return pkg

    if args.pmc_child:  # under rocprofv3 --pmc: a few frames, nothing timed, nothing printed
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        return

    groups = {"preprocess": ["preprocess"],
              "binning": ["pair_scan", "depth_sort", "rank_list", "tile_count", "ranges_order", "tile_write"],
              "render_fwd": ["render_fwd", "recolor", "second_ones"], "render_bwd": ["render_bwd"],
              "gaussian_bwd": ["gaussian_bwd"]}
    kernel_stages = ["preprocess", "render_fwd", "render_bwd", "gaussian_bwd"]

    def run_leg(tile_rect):
        """One complete measurement in one binning mode: per-stage HIP events on 5 untimed frames, W warm-up steps,
        then EXACTLY K timed steps between barrier + synchronize on both sides, max over ranks."""
        dgr._TILE_RECT = tile_rect
        dgr.release_shared_geometry()
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for i in range(3, 8):
            pkg = step(i)
        stages = _lib.profile_collect()
        _lib.profile_enable(False)
        stage_ms = {k: v[0] / 5.0 for k, v in stages.items()}  # ms per frame (5 profiled frames)
        dominant = max(kernel_stages, key=lambda k: stage_ms.get(k, 0.0))
        # D and n_contrib of the benchmark frame (reported with every number: cost is a function of D)
        with torch.no_grad():
            vis = int((pkg.radii > 0).sum().item())
        D, mean_contrib, quad_hits, pairs_valid = frame_stats(cams[0], cloud, pipe, bg)
        # untimed: ~0.7 s of back-to-back steps right before the warm-up, so that the chip's clocks have settled under
        # THIS load (the statistics above leave the GPU idle for a while; a timed region that starts on an idle chip
        # reads 10-25 % slow for its first hundreds of steps), then the W warm-up steps and the K timed ones
        # The timed loop brackets the dominant kernel with two HIP events per step.  Creating an event and recording it
        # for the first time cost ~0.1-0.2 ms each (seen as a timed loop whose first steps take 3-4 ms): the stage timer
        # therefore already runs during the settle and warm-up steps below, so that every event the timed loop takes from
        # the pool has been created AND recorded on this stream before.
        _lib.profile_reserve(2 * K + 64)
        _lib.profile_enable(True, stage=dominant)
        # A full collection of the interpreter's cyclic garbage collector walks every object torch has created at import
        # (60-100 ms, observed as ONE step of the timed loop taking that long): collect now -- BEFORE the settle steps,
        # nothing may leave the GPU idle between them and the timed loop -- and keep it off until the timing is done.
        gc.collect()
        gc.disable()
        # ... over the frames the warm-up and the timed loop will render (at least once each): the pair count differs from
        # frame to frame, and a binning capacity or scratch size the caching allocator has not seen yet is a hipMalloc of
        # milliseconds -- inside a timed region of 20 steps one of those is 20 % of the reading
        t_end = time.perf_counter() + 0.7
        nf, i = len(cams), 0
        while time.perf_counter() < t_end or i < nf:
            for _ in range(8):
                step(i % nf)
                i += 1
            torch.cuda.synchronize()
        for i in range(Wm):
            step(8 + i)
        torch.cuda.synchronize()
        _lib.profile_collect()  # returns every event used so far to the pool: the timed loop takes them from there
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trace = os.environ.get("GSPLAT_BENCH_TRACE") == "1"  # diagnostics only: a synchronise + a line every 25 steps
        worst = (0.0, -1)
        for i in range(K):
            if trace:
                h0 = time.perf_counter()
            step(8 + Wm + i)
            if trace:
                h1 = time.perf_counter() - h0
                if h1 > worst[0]:
                    worst = (h1, i)
            if trace and i % 25 == 24:
                print("[trace] slowest single step so far (host time): %.3f ms at step %d" % (worst[0] * 1e3, worst[1]), file=sys.stderr)
                torch.cuda.synchronize()
                st = torch.cuda.memory_stats(dev)
                print("[trace] leg tile_rect=%d step %d: %.4f ms/step so far, reserved %.0f MB, allocs %d, num_ooms %d, last count %s" % (
                    tile_rect, i + 1, (time.perf_counter() - t0) / (i + 1) * 1e3, st["reserved_bytes.all.current"] / 1e6,
                    st["allocation.all.allocated"], st["num_ooms"], dgr._last_count.get((dev.index, N, W, H))), file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gc.enable()
        # the shader clock the chip holds under a VALU-bound load, measured NOW, right behind the timed loop (an FMA stream
        # of ~1 ms on every SIMD: s_memtime against the constant 100 MHz counter) -- for roofline.binding
        clock_hz = _lib.clock_probe(dev) if world == 1 else None
        dom = _lib.profile_collect().get(dominant, (0.0, 0))
        _lib.profile_enable(False)
        # Per-stage times, taken NOW: right behind the timed loop, on the same settled clocks and over the same frames, one
        # stage per pass (two HIP events per call -- bracketing every stage at once adds ~2-4 us of event handling per
        # stage to the frame).  Taken before the settle steps, as until round 2, they read ~15 % high and added up to more
        # than ms_per_step.
        post = {}
        for name in sorted(stage_ms):
            _lib.profile_enable(True, stage=name)
            for i in range(8):
                step(8 + Wm + (i % max(K, 1)))
            got = _lib.profile_collect().get(name)
            if got and got[1]:
                post[name] = got[0] / 8.0  # ms per frame (a stage may run more than once per frame)
        _lib.profile_enable(False)
        stage_ms = post or stage_ms
        if use_dist:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        px = W * H
        sb = stage_bytes(N, D, px)
        dom_ms = dom[0] / max(dom[1], 1)
        achieved = sb[dominant] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        frame_bytes = sum(sb.values()) if do_bwd else sb["preprocess"] + sb["binning"] + sb["render_fwd"]
        total_frames = args.total_frames if strong else world * K
        per_rank = total_frames / world  # frames per rank (strong scaling: the average; the slowest rank sets `elapsed`)
        ms_step = elapsed / per_rank * 1e3
        grouped = {g: sum(stage_ms.get(st, 0.0) for st in ss) for g, ss in groups.items()}
        in_groups = {st for ss in groups.values() for st in ss}
        for st, v in stage_ms.items():  # stages outside the five SURVEY 8(d) groups: losses, the backward's tile order + mark fill
            if st not in in_groups:
                grouped[st] = v
        # what the step spends outside the library's stages: torch glue launches (autograd's ones-fill, grad * g, the
        # zero leaf), gaps between kernels, host pacing.  DERIVED, not measured: ms_per_step (the timed loop) minus the
        # stage times (8 frames per stage, taken right behind the loop) -- it can come out slightly negative when the
        # stage passes read a little long; the signed value is reported, `stages_ms` carries it clamped at 0 so that the
        # entries add up to ms_per_step.  Not formed for world > 1 (the step time there is the slowest rank's).
        residual = ms_step - sum(grouped.values())
        grouped["outside_stages"] = max(residual, 0.0) if world == 1 else 0.0
        stage_frac = {g: round(sb[g] / (grouped[g] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for g in sb if grouped.get(g, 0.0) > 0}
        return {
            "fps": total_frames / elapsed, "ms_per_step": ms_step, "elapsed": elapsed, "D": D, "visible": vis, "K": K,
            "stages_frac": stage_frac,
            "mean_contrib": mean_contrib, "quad_hits": quad_hits, "pairs_valid": pairs_valid, "dominant": dominant, "dom_ms": dom_ms, "stage_ms": stage_ms,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel_ms": round(dom_ms, 4), "algorithmic_bytes": sb[dominant],
                         "frame_algorithmic_bytes": frame_bytes,
                         "frame_frac": round(frame_bytes * (K / elapsed) / 1e9 / HBM_PEAK_GBS, 5)},
            "stages_ms": {g: round(v, 4) for g, v in grouped.items()},
            "outside_stages_signed": round(residual, 4) if world == 1 else None, "stage_pass_frames": 8, "clock_hz": clock_hz,
        }

    default_rect = int(os.environ.get("GSPLAT_TILE_RECT", "1"))
    main_leg = run_leg(default_rect)
    up_leg = None
    if not args.no_upstream_leg and default_rect != 0:
        up_leg = run_leg(0)
        dgr._TILE_RECT = default_rect
        dgr.release_shared_geometry()

    if rank == 0:
        roof = main_leg["roofline"]
        dominant = main_leg["dominant"]
        # ---- measured HBM traffic and VALU issue of the render kernels (rocprofv3 --pmc children of THIS run)
        per_kernel = {}
        for kname, stage in KERNEL_STAGE.items():
            c = pmc.get(kname)
            if c:
                per_kernel[kname] = {"stage": stage, "traffic": traffic_bytes(c),
                                     "valu_insts": int(c["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in c else None}
        dom_kernels = [k for k, st in KERNEL_STAGE.items() if st == dominant and k in per_kernel and per_kernel[k]["traffic"]]
        if dom_kernels:
            roof["traffic"] = sum(per_kernel[k]["traffic"]["bytes"] for k in dom_kernels)
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(roof["algorithmic_bytes"], 1), 3)
        roof["traffic_source"] = pmc_note
        # (pixel, Gaussian) pairs the render kernels evaluate: 64 pixels per (quadrant, Gaussian) entry up to the quadrant's
        # last contributor (what the backward iterates; the forward stops a little later, when all 64 pixels are done)
        pairs_eval = 64.0 * main_leg["quad_hits"]
        fl = stage_flops(pairs_eval)
        valu = {}
        for kname, stage in (("render_fwd_kernel", "render_fwd"), ("render_bwd_kernel", "render_bwd")):
            ms = main_leg["dom_ms"] if stage == dominant else main_leg["stage_ms"].get(stage, 0.0)
            if ms <= 0:
                continue
            rec = {"kernel_ms": round(ms, 4), "pairs_evaluated": pairs_eval,
                   # pairs actually composited (alpha >= 1/255 before the pixel is done; gs_pair_stats, outside the timed
                   # region) and their share of what the kernel evaluates: the head-room finer culling than 8 x 8 could have
                   "pairs_valid": float(main_leg["pairs_valid"]),
                   "pairs_valid_over_evaluated": round(main_leg["pairs_valid"] / max(pairs_eval, 1.0), 4),
                   "flops_8d": fl[stage], "tflops_8d": round(fl[stage] / (ms * 1e-3) / 1e12, 2),
                   "frac_of_fp32_vector_peak": round(fl[stage] / (ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)}
            insts = per_kernel.get(kname, {}).get("valu_insts")
            if insts:
                # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles: slots = SIMDs x clock x t / 2
                rec["insts_per_launch"] = insts
                rec["issue_slot_util"] = round(insts * 2.0 / (SIMDS * CLOCK_HZ * ms * 1e-3), 4)
            valu[stage] = rec
        if dominant in valu and "issue_slot_util" in valu[dominant]:
            # the ceiling that actually binds the dominant kernel, next to the HBM one the contract asks for
            roof["binding"] = {"ceiling": "VALU issue slots (one wave64 instruction per SIMD-32 every 2 cycles)",
                               "frac": valu[dominant]["issue_slot_util"], "kernel": dominant,
                               # the chip does not hold 2.4 GHz under a VALU-bound load: measured live in this run, right
                               # behind the timed loop (gs_clock_probe: an FMA stream of ~1 ms on every SIMD, s_memtime
                               # against s_memrealtime)
                               "measured_clock_hz": main_leg["clock_hz"],
                               "clock_source": "gs_clock_probe behind the timed loop of this run",
                               "frac_at_measured_clock": (round(valu[dominant]["issue_slot_util"] * CLOCK_HZ / main_leg["clock_hz"], 4)
                                                          if main_leg.get("clock_hz") else None)}
        roof["valu"] = {"peak_tflops": VALU_PEAK_TFLOPS, "simds": SIMDS, "clock_hz": CLOCK_HZ,
                        "note": "secondary ceiling (SURVEY.md 8d): the render kernels are VALU-issue bound, not HBM bound; "
                                "issue_slot_util = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time); "
                                "flops_8d = SURVEY 8(d)'s 26 / 71 flop per (pixel, Gaussian) pair x the pairs the kernels "
                                "evaluate (64 x quadrant-list entries up to each quadrant's last contributor)",
                        **valu}
        out = {
            "metric": "render fps (fwd+bwd) @200k Gaussians 1024x1024 SH3" if args.workload == "config3"
            else "render fps (%s) @%s" % ("fwd+bwd" if do_bwd else "fwd", args.workload),
            "value": round(main_leg["fps"], 2), "unit": "frames/s", "n_gpus": world,
            "steps": (args.total_frames + world - 1) // world if strong else K, "warmup": Wm,
            "ms_per_step": round(main_leg["ms_per_step"], 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dk Gaussians, %dx%d, SH deg %d, %s; shs+scales+rotations inputs, %s" % (
                args.workload, N // 1000, W, H, deg, "forward+backward" if do_bwd else "forward",
                ("L1 loss" if args.loss == "l1" else "0.8 L1 + 0.2 D-SSIM loss") +
                (" (L1 fused into the rasterizer: l1_target)" if (do_bwd and args.l1 == "fused") else "") +
                (", covariance + colours precomputed by the fused pre-pass" if args.prepass else "") +
                (", + densification statistics + Adam step" if args.train_step else "") +
                ("" if args.opacity == "none" else ", + opacity render (%s) with 0.1 L1 mask loss" % args.opacity)),
                "tile_rect": default_rect, "l1": (args.l1 if do_bwd else None), "long_lists": dgr._LONG_LISTS, "gaussians": N, "visible": main_leg["visible"], "width": W, "height": H,
                "sh_degree": deg, "num_rendered": main_leg["D"], "mean_n_contrib": round(main_leg["mean_contrib"], 2),
                "frames_per_rank": ([args.total_frames // world, (args.total_frames + world - 1) // world] if strong else K),
                "total_frames": args.total_frames if strong else world * K, "ranks": world,
                "backend": "none" if not use_dist else ("gloo (REHEARSAL: all ranks on one GPU)" if rehearsal else "nccl (RCCL)"),
                "rank_cpu_affinity": affinity,
                "parallelism": "frames sharded x%d" % world + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
                "broadcast_s": round(t_bcast, 6), "state_checksums": state_sums},
            "roofline": roof,
            # per frame, measured right behind the timed loop, one stage per pass; the entries add up to ms_per_step
            # (`outside_stages` = torch glue launches, gaps, host pacing); stages_frac = algorithmic bytes / time / 8 TB/s
            "stages_ms": main_leg["stages_ms"],
            "stages_frac": main_leg["stages_frac"],
            "stages_detail_ms": {k: round(v, 4) for k, v in sorted(main_leg["stage_ms"].items())},
            # `outside_stages` is derived (ms_per_step - sum of the stage times), clamped at 0 in stages_ms; the signed residual:
            "outside_stages_derived": {"signed_ms": main_leg["outside_stages_signed"], "stage_pass_frames": main_leg["stage_pass_frames"]},
        }
        if up_leg is not None:
            # the reference's own binning (3-sigma squares): tile lists / ranges / num_rendered are upstream's bit for bit
            out["upstream_rect"] = {"tile_rect": 0, "value": round(up_leg["fps"], 2), "unit": "frames/s",
                                    "ms_per_step": round(up_leg["ms_per_step"], 4), "num_rendered": up_leg["D"],
                                    "mean_n_contrib": round(up_leg["mean_contrib"], 2), "roofline": up_leg["roofline"],
                                    "stages_ms": up_leg["stages_ms"], "stages_frac": up_leg["stages_frac"]}
        if per_kernel:
            out["pmc_per_kernel"] = per_kernel
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cloud, W, H, deg, gt, do_bwd, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(cloud, W, H, deg, gt, do_bwd, threads):
    """The CPU oracle (oracle/gs_oracle.c, kind 'port': the reference has no CPU path and its CUDA
    source is absent) timed on the host cores on three frames of the same workload."""
    ncpu = len(os.sched_getaffinity(0))
    nthr = threads if threads > 0 else min(ncpu, 16)
    import helpers
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.scenes import GaussianCloud
    from oracle import gs_oracle
    gs_oracle.build()
    gs_oracle.set_num_threads(nthr)
    c = GaussianCloud(*[getattr(cloud, f).detach().cpu() for f in GaussianCloud.FIELDS], deg)
    gt_np = gt.cpu().numpy()

    def frame(i):
        sc = helpers.oracle_scene(c, orbit_camera(i, W, H))
        fw = gs_oracle.forward(sc)
        if do_bwd:
            g = (np.sign(fw["color"] - gt_np) / gt_np.size).astype(np.float32)
            gs_oracle.backward(sc, fw, g)

    frame(0)  # untimed: first touch of the oracle's buffers
    nfr = 3
    t0 = time.perf_counter()
    for i in range(1, 1 + nfr):
        frame(i)
    dt = time.perf_counter() - t0
    return {"value": round(nfr / dt, 4), "unit": "frames/s", "cores": gs_oracle.num_threads(), "kind": "port",
            "sample": "%d frames (%s) of the same scene after one untimed frame, %.1f s of wall time on %d threads" % (
                nfr, "fwd+bwd" if do_bwd else "fwd", dt, gs_oracle.num_threads())}


if __name__ == "__main__":
    main()
