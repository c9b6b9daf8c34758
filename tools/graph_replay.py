"""What the wrapper and the pair-count wait cost a fixed-shape no_grad frame -- and (opt-in) a hipGraph replay of it.


One forward-only frame = gs_forward_preprocess + gs_forward_render on a binning state of fixed capacity (the two-phase
entry points: nothing in them waits for the pair count).  Measured on the same buffers:
  eager    : the two C calls per frame, issued back to back, one synchronisation at the end
  product  : gsplat_mi355.render.render() under no_grad (gs_forward with the speculative capacity and the count wait)
  graph    : (--graph) the two calls captured once (torch.cuda.CUDAGraph = hipGraph stream capture), replayed per frame.
             ON THIS STACK (ROCm 7.2, PyTorch 2.10) THE FIRST REPLAY ENDED IN A GPU MEMORY FAULT ("write access to a
             read-only page"), config 2 and avatar50k alike, while the same calls run eagerly on the same buffers and
             streams; not pursued (a fault can take the node down) -- the eager loop already shows what there is to gain.
Usage: python tools/graph_replay.py [config2|avatar50k|...] [frames] [--graph]"""
import ctypes
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import bench
    from diff_gaussian_rasterization import GaussianRasterizationSettings, _make_args
    from gsplat_mi355 import _lib
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.render import Pipe, render
    from gsplat_mi355.scenes import synthetic_cloud
    want_graph = "--graph" in sys.argv
    argv = [x for x in sys.argv if x != "--graph"]
    wl = argv[1] if len(argv) > 1 else "config2"
    n = int(argv[2]) if len(argv) > 2 else 2000
    N, W, H, deg, tail, _ = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev, layout=bench.WORKLOAD_LAYOUT.get(wl, "box"))
    cam = orbit_camera(0, W, H, device=dev)
    bg = torch.zeros(3, device=dev)
    L = _lib.load()
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center,
        prefiltered=False, debug=False)
    keep = []
    a = _make_args(settings, cloud.xyz, cloud.shs, None, cloud.opacity, cloud.scales, cloud.rotations, None, keep)
    gb = _lib.nbytes(L.gs_geom_bytes, N)
    ib = _lib.nbytes(L.gs_image_bytes_for, ctypes.byref(a))
    geom = torch.zeros(gb, dtype=torch.uint8, device=dev)
    img = torch.zeros(ib, dtype=torch.uint8, device=dev)
    radii = torch.zeros(N, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64).pin_memory()
    color = torch.zeros(3, H, W, device=dev)

    def phase1(stream):
        _lib.check(L.gs_forward_preprocess(ctypes.byref(a), geom.data_ptr(), gb, img.data_ptr(), ib, radii.data_ptr(),
                                           count.data_ptr(), ctypes.c_void_p(stream.cuda_stream)))

    s0 = torch.cuda.current_stream(dev)
    phase1(s0)
    s0.synchronize()
    cap = int(count.item()) * 9 // 8
    bb = _lib.nbytes(L.gs_binning_bytes, cap, W, H)
    binning = torch.zeros(bb, dtype=torch.uint8, device=dev)

    def frame(stream):
        phase1(stream)
        _lib.check(L.gs_forward_render(ctypes.byref(a), geom.data_ptr(), gb, binning.data_ptr(), bb, img.data_ptr(), ib, cap,
                                       color.data_ptr(), ctypes.c_void_p(stream.cuda_stream)))

    def timed(step, label):
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%-8s %7.1f us/frame to issue  %7.1f us/frame complete  (%.0f frames/s)" % (label, t_issue / n * 1e6, t_all / n * 1e6, n / t_all))
        return t_all / n

    frame(s0)
    s0.synchronize()
    ref = color.clone()
    t_eager = timed(lambda: frame(torch.cuda.current_stream(dev)), "eager")
    pipe = Pipe()
    with torch.no_grad():
        t_prod = timed(lambda: render(cam, cloud, pipe, bg), "product")
    print("%s: eager two-phase loop / product = %.3f" % (wl, t_eager / t_prod))
    if not want_graph:
        return
    side = torch.cuda.Stream(dev)
    side.wait_stream(s0)
    with torch.cuda.stream(side):
        frame(side)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        frame(torch.cuda.current_stream(dev))
    color.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(color, ref), "the replayed frame differs"
    t_graph = timed(g.replay, "graph")
    print("%s: graph / eager = %.3f, graph / product = %.3f" % (wl, t_graph / t_eager, t_graph / t_prod))


if __name__ == "__main__":
    main()
