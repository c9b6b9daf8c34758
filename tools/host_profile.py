"""cProfile of the host side of forward-only frames (where does the Python time go when the GPU work is short?)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import bench
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.render import Pipe, render
    from gsplat_mi355.scenes import synthetic_cloud
    wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
    N, W, H, deg, tail, _ = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev)
    cams = [orbit_camera(f, W, H, device=dev) for f in range(64)]
    bg = torch.zeros(3, device=dev)
    pipe = Pipe()
    with torch.no_grad():
        for i in range(20):
            render(cams[i], cloud, pipe, bg)
        torch.cuda.synchronize()
        n = 1000
        t0 = time.perf_counter()
        for i in range(n):
            render(cams[i % 64], cloud, pipe, bg)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%s forward-only: %.1f us/frame to issue, %.1f us/frame complete" % (wl, t_issue / n * 1e6, t_all / n * 1e6))
        pr = cProfile.Profile()
        pr.enable()
        for i in range(n):
            render(cams[i % 64], cloud, pipe, bg)
        pr.disable()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
