"""Where a four-wave quadrant of the small-image forward spends its cycles (avatar-shaped frames).  Needs the variant build
  GSPLAT_VARIANT=fwd4prof GSPLAT_EXTRA_HIPCC_FLAGS=-DFWD4_PROF python 3dgs-avatar-release_amd/build.py
and GSPLAT_LIB_PATH pointing at it; renders three frames of the avatar workload, the kernel prints thread 0's cycle counts
per phase (staging incl. barriers / step loop / batch tail) for the first eight workgroups -- the heaviest tiles."""
import os
import sys

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
    wl = sys.argv[1] if len(sys.argv) > 1 else "avatar"
    N, W, H, deg, tail, _ = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev, layout=bench.WORKLOAD_LAYOUT.get(wl, "box"))
    cam = orbit_camera(0, W, H, device=dev)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        for _ in range(3):
            render(cam, cloud, Pipe(), bg)
            torch.cuda.synchronize()
            print("---- frame", flush=True)


if __name__ == "__main__":
    main()
