"""Forward-only throughput with several host threads, each rendering its own frames on its own HIP stream (the
library is stateless and works on the caller's stream; ctypes releases the GIL during the C calls).  One frame has
one host round trip (the pair count) during which its stream is idle: a second stream's kernels fill that gap.

  python tools/multistream.py --workload config2 --frames 600 --streams 1 2 3
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3dgs-avatar-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3])
    args = ap.parse_args()
    import bench
    from gsplat_mi355.camera import orbit_camera
    from gsplat_mi355.render import Pipe, render
    from gsplat_mi355.scenes import synthetic_cloud
    N, W, H, deg, tail, _ = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    cloud = synthetic_cloud(N, sh_degree=deg, seed=0, heavy_tail=tail, device=dev)
    cams = [orbit_camera(f, W, H, device=dev) for f in range(64)]
    bg = torch.zeros(3, device=dev)
    pipe = Pipe()
    with torch.no_grad():
        for i in range(8):
            ref = render(cams[0], cloud, pipe, bg).render.clone()
    torch.cuda.synchronize()

    refs = {}
    with torch.no_grad():
        for i in range(64):
            refs[i] = render(cams[i], cloud, pipe, bg).render.clone()
    torch.cuda.synchronize()

    def worker(tid, S, nframes, out):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st), torch.no_grad():
                for i in range(3):
                    render(cams[i], cloud, pipe, bg)  # per-thread warm-up (pinned word, size hint)
                st.synchronize()
                barrier.wait()
                done, last, last_i = 0, None, -1
                for i in range(tid, nframes, S):
                    last, last_i = render(cams[i % 64], cloud, pipe, bg).render, i % 64
                    done += 1
                st.synchronize()
                out[tid] = (done, bool(torch.equal(last, refs[last_i])))
        except BaseException as e:  # a dead thread would otherwise read as a faster run
            out[tid] = (0, repr(e))
            try:
                barrier.abort()
            except Exception:
                pass
            raise

    res = {}
    for S in args.streams:
        barrier = threading.Barrier(S + 1)
        out = {}
        th = [threading.Thread(target=worker, args=(t, S, args.frames, out)) for t in range(S)]
        for t in th:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert sum(v[0] for v in out.values()) == args.frames and all(v[1] is True for v in out.values()), out
        res[S] = round(args.frames / dt, 1)
        print("streams %d: %.1f frames/s (%.3f ms/frame)" % (S, args.frames / dt, dt / args.frames * 1e3), flush=True)
    print(json.dumps({"workload": args.workload, "forward_only_fps_by_streams": res}))


if __name__ == "__main__":
    main()
