"""Times knn_points (self-KNN as the AIAP loss calls it every step) and distCUDA2 on the GPU."""
import sys, time
sys.path.insert(0, "3dgs-avatar-release_amd")
import torch
from gsplat_mi355.knn import knn_points
from simple_knn._C import distCUDA2


def bench(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for dist in ("uniform", "normal"):
    for n in (50000, 200000):
        x = (torch.rand(n, 3, device="cuda") if dist == "uniform" else torch.randn(n, 3, device="cuda"))
        line = "%s N=%d:" % (dist, n)
        for K in (1, 3, 6):
            line += "  K=%d %.3f ms" % (K, bench(lambda: knn_points(x[None], x[None], K=K)))
        line += "  distCUDA2 %.3f ms" % bench(lambda: distCUDA2(x))
        print(line)
