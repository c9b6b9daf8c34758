"""One self-KNN (K = 6) and one distCUDA2 at 200k points, for rocprofv3 --kernel-trace --stats."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3dgs-avatar-release_amd"))
import torch
from gsplat_mi355.knn import knn_points
from simple_knn._C import distCUDA2
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = 200000
g = torch.Generator().manual_seed(0)
x = (torch.rand(n, 3, generator=g) if dist == "uniform" else torch.randn(n, 3, generator=g)).cuda()
for _ in range(5):
    knn_points(x[None], x[None], K=6)
    distCUDA2(x)
torch.cuda.synchronize()
