"""Times one optimizer step over the reference's six parameter groups at 200k Gaussians: torch.optim.Adam
(foreach / fused variants) against gsplat_mi355.optim.FusedAdam."""
import sys, time
sys.path.insert(0, "3dgs-avatar-release_amd")
import torch
from gsplat_mi355.optim import FusedAdam
n = 200000
shapes = [(n, 3), (n, 1, 3), (n, 15, 3), (n, 1), (n, 3), (n, 4)]
lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3]


def run(make):
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    opt = make([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)])
    for p in ps:
        p.grad = torch.randn_like(p)
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e3


print("torch.optim.Adam (default): %.3f ms" % run(lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15)))
try:
    print("torch.optim.Adam (fused=True): %.3f ms" % run(lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15, fused=True)))
except Exception as e:
    print("torch fused unavailable:", e)
print("gsplat_mi355 FusedAdam: %.3f ms" % run(lambda g: FusedAdam(g, lr=0.0, eps=1e-15)))
