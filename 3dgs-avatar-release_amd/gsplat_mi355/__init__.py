"""gsplat_mi355: MI355X-native (gfx950) differentiable Gaussian-splat rasterizer.

The compute path is the hand-written HIP library behind include/gsplat_mi355.h, reached through
ctypes (`gsplat_mi355._lib`).  Public drop-in packages built on it: `diff_gaussian_rasterization`
and `simple_knn` (same import names and symbols the reference imports at
gaussian_renderer/__init__.py:17 and scene/gaussian_model.py:20).
"""
__all__ = ["camera", "scenes"]
