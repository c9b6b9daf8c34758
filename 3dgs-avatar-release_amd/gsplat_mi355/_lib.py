"""ctypes binding of libgsplat_mi355.so (include/gsplat_mi355.h).  There is NO fallback: if the HIP
library is missing or a call fails, this raises."""
import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_LIB_PATH: another build of the same library (A/B runs of kernel variants: `GSPLAT_VARIANT=name python build.py`
# leaves it under variants/name/; tools/variant_cmp.sh, tools/ab_bench.sh)
LIB_PATH = os.environ.get("GSPLAT_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libgsplat_mi355.so")


class GsFwdArgs(ctypes.Structure):
    _fields_ = [
        ("P", c_int32), ("sh_degree", c_int32), ("M", c_int32), ("W", c_int32), ("H", c_int32),
        ("bg", c_void_p), ("means3D", c_void_p), ("shs", c_void_p), ("colors_precomp", c_void_p),
        ("opacities", c_void_p), ("scales", c_void_p), ("rotations", c_void_p), ("cov3D_precomp", c_void_p),
        ("viewmatrix", c_void_p), ("projmatrix", c_void_p), ("campos", c_void_p),
        ("scale_modifier", c_float), ("tanfovx", c_float), ("tanfovy", c_float),
        ("prefiltered", c_int32), ("debug", c_int32), ("tile_rect", c_int32), ("long_lists", c_int32),
        ("frame_stats", c_void_p), ("l1_target", c_void_p), ("l1_loss", c_void_p), ("l1_grad", c_void_p),
        ("forward_only", c_int32),
    ]


class GsSecondImage(ctypes.Structure):  # include/gsplat_mi355.h: GsSecondImage
    _fields_ = [("colors", c_void_p), ("out_color", c_void_p), ("dL_dpix", c_void_p), ("img", c_void_p),
                ("img_bytes", c_size_t), ("long_lists", c_int32)]


class GsGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dcolors", "dL_dopacity",
                                        "dL_dscales", "dL_drotations", "dL_dcov3D")]


EXPORTS = ["gs_geom_bytes", "gs_image_bytes", "gs_binning_bytes", "gs_backward_scratch_bytes",
           "gs_forward_preprocess", "gs_forward_render", "gs_forward", "gs_forward_shared", "gs_backward", "gs_mark_visible", "knn_workspace_bytes",
           "knn_dist2", "gs_geom_field", "gs_binning_field", "gs_image_field", "gs_status_string",
           "gs_last_hip_error", "gs_last_stage", "gs_build_info", "gs_profile_enable", "gs_profile_filter", "gs_profile_collect",
           "gs_l1_loss_workspace_bytes", "gs_l1_loss", "gs_bce_loss", "gs_ssim_workspace_bytes", "gs_ssim_forward", "gs_ssim_backward",
           "gs_build_covariance", "gs_build_covariance_backward", "gs_sh2rgb", "gs_sh2rgb_backward", "knn_points", "gs_densify_stats", "gs_adam_step",
           "gs_opacity_image", "gs_backward_with_opacity", "gs_tuning", "gs_profile_reserve", "gs_image_bytes_for", "gs_backward_with_second", "gs_clock_probe", "gs_pair_stats", "gs_xcc_probe"]

GS_E_WORKSPACE = -5  # include/gsplat_mi355.h
GS_E_CAPTURE = -6
GS_ADAM_MAX_TENSORS = 16


class GsAdamTensor(ctypes.Structure):  # include/gsplat_mi355.h: GsAdamTensor
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("n", c_int64), ("lr", c_float)]


_lock = threading.Lock()
_lib = None


def load():
    """Loads the HIP library (torch first, so that its bundled HIP runtime is the one both share)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (libamdhip64 must come from torch's process image)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgsplat_mi355.so not found at %s: build it with "
                               "`python 3dgs-avatar-release_amd/build.py` (there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.gs_geom_bytes.argtypes = [c_int32, POINTER(c_size_t)]
        L.gs_image_bytes.argtypes = [c_int32, c_int32, POINTER(c_size_t)]
        L.gs_image_bytes_for.argtypes = [POINTER(GsFwdArgs), POINTER(c_size_t)]
        L.gs_binning_bytes.argtypes = [c_int64, c_int32, c_int32, POINTER(c_size_t)]
        L.gs_backward_scratch_bytes.argtypes = [c_int64, c_int32, c_int32, c_int32, POINTER(c_size_t)]
        L.gs_forward_preprocess.argtypes = [POINTER(GsFwdArgs), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p,
                                            c_void_p, c_void_p]
        L.gs_forward_render.argtypes = [POINTER(GsFwdArgs), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t,
                                        c_int64, c_void_p, c_void_p]
        L.gs_forward.argtypes = [POINTER(GsFwdArgs), c_void_p, c_size_t, c_void_p, c_size_t, c_int64, c_void_p, c_size_t,
                                 c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p]
        L.gs_forward_shared.argtypes = [POINTER(GsFwdArgs), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                        c_void_p, c_size_t, c_int64, c_void_p, c_void_p]
        L.gs_backward.argtypes = [POINTER(GsFwdArgs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p,
                                  c_size_t, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(GsGrads), c_void_p]
        L.gs_opacity_image.argtypes = [POINTER(GsFwdArgs), c_void_p, c_size_t, c_void_p, c_void_p]
        L.gs_backward_with_opacity.argtypes = [POINTER(GsFwdArgs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p,
                                               c_size_t, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(GsGrads),
                                               c_void_p]
        L.gs_backward_with_second.argtypes = [POINTER(GsFwdArgs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p,
                                              c_size_t, c_int64, c_void_p, c_void_p, POINTER(GsSecondImage), c_void_p, c_size_t,
                                              POINTER(GsGrads), c_void_p]
        L.gs_mark_visible.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.knn_workspace_bytes.argtypes = [c_int32, POINTER(c_size_t)]
        L.knn_dist2.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.gs_l1_loss_workspace_bytes.argtypes = [c_int64, POINTER(c_size_t)]
        L.gs_l1_loss.argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.gs_bce_loss.argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.gs_ssim_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, POINTER(c_size_t)]
        L.gs_ssim_forward.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]
        L.gs_ssim_backward.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]
        L.gs_build_covariance.argtypes = [c_int32, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_void_p]
        L.gs_build_covariance_backward.argtypes = [c_int32, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                                   c_void_p]
        L.gs_sh2rgb.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]
        L.gs_sh2rgb_backward.argtypes = [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]
        L.knn_points.argtypes = [c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.gs_densify_stats.argtypes = [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.gs_adam_step.argtypes = [c_int32, POINTER(GsAdamTensor), c_double, c_double, c_double, c_int64, c_void_p]
        L.gs_geom_field.argtypes = [c_void_p, c_int32, c_int32, POINTER(c_void_p)]
        L.gs_binning_field.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, POINTER(c_void_p)]
        L.gs_image_field.argtypes = [c_void_p, c_int32, c_int32, c_int32, POINTER(c_void_p)]
        L.gs_clock_probe.argtypes = [c_void_p, c_int32, c_void_p]
        L.gs_xcc_probe.argtypes = [c_void_p, c_int32, c_void_p]
        L.gs_pair_stats.argtypes = [POINTER(GsFwdArgs), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_int64, c_void_p,
                                    c_void_p]
        L.gs_tuning.argtypes = [c_char_p, c_int]
        L.gs_profile_reserve.argtypes = [c_int]
        L.gs_profile_enable.argtypes = [c_int]
        L.gs_profile_filter.argtypes = [c_char_p]
        L.gs_profile_collect.argtypes = [c_int, POINTER(c_char_p), POINTER(c_float), POINTER(c_int32), POINTER(c_int32)]
        for name in EXPORTS:
            getattr(L, name).restype = c_int
        L.gs_status_string.restype = c_char_p
        L.gs_status_string.argtypes = [c_int]
        L.gs_last_stage.restype = c_char_p
        L.gs_build_info.restype = c_char_p
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        L = load()
        msg = L.gs_status_string(rc).decode()
        if rc == -4:
            msg += " %d in stage '%s'" % (L.gs_last_hip_error(), L.gs_last_stage().decode())
        raise RuntimeError("gsplat_mi355: " + msg)


tuning_listeners = []  # called after every change of a tuning switch ("small_tiles" and "fwd4" change the image state's size:
#                        whoever memoises the library's size queries forgets them here)


def tuning(name, value):
    """Process-wide tuning switch of the library (A/B measurements)."""
    check(load().gs_tuning(name.encode(), int(value)))
    for fn in tuning_listeners:
        fn()


def profile_enable(on, stage=None):
    """Per-stage hipEvent timing on/off; `stage` restricts it to one stage (two events per call)."""
    check(load().gs_profile_filter(stage.encode() if stage else None))
    check(load().gs_profile_enable(1 if on else 0))


def profile_reserve(n_events):
    """Pre-creates HIP events for the stage timers (event creation is slow: keep it out of timed regions)."""
    check(load().gs_profile_reserve(int(n_events)))


def profile_collect(max_stages=32):
    """{stage: (total_ms, launches)} since the last collect (waits for the recorded events)."""
    names = (c_char_p * max_stages)()
    ms = (c_float * max_stages)()
    cnt = (c_int32 * max_stages)()
    n = c_int32(0)
    check(load().gs_profile_collect(max_stages, names, ms, cnt, ctypes.byref(n)))
    return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(n.value)}


def xcc_probe(dev, n_blocks=4096):
    """The XCD every workgroup of a launch of `n_blocks` workgroups runs on (gs_xcc_probe), as a CPU int tensor."""
    import torch
    t = torch.zeros(int(n_blocks), dtype=torch.int32, device=dev)
    with on_device(dev):
        check(load().gs_xcc_probe(t.data_ptr(), int(n_blocks), stream_ptr(dev)))
    return t.cpu()


def clock_probe(dev, iters=8192):
    """Shader clock in Hz under a VALU-bound load (an FMA stream of ~1 ms on every SIMD), measured now on `dev`."""
    import torch
    t = torch.zeros(4, dtype=torch.int64, device=dev)
    with on_device(dev):
        check(load().gs_clock_probe(t.data_ptr(), int(iters), stream_ptr(dev)))
    c = t.cpu()
    return float(c[0]) / max(float(c[1]), 1.0) * 1.0e8


def nbytes(fn, *args):
    out = c_size_t(0)
    check(fn(*args, ctypes.byref(out)))
    return int(out.value)


class _OnDevice(object):
    __slots__ = ("idx", "prev")

    def __init__(self, idx):
        self.idx = idx
        self.prev = -1

    def __enter__(self):
        self.prev = _exchange_device(self.idx)

    def __exit__(self, *exc):
        _maybe_exchange_device(self.prev)
        return False


_exchange_device = _maybe_exchange_device = None


def on_device(dev):
    """`with on_device(dev):` -- torch.cuda.device(dev) without its Python-side index parsing (~5 us per use: as much as a
    kernel launch on a forward-only frame).  Falls back to torch.cuda.device where torch lacks the two C helpers."""
    global _exchange_device, _maybe_exchange_device
    import torch
    if _exchange_device is None:
        _exchange_device = getattr(torch.cuda, "_exchange_device", False)
        _maybe_exchange_device = getattr(torch.cuda, "_maybe_exchange_device", False)
    if dev.index is None or not _exchange_device or not _maybe_exchange_device:
        return torch.cuda.device(dev)
    return _OnDevice(dev.index)


_raw_stream = None


def stream_handle(dev):
    """The current HIP stream of `dev` as an integer handle (what torch.cuda.current_stream(dev).cuda_stream returns,
    without building a Stream object and parsing the device: ~5 us per call, seven calls per training step)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream and dev.index is not None:
        return int(_raw_stream(dev.index))
    return int(torch.cuda.current_stream(dev).cuda_stream)


def stream_ptr(dev):
    return c_void_p(stream_handle(dev))


def ptr(t):
    """Device pointer of a tensor, or None (NULL = absent) for None / empty tensors."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()
