"""diff_gaussian_rasterization -- MI355X-native implementation behind the import name the reference
uses (gaussian_renderer/__init__.py:17: `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`).

Same public surface as the upstream package of the API generation the reference's call sites pin
(12-field settings tuple ending in `prefiltered, debug`; `(color, radii)` return;
gaussian_renderer/__init__.py:85-98,121-129): `GaussianRasterizationSettings`, `GaussianRasterizer`
(with `markVisible`), `rasterize_gaussians`.  The bodies call the hand-written HIP library through
the C ABI of include/gsplat_mi355.h; there is no CPU or PyTorch fallback.
"""
import ctypes
import os
import threading
import weakref
from typing import NamedTuple

import torch
import torch.nn as nn

from gsplat_mi355 import _lib

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_tls = threading.local()
_SHARE = os.environ.get("GSPLAT_SHARE_GEOMETRY", "1") != "0"
# 0: wait for the pair count between the two forward phases, as upstream does (A/B switch; default: phase 2 is enqueued
# against the previous frame's count before the host knows this frame's)
_SPECULATE = os.environ.get("GSPLAT_SPECULATE", "1") != "0"


class _GeomEntry(object):
    """Geometry state of ONE forward call (preprocess, sorts, binning), offered to the call that follows it."""
    __slots__ = ("key", "geom", "binning", "img", "num_rendered", "capacity", "radii", "color_ref", "means2D_id",
                 "allow_second", "second", "long_lists", "__weakref__")

    def __init__(self, key, geom, binning, img, num_rendered, capacity, radii):
        self.key, self.geom, self.binning, self.img, self.radii = key, geom, binning, img, radii
        self.num_rendered, self.capacity = num_rendered, capacity  # the frame's pair count / what `binning` is carved for
        # for the fused backward of a second render of this geometry (_second_render_dependency): the producing call's
        # image (weak), its means2D, whether it can take a second image along, and what the second call left for it
        self.color_ref, self.means2D_id, self.allow_second, self.second, self.long_lists = None, None, False, None, 0

    def release(self):
        self.key = self.geom = self.binning = self.img = self.radii = None


class _GeomCache(object):
    """The reference rasterizes twice per step with identical geometry: the colour pass and, right behind it, an
    opacity pass with colours = 1 (gaussian_renderer/__init__.py:121-142).  The second call can skip preprocess,
    sorts and binning.  Sharing is restricted to exactly that pattern:

    * a hit needs the SAME tensor objects (identity, plus storage address, shape, strides and autograd version
      counter) for means3D / opacities / scales / rotations / cov3D and the camera tensors, equal scalar settings,
      the same device and stream;
    * only the call IMMEDIATELY after the one that produced the state can hit, and only once: any other forward on
      the device replaces or drops the offer;
    * the offer dies when the producing call's backward starts, or when its autograd graph is freed.  The cache
      itself holds only a weak reference; the state is owned by the producing call's autograd node, so nothing is
      retained beyond what upstream's ctx retains, and nothing at all without a graph (no_grad inference renders
      every call in full).

    A parameter update between two renders therefore never meets stale geometry: the optimiser step comes after
    the backward (which kills the offer), and the library's own raw in-place writers (FusedAdam, densify_stats)
    also bump the version counters of what they write."""

    def __init__(self):
        self.offer = {}  # device index -> weakref to the _GeomEntry on offer
        self.hits = 0    # calls served from a previous call's geometry (diagnostics / tests)

    @staticmethod
    def _sig(t):
        if t is None or t.numel() == 0:
            return None
        return (id(t), t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()))

    def key(self, settings, means3D, opacities, scales, rotations, cov3D, stream_handle=None, capturing=False):
        tensors = (means3D, opacities, scales, rotations, cov3D, settings.viewmatrix, settings.projmatrix, settings.campos)
        return dict(scalars=(int(settings.image_height), int(settings.image_width), float(settings.tanfovx),
                             float(settings.tanfovy), float(settings.scale_modifier), int(means3D.shape[0]), _TILE_RECT,
                             # the cached state is only valid in stream order: a hit must come from the same stream
                             _lib.stream_handle(means3D.device) if stream_handle is None
                             else int(stream_handle),
                             bool(capturing)),  # (a state made under capture only exists inside that graph, and vice versa)
                    sigs=[self._sig(t) for t in tensors],
                    # the objects themselves: an id() can only be trusted while its object is alive
                    objs=[t for t in tensors if t is not None and t.numel() > 0])

    def take(self, dev, key):
        """The offer for this device if it matches `key` (consumed either way: single use)."""
        ref = self.offer.pop(dev.index, None)
        e = ref() if ref is not None else None
        if e is None or e.key is None:
            return None
        if e.key["scalars"] != key["scalars"] or e.key["sigs"] != key["sigs"]:
            return None
        self.hits += 1
        return e

    def put(self, dev, entry):
        self.offer[dev.index] = weakref.ref(entry)

    def clear(self):
        self.offer.clear()


_geom_cache = _GeomCache()


def release_shared_geometry():
    """Withdraws any geometry state on offer to a following call (it is otherwise withdrawn by the next forward, by the
    producing call's backward, or when its autograd graph is freed)."""
    _geom_cache.clear()


_sizes = {}  # memo of the library's size queries (pure functions of their arguments and of the tuning switches)
_lib.tuning_listeners.append(_sizes.clear)


def _size(fn_name, *args):
    k = (fn_name,) + args
    v = _sizes.get(k)
    if v is None:
        if len(_sizes) > 4096:
            _sizes.clear()
        v = _sizes[k] = _lib.nbytes(getattr(_lib.load(), fn_name), *args)
    return v


_last_count = {}  # (device, P, W, H) -> num_rendered of the previous call: a sizing hint only


def _capacity_for(count):
    """Pairs to carve the binning state (and the backward scratch) for, given the previous frame's pair count: 1/8 of
    slack, rounded UP to a multiple of 2^(bit length - 5) -- steps of 3-6 %.  Consecutive frames of a sequence, whose
    counts drift by a fraction of a percent, then ask the caching allocator for the SAME sizes frame after frame (the
    scratch is gigabytes at 500k Gaussians / 2048^2: a new size every frame makes it split and re-allocate blocks)."""
    want = count + count // 8
    step = 1 << max(want.bit_length() - 5, 0)
    return (want + step - 1) // step * step


_is_capturing = None


def _stream_capturing():
    """True while the current stream is being captured into a graph (torch.cuda.graph): the forward then takes the
    capture-safe two-phase form (kernel launches only, no host wait for the pair count)."""
    global _is_capturing
    if _is_capturing is None:
        _is_capturing = getattr(torch._C, "_cuda_isCurrentStreamCapturing", None) or torch.cuda.is_current_stream_capturing
    return bool(_is_capturing())


# forwards issued under stream capture since the last call of captured_forwards(clear=True): what a caller needs to find
# out, after a replay, whether the frame's pair count fitted the capacity the graph was captured with
_captured = []


def captured_forwards(clear=False):
    """[(pair_count, capacity)]: for every rasterizer forward captured into a graph, a device int64 tensor (one element,
    a view of that call's geometry state: the frame's num_rendered after a replay) and the fixed pair capacity of its
    binning state.  A replay whose count exceeds the capacity has rendered an EMPTY frame (nothing out of bounds):
    compare after synchronising, outside the graph, and re-capture with `GSPLAT_CAPTURE_SLACK` raised if it happens."""
    out = list(_captured)
    if clear:
        del _captured[:]
    return out


_CAPTURE_SLACK = float(os.environ.get("GSPLAT_CAPTURE_SLACK", "0.25"))  # head-room over the last eager frame's pair count


def _pinned_count(device):
    """A per-thread, per-device pinned int64 the library copies num_rendered into."""
    cache = getattr(_tls, "pinned", None)
    if cache is None:
        cache = _tls.pinned = {}
    key = device.index
    if key not in cache:
        cache[key] = torch.zeros(1, dtype=torch.int64).pin_memory()
    return cache[key]


def _f32c(t, name):
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization: `%s` must be a GPU tensor (this build has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("diff_gaussian_rasterization: `%s` must be float32" % name)
    return t.contiguous()


# Which tiles a Gaussian is binned into (GsFwdArgs.tile_rect): 1 = the bounding box of its alpha >= 1/255 region
# (default: same outputs, ~40 % fewer (tile, Gaussian) pairs), 0 = upstream's 3-sigma square (GSPLAT_TILE_RECT=0).
_TILE_RECT = int(os.environ.get("GSPLAT_TILE_RECT", "1"))


# GsFwdArgs.long_lists: the machinery for frames of few, long tile lists (a trained avatar: four waves per quadrant in the
# forward on the tiles whose list is long against the frame's total, backward in chunks) is always used on images of up
# to 2048 tiles.  On larger ones it is a setting fixed for the process -- GSPLAT_LONG_LISTS = "0" (default) / "1" -- so that
# a frame's bits never depend on what was rendered before it (the two settings agree to fp32 rounding, not bit for bit).
# "auto" is an opt-in: on when the PREVIOUS frame of the same shape had such tiles (the forward reports them in a pinned
# word that is read without waiting for that frame) -- faster on avatar-shaped frames above 2048 tiles, but the first
# frame of a shape can then differ from the following ones in the last bits.
_LONG_LISTS = os.environ.get("GSPLAT_LONG_LISTS", "0")
_frame_stats = {}  # (device index, W, H) -> pinned int64[2]: [tiles with a long list, longest list] of the last frame


def _stats_words(key):
    st = _frame_stats.get(key)
    if st is None:
        st = _frame_stats[key] = torch.zeros(2, dtype=torch.int64).pin_memory()
    return st


def _make_args(settings, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, keep, long_lists=None):
    a = _lib.GsFwdArgs()
    P = int(means3D.shape[0])
    a.P = P
    a.sh_degree = int(settings.sh_degree)
    a.M = int(sh.shape[1]) if sh is not None else 0
    a.W, a.H = int(settings.image_width), int(settings.image_height)
    dev = means3D.device
    bg = _f32c(settings.bg.to(dev), "bg")
    view = _f32c(settings.viewmatrix.to(dev), "viewmatrix")
    proj = _f32c(settings.projmatrix.to(dev), "projmatrix")
    campos = _f32c(settings.campos.to(dev), "campos")
    keep.extend([bg, view, proj, campos])
    a.bg, a.viewmatrix, a.projmatrix, a.campos = bg.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr()
    a.means3D = _lib.ptr(means3D)
    a.shs = _lib.ptr(sh)
    a.colors_precomp = _lib.ptr(colors_precomp)
    a.opacities = _lib.ptr(opacities)
    a.scales = _lib.ptr(scales)
    a.rotations = _lib.ptr(rotations)
    a.cov3D_precomp = _lib.ptr(cov3Ds_precomp)
    a.scale_modifier = float(settings.scale_modifier)
    a.tanfovx, a.tanfovy = float(settings.tanfovx), float(settings.tanfovy)
    a.prefiltered = int(bool(settings.prefiltered))
    a.debug = int(bool(settings.debug))
    a.tile_rect = _TILE_RECT
    st = _stats_words((dev.index, a.W, a.H))  # (not keyed by P: densification changes it every few hundred steps)
    if long_lists is None:  # a forward: decided from the previous frame of this shape (stale or missing words: a guess as good)
        long_lists = 1 if (_LONG_LISTS == "1" or (_LONG_LISTS == "auto" and int(st[0]) > 0)) else 0
        a.frame_stats = st.data_ptr()
    a.long_lists = int(long_lists)
    return a


def _dump_snapshot(path, raster_settings, tensors):
    """What upstream's debug mode does when the native call raises: the call's arguments, copied to the CPU, are saved
    with torch.save (upstream: snapshot_fw.dump / snapshot_bw.dump in the working directory)."""
    try:
        cpu = [t.detach().cpu().clone() if isinstance(t, torch.Tensor) else t for t in tensors]
        settings = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in raster_settings._asdict().items()}
        torch.save({"settings": settings, "tensors": cpu}, path)
    except Exception:  # the dump is best effort: the original error is what the caller must see
        pass


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, with_opacity=False, l1_target=None):
    """(color, radii), as upstream.  with_opacity=True (an extension; see GaussianRasterizer.forward) adds the
    opacity render as a third result, computed and differentiated inside the same pass; l1_target (an extension, too)
    adds mean |color - l1_target| as the last result, computed by the render launch and differentiated in the
    backward's own pixel prologue."""
    if l1_target is not None:
        l1_target = _f32c(l1_target, "l1_target")
        if tuple(l1_target.shape) != (3, int(raster_settings.image_height), int(raster_settings.image_width)):
            raise ValueError("diff_gaussian_rasterization: l1_target must be a (3, H, W) image")
    dep = None
    if (_FUSE_SECOND and _SHARE and not with_opacity and torch.is_grad_enabled() and colors_precomp is not None
            and colors_precomp.numel() and not colors_precomp.requires_grad and (sh is None or sh.numel() == 0)):
        dep = _second_render_dependency(means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
    if not torch.is_grad_enabled():
        # no_grad (the reference's render loop, render.py:51-62): no autograd node would be built, and Function.apply
        # costs ~10 us of Python per call for building none -- a tenth of a forward-only frame at 50k Gaussians
        color, radii, opacity, l1 = _RasterizeGaussians.forward(_InferenceCtx(), means3D, means2D, sh, colors_precomp, opacities,
                                                                scales, rotations, cov3Ds_precomp, raster_settings,
                                                                bool(with_opacity), None, l1_target)
    else:
        color, radii, opacity, l1 = _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                                               rotations, cov3Ds_precomp, raster_settings, bool(with_opacity),
                                                               dep, l1_target)
    out = (color, radii, opacity) if with_opacity else (color, radii)
    return out + (l1,) if l1_target is not None else out


# The reference's render() rasterizes the opacity image with a second call on the same geometry, colours = constant ones
# (gaussian_renderer/__init__.py:132-142), and each call has its own backward: two full passes over the tile lists.  The
# two images share alpha and T, so their gradients w.r.t. the shared inputs can come out of ONE pass
# (gs_backward_with_second).  When the second call meets the first call's geometry on offer and its colours need no
# gradient, it takes the FIRST call's image as an extra autograd input: its own backward then runs before the first
# call's (a true dependency, not an ordering accident), hands its gradient image over and returns no gradients; the first
# call's backward returns the sum of both.  GSPLAT_FUSE_SECOND_BACKWARD=0 keeps the two backwards apart.
_FUSE_SECOND = os.environ.get("GSPLAT_FUSE_SECOND_BACKWARD", "1") != "0"


def _second_render_dependency(means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, settings):
    """The image of the call that just ran, if THIS call is a second render of its geometry (what `take` will find)."""
    if not means3D.is_cuda:
        return None
    ref = _geom_cache.offer.get(means3D.device.index)
    e = ref() if ref is not None else None
    if e is None or e.key is None or not e.allow_second or e.color_ref is None or e.means2D_id != id(means2D):
        return None
    img1 = e.color_ref()
    if img1 is None or not img1.requires_grad:
        return None
    def c(t):
        return t if (t is None or t.numel() == 0 or (t.dtype == torch.float32 and t.is_contiguous())) else None
    if any(c(t) is None and t is not None and t.numel() for t in (means3D, opacities, scales, rotations, cov3Ds_precomp)):
        return None  # (the forward would render from converted copies: no identity to match)
    key = _geom_cache.key(settings, means3D, opacities, scales, rotations, cov3Ds_precomp)
    if e.key["scalars"] != key["scalars"] or e.key["sigs"] != key["sigs"]:
        return None
    return img1


class _InferenceCtx(object):
    """What _forward / _finish touch of an autograd context, for a call under no_grad: nothing is saved, no node exists."""
    needs_input_grad = (False,) * 12

    def set_materialize_grads(self, value):
        pass

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, with_opacity=False, dep=None, l1_target=None):
        if not raster_settings.debug:
            return _RasterizeGaussians._forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                                cov3Ds_precomp, raster_settings, with_opacity, dep, l1_target)
        # upstream's debug mode: the arguments are kept aside and written to snapshot_fw.dump if the native call fails
        args = (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        try:
            return _RasterizeGaussians._forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                                cov3Ds_precomp, raster_settings, with_opacity, dep, l1_target)
        except Exception:
            _dump_snapshot("snapshot_fw.dump", raster_settings, args)
            print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise

    @staticmethod
    def _forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                 raster_settings, with_opacity=False, dep=None, l1_target=None):
        ctx.with_opacity = bool(with_opacity)
        ctx.l1_target = None
        l1 = None  # (never kept on ctx: an output held by its own node is a reference cycle around the whole saved state)
        ctx.defer_to = None
        means2D_in = means2D
        # outputs nothing downstream differentiates (always: radii) arrive as None in backward instead of as freshly
        # filled zero tensors -- one fill launch over P ints per step otherwise
        ctx.set_materialize_grads(False)
        L = _lib.load()
        means3D = _f32c(means3D, "means3D")
        if means3D is None:
            raise RuntimeError("diff_gaussian_rasterization: means3D is empty")
        sh = _f32c(sh, "shs")
        colors_precomp = _f32c(colors_precomp, "colors_precomp")
        opacities = _f32c(opacities, "opacities")
        scales = _f32c(scales, "scales")
        rotations = _f32c(rotations, "rotations")
        cov3Ds_precomp = _f32c(cov3Ds_precomp, "cov3D_precomp")
        dev = means3D.device
        P = int(means3D.shape[0])
        W, H = int(raster_settings.image_width), int(raster_settings.image_height)
        keep = []
        with _lib.on_device(dev):
            a = _make_args(raster_settings, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                           keep)
            stream_h = _lib.stream_handle(dev)
            sptr = ctypes.c_void_p(stream_h)
            capturing = _stream_capturing()
            if capturing:
                a.frame_stats = None  # (a kernel storing into pinned host memory is not what a graph should replay)
            if l1_target is not None:
                # the fused L1 loss: its value is a by-product of the render launch, its gradient of the backward's prologue
                ctx.l1_target = l1_target
                l1 = torch.empty((), dtype=torch.float32, device=dev)
                a.l1_target, a.l1_loss = l1_target.data_ptr(), l1.data_ptr()
            # sharing needs an autograd node to own the state (see _GeomCache): without one every call renders in full
            # (needs_input_grad reflects the inputs' requires_grad flags also under no_grad -- render()'s means2D leaf
            # always has one -- where no node exists to own anything: inference frames skip the bookkeeping altogether)
            # (whether a node exists is read off the context's type -- an _InferenceCtx is what rasterize_gaussians passes
            # under no_grad -- not off thread-local state a direct _RasterizeGaussians.apply call would find stale)
            inference = isinstance(ctx, _InferenceCtx)
            if inference:
                a.forward_only = 1  # no backward can follow: the render launch need not prepare the backward's row marks
            share = _SHARE and not inference and any(ctx.needs_input_grad)
            gkey = (_geom_cache.key(raster_settings, means3D, opacities, scales, rotations, cov3Ds_precomp, stream_h,
                                    capturing)
                    if share else None)
            hit = _geom_cache.take(dev, gkey) if (share and l1_target is None) else None  # (a shared-geometry render has no fused loss)
            if not share or l1_target is not None:
                _geom_cache.offer.pop(dev.index, None)
            ctx.geom_entry = None
            if hit is not None:
                # a second render of that call's geometry runs with ITS long_lists (the statistics word may have changed
                # since): the two image states then have one layout, which the all-ones render and the one-pass backward need
                a.long_lists = int(hit.long_lists)
            geom_bytes = _size("gs_geom_bytes", P)
            ik = ("img", W, H, int(a.long_lists))
            img_bytes = _sizes.get(ik)
            if img_bytes is None:
                img_bytes = _sizes[ik] = _lib.nbytes(L.gs_image_bytes_for, ctypes.byref(a))
            geom = torch.empty(geom_bytes, dtype=torch.uint8, device=dev)
            img = torch.empty(img_bytes, dtype=torch.uint8, device=dev)
            radii = torch.empty(P, dtype=torch.int32, device=dev)
            if hit is not None:
                # same geometry and camera as the call just before: new colours only
                num_rendered, capacity = hit.num_rendered, hit.capacity
                binning = hit.binning
                if (dep is not None and hit.allow_second and hit.color_ref is not None and hit.color_ref() is dep
                        and hit.long_lists == int(a.long_lists)):
                    ctx.defer_to = hit  # this call's backward hands its gradient to the producing call's (below)
                bin_bytes = binning.numel()
                radii.copy_(hit.radii)
                color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
                check_qlist = binning.clone() if (raster_settings.debug and bin_bytes) else None
                _lib.check(L.gs_forward_shared(ctypes.byref(a), hit.geom.data_ptr(), hit.img.data_ptr(),
                                               geom.data_ptr(), geom_bytes, binning.data_ptr(), bin_bytes, img.data_ptr(),
                                               img_bytes, capacity, color.data_ptr(), sptr))
                if check_qlist is not None and not torch.equal(check_qlist, binning):
                    # the shared binning state belongs to the first call's autograd node as well: the second forward
                    # re-records the quadrant lists, which must come out identical for identical geometry
                    raise RuntimeError("diff_gaussian_rasterization: the shared-geometry render changed the binning "
                                       "state it shares with the previous call")
                return _RasterizeGaussians._finish(ctx, raster_settings, num_rendered, capacity, means3D, sh, colors_precomp,
                                                   opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning,
                                                   img, color, dev, a, sptr, keep)
            color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
            if capturing:
                # hipGraph capture (torch.cuda.graph): gs_forward waits for the pair count on the host, which a capture
                # cannot do.  The capture-safe form is the two-phase pair at a FIXED capacity -- kernel nodes only; the
                # count stays on the device (captured_forwards() hands out a view of it)
                ck = (dev.index, P, W, H)
                if ck not in _last_count:
                    raise RuntimeError("diff_gaussian_rasterization: run one eager frame of this shape (P=%d, %dx%d) before "
                                       "capturing it into a graph: the binning state of a captured frame has a fixed "
                                       "capacity, sized from the last eager frame's pair count" % (P, W, H))
                guess = _last_count[ck]
                capacity = _capacity_for(max(int(guess * (1.0 + _CAPTURE_SLACK)), 1024))
                bin_bytes = _size("gs_binning_bytes", capacity, W, H)
                binning = torch.empty(bin_bytes, dtype=torch.uint8, device=dev)
                _lib.check(L.gs_forward_preprocess(ctypes.byref(a), geom.data_ptr(), geom_bytes, img.data_ptr(), img_bytes,
                                                   radii.data_ptr(), None, sptr))
                _lib.check(L.gs_forward_render(ctypes.byref(a), geom.data_ptr(), geom_bytes, binning.data_ptr(), bin_bytes,
                                               img.data_ptr(), img_bytes, capacity, color.data_ptr(), sptr))
                cptr = ctypes.c_void_p()
                _lib.check(L.gs_geom_field(geom.data_ptr(), P, 5, ctypes.byref(cptr)))
                off = int(cptr.value) - geom.data_ptr()
                _captured.append((geom[off:off + 8].view(torch.int64), capacity))
                num_rendered = capacity  # (the frame's own count is only known on the device)
                if share:
                    ctx.geom_entry = _GeomEntry(gkey, geom, binning, img, num_rendered, capacity, radii)
                    ctx.geom_entry.color_ref = weakref.ref(color)
                    ctx.geom_entry.means2D_id = id(means2D_in)
                    ctx.geom_entry.allow_second = not ctx.with_opacity
                    ctx.geom_entry.long_lists = int(a.long_lists)
                    _geom_cache.put(dev, ctx.geom_entry)
                return _RasterizeGaussians._finish(ctx, raster_settings, num_rendered, capacity, means3D, sh, colors_precomp,
                                                   opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning, img,
                                                   color, dev, a, sptr, keep, l1)
            count = _pinned_count(dev)
            # The binning state is sized by the pair count, which only phase 1 produces.  A buffer for the
            # previous frame's count (+ 1/8) is handed to gs_forward, which enqueues phase 2 right behind phase 1
            # against that capacity (the kernels read the count on the device) and only then waits for the count:
            # the GPU never idles for it.  Only when the count exceeds the capacity (or there is no estimate yet)
            # does control come back here to allocate and run phase 2 again.
            guess = _last_count.get((dev.index, P, W, H), 0)
            capacity = _capacity_for(guess) if (guess and _SPECULATE) else 0
            bin_bytes = _size("gs_binning_bytes", capacity, W, H) if capacity else 0
            binning = torch.empty(bin_bytes, dtype=torch.uint8, device=dev) if capacity else None
            nr = ctypes.c_int64(0)
            rc = L.gs_forward(ctypes.byref(a), geom.data_ptr(), geom_bytes, _lib.ptr(binning), bin_bytes, capacity,
                              img.data_ptr(), img_bytes, radii.data_ptr(), count.data_ptr(), color.data_ptr(),
                              ctypes.byref(nr), sptr)
            num_rendered = int(nr.value)
            if rc == _lib.GS_E_WORKSPACE:
                capacity = num_rendered
                bin_bytes = _size("gs_binning_bytes", capacity, W, H)
                binning = torch.empty(bin_bytes, dtype=torch.uint8, device=dev)
                rc = L.gs_forward_render(ctypes.byref(a), geom.data_ptr(), geom_bytes, binning.data_ptr(), bin_bytes,
                                         img.data_ptr(), img_bytes, capacity, color.data_ptr(), sptr)
            _lib.check(rc)
            if len(_last_count) > 256:  # (P changes with every densification: keep the table small)
                _last_count.clear()
            _last_count[(dev.index, P, W, H)] = num_rendered
            if share:
                # offered to the next call; owned by this call's autograd node (ctx), not by the cache
                ctx.geom_entry = _GeomEntry(gkey, geom, binning, img, num_rendered, capacity, radii)
                ctx.geom_entry.color_ref = weakref.ref(color)  # (the tensor this call returns: autograd tracks it)
                ctx.geom_entry.means2D_id = id(means2D_in)
                ctx.geom_entry.allow_second = not ctx.with_opacity
                ctx.geom_entry.long_lists = int(a.long_lists)
                _geom_cache.put(dev, ctx.geom_entry)
            return _RasterizeGaussians._finish(ctx, raster_settings, num_rendered, capacity, means3D, sh, colors_precomp,
                                               opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning, img, color,
                                               dev, a, sptr, keep, l1)

    @staticmethod
    def _finish(ctx, raster_settings, num_rendered, capacity, means3D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, radii, geom, binning, img, color, dev, a, sptr, keep=None, l1=None):
        opacity = None
        if ctx.with_opacity:
            # the opacity render is (1 - final_T) + final_T * bg[0]: the forward that just ran holds final_T
            L = _lib.load()
            H, W = int(raster_settings.image_height), int(raster_settings.image_width)
            opacity = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            _lib.check(L.gs_opacity_image(ctypes.byref(a), img.data_ptr(), img.numel(), opacity.data_ptr(), sptr))
        ctx.raster_settings = raster_settings
        ctx.long_lists = int(a.long_lists)  # the backward must be told the same (GsFwdArgs.long_lists)
        ctx.fwd_args = (a, keep)  # the argument block (pointers into the saved tensors) serves the backward as it is
        ctx.num_rendered = num_rendered  # the frame's pair count (upstream's num_rendered)
        ctx.capacity = capacity          # pairs the binning state is carved for (>= num_rendered): what the backward is given
        ctx.present = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(means3D, sh if sh is not None else empty,
                              colors_precomp if colors_precomp is not None else empty, opacities,
                              scales if scales is not None else empty, rotations if rotations is not None else empty,
                              cov3Ds_precomp if cov3Ds_precomp is not None else empty, radii, geom, binning, img, color)
        ctx.mark_non_differentiable(radii)
        return color, radii, opacity, l1

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, grad_out_opacity=None, grad_l1=None):
        if not ctx.raster_settings.debug:
            return _RasterizeGaussians._backward(ctx, grad_out_color, _grad_radii, grad_out_opacity, grad_l1)
        try:
            return _RasterizeGaussians._backward(ctx, grad_out_color, _grad_radii, grad_out_opacity, grad_l1)
        except Exception:
            _dump_snapshot("snapshot_bw.dump", ctx.raster_settings, tuple(ctx.saved_tensors[:8]) + (grad_out_color,))
            print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
            raise

    @staticmethod
    def _backward(ctx, grad_out_color, _grad_radii, grad_out_opacity=None, grad_l1=None):
        L = _lib.load()
        tgt = getattr(ctx, "defer_to", None)
        if tgt is not None and tgt.geom is not None and grad_out_color is not None:
            # a second render of another call's geometry (see _second_render_dependency): that call's backward, which runs
            # after this one, differentiates both images in one pass
            saved = ctx.saved_tensors
            tgt.second = dict(colors=saved[2], out_color=saved[11], grad=_f32c(grad_out_color, "grad_out_color"), img=saved[10],
                              long_lists=ctx.long_lists)
            return (None,) * 12
        entry = getattr(ctx, "geom_entry", None)
        second = None
        if entry is not None:  # a render after this backward (e.g. after an optimiser step) must not meet this state
            second, entry.second = entry.second, None
            entry.release()
            ctx.geom_entry = None
        (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning, img,
         color) = ctx.saved_tensors
        has_sh, has_col, has_sr, has_cov = ctx.present
        settings = ctx.raster_settings
        dev = means3D.device
        P = int(means3D.shape[0])
        W, H = int(settings.image_width), int(settings.image_height)
        D = ctx.capacity
        use_l1 = ctx.l1_target is not None and grad_l1 is not None
        if grad_out_color is None and not use_l1:  # only the opacity render was used downstream
            grad_out_color = torch.zeros(3, H, W, dtype=torch.float32, device=dev)
        g = _f32c(grad_out_color, "grad_out_color") if grad_out_color is not None else None
        g_l1 = _f32c(grad_l1, "grad_l1") if use_l1 else None
        g_op = _f32c(grad_out_opacity, "grad_out_opacity") if (ctx.with_opacity and grad_out_opacity is not None) else None
        with _lib.on_device(dev):
            a, _keep = ctx.fwd_args  # the forward's argument block: the same tensors (saved above), the same long_lists
            if (a.means3D != means3D.data_ptr() or a.opacities != opacities.data_ptr()
                    or (has_sh and a.shs != sh.data_ptr()) or (has_sr and a.scales != scales.data_ptr())):
                # the saved tensors came back in other storage (saved-tensor hooks: offloading, checkpointing)
                _keep = []
                a = _make_args(settings, means3D, sh if has_sh else None, colors_precomp if has_col else None, opacities,
                               scales if has_sr else None, rotations if has_sr else None,
                               cov3Ds_precomp if has_cov else None, _keep, long_lists=ctx.long_lists)
            # the fused L1 loss: dL/d(image) of it is formed per pixel inside the backward's render pass (no gradient image)
            a.l1_target = ctx.l1_target.data_ptr() if use_l1 else None
            a.l1_loss = None  # (the forward's output; the backward does not touch it)
            a.l1_grad = g_l1.data_ptr() if use_l1 else None
            gp = g.data_ptr() if g is not None else None
            sptr = _lib.stream_ptr(dev)
            scratch_bytes = _size("gs_backward_scratch_bytes", D, P, W, H)
            scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
            M = int(sh.shape[1]) if has_sh else 0
            f = dict(dtype=torch.float32, device=dev)
            d_means3D = torch.empty(P, 3, **f)
            d_means2D = torch.empty(P, 3, **f)
            d_colors = torch.empty(P, 3, **f)
            d_opacity = torch.empty(P, 1, **f)
            d_cov3D = torch.empty(P, 6, **f)
            d_sh = torch.empty(P, M, 3, **f) if has_sh else None
            d_scales = torch.empty(P, 3, **f) if has_sr else None
            d_rot = torch.empty(P, 4, **f) if has_sr else None
            gr = _lib.GsGrads(_lib.ptr(d_means3D), _lib.ptr(d_means2D), _lib.ptr(d_sh), _lib.ptr(d_colors),
                              _lib.ptr(d_opacity), _lib.ptr(d_scales), _lib.ptr(d_rot), _lib.ptr(d_cov3D))
            if second is not None and g_op is None:
                si = _lib.GsSecondImage(second["colors"].data_ptr(), second["out_color"].data_ptr(), second["grad"].data_ptr(),
                                        second["img"].data_ptr(), second["img"].numel(), int(second["long_lists"]))
                _lib.check(L.gs_backward_with_second(ctypes.byref(a), radii.data_ptr(), geom.data_ptr(), geom.numel(),
                                                     binning.data_ptr(), binning.numel(), img.data_ptr(), img.numel(), D,
                                                     color.data_ptr(), gp, ctypes.byref(si), scratch.data_ptr(),
                                                     scratch_bytes, ctypes.byref(gr), sptr))
            elif g_op is None:
                _lib.check(L.gs_backward(ctypes.byref(a), radii.data_ptr(), geom.data_ptr(), geom.numel(),
                                         binning.data_ptr(), binning.numel(), img.data_ptr(), img.numel(), D,
                                         color.data_ptr(), gp, scratch.data_ptr(), scratch_bytes,
                                         ctypes.byref(gr), sptr))
            else:
                # the opacity render's gradient rides along as a fourth channel of the same backward pass
                _lib.check(L.gs_backward_with_opacity(ctypes.byref(a), radii.data_ptr(), geom.data_ptr(), geom.numel(),
                                                      binning.data_ptr(), binning.numel(), img.data_ptr(), img.numel(), D,
                                                      color.data_ptr(), gp, g_op.data_ptr(), scratch.data_ptr(),
                                                      scratch_bytes, ctypes.byref(gr), sptr))
        return (d_means3D, d_means2D, d_sh, d_colors if has_col else None, d_opacity, d_scales, d_rot,
                d_cov3D if has_cov else None, None, None, None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: z_view > 0.2 (upstream mark_visible / checkFrustum)."""
        L = _lib.load()
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions, "positions")
            P = int(pos.shape[0])
            out = torch.zeros(P, dtype=torch.uint8, device=pos.device)
            view = _f32c(rs.viewmatrix.to(pos.device), "viewmatrix")
            proj = _f32c(rs.projmatrix.to(pos.device), "projmatrix")
            with _lib.on_device(pos.device):
                sptr = _lib.stream_ptr(pos.device)
                _lib.check(L.gs_mark_visible(P, pos.data_ptr(), view.data_ptr(), proj.data_ptr(), _lib.ptr(out), sptr))
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, with_opacity=False, l1_target=None):
        """Upstream signature and results: (color[3,H,W], radii[N]).  `with_opacity=True` is an extension: a third
        result, the opacity render [1,H,W] -- what the reference gets from a second call with colours = 1
        (gaussian_renderer/__init__.py:132-142, `[:1]`) -- produced and differentiated inside the same pass.
        `l1_target` (a (3,H,W) image) is an extension, too: a LAST result, the scalar mean |color - l1_target| -- the
        reference's `l1_loss(image, gt_image)` (train.py:121, utils/loss_utils.py:21-22) -- computed by the render launch
        from the colours it has just composited; its gradient w.r.t. the image is formed per pixel inside the backward
        (added to whatever gradient `color` itself receives), so no gradient image is written or read for it."""
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings, with_opacity=with_opacity, l1_target=l1_target)
