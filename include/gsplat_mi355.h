/*
 * gsplat_mi355.h -- C ABI of libgsplat_mi355.so: the MI355X (gfx950) differentiable
 * Gaussian-splat rasterizer and the distCUDA2 K-NN initialiser.
 *
 * Drop-in boundary.  The reference reaches this path through two Python imports of
 * third-party torch C++ extensions whose source is NOT vendored in the reference tree:
 *   - `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
 *     (gaussian_renderer/__init__.py:17; settings built at :85-98, module at :100, calls at
 *     :121-129 and :133-141)  ->  upstream `_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`,
 *     `_C.mark_visible`
 *   - `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20, call at :186)
 * Each entry point below names the upstream binding it replaces.  Plain pointers and sizes only:
 * no torch types.  Every pointer is a DEVICE pointer unless it says "host".  The library never
 * allocates or frees device memory and never synchronises the stream except where stated; all
 * work is enqueued on the caller's `stream` (a hipStream_t passed as void*), so calls are
 * re-entrant across streams and devices (the caller selects the device).
 *
 * All entry points return 0 on success, a negative GS_E_* code on failure; `gs_status_string`
 * turns a code into text.  A NULL optional pointer means "absent", mirroring the upstream
 * wrapper's empty-tensor convention (gaussian_renderer/__init__.py:107-129).
 */
#ifndef GSPLAT_MI355_H
#define GSPLAT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK 0
#define GS_E_BAD_ARG (-1)      /* NULL required pointer, non-positive size, bad sh degree, M > 16, rotations / dL_drotations not 16-byte aligned */
#define GS_E_EXCLUSIVE (-2)    /* both / neither of (shs | colors_precomp) or (scales+rotations | cov3D_precomp) */
#define GS_E_TOO_LARGE (-3)    /* num_rendered or tile count exceeds the 32-bit index space  */
#define GS_E_HIP (-4)          /* a HIP launch or API call failed (see gs_last_hip_error)    */
#define GS_E_WORKSPACE (-5)    /* a caller-provided buffer is smaller than gs_*_bytes says    */
#define GS_E_CAPTURE (-6)      /* `stream` is being captured into a hipGraph and this call, with these arguments, is not
                                * capture-safe (see "Stream capture" below); nothing was enqueued */

/* ---- Stream capture (hipGraph).  Every entry point that takes a stream asks hipStreamIsCapturing first.  A call that
 * would wait for the GPU, copy to or store into host memory, or enqueue anything but kernel launches returns
 * GS_E_CAPTURE without enqueuing anything -- it never leaves the capture half-built or faults on replay.  Capture-safe
 * (kernel launches only, every pointer a device pointer, no host wait):
 *   gs_forward_preprocess  with count_host_pinned == NULL (the count stays in the geom state: gs_geom_field 5)
 *   gs_forward_render      with a->frame_stats == NULL, at a fixed capacity (a frame whose count exceeds it renders
 *                          empty: read the count afterwards, outside the graph)
 *   gs_forward_shared (P > 0), gs_opacity_image, gs_backward, gs_backward_with_opacity, gs_backward_with_second,
 *   gs_mark_visible, gs_l1_loss, gs_bce_loss, gs_ssim_*, gs_build_covariance*, gs_sh2rgb* (view_noise_host == NULL),
 *   gs_densify_stats
 * -- all of them with a->debug == 0 and the stage timer (gs_profile_enable) off.  Not capture-safe: gs_forward (it waits
 * for the pair count on the host), gs_adam_step (the step number is a host scalar: a replay would repeat the captured
 * step's bias correction), knn_dist2 / knn_points (their sorts clear tables with memset nodes: untested under replay),
 * anything in debug mode. */

/* Arguments of one rasterizer call: the fields of GaussianRasterizationSettings
 * (gaussian_renderer/__init__.py:85-98) plus the tensors of GaussianRasterizer.forward
 * (:121-129), flattened.  Shapes (fp32, contiguous): means3D[P,3], opacities[P], shs[P,M,3]
 * (coefficient-major, channel-minor; scene/gaussian_model.py:145-148), colors_precomp[P,3],
 * scales[P,3], rotations[P,4] (w,x,y,z; utils/general_utils.py:94-97; 16-byte aligned), cov3D_precomp[P,6]
 * ([xx,xy,xz,yy,yz,zz]; utils/general_utils.py:73-85), viewmatrix[16] and projmatrix[16]
 * (row-vector convention, scene/cameras.py:35-39), campos[3], bg[3]. */
typedef struct GsFwdArgs {
    int32_t P;          /* number of Gaussians                                  */
    int32_t sh_degree;  /* active SH degree 0..3 (settings.sh_degree)           */
    int32_t M;          /* SH coefficients present per channel: shs.shape[1]    */
    int32_t W, H;       /* image_width, image_height                            */
    const float* bg;
    const float* means3D;
    const float* shs;            /* NULL if colors_precomp given */
    const float* colors_precomp; /* NULL if shs given            */
    const float* opacities;
    const float* scales;         /* NULL if cov3D_precomp given  */
    const float* rotations;      /* NULL if cov3D_precomp given  */
    const float* cov3D_precomp;  /* NULL if scales/rotations     */
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    float scale_modifier, tanfovx, tanfovy;
    int32_t prefiltered; /* accepted for API parity; culled points are skipped either way */
    int32_t debug;       /* !=0: synchronise + check after every kernel, name the failing stage */
    int32_t tile_rect;   /* which tiles a Gaussian is binned into.  0: upstream's square of half-width ceil(3 sigma_max)
                          * around the centre.  1: the bounding box of the region where its alpha can reach 1/255
                          * (half-widths sqrt(2 ln(255 opacity) Sigma_xx), sqrt(... Sigma_yy)), intersected with the
                          * square: every tile left out contributes nothing to any pixel, so colour, radii and all
                          * gradients are those of mode 0 while num_rendered and the tile lists are ~40 % shorter */
    int32_t long_lists;  /* 0: the machinery for frames of few, long tile lists (four waves per quadrant in the forward on
                          * the tiles whose list is long against the frame's total, backward in chunks from checkpoints of
                          * the forward) is used on images of up to 2048 tiles only.  1: on this image whatever its size --
                          * the image state is then larger (gs_image_bytes_for).  A frame that fills the chip with one wave
                          * per quadrant is ~10 % slower with it, one of few long lists (a trained avatar filling a sixth of
                          * 1024 x 1024) 1.45 x faster; outputs agree to fp32 rounding.  The SAME value must be passed to the
                          * backward (and to gs_forward_shared) of a forward */
    int64_t* frame_stats; /* NULL, or two words the forward writes (device-visible host memory or device memory), for the
                          * caller to choose long_lists for the NEXT frame: [0] tiles whose list is long against this
                          * frame's total (the tiles the four-wave forward takes), [1] the longest tile list */
    /* ---- L1 image loss fused into the rasterizer (train.py:121 `Ll1 = l1_loss(image, gt_image)`,
     * utils/loss_utils.py:21-22).  l1_target (NULL = off): the target image [3,H,W].  The forward's render launch then also
     * adds up |out_color - l1_target| over the pixels it has just composited and l1_loss[0] (device float, required with
     * l1_target) receives the mean over the 3 H W elements -- the image is not read again.  The backward (the same
     * argument block) forms dL/d out_color of that loss itself, per pixel, in the prologue of its render pass:
     * sign(out_color - l1_target) / (3 H W) times l1_grad[0] (device float = dLoss/d l1_loss; NULL = 1), ADDED to the
     * dL_dpix the caller passes (which may then be NULL) -- no gradient image is written or read.  Supported by
     * gs_forward / gs_forward_render and every gs_backward*; gs_forward_shared ignores it. */
    const float* l1_target;
    float* l1_loss;
    const float* l1_grad;
    int32_t forward_only; /* !=0: no backward will follow this forward (a frame rendered under no_grad): the render launch does
                           * not prepare the backward's row marks on the side (55 MB of streaming stores at config 3).  A
                           * backward that is run on the state anyway prepares them itself, as after a first backward */
} GsFwdArgs;

/* The eight gradient outputs of upstream `rasterize_gaussians_backward`, in the order the
 * autograd wrapper returns them.  Every non-NULL array is written IN FULL by the call (zeros for
 * culled Gaussians): the caller does not need to pre-zero.  dL_dsh may be NULL when shs is absent,
 * dL_dscales / dL_drotations when cov3D_precomp was given. */
typedef struct GsGrads {
    float* dL_dmeans3D;  /* [P,3] */
    float* dL_dmeans2D;  /* [P,3]  x,y = d/d(NDC centre), z = 0 (consumed as .grad[:, :2], scene/gaussian_model.py:464-466) */
    float* dL_dsh;       /* [P,M,3] */
    float* dL_dcolors;   /* [P,3]  gradient of colors_precomp (or of the SH colour before the clamp mask) */
    float* dL_dopacity;  /* [P,1] */
    float* dL_dscales;   /* [P,3] */
    float* dL_drotations;/* [P,4] */
    float* dL_dcov3D;    /* [P,6] */
} GsGrads;

/* ---- state-buffer sizes (the caller owns every allocation; upstream grew torch byte tensors
 * through a resize callback: geomBuffer / binningBuffer / imgBuffer) ---- */
int gs_geom_bytes(int32_t P, size_t* out);
int gs_image_bytes(int32_t W, int32_t H, size_t* out);
int gs_binning_bytes(int64_t num_rendered, int32_t W, int32_t H, size_t* out);
int gs_backward_scratch_bytes(int64_t num_rendered, int32_t P, int32_t W, int32_t H, size_t* out);

/* ---- forward, phase 1 (replaces the first half of upstream rasterize_gaussians: preprocess +
 * prefix sum).  Runs: per-Gaussian preprocess (cull, EWA projection, conic, radius, tile rect,
 * SH->RGB), a stable depth sort of the Gaussians, and the prefix sum of tiles touched.
 * Writes radii[P] (int32).  The number of (tile, Gaussian) pairs `num_rendered` is left in the geom
 * state and, if `count_host_pinned` is non-NULL, copied asynchronously (same stream) into that
 * HOST-pinned int64; the caller synchronises before reading it.  No implicit synchronisation. */
int gs_forward_preprocess(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* img, size_t img_bytes,
                          int32_t* radii, int64_t* count_host_pinned, void* stream);

/* ---- forward, phase 2 (second half of upstream rasterize_gaussians: duplicateWithKeys, sort,
 * identifyTileRanges, render).  `num_rendered` is the number of pairs the binning state is carved for: the value
 * phase 1 produced, or any larger capacity (the kernels read the frame's own count from the geom state).  Writes
 * out_color[3,H,W]. */
int gs_forward_render(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                      void* img, size_t img_bytes, int64_t num_rendered, float* out_color, void* stream);

/* ---- forward, both phases in one call (the whole of upstream rasterize_gaussians), with NO GPU idle stretch for the
 * pair count.  Upstream blocks on a device-to-host copy of num_rendered between its two halves, because the count
 * sizes the binning buffer.  Here the caller passes a binning state sized for `capacity` pairs
 * (gs_binning_bytes(capacity); e.g. the previous frame's count + 1/8); phase 2 is enqueued right behind phase 1 with
 * grids sized by the capacity, its kernels reading the count on the device, and only then does the host wait for the
 * count (stored by the device straight into the HOST-pinned `count_host_pinned`) -- the GPU is already busy with
 * phase 2.  Returns GS_OK with *num_rendered set when the count fits the capacity.  Returns GS_E_WORKSPACE with
 * *num_rendered set when it does not (phase 2 has then rendered an empty frame, nothing out of bounds), or when
 * capacity is 0 (only phase 1 ran): the caller allocates gs_binning_bytes(*num_rendered) and calls
 * gs_forward_render.  A state carved for `capacity` pairs is passed on with num_rendered = capacity to
 * gs_backward / gs_forward_shared / gs_binning_field (the carve is a function of that number). */
int gs_forward(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, int64_t capacity,
               void* img, size_t img_bytes, int32_t* radii, int64_t* count_host_pinned, float* out_color,
               int64_t* num_rendered, void* stream);

/* ---- shared-geometry forward (no upstream counterpart).  The reference's render()
 * rasterizes twice per step with identical geometry -- colour pass, then an opacity pass with
 * colors = 1 (gaussian_renderer/__init__.py:121-142).  Given the geom / binning / image state of a
 * previous gs_forward_* call on the SAME means3D, opacities, covariance inputs and camera, this renders
 * new colours (a->shs or a->colors_precomp) without repeating preprocess, sorts and binning: it fills
 * a fresh geom / image state (usable by gs_backward) and shares the binning state.  The image is composited from the
 * quadrant lists the first render recorded (same bits as a stand-alone render); if colors_precomp is all ones -- found
 * out on the device, one word of geom_src is written -- it is instead written as 1 - T of the first render (+ T bg),
 * equal to the composited image to fp32 rounding.  Pass the SAME a->long_lists as to the first render.  Its gradients
 * can be had together with the first render's in one pass: gs_backward_with_second. */
int gs_forward_shared(const GsFwdArgs* a, const void* geom_src, const void* img_src, void* geom, size_t geom_bytes,
                      void* binning, size_t binning_bytes, void* img, size_t img_bytes, int64_t num_rendered,
                      float* out_color, void* stream);

/* ---- backward (replaces upstream rasterize_gaussians_backward).  `out_color` is the forward's
 * output image, `radii` the forward's radii, `dL_dpix` = dL/d out_color [3,H,W] (NULL allowed when a->l1_target is set: the
 * fused L1 loss is then the only consumer of the image).  `num_rendered` is the number of
 * pairs the forward's binning state was carved for (the capacity given to gs_forward, or the count given to
 * gs_forward_render).  `scratch` holds gs_backward_scratch_bytes(num_rendered, P, W, H) bytes.
 * The binning state also holds the mark word of every gradient row the backward writes (all "unwritten" on entry): the
 * forward's render launch sets them on the side and says so in a state word; a backward that finds them used by an
 * earlier backward of the same forward sets them itself.  A backward therefore WRITES that part of `binning` (which is
 * why the pointer is not const; the lists and ranges it only reads); two backwards of one forward must not run concurrently. */
int gs_backward(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes, void* binning,
                size_t binning_bytes, const void* img, size_t img_bytes, int64_t num_rendered,
                const float* out_color, const float* dL_dpix, void* scratch, size_t scratch_bytes,
                const GsGrads* grads, void* stream);

/* ---- opacity render fused into the colour render (second form).  The reference obtains
 * its opacity image with a SECOND rasterizer call with colours = 1 (gaussian_renderer/__init__.py:132-142; used by
 * the mask loss, train.py:143-153, lambda_mask = 0.1 in configs/config.yaml).  That image is
 * (1 - final_T) + final_T * bg[0] per pixel and the forward already holds final_T: gs_opacity_image writes it
 * ([H,W] floats) from the image state of a finished forward, and gs_backward_with_opacity takes the gradient of
 * that image as a fourth channel of the same backward pass (its background value is bg[0], as in the reference) --
 * one render and one backward instead of two of each. ---- */
int gs_opacity_image(const GsFwdArgs* a, const void* img, size_t img_bytes, float* opacity, void* stream);
int gs_backward_with_opacity(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes,
                             void* binning, size_t binning_bytes, const void* img, size_t img_bytes,
                             int64_t num_rendered, const float* out_color, const float* dL_dpix,
                             const float* dL_dopacity_img, void* scratch, size_t scratch_bytes, const GsGrads* grads,
                             void* stream);

/* A SECOND image rendered from the same geometry with other colours (gs_forward_shared: the reference's opacity pass,
 * gaussian_renderer/__init__.py:132-142) differentiated in the same pass as the first: alpha and T are shared, so the
 * second image adds one dot product per (pixel, Gaussian) step and one term to Gtot instead of a whole second backward.
 * The gradients are the SUM of both images' gradients w.r.t. the shared inputs; the second image's colours get none
 * (they must be constants); dL_dcolors / dL_dsh are the first image's.  `img` = the image state gs_forward_shared
 * filled for the second render (its checkpoints; a word of it says whether that render's colours were all (1, 1, 1), in
 * which case the pass needs no second colours at all: the reference's case), `long_lists` the value that render was
 * given. */
typedef struct GsSecondImage {
    const float* colors;    /* [P,3] colors_precomp of the second render */
    const float* out_color; /* [3,H,W] its result */
    const float* dL_dpix;   /* [3,H,W] its gradient */
    const void* img;        /* its image state */
    size_t img_bytes;
    int32_t long_lists;
} GsSecondImage;
int gs_backward_with_second(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes,
                            void* binning, size_t binning_bytes, const void* img, size_t img_bytes, int64_t D,
                            const float* out_color, const float* dL_dpix, const GsSecondImage* second, void* scratch,
                            size_t scratch_bytes, const GsGrads* grads, void* stream);

/* ---- upstream mark_visible / GaussianRasterizer.markVisible: present[i] = (z_view > 0.2) ---- */
int gs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    uint8_t* present, void* stream);

/* ---- simple_knn._C.distCUDA2 (scene/gaussian_model.py:186): mean squared distance to the three
 * nearest other points.  points[P,3] fp32 -> mean_d2[P] fp32. ---- */
int knn_workspace_bytes(int32_t P, size_t* out);
int knn_dist2(int32_t P, const float* points, float* mean_d2, void* workspace, size_t workspace_bytes, void* stream);

/* ---- pre-rasterizer per-Gaussian chains.
 * gs_build_covariance: scene/gaussian_model.py:28-32 build_covariance_from_scaling_rotation =
 *   strip_symmetric(L L^T), L = R diag(scaling_modifier * scaling) (utils/general_utils.py:73-85,194-207);
 *   `rotation` is N quaternions (w,x,y,z), normalised as build_rotation does (general_utils.py:87-108), or,
 *   with rotation_is_matrix, N row-major 3x3 matrices (the rigid deformer's rotation_precomp,
 *   models/deformer/rigid.py:229-231).  cov6 = [xx, xy, xz, yy, yz, zz].  The backward gives what autograd
 *   derives for that chain (d/dscaling, d/drotation in the input's own parametrisation).
 * gs_sh2rgb: models/texture/texture.py:21-38 SH2RGB.forward: direction xyz - campos, optionally rotated by the
 *   transpose of fwd_rotation[N,3,3] (cano_view_dir: T_fwd[:, :3, :3]) and multiplied from the right by a 3x3
 *   view-noise matrix given as 9 HOST floats (NULL = none), normalised with +1e-12, eval_sh of degree
 *   sh_degree over shs[N,M,3], +0.5, clamp at 0.  `clamped` (N bytes, bit c = channel c clamped) feeds the
 *   backward, which returns d/dshs and d/dxyz (fwd_rotation carries no gradient: it is detached upstream,
 *   rigid.py:223). ---- */
int gs_build_covariance(int32_t N, const float* scaling, float scaling_modifier, const float* rotation,
                        int32_t rotation_is_matrix, float* cov6, void* stream);
int gs_build_covariance_backward(int32_t N, const float* scaling, float scaling_modifier, const float* rotation,
                                 int32_t rotation_is_matrix, const float* dL_dcov6, float* dL_dscaling,
                                 float* dL_drotation, void* stream);
int gs_sh2rgb(int32_t N, int32_t sh_degree, int32_t M, const float* shs, const float* xyz, const float* campos,
              const float* fwd_rotation, const float* view_noise_host, float* colors, uint8_t* clamped, void* stream);
int gs_sh2rgb_backward(int32_t N, int32_t sh_degree, int32_t M, const float* shs, const float* xyz, const float* campos,
                       const float* fwd_rotation, const float* view_noise_host, const uint8_t* clamped,
                       const float* dL_dcolors, float* dL_dshs, float* dL_dxyz, void* stream);

/* ---- image-side L1 loss: the reference computes
 * torch.abs(network_output - gt).mean() (utils/loss_utils.py:21-22, called at train.py:121) and lets
 * autograd derive d(loss)/d(network_output).  One call here: loss[0] = mean |x - y| and
 * dL_dx[i] = sign(x[i] - y[i]) / n, for n fp32 elements (pointers 16-byte aligned); deterministic. ---- */
int gs_l1_loss_workspace_bytes(int64_t n, size_t* out);
int gs_l1_loss(int64_t n, const float* x, const float* y, float* loss, float* dL_dx, void* workspace,
               size_t workspace_bytes, void* stream);

/* Mask loss, BCE form (train.py:146-148, mask_loss_type = 'bce'; the 'l1' form is gs_l1_loss):
 * loss[0] = mean of binary_cross_entropy(clamp(x, 1e-3, 1 - 1e-3), y), dL_dx its gradient w.r.t. x (zero where the
 * clamp is active).  Workspace: gs_l1_loss_workspace_bytes(n). */
int gs_bce_loss(int64_t n, const float* x, const float* y, float* loss, float* dL_dx, void* workspace,
                size_t workspace_bytes, void* stream);

/* SSIM half of row N2: utils/loss_utils.py:27-67 `ssim(img1, img2)` (window 11, sigma 1.5, zero padding,
 * C1 = 0.01^2, C2 = 0.03^2, mean over all C*H*W elements; train.py:123 uses 1 - ssim as the D-SSIM loss).
 * gs_ssim_forward writes ssim_out[0] and, when the three map pointers are non-NULL (all or none), the
 * per-element partial derivatives of the SSIM map w.r.t. the window mean, variance and covariance, each
 * C*H*W floats.  gs_ssim_backward turns them into dL/dimg1 given the device scalar dL/dssim. ---- */
int gs_ssim_workspace_bytes(int32_t C, int32_t H, int32_t W, size_t* out);
int gs_ssim_forward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, float* ssim_out,
                    float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* workspace, size_t workspace_bytes,
                    void* stream);
int gs_ssim_backward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, const float* dm_dmu1,
                     const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dssim, float* dL_dimg1,
                     void* stream);

/* ---- K nearest neighbours: pytorch3d.ops.knn_points as the reference calls it
 * (utils/loss_utils.py:76-79,92-96: K = 5 / 6 self-KNN of the canonical Gaussians for the AIAP loss;
 * models/deformer/rigid.py:43: nearest SMPL vertex, K = 1).  For every query the K (<= 8) nearest points of
 * `ref`: squared distances ascending and their indices into `ref` (ties: smaller index first; -1 / FLT_MAX when
 * ref has fewer than K points).  A query that is itself in `ref` is returned as its own first neighbour, as
 * pytorch3d does.  Exact (no approximation).  Workspace: knn_workspace_bytes(Nr). ---- */
int knn_points(int32_t Nq, const float* queries, int32_t Nr, const float* ref, int32_t K, float* dists, int64_t* idx,
               void* workspace, size_t workspace_bytes, void* stream);

/* ---- training-step bookkeeping after the backward pass.
 * gs_densify_stats: train.py:219-220 + scene/gaussian_model.py:464-466, for every Gaussian with radii > 0:
 *   max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |viewspace_grad[:2]|; denom += 1
 *   (viewspace_grad is the (N,3) gradient of the screen-space points the rasterizer returns for means2D).
 * gs_adam_step: torch.optim.Adam as scene/gaussian_model.py:201-216 sets it up (one learning rate per tensor,
 *   shared betas / eps, no weight decay, no amsgrad), step number `step` >= 1, all tensors in ONE launch;
 *   exp_avg / exp_avg_sq are the optimizer's state tensors and are updated in place, as is param. ---- */
#define GS_ADAM_MAX_TENSORS 16
typedef struct GsAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;   /* elements */
    float lr;
} GsAdamTensor;
int gs_densify_stats(int32_t N, const int32_t* radii, const float* viewspace_grad, float* max_radii2D,
                     float* xyz_gradient_accum, float* denom, void* stream);
int gs_adam_step(int32_t n_tensors, const GsAdamTensor* tensors, double beta1, double beta2, double eps, int64_t step,
                 void* stream);

/* ---- introspection for parity tests: device pointers INTO the opaque state buffers.  `field`:
 *  geom:    0 depths f32[P]        1 tiles_touched u32[P]   2 splat records f32[P,12]
 *           (x, y, conicA, conicB, conicC, opacity, r, g, b, first pair u32, rect_min u32 (x | y<<16), rect_size u32 (w | h<<16))
 *           3 clamped bitmask u32[P]   4 depth-sorted Gaussian index u32[P]   5 num_rendered u64[1]
 *  binning: 0 point_list u32[D] (tile after tile, (depth, index) order inside a tile; tile t owns ranges[t] of it)
 *  image:   0 ranges u32[tiles,2]   1 n_contrib u32[H,W]   2 final_T f32[H,W]
 *           3 per-quadrant compacted count up to the last contributor u32[tiles,4]
 *           4 per-pixel last contributor in compacted coordinates u32[H,W]
 *           5 launch order of the tiles u32[tiles], heaviest first; bit 31 = rendered by four waves per quadrant, or in chunks
 *           (gs_tuning "fwd4" = 2 only) 6 chunk work header u32[16] (units, entries per chunk, .., [4..11] items per XCD)
 *           7 units u32[.,2] {tile, chunk | chunks of the tile << 16}   8 per (unit, quadrant) hits + 1 | dead << 31
 *           9 per (unit, quadrant) record f32[8,64] */
int gs_geom_field(void* geom, int32_t P, int32_t field, void** out);
int gs_binning_field(void* binning, int64_t num_rendered, int32_t W, int32_t H, int32_t field, void** out);
int gs_image_field(void* img, int32_t W, int32_t H, int32_t field, void** out);
/* size of the image state for a call with these arguments (W, H, long_lists); gs_image_bytes(W, H) = long_lists 0 */
int gs_image_bytes_for(const GsFwdArgs* a, size_t* out);

/* ---- per-stage timing (the reference only timed whole calls with CUDA events: render.py:46-62,
 * train.py:79-88,181-185).  When enabled (process-wide: autograd runs the backward on its own host thread), every stage launched by this library is
 * bracketed by a pair of hipEvents recorded ON THE STAGE'S OWN STREAM.  gs_profile_collect waits for
 * the recorded events, sums the elapsed milliseconds and launch counts per stage name (first `max`
 * distinct stages, names are static strings) and clears the record. ---- */
int gs_profile_reserve(int n_events); /* pre-create events so that none is created inside a timed region */
int gs_profile_enable(int on);
int gs_profile_filter(const char* stage); /* NULL or "" = every stage; else only the named stage is timed */
int gs_profile_collect(int max, const char** names, float* ms, int32_t* launches, int32_t* n_out);

/* Report counter (bench.py's `pairs_valid`; never in a timed path): from the state of a finished forward, counts[0] = the
 * (pixel, Gaussian) pairs actually composited -- alpha >= 1/255 at that pixel, before the pixel was done -- and counts[1] =
 * the pairs a per-pixel walk up to each pixel's last contributor visits (the sum of n_contrib).  `counts`: two device
 * uint64.  Against 64 x (quadrant-list entries up to each quadrant's last contributor), which is what the render kernels
 * evaluate, counts[0] says how much of their work is useful. */
int gs_pair_stats(const GsFwdArgs* a, const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
                  const void* img, size_t img_bytes, int64_t num_rendered, uint64_t* counts, void* stream);

/* Shader clock under load (bench.py's `roofline.binding`): runs an FMA stream on every SIMD for `iters` x 32 instructions per
 * wave (8192 iterations ~ 1 ms) and adds up, over the workgroups, ticks[0] = shader-clock cycles (s_memtime) and ticks[1] =
 * ticks of the constant 100 MHz counter (s_memrealtime) spent in it: clock = ticks[0] / ticks[1] x 100 MHz.  `ticks`: four
 * device uint64 (two results, two scratch words).  Not capture-safe. */
int gs_clock_probe(uint64_t* ticks, int32_t iters, void* stream);

/* Which XCD (accelerator die with its own L2) every workgroup of a launch of `n_blocks` workgroups of 64 threads runs on:
 * xcc[b] = HW_REG_XCC_ID of workgroup b.  The chunk-parallel forward (gs_tuning "fwd4" = 2) numbers its workers on the
 * premise that the dispatcher deals workgroups round-robin over the XCDs, xcc[b] == b % 8 on an MI355X in SPX mode; it
 * reads the register and claims its items, so a different deal costs speed, not the image -- this call is how the -m gpu
 * suite checks the premise.  `xcc`: n_blocks device uint32. */
int gs_xcc_probe(uint32_t* xcc, int32_t n_blocks, void* stream);

/* process-wide tuning switches for experiments and A/B measurements.  Without effect on the results: "xcd_map" (1: the
 * four quadrant waves of a tile on one XCD), "depth_sort" (1: bucket sort, 0: LSD radix), "nt_stores" (1: the backward's
 * row-mark fill is written with streaming stores), "fwd_marks" (1: the forward's render launch sets the backward's row
 * marks on the side, 0: every backward sets them itself), "bwd_order" (1: the backward orders the tiles by the forward's
 * per-quadrant last contributors, 0: walks them in the forward's launch order).  With an effect of fp32 rounding (which kernels render a frame of few,
 * long tile lists; flip them between frames only, "small_tiles" also changes the image state's size): "fwd4" (0: one wave per quadrant everywhere; 1 (default): four
 * waves per quadrant, four entries per step on the marked tiles; 2: the marked tiles' lists cut into chunks, a wave per
 * (chunk, quadrant) -- also changes the image state's size; the same list, contributors and stop rule, transmittance and
 * colour sums associated per chunk), "fwdc_ch" (entries per chunk, a power of two >= 64; 256), "fwdc_div" (a tile is marked
 * when its list is longer than the frame's pairs / this; 320), "bwd_chunks" (1: backward in chunks from the forward's
 * checkpoints), "small_tiles" (images of up to this many tiles use both whatever GsFwdArgs.long_lists says; 2048).  "shared_qlist"
 * (1: gs_forward_shared renders from the recorded quadrant lists, 0: from the tiles' lists; same bits).  "ones_fast" (1: a
 * second render whose colours are all ones is written as 1 - T of the first; 0: composited; equal to fp32 rounding) */
int gs_tuning(const char* name, int value);
const char* gs_status_string(int code);
int gs_last_hip_error(void); /* hipError_t of the most recent GS_E_HIP on this thread */
const char* gs_last_stage(void); /* name of the stage that failed (debug mode names every kernel) */
const char* gs_build_info(void); /* "gfx950 ..." */

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_MI355_H */
