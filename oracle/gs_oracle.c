/*
 * gs_oracle.c -- CPU restatement of the differentiable Gaussian-splat rasterizer
 * (diff_gaussian_rasterization) and of simple_knn.distCUDA2.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity checker for the HIP path and the
 * "cpu_baseline" leg of bench.py.  Nothing in the shipped package imports, links or calls it.
 *
 * PARITY UNPINNED: the reference tree (/root/reference) holds no source for this path
 * (submodules/diff-gaussian-rasterization and submodules/simple-knn are empty, un-vendored git
 * submodules without a commit pin -- .gitmodules:1-6, environment.yml:25-26) and no tests or
 * golden vectors for it.  What the reference DOES pin, and what this file follows:
 *   - the API generation and argument meaning: gaussian_renderer/__init__.py:85-98,121-141
 *   - matrix conventions (row-vector, transposed W2C, full_proj = view @ proj^T):
 *     scene/cameras.py:35-40, utils/graphics_utils.py:38-71
 *   - quaternion order (w,x,y,z) and rotation matrix: utils/general_utils.py:87-108
 *   - cov3D six-vector layout [xx,xy,xz,yy,yz,zz]: utils/general_utils.py:73-85
 *   - covariance = (R S)(R S)^T with S = diag(mod*scale): scene/gaussian_model.py:28-32,
 *     utils/general_utils.py:194-207
 *   - the SH polynomial and constants: utils/sh_utils.py:27-44,58-101, "+0.5, clamp_min 0":
 *     models/texture/texture.py:35-37
 *   - distCUDA2 call site and its clamp: scene/gaussian_model.py:186
 * The kernel arithmetic itself is the published algorithm of the third-party dependency
 * graphdeco-inria/diff-gaussian-rasterization (2023 API generation: 12-field settings tuple,
 * (color, radii) return) as restated in SURVEY.md section 8a rows A4-A10.
 *
 * Arithmetic: fp32, one rounding per operation, in the order written (build with
 * -ffp-contract=off, no fast-math).  Gradient accumulators over (pixel, Gaussian) pairs are kept
 * in double: the reference sums float atomics in arbitrary order, so the order-free sum is the
 * thing to check against.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define BLOCK_X 16
#define BLOCK_Y 16

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    int P, deg, M, W, H;
    const float *bg, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations,
        *cov3D_precomp, *viewmatrix, *projmatrix, *campos;
    float scale_modifier, tanfovx, tanfovy;
    int prefiltered;
    int tile_rect; /* 0: upstream's 3-sigma square; 1: bounding box of the alpha >= 1/255 region (see or_preprocess) */
} OrArgs;

/* p_view = [p,1] @ viewmatrix (row-vector convention, flat index = row*4 + col):
 * scene/cameras.py:35 stores W2C transposed, so translation sits at 12..14. */
static void xform4x3(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Rotation of a (w,x,y,z) quaternion, NOT renormalised by the kernel (the caller's activation
 * normalises: scene/gaussian_model.py:137-138); entries as utils/general_utils.py:99-107. */
static void quat_to_R(const float* q, float R[3][3]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = (R S)(R S)^T, S = diag(mod * scale); six-vector [xx,xy,xz,yy,yz,zz]
 * (scene/gaussian_model.py:28-32, utils/general_utils.py:73-85,194-207). */
static void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* c6) {
    float R[3][3];
    quat_to_R(q, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float L[3][3]; /* L = R * S */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * s[j];
    /* Sigma_ij = sum_k L_ik L_jk */
    float S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[i][j] = L[i][0] * L[j][0] + L[i][1] * L[j][1] + L[i][2] * L[j][2];
    c6[0] = S[0][0]; c6[1] = S[0][1]; c6[2] = S[0][2];
    c6[3] = S[1][1]; c6[4] = S[1][2]; c6[5] = S[2][2];
}

/* EWA projection: cov2D = (J Rv) Sigma (J Rv)^T, + 0.3 px^2 low-pass on the diagonal.
 * Rv(i,j) = viewmatrix[j*4+i] (rotation of W2C).  Returns M = J Rv (2x3) as well, used by the
 * backward pass. */
static void cov2d(const float* mean, float fx, float fy, float tanfovx, float tanfovy,
                  const float* c6, const float* V, float* cov /*a,b,c*/, float M[2][3],
                  float* t_out /*clamped t*/, float* txtz_tytz) {
    float t[3];
    xform4x3(mean, V, t);
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    for (int k = 0; k < 3; k++) {
        /* Rv(0,k) = V[4k+0], Rv(1,k) = V[4k+1], Rv(2,k) = V[4k+2] */
        M[0][k] = J00 * V[4 * k + 0] + J02 * V[4 * k + 2];
        M[1][k] = J11 * V[4 * k + 1] + J12 * V[4 * k + 2];
    }
    float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float MS[2][3];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) MS[i][j] = M[i][0] * S[0][j] + M[i][1] * S[1][j] + M[i][2] * S[2][j];
    cov[0] = MS[0][0] * M[0][0] + MS[0][1] * M[0][1] + MS[0][2] * M[0][2] + 0.3f;
    cov[1] = MS[0][0] * M[1][0] + MS[0][1] * M[1][1] + MS[0][2] * M[1][2];
    cov[2] = MS[1][0] * M[1][0] + MS[1][1] * M[1][1] + MS[1][2] * M[1][2] + 0.3f;
    if (t_out) { t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2]; }
    if (txtz_tytz) { txtz_tytz[0] = txtz; txtz_tytz[1] = tytz; }
}

/* SH -> RGB: polynomial and constants of utils/sh_utils.py:58-101; "+0.5", clamp at 0 with the
 * clamp recorded (models/texture/texture.py:36-37 minus its 1e-12).  shs layout (N, M, 3),
 * coefficient-major / channel-minor (scene/gaussian_model.py:145-148). */
static void sh_to_rgb(int deg, int M, const float* mean, const float* campos, const float* sh /*M*3*/,
                      float* rgb, uint8_t* clamped) {
    float d[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] / len, y = d[1] / len, z = d[2] / len;
    (void)M;
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k)*3 + c]
        float r = SH_C0 * SH(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                    SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                    SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                        SH_C3[5] * z * (xx - yy) * SH(14) + SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        clamped[c] = (r < 0.f);
        rgb[c] = fmaxf(r, 0.f);
    }
}

/* ---- A4: preprocess ---------------------------------------------------------------------- */
/* outputs (all length P unless noted): depths f32, radii i32, xy f32[2P], conic_opacity f32[4P],
 * rgb f32[3P], clamped u8[3P], cov3D f32[6P], tiles_touched u32, rect i32[4P] (minx,miny,maxx,maxy) */
int or_preprocess(const OrArgs* a, float* depths, int* radii, float* xy, float* conic_opacity,
                  float* rgb, uint8_t* clamped, float* cov3D, uint32_t* tiles_touched, int* rect) {
    const int P = a->P, W = a->W, H = a->H;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const float fy = H / (2.0f * a->tanfovy), fx = W / (2.0f * a->tanfovx);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        depths[i] = 0.f;
        xy[2 * i] = xy[2 * i + 1] = 0.f;
        for (int k = 0; k < 4; k++) { conic_opacity[4 * i + k] = 0.f; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        const float* p = a->means3D + 3 * i;
        float pv[3];
        xform4x3(p, a->viewmatrix, pv);
        if (pv[2] <= 0.2f) continue; /* near cull */
        float ph[4];
        xform4x4(p, a->projmatrix, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
        float* c6 = cov3D + 6 * i;
        if (a->cov3D_precomp) {
            for (int k = 0; k < 6; k++) c6[k] = a->cov3D_precomp[6 * i + k];
        } else {
            cov3d_from_scale_rot(a->scales + 3 * i, a->scale_modifier, a->rotations + 4 * i, c6);
        }
        float cov[3], M[2][3];
        cov2d(p, fx, fy, a->tanfovx, a->tanfovy, c6, a->viewmatrix, cov, M, NULL, NULL);
        float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        float l1 = mid + sq, l2 = mid - sq;
        float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        /* ndc2Pix is evaluated in double upstream ((v + 1.0) * S - 1.0) * 0.5 */
        float px = (float)((((double)pp[0] + 1.0) * (double)W - 1.0) * 0.5);
        float py = (float)((((double)pp[1] + 1.0) * (double)H - 1.0) * 0.5);
        int r = (int)my_radius;
        int minx = (int)((px - r) / BLOCK_X), miny = (int)((py - r) / BLOCK_Y);
        int maxx = (int)((px + r + BLOCK_X - 1) / BLOCK_X), maxy = (int)((py + r + BLOCK_Y - 1) / BLOCK_Y);
        minx = minx < 0 ? 0 : (minx > gx ? gx : minx);
        miny = miny < 0 ? 0 : (miny > gy ? gy : miny);
        maxx = maxx < 0 ? 0 : (maxx > gx ? gx : maxx);
        maxy = maxy < 0 ? 0 : (maxy > gy ? gy : maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (a->tile_rect) {
            /* Not upstream: bin into the axis-aligned bounding box of { alpha >= 1/255 } =
             * { d^T Sigma^-1 d <= 2 ln(255 opacity) } only -- half-widths sqrt(thr Sigma_xx), sqrt(thr Sigma_yy) --
             * intersected with the square.  Tiles left out cannot contribute (alpha < 1/255 on every pixel), so
             * colour and gradients are unchanged; the Gaussian stays visible (radii, depth, colour).  ln is bounded
             * from above with exactly rounded operations only so that the rectangle is reproducible bit for bit. */
            const float x = 255.0f * a->opacities[i];
            if (x >= 1.0f) {
                uint32_t u; memcpy(&u, &x, 4);
                const int e = (int)(u >> 23) - 127;
                uint32_t mu = (u & 0x007FFFFFu) | 0x3F800000u;
                float m; memcpy(&m, &mu, 4);
                const float lm = (m < 1.5f) ? (m - 1.0f) : (0.405465126f + (m - 1.5f) * 0.666666687f);
                const float thr = 2.0f * ((float)e * 0.693147182f + lm) + 0.002f;
                /* the pixels evaluate the form with the rounded fp32 conic: for an ill-conditioned covariance its level
                 * set is the exact ellipse scaled by up to sqrt(1 + k 2^-24 a c / det); widen the threshold by that
                 * (k = 32), and keep the square when the bound says nothing */
                const float cond = (cov[0] * cov[2]) / det;
                const float widen = 1.0f + cond * 1.9073486e-6f;
                float hx = 1.0e7f, hy = 1.0e7f;
                if (det > 0.0f && widen <= 2.0f) {
                    const float thr_w = thr * widen;
                    hx = fminf(sqrtf(thr_w * cov[0]), 1.0e7f);
                    hy = fminf(sqrtf(thr_w * cov[2]), 1.0e7f);
                }
                int sminx = (int)((px - hx) / BLOCK_X), sminy = (int)((py - hy) / BLOCK_Y);
                int smaxx = (int)((px + hx) / BLOCK_X) + 1, smaxy = (int)((py + hy) / BLOCK_Y) + 1;
                if (sminx > minx) minx = sminx;
                if (sminy > miny) miny = sminy;
                if (smaxx < maxx) maxx = smaxx;
                if (smaxy < maxy) maxy = smaxy;
                if (maxx < minx) maxx = minx;
                if (maxy < miny) maxy = miny;
            } else {
                maxx = minx;
                maxy = miny;
            }
        }
        if (a->colors_precomp) {
            for (int k = 0; k < 3; k++) rgb[3 * i + k] = a->colors_precomp[3 * i + k];
        } else {
            sh_to_rgb(a->deg, a->M, p, a->campos, a->shs + (size_t)i * a->M * 3, rgb + 3 * i, clamped + 3 * i);
        }
        depths[i] = pv[2];
        radii[i] = r;
        xy[2 * i] = px; xy[2 * i + 1] = py;
        conic_opacity[4 * i + 0] = conic[0]; conic_opacity[4 * i + 1] = conic[1];
        conic_opacity[4 * i + 2] = conic[2]; conic_opacity[4 * i + 3] = a->opacities[i];
        tiles_touched[i] = (uint32_t)((maxy - miny) * (maxx - minx));
        rect[4 * i + 0] = minx; rect[4 * i + 1] = miny; rect[4 * i + 2] = maxx; rect[4 * i + 3] = maxy;
    }
    return 0;
}

/* ---- A10: mark_visible --------------------------------------------------------------------- */
int or_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > 0.2f;
    }
    return 0;
}

/* ---- A5: scan / duplicateWithKeys / stable sort / identifyTileRanges ----------------------- */
int64_t or_scan(int P, const uint32_t* tiles_touched, uint32_t* offsets /*inclusive*/) {
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; offsets[i] = acc; }
    return P > 0 ? (int64_t)acc : 0;
}

void or_duplicate_with_keys(int P, int W, const float* depths, const int* radii, const int* rect,
                            const uint32_t* offsets, uint64_t* keys, uint32_t* vals) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        uint32_t off = (i == 0) ? 0 : offsets[i - 1];
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; x++) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32;
                key |= dbits;
                keys[off] = key;
                vals[off] = (uint32_t)i;
                off++;
            }
    }
}

typedef struct { uint64_t k; uint32_t v; uint32_t pos; } KV;
static int kv_cmp(const void* a, const void* b) {
    const KV* x = (const KV*)a; const KV* y = (const KV*)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0); /* stable */
}
int or_sort_pairs(int64_t D, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout) {
    KV* t = (KV*)malloc(sizeof(KV) * (size_t)(D > 0 ? D : 1));
    if (!t) return -1;
    for (int64_t j = 0; j < D; j++) { t[j].k = kin[j]; t[j].v = vin[j]; t[j].pos = (uint32_t)j; }
    qsort(t, (size_t)D, sizeof(KV), kv_cmp);
    for (int64_t j = 0; j < D; j++) { kout[j] = t[j].k; vout[j] = t[j].v; }
    free(t);
    return 0;
}

void or_tile_ranges(int64_t D, const uint64_t* keys, int ntiles, uint32_t* ranges /*ntiles*2*/) {
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)ntiles);
    for (int64_t j = 0; j < D; j++) {
        uint32_t cur = (uint32_t)(keys[j] >> 32);
        if (j == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys[j - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)j; ranges[2 * cur] = (uint32_t)j; }
        }
        if (j == D - 1) ranges[2 * cur + 1] = (uint32_t)D;
    }
}

/* ---- threshold decisions of the compositing loop: margins, overrides, attribution ------------------------
 * The loop below takes three yes/no decisions per (pixel, Gaussian) pair: power > 0, alpha < 1/255, T (1 - alpha) < 1e-4.
 * A device kernel evaluates the same expressions with another exp (v_exp_f32 vs glibc expf), in the log2 domain and
 * with FMA contraction: for a pair whose value sits within rounding of a threshold it can decide the other way, and
 * the pixel then differs by that pair's whole contribution -- not by rounding.  To make "the only differences are
 * such flips" a CHECKED statement (tests/test_gpu_parity.py) the oracle can
 *   - report per pixel the smallest relative margin of any decision it took (or_render_forward_ex, `margin`),
 *   - take a table of OVERRIDES: decisions of named (pixel, list entry) pairs forced to a given outcome, in the forward
 *     and in the backward,
 *   - and, given what the device produced for a pixel (colour, final T, last contributor), SEARCH for the smallest set
 *     of decisions -- each within a stated margin of its threshold -- whose flipping reproduces it (or_explain_pixels).
 * None of this changes the arithmetic of a pair: or_render_forward is or_render_forward_ex without overrides. */
#define OV_SKIP 1u /* the pair is skipped whatever its power / alpha tests say     */
#define OV_KEEP 2u /* the pair passes the power and the alpha test whatever they say */
#define OV_STOP 4u /* the pixel is done at this pair: the T test fails             */
#define OV_GO 8u   /* the T test passes                                            */

typedef struct { uint32_t j; uint8_t act; float margin; } OrCand;

/* overrides of one pixel: the slice [lo, hi) of the table sorted by key = pixel << 32 | list index */
static void ov_slice(int n, const uint64_t* key, uint64_t pid, int* lo_out, int* hi_out) {
    int lo = 0, hi = n;
    const uint64_t k0 = pid << 32;
    while (lo < hi) { int m = (lo + hi) >> 1; if (key[m] < k0) lo = m + 1; else hi = m; }
    int e = lo;
    while (e < n && (key[e] >> 32) == pid) e++;
    *lo_out = lo; *hi_out = e;
}
static inline uint8_t ov_act(const uint64_t* key, const uint8_t* act, int lo, int hi, uint32_t j) {
    for (int k = lo; k < hi; k++) if ((uint32_t)key[k] == j) return act[k];
    return 0;
}

/* One pixel, front to back (A6).  okey/oact[lo, hi): this pixel's overrides.  cands (may be NULL): the decisions
 * taken within eps[] = (power, alpha, T) relative margin of their threshold and not overridden, with the action that
 * would flip each.  margin (may be NULL): the smallest relative margin of any decision taken. */
static void pixel_walk(const uint32_t* point_list, uint32_t r0, uint32_t r1, const float* xy, const float* conic_opacity,
                       const float* rgb, float pxf, float pyf, const uint64_t* okey, const uint8_t* oact, int lo, int hi,
                       float C[3], float* T_out, uint32_t* last_out, float* margin, OrCand* cands, int* ncand, int maxc,
                       const float* eps, uint32_t* blended_out) {
    float T = 1.0f;
    C[0] = C[1] = C[2] = 0.f;
    uint32_t contributor = 0, last = 0, blended = 0;
    float mg = INFINITY;
    for (uint32_t j = r0; j < r1; j++) {
        contributor++;
        const uint8_t act = lo < hi ? ov_act(okey, oact, lo, hi, j) : 0;
        uint32_t g = point_list[j];
        float dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
        const float* co = conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (act & OV_SKIP) continue;
        const int keep = (act & OV_KEEP) != 0;
        if (!keep && (margin || cands)) {
            /* |power| against the size of the terms it is the sum of (nothing to decide when they are all zero) */
            float terms = 0.5f * (fabsf(co[0]) * dx * dx + fabsf(co[2]) * dy * dy) + fabsf(co[1] * dx * dy);
            float relp = terms > 0.f ? fabsf(power) / terms : INFINITY;
            if (relp < mg) mg = relp;
            if (cands && relp <= eps[0] && *ncand < maxc) cands[(*ncand)++] = (OrCand){j, power > 0.0f ? OV_KEEP : OV_SKIP, relp};
        }
        if (!keep && power > 0.0f) continue;
        float alpha = fminf(0.99f, co[3] * expf(power));
        if (!keep && (margin || cands)) {
            float rela = fabsf(alpha * 255.0f - 1.0f);
            if (rela < mg) mg = rela;
            if (cands && rela <= eps[1] && *ncand < maxc) cands[(*ncand)++] = (OrCand){j, alpha < 1.0f / 255.0f ? OV_KEEP : OV_SKIP, rela};
        }
        if (!keep && alpha < 1.0f / 255.0f) continue;
        float test_T = T * (1 - alpha);
        int stop = test_T < 0.0001f;
        if (act & OV_STOP) stop = 1;
        else if (act & OV_GO) stop = 0;
        else if (margin || cands) {
            float relT = fabsf(test_T * 10000.0f - 1.0f);
            if (relT < mg) mg = relT;
            if (cands && relT <= eps[2] && *ncand < maxc) cands[(*ncand)++] = (OrCand){j, stop ? OV_GO : OV_STOP, relT};
        }
        if (stop) break; /* done: this Gaussian is NOT blended */
        for (int c = 0; c < 3; c++) C[c] += rgb[3 * g + c] * alpha * T;
        T = test_T;
        last = contributor;
        blended++;
    }
    *T_out = T;
    *last_out = last;
    if (blended_out) *blended_out = blended;
    if (margin) *margin = mg;
}

/* ---- A6: render forward --------------------------------------------------------------------- */
int or_render_forward_ex(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                         const float* xy, const float* conic_opacity, const float* rgb, const float* bg,
                         int n_over, const uint64_t* over_key, const uint8_t* over_act,
                         float* out_color /*3HW*/, float* final_T /*HW*/, uint32_t* n_contrib /*HW*/, float* margin /*HW or NULL*/,
                         uint32_t* n_blended /*HW or NULL: pairs actually composited per pixel*/) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                size_t pid = (size_t)pyi * W + pxi;
                int lo = 0, hi = 0;
                if (n_over > 0) ov_slice(n_over, over_key, (uint64_t)pid, &lo, &hi);
                float T, C[3];
                uint32_t last;
                pixel_walk(point_list, r0, r1, xy, conic_opacity, rgb, (float)pxi, (float)pyi, over_key, over_act, lo, hi, C, &T,
                           &last, margin ? margin + pid : NULL, NULL, NULL, 0, NULL, n_blended ? n_blended + pid : NULL);
                final_T[pid] = T;
                n_contrib[pid] = last;
                for (int c = 0; c < 3; c++) out_color[(size_t)c * H * W + pid] = C[c] + T * bg[c];
            }
    }
    return 0;
}
int or_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                      const float* xy, const float* conic_opacity, const float* rgb, const float* bg,
                      float* out_color /*3HW*/, float* final_T /*HW*/, uint32_t* n_contrib /*HW*/) {
    return or_render_forward_ex(W, H, ranges, point_list, xy, conic_opacity, rgb, bg, 0, NULL, NULL, out_color, final_T,
                                n_contrib, NULL, NULL);
}

/* Attribution.  For each of the `npix` listed pixels: what the device produced (dev_color[3 npix] channel-major,
 * dev_T, dev_last = its n_contrib).  Searches, by iterative deepening up to `max_flips` decisions, for a set of
 * decisions each within eps[] of its threshold whose flipping makes this oracle's walk of the pixel reproduce the
 * device's result: the same last contributor, final T within tol_T (relative), every channel within tol_c (absolute).
 * status[p] = number of flips used (0: the plain walk already matches), or -1: NOT explained.  The flips found are
 * appended to (out_key, out_act, out_margin) -- at most `cap` of them; returns their number, or -1 on overflow. */
typedef struct {
    const uint32_t* point_list; uint32_t r0, r1; const float *xy, *co, *rgb, *bg; float pxf, pyf;
    const float* dev_c; float dev_T; uint32_t dev_last; float tol_c, tol_T; const float* eps; uint64_t pid;
} ExplainCtx;
static int explain_rec(const ExplainCtx* x, uint64_t* fkey, uint8_t* fact, float* fmargin, int nf, int depth_left) {
    OrCand cands[24];
    int nc = 0;
    float C[3], T;
    uint32_t last;
    /* (the forced table of this pixel, sorted by list index: insertion keeps it so) */
    pixel_walk(x->point_list, x->r0, x->r1, x->xy, x->co, x->rgb, x->pxf, x->pyf, fkey, fact, 0, nf, C, &T, &last, NULL,
               depth_left > 0 ? cands : NULL, &nc, 24, x->eps, NULL);
    int ok = last == x->dev_last && fabsf(T - x->dev_T) <= x->tol_T * fmaxf(T, x->dev_T);
    for (int c = 0; ok && c < 3; c++) ok = fabsf(C[c] + T * x->bg[c] - x->dev_c[c]) <= x->tol_c;
    if (ok) return nf;
    if (depth_left == 0) return -1;
    for (int k = 0; k < nc; k++) {
        /* force the other outcome of decision k: a new table with it merged in */
        uint64_t key2[8]; uint8_t act2[8]; float mg2[8];
        int n2 = 0, merged = 0;
        const uint64_t kk = (x->pid << 32) | cands[k].j;
        for (int i = 0; i < nf; i++) {
            if (!merged && fkey[i] == kk) { key2[n2] = kk; act2[n2] = fact[i] | cands[k].act; mg2[n2] = fmaxf(fmargin[i], cands[k].margin); n2++; merged = 1; continue; }
            if (!merged && fkey[i] > kk) { key2[n2] = kk; act2[n2] = cands[k].act; mg2[n2] = cands[k].margin; n2++; merged = 1; }
            key2[n2] = fkey[i]; act2[n2] = fact[i]; mg2[n2] = fmargin[i]; n2++;
        }
        if (!merged) { key2[n2] = kk; act2[n2] = cands[k].act; mg2[n2] = cands[k].margin; n2++; }
        int r = explain_rec(x, key2, act2, mg2, n2, depth_left - 1);
        if (r >= 0) { memcpy(fkey, key2, sizeof(uint64_t) * r); memcpy(fact, act2, r); memcpy(fmargin, mg2, sizeof(float) * r); return r; }
    }
    return -1;
}
int or_explain_pixels(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                      const float* conic_opacity, const float* rgb, const float* bg, int npix, const uint32_t* pids,
                      const float* dev_color /*3*npix*/, const float* dev_T, const uint32_t* dev_last, const float* eps /*3*/,
                      float tol_c, float tol_T, int max_flips, int cap, uint64_t* out_key, uint8_t* out_act,
                      float* out_margin, int* status /*npix*/) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    (void)H;
    if (max_flips > 6) max_flips = 6;
    int nout = 0;
    for (int p = 0; p < npix; p++) {
        const uint32_t pid = pids[p];
        const int pxi = (int)(pid % (uint32_t)W), pyi = (int)(pid / (uint32_t)W);
        const int tile = (pyi / BLOCK_Y) * gx + pxi / BLOCK_X;
        const float dc[3] = {dev_color[p], dev_color[(size_t)npix + p], dev_color[2 * (size_t)npix + p]};
        ExplainCtx x = {point_list, ranges[2 * tile], ranges[2 * tile + 1], xy, conic_opacity, rgb, bg, (float)pxi, (float)pyi,
                        dc, dev_T[p], dev_last[p], tol_c, tol_T, eps, (uint64_t)pid};
        uint64_t fkey[8]; uint8_t fact[8]; float fmg[8];
        int found = -1;
        for (int d = 0; d <= max_flips && found < 0; d++) found = explain_rec(&x, fkey, fact, fmg, 0, d);
        status[p] = found;
        if (found > 0) {
            if (nout + found > cap) return -1;
            for (int i = 0; i < found; i++) { out_key[nout] = fkey[i]; out_act[nout] = fact[i]; out_margin[nout] = fmg[i]; nout++; }
        }
    }
    return nout;
}

/* ---- A7: render backward -------------------------------------------------------------------- */
/* Back-to-front per pixel with T recovered by division, exactly the recurrence of the reference
 * kernel; the nine per-Gaussian sums are accumulated in double per list entry, then folded per
 * Gaussian in list order.  Outputs (double): dL_dmean2D[2P], dL_dconic[3P] (A,B,C), dL_dopacity[P],
 * dL_dcolor[3P]. */
int or_render_backward_ex(int P, int W, int H, int64_t D, const uint32_t* ranges, const uint32_t* point_list,
                          const float* xy, const float* conic_opacity, const float* rgb, const float* bg,
                          const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                          int n_over, const uint64_t* over_key, const uint8_t* over_act,
                          double* dL_dmean2D, double* dL_dconic, double* dL_dopacity, double* dL_dcolor);
int or_render_backward(int P, int W, int H, int64_t D, const uint32_t* ranges, const uint32_t* point_list,
                       const float* xy, const float* conic_opacity, const float* rgb, const float* bg,
                       const float* final_T, const uint32_t* n_contrib, const float* dL_dpix /*3HW*/,
                       double* dL_dmean2D, double* dL_dconic, double* dL_dopacity, double* dL_dcolor) {
    return or_render_backward_ex(P, W, H, D, ranges, point_list, xy, conic_opacity, rgb, bg, final_T, n_contrib, dL_dpix, 0,
                                 NULL, NULL, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
}
/* ... with a table of overridden decisions (see "threshold decisions" above): the SKIP / KEEP outcomes are applied
 * here, the STOP / GO outcomes are already in the n_contrib the overridden forward produced. */
int or_render_backward_ex(int P, int W, int H, int64_t D, const uint32_t* ranges, const uint32_t* point_list,
                          const float* xy, const float* conic_opacity, const float* rgb, const float* bg,
                          const float* final_T, const uint32_t* n_contrib, const float* dL_dpix /*3HW*/,
                          int n_over, const uint64_t* over_key, const uint8_t* over_act,
                          double* dL_dmean2D, double* dL_dconic, double* dL_dopacity, double* dL_dcolor) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    double* E = (double*)calloc((size_t)(D > 0 ? D : 1) * 9, sizeof(double));
    if (!E) return -1;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                size_t pid = (size_t)pyi * W + pxi;
                float pxf = (float)pxi, pyf = (float)pyi;
                const float T_final = final_T[pid];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pid];
                float accum_rec[3] = {0, 0, 0}, dLp[3], last_color[3] = {0, 0, 0};
                for (int c = 0; c < 3; c++) dLp[c] = dL_dpix[(size_t)c * H * W + pid];
                float last_alpha = 0.f;
                int olo = 0, ohi = 0;
                if (n_over > 0) ov_slice(n_over, over_key, (uint64_t)pid, &olo, &ohi);
                for (uint32_t jj = r1; jj > r0; jj--) {
                    uint32_t j = jj - 1;
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint8_t act = olo < ohi ? ov_act(over_key, over_act, olo, ohi, j) : 0;
                    if (act & OV_SKIP) continue;
                    const int keep = (act & OV_KEEP) != 0;
                    uint32_t g = point_list[j];
                    float dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
                    const float* co = conic_opacity + 4 * g;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (!keep && power > 0.0f) continue;
                    float G = expf(power);
                    float alpha = fminf(0.99f, co[3] * G);
                    if (!keep && alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    double* e = E + (size_t)j * 9;
                    for (int c = 0; c < 3; c++) {
                        float col = rgb[3 * g + c];
                        accum_rec[c] = last_alpha * last_color[c] + (1.f - last_alpha) * accum_rec[c];
                        last_color[c] = col;
                        dL_dalpha += (col - accum_rec[c]) * dLp[c];
                        e[6 + c] += (double)(dchannel_dcolor * dLp[c]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot = 0.f;
                    for (int c = 0; c < 3; c++) bg_dot += bg[c] * dLp[c];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    float dL_dG = co[3] * dL_dalpha;
                    float gdx = G * dx, gdy = G * dy;
                    float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    float dG_ddely = -gdy * co[2] - gdx * co[1];
                    e[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    e[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    e[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    e[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    e[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    e[5] += (double)(G * dL_dalpha);
                }
            }
    }
    memset(dL_dmean2D, 0, sizeof(double) * 2 * (size_t)P);
    memset(dL_dconic, 0, sizeof(double) * 3 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(double) * (size_t)P);
    memset(dL_dcolor, 0, sizeof(double) * 3 * (size_t)P);
    for (int64_t j = 0; j < D; j++) {
        uint32_t g = point_list[j];
        const double* e = E + (size_t)j * 9;
        dL_dmean2D[2 * g] += e[0]; dL_dmean2D[2 * g + 1] += e[1];
        dL_dconic[3 * g] += e[2]; dL_dconic[3 * g + 1] += e[3]; dL_dconic[3 * g + 2] += e[4];
        dL_dopacity[g] += e[5];
        dL_dcolor[3 * g] += e[6]; dL_dcolor[3 * g + 1] += e[7]; dL_dcolor[3 * g + 2] += e[8];
    }
    free(E);
    return 0;
}

/* ---- A8: preprocess backward ---------------------------------------------------------------- */
/* Inputs: per-Gaussian 2-D gradients (float, as the HIP path would hold them) and the forward
 * state (radii, cov3D, clamped).  Outputs: dL_dmeans3D[3P], dL_dsh[P*M*3], dL_dcov3D[6P],
 * dL_dscale[3P], dL_drot[4P] -- all written in full (zeros where radii <= 0). */
int or_preprocess_backward(const OrArgs* a, const int* radii, const float* cov3D, const uint8_t* clamped,
                           const float* dL_dmean2D /*2P*/, const float* dL_dconic /*3P*/,
                           const float* dL_dcolor /*3P*/, float* dL_dmeans3D, float* dL_dsh,
                           float* dL_dcov3D, float* dL_dscale, float* dL_drot) {
    const int P = a->P, W = a->W, H = a->H, M = a->M;
    const float fy = H / (2.0f * a->tanfovy), fx = W / (2.0f * a->tanfovx);
    const float* V = a->viewmatrix;
    const float* proj = a->projmatrix;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float* gm = dL_dmeans3D + 3 * i;
        gm[0] = gm[1] = gm[2] = 0.f;
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0.f;
        if (dL_dsh) for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = 0.f;
        if (dL_dscale) for (int k = 0; k < 3; k++) dL_dscale[3 * i + k] = 0.f;
        if (dL_drot) for (int k = 0; k < 4; k++) dL_drot[4 * i + k] = 0.f;
        if (!(radii[i] > 0)) continue;
        const float* mean = a->means3D + 3 * i;
        const float* c6 = cov3D + 6 * i;
        /* (i) conic -> cov2D, (ii) cov2D -> cov3D, (iii) cov2D -> mean through J */
        float cov[3], Mx[2][3], t[3], tt[2];
        cov2d(mean, fx, fy, a->tanfovx, a->tanfovy, c6, V, cov, Mx, t, tt);
        const float limx = 1.3f * a->tanfovx, limy = 1.3f * a->tanfovy;
        const float x_grad_mul = (tt[0] < -limx || tt[0] > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tt[1] < -limy || tt[1] > limy) ? 0.f : 1.f;
        float ca = cov[0], cb = cov[1], cc = cov[2];
        float gA = dL_dconic[3 * i], gB = dL_dconic[3 * i + 1], gC = dL_dconic[3 * i + 2];
        float denom = ca * cc - cb * cb;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float* gc = dL_dcov3D + 6 * i;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * gA + 2 * cb * cc * gB + (denom - ca * cc) * gC);
            dL_dc = denom2inv * (-ca * ca * gC + 2 * ca * cb * gB + (denom - ca * cc) * gA);
            dL_db = denom2inv * 2 * (cb * cc * gA - (denom + 2 * cb * cb) * gB + ca * cb * gC);
            const float* m0 = Mx[0]; const float* m1 = Mx[1];
            gc[0] = (m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc);
            gc[3] = (m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc);
            gc[5] = (m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc);
            gc[1] = 2 * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2 * m1[0] * m1[1] * dL_dc;
            gc[2] = 2 * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2 * m1[0] * m1[2] * dL_dc;
            gc[4] = 2 * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2 * m1[1] * m1[2] * dL_dc;
        }
        float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float dM[2][3];
        for (int k = 0; k < 3; k++) {
            float m0S = Mx[0][0] * S[k][0] + Mx[0][1] * S[k][1] + Mx[0][2] * S[k][2];
            float m1S = Mx[1][0] * S[k][0] + Mx[1][1] * S[k][1] + Mx[1][2] * S[k][2];
            dM[0][k] = 2 * m0S * dL_da + m1S * dL_db;
            dM[1][k] = 2 * m1S * dL_dc + m0S * dL_db;
        }
        /* M = J Rv, Rv(i,k) = V[4k+i] */
        float dJ00 = V[0] * dM[0][0] + V[4] * dM[0][1] + V[8] * dM[0][2];
        float dJ02 = V[2] * dM[0][0] + V[6] * dM[0][1] + V[10] * dM[0][2];
        float dJ11 = V[1] * dM[1][0] + V[5] * dM[1][1] + V[9] * dM[1][2];
        float dJ12 = V[2] * dM[1][0] + V[6] * dM[1][1] + V[10] * dM[1][2];
        float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = x_grad_mul * -fx * tz2 * dJ02;
        float dty = y_grad_mul * -fy * tz2 * dJ12;
        float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
        gm[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        gm[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        gm[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
        /* (iv) mean2D -> mean3D through the perspective divide */
        float mh[4];
        xform4x4(mean, proj, mh);
        float mw = 1.0f / (mh[3] + 0.0000001f);
        float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * mw * mw;
        float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * mw * mw;
        float g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
        gm[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
        gm[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
        gm[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
        /* (v) SH backward */
        if (a->shs) {
            const float* sh = a->shs + (size_t)i * M * 3;
            float* gsh = dL_dsh + (size_t)i * M * 3;
            float dor[3] = {mean[0] - a->campos[0], mean[1] - a->campos[1], mean[2] - a->campos[2]};
            float len = sqrtf(dor[0] * dor[0] + dor[1] * dor[1] + dor[2] * dor[2]);
            float x = dor[0] / len, y = dor[1] / len, z = dor[2] / len;
            float gR[3];
            for (int c = 0; c < 3; c++) gR[c] = clamped[3 * i + c] ? 0.f : dL_dcolor[3 * i + c];
            float ddir[3] = {0, 0, 0};
            const int deg = a->deg;
            for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k)*3 + c]
#define GSH(k) gsh[(k)*3 + c]
                float dx = 0, dy = 0, dz = 0;
                GSH(0) = SH_C0 * gR[c];
                if (deg > 0) {
                    GSH(1) = -SH_C1 * y * gR[c]; GSH(2) = SH_C1 * z * gR[c]; GSH(3) = -SH_C1 * x * gR[c];
                    dx = -SH_C1 * SH(3); dy = -SH_C1 * SH(1); dz = SH_C1 * SH(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
                        GSH(4) = SH_C2[0] * xy_ * gR[c]; GSH(5) = SH_C2[1] * yz * gR[c];
                        GSH(6) = SH_C2[2] * (2.f * zz - xx - yy) * gR[c];
                        GSH(7) = SH_C2[3] * xz * gR[c]; GSH(8) = SH_C2[4] * (xx - yy) * gR[c];
                        dx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                        dy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                        dz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                        if (deg > 2) {
                            GSH(9) = SH_C3[0] * y * (3.f * xx - yy) * gR[c];
                            GSH(10) = SH_C3[1] * xy_ * z * gR[c];
                            GSH(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * gR[c];
                            GSH(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gR[c];
                            GSH(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * gR[c];
                            GSH(14) = SH_C3[5] * z * (xx - yy) * gR[c];
                            GSH(15) = SH_C3[6] * x * (xx - 3.f * yy) * gR[c];
                            dx += SH_C3[0] * SH(9) * 3.f * 2.f * xy_ + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy_ +
                                  SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy);
                            dy += SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz + SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy_ +
                                  SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy_;
                            dz += SH_C3[1] * SH(10) * xy_ + SH_C3[2] * SH(11) * 4.f * 2.f * yz + SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy);
                        }
                    }
                }
#undef SH
#undef GSH
                ddir[0] += dx * gR[c]; ddir[1] += dy * gR[c]; ddir[2] += dz * gR[c];
            }
            /* through dir = v/|v| */
            float sum2 = dor[0] * dor[0] + dor[1] * dor[1] + dor[2] * dor[2];
            float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            gm[0] += ((+sum2 - dor[0] * dor[0]) * ddir[0] - dor[1] * dor[0] * ddir[1] - dor[2] * dor[0] * ddir[2]) * invsum32;
            gm[1] += (-dor[0] * dor[1] * ddir[0] + (sum2 - dor[1] * dor[1]) * ddir[1] - dor[2] * dor[1] * ddir[2]) * invsum32;
            gm[2] += (-dor[0] * dor[2] * ddir[0] - dor[1] * dor[2] * ddir[1] + (sum2 - dor[2] * dor[2]) * ddir[2]) * invsum32;
        }
        /* (vi) cov3D -> scale, rotation.  Sigma = L L^T, L = R S.  As in the reference kernel the
         * scale gradient is taken w.r.t. s = mod*scale and NOT multiplied by mod (exact for
         * mod = 1, the only value the reference passes: gaussian_renderer/__init__.py:64). */
        if (a->scales) {
            const float* q = a->rotations + 4 * i;
            float R[3][3];
            quat_to_R(q, R);
            float s[3] = {a->scale_modifier * a->scales[3 * i], a->scale_modifier * a->scales[3 * i + 1],
                          a->scale_modifier * a->scales[3 * i + 2]};
            /* dL/dSigma symmetric with halved off-diagonals (six-vector carries the x2) */
            float dS[3][3] = {{gc[0], 0.5f * gc[1], 0.5f * gc[2]},
                              {0.5f * gc[1], gc[3], 0.5f * gc[4]},
                              {0.5f * gc[2], 0.5f * gc[4], gc[5]}};
            /* dL/dL = 2 dS L, L = R diag(s) */
            float L[3][3], dL[3][3];
            for (int r_ = 0; r_ < 3; r_++) for (int c_ = 0; c_ < 3; c_++) L[r_][c_] = R[r_][c_] * s[c_];
            for (int r_ = 0; r_ < 3; r_++)
                for (int c_ = 0; c_ < 3; c_++)
                    dL[r_][c_] = 2.0f * (dS[r_][0] * L[0][c_] + dS[r_][1] * L[1][c_] + dS[r_][2] * L[2][c_]);
            /* dL/ds_c = sum_r R[r][c] dL[r][c];  dL/dR[r][c] = dL[r][c] s_c */
            float dR[3][3];
            for (int c_ = 0; c_ < 3; c_++) {
                dL_dscale[3 * i + c_] = R[0][c_] * dL[0][c_] + R[1][c_] * dL[1][c_] + R[2][c_] * dL[2][c_];
                for (int r_ = 0; r_ < 3; r_++) dR[r_][c_] = dL[r_][c_] * s[c_];
            }
            float r = q[0], x = q[1], y = q[2], z = q[3];
            float* gq = dL_drot + 4 * i;
            gq[0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
            gq[1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) - 4 * x * (dR[2][2] + dR[1][1]);
            gq[2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[2][1] + dR[1][2]) - 4 * y * (dR[2][2] + dR[0][0]);
            gq[3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[2][1] + dR[1][2]) - 4 * z * (dR[1][1] + dR[0][0]);
        }
    }
    return 0;
}

/* ---- A9: distCUDA2 --------------------------------------------------------------------------- */
/* Exact mean of the squared distances to the 3 nearest OTHER points (self excluded by index, so
 * duplicated points give 0 and the caller's clamp 1e-7 applies: scene/gaussian_model.py:186).
 * Brute force O(N^2); distance d.x*d.x + d.y*d.y + d.z*d.z in fp32, best-3 kept ascending,
 * result (b0 + b1 + b2) / 3.0f. */
int or_dist2(int P, const float* pts, float* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        const float* p = pts + 3 * i;
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = p[0] - pts[3 * j], dy = p[1] - pts[3 * j + 1], dz = p[2] - pts[3 * j + 2];
            float d = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++)
                if (best[k] > d) { float tmp = best[k]; best[k] = d; d = tmp; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    return 0;
}

/* N2: image L1 loss of utils/loss_utils.py:21-22, torch.abs(network_output - gt).mean(), and the
 * gradient autograd derives for it: sign(x - y) / n (sign(0) = 0; 1/n rounded to fp32 as torch's mean
 * backward does).  The sum is kept in double (the order of a float reduction is unspecified). */
double or_l1_loss(long long n, const float* x, const float* y, float* grad) {
    double acc = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (long long i = 0; i < n; i++) {
        const float d = x[i] - y[i];
        acc += fabs((double)d);
        if (grad) grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
    return acc / (double)n;
}

/* N2: SSIM of utils/loss_utils.py:27-67 -- 11x11 window = outer product of the normalised 1-D Gaussian
 * (sigma 1.5; 1-D weights and their products held in fp32 as the reference's tensors are), conv2d with zero
 * padding 5 per channel, C1 = 0.01^2, C2 = 0.03^2, mean over all C*H*W elements -- and the gradient of that
 * mean w.r.t. img1 (what autograd derives).  Accumulation in double: direct 121-tap sums. */
double or_ssim(int C, int H, int W, const float* img1, const float* img2, float* grad1) {
    /* the divisor is torch's gauss.sum(): the correctly rounded sum of the eleven fp32 values (3.7592328; a sequential
     * fp32 sum gives 3.7592325) -- pinned by the reference's own window, tests/golden/losses.npz */
    float g1[11];
    double sd = 0.0;
    for (int k = 0; k < 11; k++) { g1[k] = (float)exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); sd += (double)g1[k]; }
    const float s = (float)sd;
    for (int k = 0; k < 11; k++) g1[k] /= s;
    float w2[11][11];
    for (int i = 0; i < 11; i++)
        for (int j = 0; j < 11; j++) w2[i][j] = g1[i] * g1[j];
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    const size_t n = (size_t)C * H * W;
    double* dmu = (double*)malloc(n * sizeof(double));
    double* ds1 = (double*)malloc(n * sizeof(double));
    double* ds12 = (double*)malloc(n * sizeof(double));
    double total = 0.0;
    for (int c = 0; c < C; c++) {
        const float* a = img1 + (size_t)c * H * W;
        const float* b = img2 + (size_t)c * H * W;
        double chan = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : chan)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
                for (int i = 0; i < 11; i++) {
                    const int yy = y + i - 5;
                    if (yy < 0 || yy >= H) continue;
                    for (int j = 0; j < 11; j++) {
                        const int xx = x + j - 5;
                        if (xx < 0 || xx >= W) continue;
                        const double w = w2[i][j], p = a[(size_t)yy * W + xx], q = b[(size_t)yy * W + xx];
                        mu1 += w * p; mu2 += w * q; e11 += w * p * p; e22 += w * q * q; e12 += w * p * q;
                    }
                }
                const double s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
                const double A = 2 * mu1 * mu2 + C1, B = 2 * s12 + C2, Cc = mu1 * mu1 + mu2 * mu2 + C1, D = s1 + s2 + C2;
                const double v = A * B / (Cc * D);
                chan += v;
                const size_t o = (size_t)c * H * W + (size_t)y * W + x;
                ds1[o] = -v / D;
                ds12[o] = 2 * A / (Cc * D);
                dmu[o] = 2 * mu2 * B / (Cc * D) - 2 * mu1 * v / Cc - 2 * mu1 * ds1[o] - mu2 * ds12[o];
            }
        total += chan;
    }
    if (grad1) {
        for (int c = 0; c < C; c++) {
            const float* a = img1 + (size_t)c * H * W;
            const float* b = img2 + (size_t)c * H * W;
#pragma omp parallel for schedule(static)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    double f0 = 0, f1 = 0, f2 = 0;
                    for (int i = 0; i < 11; i++) {
                        const int yy = y + i - 5;
                        if (yy < 0 || yy >= H) continue;
                        for (int j = 0; j < 11; j++) {
                            const int xx = x + j - 5;
                            if (xx < 0 || xx >= W) continue;
                            const size_t o = (size_t)c * H * W + (size_t)yy * W + xx;
                            const double w = w2[i][j];
                            f0 += w * dmu[o]; f1 += w * ds1[o]; f2 += w * ds12[o];
                        }
                    }
                    const size_t o = (size_t)y * W + x;
                    grad1[(size_t)c * H * W + o] = (float)((f0 + 2.0 * a[o] * f1 + b[o] * f2) / (double)n);
                }
        }
    }
    free(dmu); free(ds1); free(ds12);
    return total / (double)n;
}

/* ---- N3: the two per-Gaussian torch chains in front of the rasterizer, in double ---- */
static void n3_rotation(const float* rot, int is_matrix, int i, double R[3][3], double q[4], double* norm) {
    if (is_matrix) {
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) R[a][b] = rot[9 * (size_t)i + 3 * a + b];
        q[0] = 1; q[1] = q[2] = q[3] = 0; *norm = 1;
        return;
    }
    const float* r4 = rot + 4 * (size_t)i;  /* utils/general_utils.py:87-108: q = r / |r|, (w, x, y, z) */
    const double n = sqrt((double)r4[0] * r4[0] + (double)r4[1] * r4[1] + (double)r4[2] * r4[2] + (double)r4[3] * r4[3]);
    for (int k = 0; k < 4; k++) q[k] = r4[k] / n;
    *norm = n;
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - r * z); R[0][2] = 2 * (x * z + r * y);
    R[1][0] = 2 * (x * y + r * z); R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - r * x);
    R[2][0] = 2 * (x * z - r * y); R[2][1] = 2 * (y * z + r * x); R[2][2] = 1 - 2 * (x * x + y * y);
}

/* scene/gaussian_model.py:28-32: strip_symmetric(L L^T), L = R diag(mod * scaling) (general_utils.py:194-207).
 * dL_dcov6 != NULL: also the gradients autograd derives (strip_symmetric reads the UPPER triangle). */
int or_build_covariance(int N, const float* scaling, float mod, const float* rot, int is_matrix, float* cov6,
                        const float* dL_dcov6, float* dL_dscaling, float* dL_drot) {
    for (int i = 0; i < N; i++) {
        double R[3][3], q[4], n;
        n3_rotation(rot, is_matrix, i, R, q, &n);
        const double s[3] = {(double)mod * scaling[3 * i], (double)mod * scaling[3 * i + 1], (double)mod * scaling[3 * i + 2]};
        double L[3][3], S[3][3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) L[a][b] = R[a][b] * s[b];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) S[a][b] = L[a][0] * L[b][0] + L[a][1] * L[b][1] + L[a][2] * L[b][2];
        if (cov6) {
            float* o = cov6 + 6 * (size_t)i;
            o[0] = (float)S[0][0]; o[1] = (float)S[0][1]; o[2] = (float)S[0][2];
            o[3] = (float)S[1][1]; o[4] = (float)S[1][2]; o[5] = (float)S[2][2];
        }
        if (!dL_dcov6) continue;
        const float* g = dL_dcov6 + 6 * (size_t)i;
        const double G[3][3] = {{g[0], g[1], g[2]}, {0, g[3], g[4]}, {0, 0, g[5]}};
        double dLm[3][3], dR[3][3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double v = 0;
                for (int k = 0; k < 3; k++) v += (G[a][k] + G[k][a]) * L[k][b];
                dLm[a][b] = v;
            }
        for (int b = 0; b < 3; b++) {
            double v = 0;
            for (int a = 0; a < 3; a++) v += dLm[a][b] * R[a][b];
            dL_dscaling[3 * (size_t)i + b] = (float)(mod * v);
        }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) dR[a][b] = dLm[a][b] * s[b];
        if (is_matrix) {
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) dL_drot[9 * (size_t)i + 3 * a + b] = (float)dR[a][b];
        } else {
            const double r = q[0], x = q[1], y = q[2], z = q[3];
            double gq[4];
            gq[0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
            gq[1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) - 4 * x * (dR[2][2] + dR[1][1]);
            gq[2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[2][1] + dR[1][2]) - 4 * y * (dR[2][2] + dR[0][0]);
            gq[3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[2][1] + dR[1][2]) - 4 * z * (dR[1][1] + dR[0][0]);
            const double dot = q[0] * gq[0] + q[1] * gq[1] + q[2] * gq[2] + q[3] * gq[3];
            for (int k = 0; k < 4; k++) dL_drot[4 * (size_t)i + k] = (float)((gq[k] - q[k] * dot) / n);
        }
    }
    return 0;
}

/* models/texture/texture.py:21-38 SH2RGB.forward (+ backward when dL_dcolors != NULL).  shs is (N, M, 3);
 * R_fwd (N,3,3) or NULL; noise 3x3 or NULL (dir @ noise).  eval_sh as utils/sh_utils.py:58-101, in double. */
int or_sh2rgb(int N, int deg, int M, const float* shs, const float* xyz, const float* campos, const float* R_fwd,
              const float* noise, float* colors, uint8_t* clamped, const float* dL_dcolors, float* dL_dshs, float* dL_dxyz) {
    const double C0 = 0.28209479177387814, C1 = 0.4886025119029199;
    const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
    const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                          -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
    for (int i = 0; i < N; i++) {
        double v[3] = {(double)xyz[3 * i] - campos[0], (double)xyz[3 * i + 1] - campos[1], (double)xyz[3 * i + 2] - campos[2]};
        if (R_fwd) {
            const float* R = R_fwd + 9 * (size_t)i;
            const double w[3] = {R[0] * v[0] + R[3] * v[1] + R[6] * v[2], R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
                                 R[2] * v[0] + R[5] * v[1] + R[8] * v[2]};
            v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
        }
        if (noise) {
            const double w[3] = {v[0] * noise[0] + v[1] * noise[3] + v[2] * noise[6], v[0] * noise[1] + v[1] * noise[4] + v[2] * noise[7],
                                 v[0] * noise[2] + v[1] * noise[5] + v[2] * noise[8]};
            v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
        }
        const double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), inv = 1.0 / (len + 1e-12);
        const double x = v[0] * inv, y = v[1] * inv, z = v[2] * inv;
        const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        double basis[16] = {0}, dbx[16] = {0}, dby[16] = {0}, dbz[16] = {0};
        basis[0] = C0;
        if (deg > 0) {
            basis[1] = -C1 * y; dby[1] = -C1;
            basis[2] = C1 * z; dbz[2] = C1;
            basis[3] = -C1 * x; dbx[3] = -C1;
        }
        if (deg > 1) {
            basis[4] = C2[0] * xy; dbx[4] = C2[0] * y; dby[4] = C2[0] * x;
            basis[5] = C2[1] * yz; dby[5] = C2[1] * z; dbz[5] = C2[1] * y;
            basis[6] = C2[2] * (2 * zz - xx - yy); dbx[6] = C2[2] * -2 * x; dby[6] = C2[2] * -2 * y; dbz[6] = C2[2] * 4 * z;
            basis[7] = C2[3] * xz; dbx[7] = C2[3] * z; dbz[7] = C2[3] * x;
            basis[8] = C2[4] * (xx - yy); dbx[8] = C2[4] * 2 * x; dby[8] = C2[4] * -2 * y;
        }
        if (deg > 2) {
            basis[9] = C3[0] * y * (3 * xx - yy); dbx[9] = C3[0] * 6 * xy; dby[9] = C3[0] * (3 * xx - 3 * yy);
            basis[10] = C3[1] * xy * z; dbx[10] = C3[1] * yz; dby[10] = C3[1] * xz; dbz[10] = C3[1] * xy;
            basis[11] = C3[2] * y * (4 * zz - xx - yy); dbx[11] = C3[2] * -2 * xy; dby[11] = C3[2] * (4 * zz - xx - 3 * yy); dbz[11] = C3[2] * 8 * yz;
            basis[12] = C3[3] * z * (2 * zz - 3 * xx - 3 * yy); dbx[12] = C3[3] * -6 * xz; dby[12] = C3[3] * -6 * yz; dbz[12] = C3[3] * (6 * zz - 3 * xx - 3 * yy);
            basis[13] = C3[4] * x * (4 * zz - xx - yy); dbx[13] = C3[4] * (4 * zz - 3 * xx - yy); dby[13] = C3[4] * -2 * xy; dbz[13] = C3[4] * 8 * xz;
            basis[14] = C3[5] * z * (xx - yy); dbx[14] = C3[5] * 2 * xz; dby[14] = C3[5] * -2 * yz; dbz[14] = C3[5] * (xx - yy);
            basis[15] = C3[6] * x * (xx - 3 * yy); dbx[15] = C3[6] * (3 * xx - 3 * yy); dby[15] = C3[6] * -6 * xy;
        }
        const int nb = (deg + 1) * (deg + 1);
        const float* sh = shs + (size_t)i * M * 3;
        double ddir[3] = {0, 0, 0};
        uint8_t cl = 0;
        for (int c = 0; c < 3; c++) {
            double r = 0;
            for (int k = 0; k < nb; k++) r += basis[k] * sh[k * 3 + c];
            r += 0.5;
            if (r < 0) cl |= (uint8_t)(1u << c);
            if (colors) colors[3 * (size_t)i + c] = (float)(r < 0 ? 0 : r);
            if (dL_dcolors) {
                const double g = (r < 0) ? 0.0 : dL_dcolors[3 * (size_t)i + c];
                for (int k = 0; k < M; k++) dL_dshs[((size_t)i * M + k) * 3 + c] = (float)(k < nb ? basis[k] * g : 0.0);
                for (int k = 0; k < nb; k++) {
                    ddir[0] += dbx[k] * sh[k * 3 + c] * g;
                    ddir[1] += dby[k] * sh[k * 3 + c] * g;
                    ddir[2] += dbz[k] * sh[k * 3 + c] * g;
                }
            }
        }
        if (clamped) clamped[i] = cl;
        if (!dL_dcolors) continue;
        const double dotv = v[0] * ddir[0] + v[1] * ddir[1] + v[2] * ddir[2];
        const double k2 = len > 0 ? dotv * inv * inv / len : 0.0;
        double gv[3] = {ddir[0] * inv - v[0] * k2, ddir[1] * inv - v[1] * k2, ddir[2] * inv - v[2] * k2};
        if (noise) {
            const double w[3] = {noise[0] * gv[0] + noise[1] * gv[1] + noise[2] * gv[2], noise[3] * gv[0] + noise[4] * gv[1] + noise[5] * gv[2],
                                 noise[6] * gv[0] + noise[7] * gv[1] + noise[8] * gv[2]};
            gv[0] = w[0]; gv[1] = w[1]; gv[2] = w[2];
        }
        if (R_fwd) {
            const float* R = R_fwd + 9 * (size_t)i;
            const double w[3] = {R[0] * gv[0] + R[1] * gv[1] + R[2] * gv[2], R[3] * gv[0] + R[4] * gv[1] + R[5] * gv[2],
                                 R[6] * gv[0] + R[7] * gv[1] + R[8] * gv[2]};
            gv[0] = w[0]; gv[1] = w[1]; gv[2] = w[2];
        }
        for (int k = 0; k < 3; k++) dL_dxyz[3 * (size_t)i + k] = (float)gv[k];
    }
    return 0;
}

/* N4: K nearest reference points of every query, brute force (pytorch3d.ops.knn_points semantics as the
 * reference uses them: squared distances ascending; a query contained in ref finds itself first).  Distance
 * dx*dx + dy*dy + dz*dz in fp32; ties broken by the smaller reference index. */
int or_knn_points(int Nq, const float* q, int Nr, const float* ref, int K, float* dists, long long* idx) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < Nq; i++) {
        float bd[8];
        long long bi[8];
        for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = -1; }
        const float* p = q + 3 * (size_t)i;
        for (int j = 0; j < Nr; j++) {
            const float dx = p[0] - ref[3 * (size_t)j], dy = p[1] - ref[3 * (size_t)j + 1], dz = p[2] - ref[3 * (size_t)j + 2];
            float d = dx * dx + dy * dy + dz * dz;
            long long id = j;
            for (int k = 0; k < K; k++)
                if (bd[k] > d || (bd[k] == d && (bi[k] > id || bi[k] < 0))) {
                    const float td = bd[k]; const long long ti = bi[k];
                    bd[k] = d; bi[k] = id; d = td; id = ti;
                }
        }
        for (int k = 0; k < K; k++) { dists[(size_t)i * K + k] = bd[k]; idx[(size_t)i * K + k] = bi[k]; }
    }
    return 0;
}

void or_set_num_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int or_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
