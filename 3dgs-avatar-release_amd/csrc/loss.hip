// loss.hip -- image-side L1 loss (replaces utils/loss_utils.py:21-22,
// torch.abs(network_output - gt).mean(), and its autograd chain of ~10 elementwise kernels).
// One streaming pass writes d(loss)/d(network_output) = sign(x - y) / n and one partial sum of
// |x - y| per block; a second, single-block pass adds the partials in index order.  No atomics: the
// result is bitwise reproducible.
#include "common.h"
#include <math.h>

#define L1_THREADS 256
#define L1_MAX_BLOCKS 1024

__device__ __forceinline__ float sgn_over_n(float d, float inv_n) { return d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f); }

__global__ __launch_bounds__(L1_THREADS) void l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                int64_t n, float inv_n, float* __restrict__ grad,
                                                                float* __restrict__ partial) {
    __shared__ float ws[L1_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    float4* g4 = reinterpret_cast<float4*>(grad);
    const int64_t stride = (int64_t)gridDim.x * L1_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * L1_THREADS + tid; i < n4; i += stride) {
        const float4 a = x4[i], b = y4[i];
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        g4[i] = make_float4(sgn_over_n(d0, inv_n), sgn_over_n(d1, inv_n), sgn_over_n(d2, inv_n), sgn_over_n(d3, inv_n));
    }
    if (blockIdx.x == 0 && tid < (int)(n & 3)) {  // the last n mod 4 elements
        const int64_t i = (n4 << 2) + tid;
        const float d = x[i] - y[i];
        acc += fabsf(d);
        grad[i] = sgn_over_n(d, inv_n);
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k, 64);
    if (lane == 0) ws[wid] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(L1_THREADS) void l1_final_kernel(const float* __restrict__ partial, int nparts, float inv_n,
                                                              float* __restrict__ loss) {
    __shared__ float ws[L1_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float acc = 0.f;
    for (int i = tid; i < nparts; i += L1_THREADS) acc += partial[i];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k, 64);
    if (lane == 0) ws[wid] = acc;
    __syncthreads();
    if (tid == 0) loss[0] = ((ws[0] + ws[1]) + (ws[2] + ws[3])) * inv_n;
}

// Mask loss, BCE form (train.py:146-148): F.binary_cross_entropy(torch.clamp(x, 1e-3, 1 - 1e-3), y), mean over all
// elements, and its gradient w.r.t. x: (p - y) / (p (1 - p)) / n inside the clamp range, 0 outside.  Same
// two-stage ordered reduction as the L1 loss.
__global__ __launch_bounds__(L1_THREADS) void bce_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 int64_t n, float inv_n, float* __restrict__ grad,
                                                                 float* __restrict__ partial) {
    __shared__ float ws[L1_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * L1_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * L1_THREADS + tid; i < n; i += stride) {
        const float xv = x[i], t = y[i];
        const float p = fminf(fmaxf(xv, 1.0e-3f), 1.0f - 1.0e-3f);
        acc -= t * logf(p) + (1.0f - t) * logf(1.0f - p);
        const bool inside = xv >= 1.0e-3f && xv <= 1.0f - 1.0e-3f;  // clamp passes the gradient on its closed range
        grad[i] = inside ? (p - t) / (p * (1.0f - p)) * inv_n : 0.f;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k, 64);
    if (lane == 0) ws[wid] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

static int l1_blocks(int64_t n) {
    const int64_t want = ((n >> 2) + L1_THREADS - 1) / L1_THREADS;
    return (int)(want < 1 ? 1 : (want > L1_MAX_BLOCKS ? L1_MAX_BLOCKS : want));
}

size_t l1_ws_bytes(int64_t n) { return (size_t)l1_blocks(n) * sizeof(float); }

int launch_bce_loss(const float* x, const float* y, int64_t n, float* loss, float* grad, float* partial, hipStream_t s) {
    const int blocks = l1_blocks(n);
    const float inv_n = 1.0f / (float)n;
    StageScope st("bce_loss", s);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(blocks), dim3(L1_THREADS), 0, s, x, y, n, inv_n, grad, partial);
    GS_LAUNCH_CHECK("bce_loss.partial", 0, s);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(L1_THREADS), 0, s, partial, blocks, inv_n, loss);
    GS_LAUNCH_CHECK("bce_loss.final", 0, s);
    return GS_OK;
}

int launch_l1_loss(const float* x, const float* y, int64_t n, float* loss, float* grad, float* partial, hipStream_t s) {
    const int blocks = l1_blocks(n);
    const float inv_n = 1.0f / (float)n;
    StageScope st("l1_loss", s);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(L1_THREADS), 0, s, x, y, n, inv_n, grad, partial);
    GS_LAUNCH_CHECK("l1_loss.partial", 0, s);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(L1_THREADS), 0, s, partial, blocks, inv_n, loss);
    GS_LAUNCH_CHECK("l1_loss.final", 0, s);
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// SSIM (utils/loss_utils.py:27-67: 11x11 Gaussian window, sigma 1.5, zero
// padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements) with its gradient w.r.t. the first image.
// The reference runs five depthwise conv2d launches plus ~15 elementwise ones and lets autograd replay
// them backwards.  Here: one forward kernel (32x32 output tile per workgroup, 42x42 halo tiles of both
// images in LDS, the 121-tap window applied as two 11-tap passes, five moments at once) that also
// emits the three partial-derivative maps the backward needs, and one backward kernel that filters
// those three maps with the same window and combines them with the images.  Reductions are two-stage
// and ordered: bitwise reproducible.
// ---------------------------------------------------------------------------------------------
#define SS_T 32            // output tile edge (32 x 32 outputs per workgroup of 256 threads: four per thread)
#define SS_NT 256
#define SS_R 5             // window radius
#define SS_H (SS_T + 2 * SS_R)  // halo tile edge (42)
#define SS_C1 0.0001f
#define SS_C2 0.0009f

struct SsimWindow { float w[11]; };

__global__ __launch_bounds__(SS_NT) void ssim_fwd_kernel(int H, int W, const float* __restrict__ img1,
                                                         const float* __restrict__ img2, SsimWindow win,
                                                         float* __restrict__ dm_dmu1, float* __restrict__ dm_ds1,
                                                         float* __restrict__ dm_ds12, float* __restrict__ partial) {
    __shared__ float t1[SS_H][SS_H + 1], t2[SS_H][SS_H + 1];
    __shared__ float hx[5][SS_H][SS_T + 1];
    __shared__ float ws[SS_NT / 64];
    const int tid = threadIdx.x, tx = tid % SS_T, ty = tid / SS_T;  // ty in 0..7: rows ty, ty + 8, ty + 16, ty + 24
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int e = tid; e < SS_H * SS_H; e += SS_NT) {
        const int r = e / SS_H, c = e - r * SS_H;
        const int y = y0 + r - SS_R, x = x0 + c - SS_R;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;  // zero padding (conv2d padding = 5)
        t1[r][c] = in ? img1[plane + (size_t)y * W + x] : 0.f;
        t2[r][c] = in ? img2[plane + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    // horizontal pass: 42 rows x 32 columns, five moments
    for (int e = tid; e < SS_H * SS_T; e += SS_NT) {
        const int r = e / SS_T, c = e - r * SS_T;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float a = t1[r][c + k], b = t2[r][c + k], w = win.w[k];
            m1 += w * a;
            m2 += w * b;
            e11 += w * (a * a);
            e22 += w * (b * b);
            e12 += w * (a * b);
        }
        hx[0][r][c] = m1; hx[1][r][c] = m2; hx[2][r][c] = e11; hx[3][r][c] = e22; hx[4][r][c] = e12;
    }
    __syncthreads();
    float val = 0.f;
#pragma unroll
    for (int q = 0; q < SS_T * SS_T / SS_NT; q++) {
        const int oy = ty + q * (SS_NT / SS_T);
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.w[k];
            mu1 += w * hx[0][oy + k][tx];
            mu2 += w * hx[1][oy + k][tx];
            e11 += w * hx[2][oy + k][tx];
            e22 += w * hx[3][oy + k][tx];
            e12 += w * hx[4][oy + k][tx];
        }
        const int x = x0 + tx, y = y0 + oy;
        if (x < W && y < H) {
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
            const float A = 2.f * mu12 + SS_C1, B = 2.f * s12 + SS_C2;
            const float C = mu1_sq + mu2_sq + SS_C1, D = s1 + s2 + SS_C2;
            const float inv_cd = 1.0f / (C * D);
            const float v = A * B * inv_cd;
            val += v;
            if (dm_dmu1) {
                // ssim as a function of (mu1, E[x^2], E[xy]) of this window; s1 and s12 depend on mu1 too
                const float d_s1 = -v / D;              // d ssim / d s1   = -A B / (C D^2)
                const float d_s12 = 2.f * A * inv_cd;   // d ssim / d s12  =  2 A / (C D)
                const float d_mu1 = 2.f * mu2 * B * inv_cd - 2.f * mu1 * v / C - 2.f * mu1 * d_s1 - mu2 * d_s12;
                const size_t o = plane + (size_t)y * W + x;
                dm_dmu1[o] = d_mu1;
                dm_ds1[o] = d_s1;
                dm_ds12[o] = d_s12;
            }
        }
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) val += __shfl_xor(val, k, 64);
    if ((tid & 63) == 0) ws[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(256) void ssim_final_kernel(const float* __restrict__ partial, int nparts, float inv_n,
                                                         float* __restrict__ out) {
    __shared__ float ws[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float acc = 0.f;
    for (int i = tid; i < nparts; i += 256) acc += partial[i];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k, 64);
    if (lane == 0) ws[wid] = acc;
    __syncthreads();
    if (tid == 0) out[0] = ((ws[0] + ws[1]) + (ws[2] + ws[3])) * inv_n;
}

// dL/dimg1(p) = g/n * sum_q w(q - p) [ dm_dmu1(q) + 2 img1(p) dm_ds1(q) + img2(p) dm_ds12(q) ]
__global__ __launch_bounds__(SS_NT) void ssim_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                         const float* __restrict__ img2, SsimWindow win,
                                                         const float* __restrict__ dm_dmu1,
                                                         const float* __restrict__ dm_ds1,
                                                         const float* __restrict__ dm_ds12,
                                                         const float* __restrict__ dL_dssim, float inv_n,
                                                         float* __restrict__ dL_dimg1) {
    __shared__ float t[3][SS_H][SS_H + 1];
    __shared__ float hx[3][SS_H][SS_T + 1];
    const int tid = threadIdx.x, tx = tid % SS_T, ty = tid / SS_T;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int e = tid; e < SS_H * SS_H; e += SS_NT) {
        const int r = e / SS_H, c = e - r * SS_H;
        const int y = y0 + r - SS_R, x = x0 + c - SS_R;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;  // windows centred outside the image do not exist
        const size_t o = plane + (size_t)y * W + x;
        t[0][r][c] = in ? dm_dmu1[o] : 0.f;
        t[1][r][c] = in ? dm_ds1[o] : 0.f;
        t[2][r][c] = in ? dm_ds12[o] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < SS_H * SS_T; e += SS_NT) {
        const int r = e / SS_T, c = e - r * SS_T;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.w[k];
            a0 += w * t[0][r][c + k];
            a1 += w * t[1][r][c + k];
            a2 += w * t[2][r][c + k];
        }
        hx[0][r][c] = a0; hx[1][r][c] = a1; hx[2][r][c] = a2;
    }
    __syncthreads();
    const float g = dL_dssim[0] * inv_n;
#pragma unroll
    for (int q = 0; q < SS_T * SS_T / SS_NT; q++) {
        const int oy = ty + q * (SS_NT / SS_T);
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = win.w[k];
            f0 += w * hx[0][oy + k][tx];
            f1 += w * hx[1][oy + k][tx];
            f2 += w * hx[2][oy + k][tx];
        }
        const int x = x0 + tx, y = y0 + oy;
        if (x < W && y < H) {
            const size_t o = plane + (size_t)y * W + x;
            dL_dimg1[o] = g * (f0 + 2.f * img1[o] * f1 + img2[o] * f2);
        }
    }
}

static SsimWindow ssim_window() {
    // utils/loss_utils.py:27-29: exp(-(x - 5)^2 / (2 * 1.5^2)) evaluated in double, stored as fp32, normalised in fp32
    SsimWindow w;
    float s = 0.f;
    for (int k = 0; k < 11; k++) {
        w.w[k] = (float)exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5));
        s += w.w[k];
    }
    for (int k = 0; k < 11; k++) w.w[k] /= s;
    return w;
}

static inline int ssim_tiles(int v) { return (v + SS_T - 1) / SS_T; }
size_t ssim_ws_bytes(int C, int H, int W) { return (size_t)C * ssim_tiles(H) * ssim_tiles(W) * sizeof(float); }

int launch_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_out, float* dm_dmu1,
                        float* dm_ds1, float* dm_ds12, float* partial, hipStream_t s) {
    const dim3 grid(ssim_tiles(W), ssim_tiles(H), C);
    StageScope st("ssim_fwd", s);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(SS_NT), 0, s, H, W, img1, img2, ssim_window(), dm_dmu1, dm_ds1,
                       dm_ds12, partial);
    GS_LAUNCH_CHECK("ssim.forward", 0, s);
    hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(256), 0, s, partial, (int)(grid.x * grid.y * grid.z),
                       1.0f / ((float)C * (float)H * (float)W), ssim_out);
    GS_LAUNCH_CHECK("ssim.final", 0, s);
    return GS_OK;
}

int launch_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                         const float* dm_ds1, const float* dm_ds12, const float* dL_dssim, float* dL_dimg1,
                         hipStream_t s) {
    const dim3 grid(ssim_tiles(W), ssim_tiles(H), C);
    StageScope st("ssim_bwd", s);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(SS_NT), 0, s, H, W, img1, img2, ssim_window(), dm_dmu1, dm_ds1,
                       dm_ds12, dL_dssim, 1.0f / ((float)C * (float)H * (float)W), dL_dimg1);
    GS_LAUNCH_CHECK("ssim.backward", 0, s);
    return GS_OK;
}
