// optim.hip -- the per-Gaussian bookkeeping that follows the backward pass in the reference's training
// step:
//   * densification statistics: train.py:219-220 + scene/gaussian_model.py:464-466 -- three boolean-mask
//     indexed updates in torch (each a nonzero() with a host sync, gathers and scatters);
//   * Adam: scene/gaussian_model.py:201-216 builds torch.optim.Adam over six parameter groups
//     (lr per group, eps 1e-15): the foreach implementation launches ~10 kernels per group.
// Here: one streaming kernel for the statistics and ONE launch for the Adam step of all tensors.
#include "common.h"

__global__ __launch_bounds__(256) void densify_stats_kernel(int N, const int32_t* __restrict__ radii,
                                                            const float* __restrict__ viewspace_grad,
                                                            float* __restrict__ max_radii2D,
                                                            float* __restrict__ xyz_gradient_accum,
                                                            float* __restrict__ denom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = radii[i];
    if (r <= 0) return;  // visibility_filter = radii > 0
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = viewspace_grad[3 * (size_t)i], gy = viewspace_grad[3 * (size_t)i + 1];
    xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy);  // torch.norm(grad[:, :2], dim=-1)
    denom[i] += 1.0f;
}

int launch_densify_stats(int N, const int32_t* radii, const float* viewspace_grad, float* max_radii2D,
                         float* xyz_gradient_accum, float* denom, hipStream_t s) {
    StageScope st("densify_stats", s);
    hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, radii, viewspace_grad, max_radii2D,
                       xyz_gradient_accum, denom);
    GS_LAUNCH_CHECK("densify_stats", 0, s);
    return GS_OK;
}

// torch.optim.Adam (no amsgrad, no weight decay, maximize = false), the arithmetic of its single-tensor
// path in fp32:  m += (g - m) (1 - b1);  v = v b2 + ((1 - b2) g) g;  p += -(lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps))
struct AdamBatch {
    GsAdamTensor t[GS_ADAM_MAX_TENSORS];
    long long start[GS_ADAM_MAX_TENSORS + 1];  // first 1024-element chunk of every tensor
    int n;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamBatch b, float w1, float beta2, float w2, float eps, float bc1,
                                                   float bc2_sqrt) {
    // which tensor does this chunk belong to (wave-uniform, at most GS_ADAM_MAX_TENSORS steps)
    const long long chunk = blockIdx.x;
    int k = 0;
    while (k + 1 < b.n && chunk >= b.start[k + 1]) k++;
    const GsAdamTensor T = b.t[k];
    const long long base = (chunk - b.start[k]) * 1024;
    const float neg_step = -(T.lr / bc1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long long i = base + j * 256 + threadIdx.x;
        if (i >= T.n) break;
        const float g = T.grad[i];
        float m = T.exp_avg[i], v = T.exp_avg_sq[i];
        m = m + (g - m) * w1;
        v = v * beta2 + (w2 * g) * g;
        const float den = sqrtf(v) / bc2_sqrt + eps;
        T.param[i] = T.param[i] + neg_step * (m / den);
        T.exp_avg[i] = m;
        T.exp_avg_sq[i] = v;
    }
}

int launch_adam(int n, const GsAdamTensor* tensors, double beta1, double beta2, double eps, int64_t step, hipStream_t s) {
    AdamBatch b;
    b.n = n;
    long long chunks = 0;
    for (int k = 0; k < n; k++) {
        b.t[k] = tensors[k];
        b.start[k] = chunks;
        chunks += (tensors[k].n + 1023) / 1024;
    }
    b.start[n] = chunks;
    if (chunks == 0) return GS_OK;
    // every scalar is formed in double, as torch forms them from Python floats, and rounded to fp32 once
    // (1 - beta2 taken from an fp32 beta2 would be off by 1e-5)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    StageScope st("adam", s);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(256), 0, s, b, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)bc1, (float)sqrt(bc2));
    GS_LAUNCH_CHECK("adam", 0, s);
    return GS_OK;
}
