// loss.hip -- image-side L1 loss (replaces utils/loss_utils.py:21-22,
// torch.abs(network_output - gt).mean(), and its autograd chain of ~10 elementwise kernels).
// One streaming pass writes d(loss)/d(network_output) = sign(x - y) / n and one partial sum of
// |x - y| per block; a second, single-block pass adds the partials in index order.  No atomics: the
// result is bitwise reproducible.
#include "common.h"

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

static int l1_blocks(int64_t n) {
    const int64_t want = ((n >> 2) + L1_THREADS - 1) / L1_THREADS;
    return (int)(want < 1 ? 1 : (want > L1_MAX_BLOCKS ? L1_MAX_BLOCKS : want));
}

size_t l1_ws_bytes(int64_t n) { return (size_t)l1_blocks(n) * sizeof(float); }

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
