// debug_stats.hip -- counters for reports, outside every timed path (gs_pair_stats).
//
// How many of the (pixel, Gaussian) pairs the render kernels evaluate are USEFUL?  The kernels walk, per 8x8 quadrant, the
// Gaussians whose alpha >= 1/255 footprint reaches the quadrant, up to the quadrant's last contributor: 64 pairs per
// entry.  A pair is useful when it is actually composited: alpha >= 1/255 at that pixel, before the pixel is done.  This
// kernel counts those per frame from the finished forward state (tile lists, records, n_contrib) with a plain per-pixel
// walk -- one workgroup per tile, one thread per pixel, the list staged through LDS 256 entries at a time -- and, beside
// them, the pairs a per-pixel walk up to each pixel's last contributor visits (sum of n_contrib).
#include "common.h"

__global__ __launch_bounds__(256) void pair_stats_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ point_list,
                                                         const uint2* __restrict__ ranges, const uint32_t* __restrict__ n_contrib,
                                                         int W, int H, int gx, unsigned long long* __restrict__ out) {
    __shared__ float4 s0[256];
    __shared__ float2 s1[256];
    __shared__ unsigned long long red[2][4];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int px = (tile % gx) * TILE + (tid & 15), py = (tile / gx) * TILE + (tid >> 4);
    const bool inside = px < W && py < H;
    const uint2 r = ranges[tile];
    const uint32_t n = r.y - r.x;
    const uint32_t last = inside ? n_contrib[(size_t)py * W + px] : 0u;  // 1-based position of the pixel's last contributor
    const float pxf = (float)px, pyf = (float)py;
    float T = 1.0f;
    unsigned long long valid = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        __syncthreads();
        if (base + tid < n) {
            const uint32_t id = point_list[r.x + base + tid];
            const float4 p0 = rec[(size_t)id * 3], p1 = rec[(size_t)id * 3 + 1];
            s0[tid] = p0;                       // x, y, conic A, conic B
            s1[tid] = make_float2(p1.x, p1.y);  // conic C, opacity
        }
        __syncthreads();
        const uint32_t m = min(256u, n - base);
        for (uint32_t k = 0; k < m && base + k < last; k++) {
            const float4 a = s0[k];
            const float2 b = s1[k];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, b.y * __expf(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) break;
            T = test_T;
            valid++;
        }
    }
    valid = wave_sum(valid);
    const unsigned long long walked = wave_sum((unsigned long long)last);
    if ((tid & 63) == 0) { red[0][tid >> 6] = valid; red[1][tid >> 6] = walked; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&out[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));  // (integer adds: the order does not matter)
        atomicAdd(&out[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

int launch_pair_stats(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* n_contrib, int W, int H,
                      unsigned long long* out, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int rc = gs_zero_async(out, 16, "pair_stats.zero", s);
    if (rc != GS_OK) return rc;
    hipLaunchKernelGGL(pair_stats_kernel, dim3(gx * gy), dim3(256), 0, s, reinterpret_cast<const float4*>(rec), point_list,
                       reinterpret_cast<const uint2*>(ranges), n_contrib, W, H, gx, out);
    GS_LAUNCH_CHECK("pair_stats", 0, s);
    return GS_OK;
}
