// render_fwd.hip -- per-tile front-to-back alpha compositing (replaces
// upstream renderCUDA forward).
//
// CDNA4 mapping: ONE wave64 owns one 16x16 tile; lane (lx = lane & 15, ly = lane >> 4) owns the
// four vertically adjacent pixels (16*tx + lx, 16*ty + 4*ly + k), k = 0..3.  The tile's depth-ordered
// list is staged 64 splat records (48 B each, gathered by the 64 lanes in parallel, next batch
// prefetched into registers) at a time through LDS; the inner loop reads one record per
// iteration at a wave-uniform LDS address (broadcast) and amortises it over the lane's 4 pixels.
// No workgroup barrier spans more than this one wave, and the early-out is a wave ballot.
#include "common.h"
#include "blend.h"

__global__ __launch_bounds__(64) void render_fwd_kernel(const float4* __restrict__ rec,
                                                        const uint32_t* __restrict__ point_list,
                                                        const uint2* __restrict__ ranges, const float* __restrict__ bg,
                                                        int W, int H, int gx, float* __restrict__ out_color,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ float4 srec[64 * 3];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int lx = lane & 15, ly = lane >> 4;
    const int px = tx * TILE + lx, py0 = ty * TILE + ly * 4;
    const float pxf = (float)px;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float T[4], C[4][3];
    uint32_t last[4];
    bool done[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T[k] = 1.0f;
        C[k][0] = C[k][1] = C[k][2] = 0.f;
        last[k] = 0;
        done[k] = !(px < W && (py0 + k) < H);
    }

    // prefetch batch 0
    float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
    if (lane < n) {
        const uint32_t id = point_list[range.x + lane];
        p0 = rec[(size_t)id * 3];
        p1 = rec[(size_t)id * 3 + 1];
        p2 = rec[(size_t)id * 3 + 2];
    }
    for (int base = 0; base < n; base += 64) {
        const bool lane_live = !(done[0] && done[1] && done[2] && done[3]);
        if (__ballot(lane_live) == 0ull) break;
        const int cnt = min(64, n - base);
        __syncthreads();
        srec[lane * 3] = p0;
        srec[lane * 3 + 1] = p1;
        srec[lane * 3 + 2] = p2;
        __syncthreads();
        if (base + 64 + lane < n) {
            const uint32_t id = point_list[range.x + base + 64 + lane];
            p0 = rec[(size_t)id * 3];
            p1 = rec[(size_t)id * 3 + 1];
            p2 = rec[(size_t)id * 3 + 2];
        }
        for (int j = 0; j < cnt; j++) {
            const float4 a = srec[j * 3], b = srec[j * 3 + 1];
            const float cz = srec[j * 3 + 2].x;
            const float dx = a.x - pxf;
            const uint32_t contributor = (uint32_t)(base + j + 1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (done[k]) continue;
                const float dy = a.y - (float)(py0 + k);
                float alpha, G;
                if (!splat_alpha(dx, dy, a.z, a.w, b.x, b.y, alpha, G)) continue;
                const float test_T = T[k] * (1.f - alpha);
                if (test_T < 0.0001f) { done[k] = true; continue; }
                const float w = alpha * T[k];
                C[k][0] += b.z * w;
                C[k][1] += b.w * w;
                C[k][2] += cz * w;
                T[k] = test_T;
                last[k] = contributor;
            }
        }
    }
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int py = py0 + k;
        if (px < W && py < H) {
            const size_t pid = (size_t)py * W + px;
            final_T[pid] = T[k];
            n_contrib[pid] = last[k];
            out_color[pid] = C[k][0] + T[k] * bg0;
            out_color[HW + pid] = C[k][1] + T[k] * bg1;
            out_color[2 * HW + pid] = C[k][2] + T[k] * bg2;
        }
    }
}

int launch_render_forward(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const float* bg, int W,
                          int H, float* out_color, float* final_T, uint32_t* n_contrib, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(render_fwd_kernel, dim3(gx * gy), dim3(64), 0, s, reinterpret_cast<const float4*>(rec),
                       point_list, reinterpret_cast<const uint2*>(ranges), bg, W, H, gx, out_color, final_T, n_contrib);
    GS_LAUNCH_CHECK("render_forward", 0, s);
    return GS_OK;
}
