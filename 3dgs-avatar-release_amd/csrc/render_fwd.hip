// render_fwd.hip -- per-tile front-to-back alpha compositing (replaces
// upstream renderCUDA forward).
//
// CDNA4 mapping: ONE wave64 owns one 16x16 tile, as four 8x8 quadrants; lane l owns pixel
// (l & 7, l >> 3) of each quadrant, so one wave-uniform read of a staged entry serves 256 pixels.
// The tile's depth-ordered list is staged 64 entries at a time: each lane gathers one 48-byte splat
// record (the next batch is prefetched into registers), converts it (log2-domain conic, alpha
// threshold, QUADRANT MASK = which 8x8 quadrants the Gaussian's alpha >= 1/255 footprint can reach)
// and parks it in LDS.  The inner loop then runs only the quadrants in the mask that still have a
// live pixel -- a wave-uniform (scalar) branch, so skipped quadrants cost nothing.
// No workgroup barrier spans more than this one wave; the early-out is a wave ballot.
#include "common.h"
#include "blend.h"

__global__ __launch_bounds__(64) void render_fwd_kernel(const float4* __restrict__ rec,
                                                        const uint32_t* __restrict__ point_list,
                                                        const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ order,
                                                        const float* __restrict__ bg, int W, int H, int gx,
                                                        float* __restrict__ out_color, float* __restrict__ final_T,
                                                        uint32_t* __restrict__ n_contrib,
                                                        uint32_t* __restrict__ tile_nmax) {
    __shared__ float4 srec[64 * 3];
    const int tile = (int)order[blockIdx.x];  // heaviest tiles first (tile_order_kernel)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int X0 = tx * TILE, Y0 = ty * TILE;
    const int px0 = X0 + (lane & 7), py0 = Y0 + (lane >> 3);
    const float pxf = (float)px0, pyf = (float)py0;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float T[4], C[4][3];
    uint32_t last[4];
    bool done[4];
    uint32_t qlive = 0;  // wave-uniform: quadrants that still have a pixel accepting contributions
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T[k] = 1.0f;
        C[k][0] = C[k][1] = C[k][2] = 0.f;
        last[k] = 0;
        done[k] = !((px0 + 8 * (k & 1)) < W && (py0 + 8 * (k >> 1)) < H);
        if (__ballot(!done[k]) != 0ull) qlive |= 1u << k;
    }

    float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
    if (lane < n) {
        const uint32_t id = point_list[range.x + lane];
        p0 = rec[(size_t)id * 3];
        p1 = rec[(size_t)id * 3 + 1];
        p2 = rec[(size_t)id * 3 + 2];
    }
    for (int base = 0; base < n && qlive != 0; base += 64) {
        const int cnt = min(64, n - base);
        __syncthreads();
        {
            const Staged s = stage_entry(p0, p1, p2, X0, Y0);
            srec[lane * 3] = s.a;
            srec[lane * 3 + 1] = s.b;
            srec[lane * 3 + 2] = s.c;
        }
        __syncthreads();
        if (base + 64 + lane < n) {
            const uint32_t id = point_list[range.x + base + 64 + lane];
            p0 = rec[(size_t)id * 3];
            p1 = rec[(size_t)id * 3 + 1];
            p2 = rec[(size_t)id * 3 + 2];
        }
        float4 na = srec[0], nb = srec[1], nc = srec[2];  // software-pipelined LDS reads
        for (int j = 0; j < cnt; j++) {
            const float4 a = na, b = nb, c = nc;
            {
                const int jn = min(j + 1, 63);
                na = srec[jn * 3];
                nb = srec[jn * 3 + 1];
                nc = srec[jn * 3 + 2];
            }
            const uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(b.w)) & qlive;
            if (m == 0) continue;
            const uint32_t contributor = (uint32_t)(base + j + 1);
            float dx[2], dy[2], ax[2], cy[2], bx[2];
            dx[0] = a.x - pxf;
            dx[1] = dx[0] - 8.f;
            dy[0] = a.y - pyf;
            dy[1] = dy[0] - 8.f;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                ax[h] = a.z * dx[h] * dx[h];
                cy[h] = b.x * dy[h] * dy[h];
                bx[h] = a.w * dx[h];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!(m & (1u << k))) continue;  // wave-uniform (scalar branch)
                // branch-free per-pixel body: everything is computed, then selected
                const float power2 = bx[k & 1] * dy[k >> 1] + (ax[k & 1] + cy[k >> 1]);
                const float G = __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, b.y * G);
                const bool valid = !done[k] && (power2 <= 0.0f) && (power2 >= b.z) && (alpha >= (1.0f / 255.0f));
                const float test_T = T[k] * (1.f - alpha);
                const bool kill = valid && (test_T < 0.0001f);
                const bool blend = valid && !kill;
                const float w = blend ? alpha * T[k] : 0.f;
                C[k][0] += c.x * w;
                C[k][1] += c.y * w;
                C[k][2] += c.z * w;
                T[k] = blend ? test_T : T[k];
                last[k] = blend ? contributor : last[k];
                done[k] = done[k] || kill;
                if (__ballot(kill) != 0ull) {
                    if (__ballot(!done[k]) == 0ull) qlive &= ~(1u << k);
                }
            }
            if (qlive == 0) break;
        }
    }
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t HW = (size_t)H * W;
    {
        const uint32_t nm = wave_max_u32(max(max(last[0], last[1]), max(last[2], last[3])));
        if (lane == 0) tile_nmax[tile] = nm;  // the backward's work estimate for this tile
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
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

int launch_render_forward(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* order,
                          const float* bg, int W, int H, float* out_color, float* final_T, uint32_t* n_contrib,
                          uint32_t* tile_nmax, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(render_fwd_kernel, dim3(gx * gy), dim3(64), 0, s, reinterpret_cast<const float4*>(rec),
                       point_list, reinterpret_cast<const uint2*>(ranges), order, bg, W, H, gx, out_color, final_T,
                       n_contrib, tile_nmax);
    GS_LAUNCH_CHECK("render_forward", 0, s);
    return GS_OK;
}
