// render_bwd.hip -- backward of the per-tile compositing (replaces upstream
// renderCUDA backward and its 9 float atomicAdds per (pixel, Gaussian) pair).
//
// CDNA4 formulation:
//  * Same ownership as the forward: one wave64 per tile, 4 pixels per lane.
//  * FRONT-to-back re-traversal.  With g = dL/dpixel, Gtot = out_color . g (out_color already holds
//    T_final * bg) and the running inclusive prefix Pfx_i = sum_{j<=i} (c_j . g) alpha_j T_j,
//        dL/dalpha_i = T_i (c_i . g) - (Gtot - Pfx_i) / (1 - alpha_i)
//    which is the reference's back-to-front recurrence (accum_rec / T division) rewritten so that T is
//    rebuilt by the same multiplications the forward did.
//  * The nine per-Gaussian sums are first accumulated over the lane's 4 pixels, then reduced over
//    the wave with DPP adds; NO global atomics.  Each (tile, Gaussian) pair owns one 48-byte row of
//    `entry_grads`, addressed by the pair's index in EMISSION order (Gaussian-major), so the
//    per-Gaussian kernel that follows reads one contiguous segment per Gaussian.
//    (float atomics into 64 different rows per wave-instruction run ~17x below the streaming
//    rate on MI355X; plain row stores do not.)
#include "common.h"
#include "blend.h"

__global__ __launch_bounds__(64) void render_bwd_kernel(const float4* __restrict__ rec,
                                                        const uint32_t* __restrict__ point_list,
                                                        const uint2* __restrict__ ranges, int W, int H, int gx,
                                                        const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ out_color,
                                                        const float* __restrict__ dL_dpix,
                                                        float4* __restrict__ entry_grads) {
    __shared__ float4 srec[64 * 3];
    __shared__ float4 srow[64 * 3];
    __shared__ uint32_t sq[64];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int lx = lane & 15, ly = lane >> 4;
    const int px = tx * TILE + lx, py0 = ty * TILE + ly * 4;
    const float pxf = (float)px;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    const size_t HW = (size_t)H * W;

    float T[4], Pfx[4], Gtot[4], g[4][3];
    uint32_t ncon[4];
    uint32_t my_max = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T[k] = 1.0f;
        Pfx[k] = 0.f;
        const int py = py0 + k;
        if (px < W && py < H) {
            const size_t pid = (size_t)py * W + px;
            g[k][0] = dL_dpix[pid];
            g[k][1] = dL_dpix[HW + pid];
            g[k][2] = dL_dpix[2 * HW + pid];
            Gtot[k] = out_color[pid] * g[k][0] + out_color[HW + pid] * g[k][1] + out_color[2 * HW + pid] * g[k][2];
            ncon[k] = n_contrib[pid];
        } else {
            g[k][0] = g[k][1] = g[k][2] = 0.f;
            Gtot[k] = 0.f;
            ncon[k] = 0;
        }
        my_max = max(my_max, ncon[k]);
    }
    const int nmax = (int)wave_max_u32(my_max);  // entries >= nmax contribute to no pixel of the tile

    float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
    if (lane < nmax) {
        const uint32_t id = point_list[range.x + lane];
        p0 = rec[(size_t)id * 3];
        p1 = rec[(size_t)id * 3 + 1];
        p2 = rec[(size_t)id * 3 + 2];
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < nmax; base += 64) {
        const int cnt = min(64, nmax - base);
        __syncthreads();
        srec[lane * 3] = p0;
        srec[lane * 3 + 1] = p1;
        srec[lane * 3 + 2] = p2;
        srow[lane * 3] = zero4;
        srow[lane * 3 + 1] = zero4;
        srow[lane * 3 + 2] = zero4;
        {
            // row index of this pair in emission order: first pair of the Gaussian + position of this
            // tile inside the Gaussian's rectangle (y outer, x inner)
            const uint32_t off = __float_as_uint(p2.y), rmin = __float_as_uint(p2.z), rsz = __float_as_uint(p2.w);
            const uint32_t minx = rmin & 0xFFFFu, miny = rmin >> 16, w = rsz & 0xFFFFu;
            sq[lane] = off + ((uint32_t)ty - miny) * w + ((uint32_t)tx - minx);
        }
        __syncthreads();
        if (base + 64 + lane < nmax) {
            const uint32_t id = point_list[range.x + base + 64 + lane];
            p0 = rec[(size_t)id * 3];
            p1 = rec[(size_t)id * 3 + 1];
            p2 = rec[(size_t)id * 3 + 2];
        }
        for (int j = 0; j < cnt; j++) {
            const float4 a = srec[j * 3], b = srec[j * 3 + 1];
            const float cz = srec[j * 3 + 2].x;
            const float dx = a.x - pxf;
            const uint32_t entry = (uint32_t)(base + j);
            float acc[9];
#pragma unroll
            for (int c = 0; c < 9; c++) acc[c] = 0.f;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (entry >= ncon[k]) continue;
                const float dy = a.y - (float)(py0 + k);
                float alpha, G;
                if (!splat_alpha(dx, dy, a.z, a.w, b.x, b.y, alpha, G)) continue;
                any = true;
                const float w = alpha * T[k];
                const float cg = b.z * g[k][0] + b.w * g[k][1] + cz * g[k][2];
                Pfx[k] += cg * w;
                const float one_m = 1.f - alpha;
                const float dL_dalpha = T[k] * cg - (Gtot[k] - Pfx[k]) / one_m;
                T[k] *= one_m;
                acc[6] += w * g[k][0];
                acc[7] += w * g[k][1];
                acc[8] += w * g[k][2];
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                acc[0] += dL_dG * dG_ddelx;
                acc[1] += dL_dG * dG_ddely;
                acc[2] += -0.5f * gdx * dx * dL_dG;
                acc[3] += -0.5f * gdx * dy * dL_dG;
                acc[4] += -0.5f * gdy * dy * dL_dG;
                acc[5] += G * dL_dalpha;
            }
            if (__ballot(any) != 0ull) {
#pragma unroll
                for (int c = 0; c < 9; c++) acc[c] = wave_sum_to_lane63(acc[c]);
                if (lane == 63) {
                    srow[j * 3] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    srow[j * 3 + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
                    srow[j * 3 + 2] = make_float4(acc[8], 0.f, 0.f, 0.f);
                }
            }
        }
        __syncthreads();
        if (lane < cnt) {
            const size_t q = sq[lane];
            entry_grads[q * 3] = srow[lane * 3];
            entry_grads[q * 3 + 1] = srow[lane * 3 + 1];
            entry_grads[q * 3 + 2] = srow[lane * 3 + 2];
        }
    }
    // pairs past the last contributor of every pixel: zero rows (every row is written exactly once,
    // so the caller never has to clear `entry_grads`)
    for (int e = nmax + lane; e < n; e += 64) {
        const uint32_t id = point_list[range.x + e];
        const float4 r2 = rec[(size_t)id * 3 + 2];
        const uint32_t off = __float_as_uint(r2.y), rmin = __float_as_uint(r2.z), rsz = __float_as_uint(r2.w);
        const uint32_t minx = rmin & 0xFFFFu, miny = rmin >> 16, w = rsz & 0xFFFFu;
        const size_t q = off + ((uint32_t)ty - miny) * w + ((uint32_t)tx - minx);
        entry_grads[q * 3] = zero4;
        entry_grads[q * 3 + 1] = zero4;
        entry_grads[q * 3 + 2] = zero4;
    }
}

int launch_render_backward(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const float* bg, int W,
                           int H, const uint32_t* n_contrib, const float* out_color, const float* dL_dpix,
                           float* entry_grads, hipStream_t s) {
    (void)bg;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy), dim3(64), 0, s, reinterpret_cast<const float4*>(rec),
                       point_list, reinterpret_cast<const uint2*>(ranges), W, H, gx, n_contrib, out_color, dL_dpix,
                       reinterpret_cast<float4*>(entry_grads));
    GS_LAUNCH_CHECK("render_backward", 0, s);
    return GS_OK;
}
