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
                                                        const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ order, int W, int H, int gx,
                                                        const uint32_t* __restrict__ n_contrib,
                                                        const uint32_t* __restrict__ quad_nmax,
                                                        const float* __restrict__ out_color,
                                                        const float* __restrict__ dL_dpix,
                                                        float4* __restrict__ entry_grads) {
    __shared__ float4 srec[64 * 3];
    __shared__ float4 srow[64 * 3];
    const int tile = (int)order[blockIdx.x];  // heaviest tiles first (tile_order_kernel on the forward's tile_nmax)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int X0 = tx * TILE, Y0 = ty * TILE;
    const int px0 = X0 + (lane & 7), py0 = Y0 + (lane >> 3);
    const float pxf = (float)px0, pyf = (float)py0;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    const size_t HW = (size_t)H * W;

    float T[4], Pfx[4], Gtot[4], g[4][3];
    uint32_t ncon[4];
    uint32_t qmax[4];  // wave-uniform: entries >= qmax[k] contribute to no pixel of quadrant k
    uint32_t nmax_u = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T[k] = 1.0f;
        Pfx[k] = 0.f;
        const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
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
        qmax[k] = quad_nmax[tile * 4 + k];  // wave-uniform (scalar load), written by the forward
        nmax_u = max(nmax_u, qmax[k]);
    }
    const int nmax = (int)nmax_u;  // entries >= nmax contribute to no pixel of the tile

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
        {
            Staged s = stage_entry(p0, p1, p2, X0, Y0);
            // row index of this pair in emission order: first pair of the Gaussian + position of this
            // tile inside the Gaussian's rectangle (y outer, x inner)
            const uint32_t off = __float_as_uint(p2.y), rmin = __float_as_uint(p2.z), rsz = __float_as_uint(p2.w);
            const uint32_t minx = rmin & 0xFFFFu, miny = rmin >> 16, w = rsz & 0xFFFFu;
            s.c.w = __uint_as_float(off + ((uint32_t)ty - miny) * w + ((uint32_t)tx - minx));
            srec[lane * 3] = s.a;
            srec[lane * 3 + 1] = s.b;
            srec[lane * 3 + 2] = s.c;
            srow[lane * 3] = zero4;
            srow[lane * 3 + 1] = zero4;
            srow[lane * 3 + 2] = zero4;
        }
        __syncthreads();
        if (base + 64 + lane < nmax) {
            const uint32_t id = point_list[range.x + base + 64 + lane];
            p0 = rec[(size_t)id * 3];
            p1 = rec[(size_t)id * 3 + 1];
            p2 = rec[(size_t)id * 3 + 2];
        }
        float4 na = srec[0], nb = srec[1], nc = srec[2];  // software-pipelined LDS reads
        for (int j = 0; j < cnt; j++) {
            const uint32_t entry = (uint32_t)(base + j);
            const float4 a = na, b = nb, c = nc;
            {
                const int jn = min(j + 1, 63);
                na = srec[jn * 3];
                nb = srec[jn * 3 + 1];
                nc = srec[jn * 3 + 2];
            }
            uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(b.w));
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (entry >= qmax[k]) m &= ~(1u << k);
            if (m == 0) continue;
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
            // accumulators (constant factors applied after the reduction):
            //  0: sum t*u   u = 2 A2 dx + B2 dy   -> dL/dmean.x = acc0 / log2e
            //  1: sum t*v   v = 2 C2 dy + B2 dx   -> dL/dmean.y = acc1 / log2e
            //  2: sum t*dx^2  3: sum t*dx*dy  4: sum t*dy^2   (t = G dL/dG)  -> dL/dconic = -0.5 * acc
            //  5: sum G dL/dalpha   6..8: sum w g_c
            float acc[9];
#pragma unroll
            for (int q = 0; q < 9; q++) acc[q] = 0.f;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!(m & (1u << k))) continue;  // wave-uniform (scalar branch)
                const float ddx = dx[k & 1], ddy = dy[k >> 1];
                const float power2 = bx[k & 1] * ddy + (ax[k & 1] + cy[k >> 1]);
                const float G = __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, b.y * G);
                const bool valid = (entry < ncon[k]) && (power2 <= 0.0f) && (power2 >= b.z) && (alpha >= (1.0f / 255.0f));
                if (__ballot(valid) == 0ull) continue;  // wave-uniform
                any = any || valid;
                // branch-free from here: invalid lanes carry w = 0, t = 0
                const float w = valid ? alpha * T[k] : 0.f;
                const float cg = c.x * g[k][0] + c.y * g[k][1] + c.z * g[k][2];
                Pfx[k] += cg * w;
                const float one_m = valid ? 1.f - alpha : 1.f;
                const float dL_dalpha = T[k] * cg - (Gtot[k] - Pfx[k]) * __builtin_amdgcn_rcpf(one_m);
                T[k] *= one_m;
                acc[6] += w * g[k][0];
                acc[7] += w * g[k][1];
                acc[8] += w * g[k][2];
                const float Gd = valid ? G * dL_dalpha : 0.f;
                const float t = b.y * Gd;  // G * dL/dG, dL/dG = opacity * dL/dalpha
                const float u = 2.f * a.z * ddx + a.w * ddy;
                const float v = 2.f * b.x * ddy + a.w * ddx;
                acc[0] += t * u;
                acc[1] += t * v;
                const float tdx = t * ddx;
                acc[2] += tdx * ddx;
                acc[3] += tdx * ddy;
                acc[4] += t * ddy * ddy;
                acc[5] += Gd;
            }
            if (__ballot(any) != 0ull) {
#pragma unroll
                for (int q = 0; q < 9; q++) acc[q] = wave_sum_to_lane63(acc[q]);
                if (lane == 63) {
                    const float il2 = 1.0f / LOG2E_F;
                    srow[j * 3] = make_float4(acc[0] * il2, acc[1] * il2, -0.5f * acc[2], -0.5f * acc[3]);
                    srow[j * 3 + 1] = make_float4(-0.5f * acc[4], acc[5], acc[6], acc[7]);
                    srow[j * 3 + 2] = make_float4(acc[8], 0.f, 0.f, 0.f);
                }
            }
        }
        __syncthreads();
        if (lane < cnt) {
            const size_t q = __float_as_uint(srec[lane * 3 + 2].w);
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

int launch_render_backward(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* order,
                           const float* bg, int W, int H, const uint32_t* n_contrib, const uint32_t* quad_nmax,
                           const float* out_color, const float* dL_dpix, float* entry_grads, hipStream_t s) {
    (void)bg;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy), dim3(64), 0, s, reinterpret_cast<const float4*>(rec),
                       point_list, reinterpret_cast<const uint2*>(ranges), order, W, H, gx, n_contrib, quad_nmax, out_color,
                       dL_dpix,
                       reinterpret_cast<float4*>(entry_grads));
    GS_LAUNCH_CHECK("render_backward", 0, s);
    return GS_OK;
}
