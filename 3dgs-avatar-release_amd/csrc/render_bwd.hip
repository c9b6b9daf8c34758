// render_bwd.hip -- backward of the per-tile compositing (replaces upstream
// renderCUDA backward and its 9 float atomicAdds per (pixel, Gaussian) pair).
//
// CDNA4 formulation -- GAUSSIAN-PARALLEL, no cross-lane reduction and no atomics at all:
//  * One wave64 per 8x8 quadrant.  The forward recorded the quadrant's compacted list (qlist) of the
//    Gaussians whose footprint reaches it, up to its last contributor.
//  * LANES ARE LIST ENTRIES, the 64 pixels stream through them.  Entry e lives in lane e mod 64 for the
//    64 steps e .. e+63; at step s that lane works on pixel s - e.  A pixel's running state (T, Pfx)
//    therefore moves one lane per step -- a wave rotate (DPP wave_ror:1) -- and visits the entries
//    front to back, exactly like the forward.  Each lane keeps its Gaussian's nine gradient sums
//    in registers over its 64 pixels and stores them once as a plain 48-byte row.  The pipeline is
//    skewed, so it never drains between chunks of 64 entries: only the first 63 steps of a quadrant
//    run partly empty.
//    (The pixel-parallel formulation needs a 64-lane reduction of nine values per list entry; DPP
//    adds issue at half rate on gfx950 (tools/dpp_rate.hip), which made that reduction ~2/3 of the
//    kernel.)
//  * Per-pixel constants (dL/dpixel, Gtot, position, last contributor) sit in LDS and are read at
//    the lane's current pixel index; entries are staged 64 at a time through a small LDS array, one
//    chunk ahead, the chunk after that already in flight in registers.
//  * FRONT-to-back recurrence.  With g = dL/dpixel, Gtot = out_color . g (out_color already holds
//    T_final * bg) and the running inclusive prefix Pfx_i = sum_{j<=i} (c_j . g) alpha_j T_j,
//        dL/dalpha_i = T_i (c_i . g) - (Gtot - Pfx_i) / (1 - alpha_i)
//    which is the reference's back-to-front recurrence (accum_rec / T division) rewritten so that T is
//    rebuilt by the same multiplications the forward did.
//  * Output: one row per (quadrant, Gaussian) at qrows[qbase + k], k = compacted index; the forward's
//    kmap tells the per-Gaussian kernel which rows belong to which (tile, Gaussian) pair.
#include "common.h"
#include "blend.h"

__device__ __forceinline__ float wave_ror1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, false));
}

__global__ __launch_bounds__(64) void render_bwd_kernel(const float4* __restrict__ rec,
                                                        const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ order, int W, int H, int gx,
                                                        const uint32_t* __restrict__ qlist,
                                                        const uint32_t* __restrict__ ncon_c,
                                                        const uint32_t* __restrict__ qcount,
                                                        const uint32_t* __restrict__ qstaged,
                                                        const float* __restrict__ out_color,
                                                        const float* __restrict__ dL_dpix,
                                                        float4* __restrict__ qrows) {
    __shared__ float4 ring[64 * 3];
    __shared__ float4 pix[64 * 2];
    const int tile = (int)order[blockIdx.x >> 2];  // heaviest tiles first (tile_order_kernel on the forward's counts)
    const int q = blockIdx.x & 3;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int QX0 = tx * TILE + 8 * (q & 1), QY0 = ty * TILE + 8 * (q >> 1);
    const int px = QX0 + (lane & 7), py = QY0 + (lane >> 3);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const uint32_t qbase = 4u * range.x + (uint32_t)q * (uint32_t)n;
    const int m = (int)qcount[tile * 4 + q];
    const int staged = (int)qstaged[tile * 4 + q];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // rows the forward announced in kmap but that lie past the quadrant's last contributor
    for (int k = m + lane; k < staged; k += 64) {
        qrows[(size_t)(qbase + k) * 3] = zero4;
        qrows[(size_t)(qbase + k) * 3 + 1] = zero4;
        qrows[(size_t)(qbase + k) * 3 + 2] = zero4;
    }
    if (m == 0) return;

    {
        float4 c0 = zero4, c1 = make_float4((float)px, (float)py, 0.f, 0.f);
        if (px < W && py < H) {
            const size_t HW = (size_t)H * W;
            const size_t pid = (size_t)py * W + px;
            const float g0 = dL_dpix[pid], g1 = dL_dpix[HW + pid], g2 = dL_dpix[2 * HW + pid];
            c0 = make_float4(g0, g1, g2, out_color[pid] * g0 + out_color[HW + pid] * g1 + out_color[2 * HW + pid] * g2);
            c1.z = __uint_as_float(ncon_c[pid]);
        }
        pix[lane * 2] = c0;
        pix[lane * 2 + 1] = c1;
    }

    // chunk 0 -> LDS now, chunk 1 -> registers (in flight during round 0)
    float4 p0 = zero4, p1 = zero4, p2 = zero4;
    if (lane < m) {
        const uint32_t id = qlist[qbase + lane];
        p0 = rec[(size_t)id * 3];
        p1 = rec[(size_t)id * 3 + 1];
        p2 = rec[(size_t)id * 3 + 2];
    }
    {
        Staged s;
        stage_entry_quad(p0, p1, p2, QX0, QY0, s);
        ring[lane * 3] = s.a;
        ring[lane * 3 + 1] = s.b;
        ring[lane * 3 + 2] = s.c;
    }
    if (64 + lane < m) {
        const uint32_t id = qlist[qbase + 64 + lane];
        p0 = rec[(size_t)id * 3];
        p1 = rec[(size_t)id * 3 + 1];
        p2 = rec[(size_t)id * 3 + 2];
    }
    __syncthreads();

    float4 ca = zero4, cb = zero4, cc = zero4;  // this lane's current entry
    bool has = false;
    uint32_t myk = 0;
    float acc[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) acc[c9] = 0.f;
    float T = 1.0f, Pfx = 0.f;       // state of the pixel currently at this lane
    int pidx = (64 - lane) & 63;     // index of that pixel: (s - lane) mod 64
    const float il2 = 1.0f / LOG2E_F;

    const int total = m + 63;
    for (int s = 0; s < total; s++) {
        const int t = s & 63;
        if (t == 0 && s > 0) {
            // round start: chunk s/64 (prefetched) -> LDS; chunk s/64 + 1 -> registers.  All lanes
            // finished reading the previous chunk from LDS during the previous round.
            __syncthreads();
            Staged sg;
            stage_entry_quad(p0, p1, p2, QX0, QY0, sg);
            ring[lane * 3] = sg.a;
            ring[lane * 3 + 1] = sg.b;
            ring[lane * 3 + 2] = sg.c;
            if (s + 64 + lane < m) {
                const uint32_t id = qlist[qbase + s + 64 + lane];
                p0 = rec[(size_t)id * 3];
                p1 = rec[(size_t)id * 3 + 1];
                p2 = rec[(size_t)id * 3 + 2];
            }
            __syncthreads();
        }
        if (lane == t) {
            // this lane has seen all 64 pixels with its entry: store the row, take the next entry
            if (has) {
                const size_t row = (size_t)(qbase + myk) * 3;
                qrows[row] = make_float4(acc[0] * il2, acc[1] * il2, -0.5f * acc[2], -0.5f * acc[3]);
                qrows[row + 1] = make_float4(-0.5f * acc[4], acc[5], acc[6], acc[7]);
                qrows[row + 2] = make_float4(acc[8], 0.f, 0.f, 0.f);
            }
            has = s < m;
            myk = (uint32_t)s;
            ca = ring[t * 3];
            cb = ring[t * 3 + 1];
            cc = ring[t * 3 + 2];
#pragma unroll
            for (int c9 = 0; c9 < 9; c9++) acc[c9] = 0.f;
        }
        {
            const float4 pc0 = pix[pidx * 2], pc1 = pix[pidx * 2 + 1];
            const float dx = ca.x - pc1.x, dy = ca.y - pc1.y;
            const float power2 = ca.z * dx * dx + (cb.x * dy * dy + ca.w * dx * dy);
            const float G = __builtin_amdgcn_exp2f(power2);
            const float alpha = fminf(0.99f, cb.y * G);
            const bool valid = has && (myk < __float_as_uint(pc1.z)) && (power2 <= 0.0f) && (power2 >= cb.z) &&
                               (alpha >= (1.0f / 255.0f));
            // branch-free: invalid lanes carry wgt = 0, t = 0 and leave the pixel state untouched
            const float wgt = valid ? alpha * T : 0.f;
            const float cg = cc.x * pc0.x + cc.y * pc0.y + cc.z * pc0.z;
            Pfx += cg * wgt;
            const float one_m = valid ? 1.f - alpha : 1.f;
            const float dL_dalpha = T * cg - (pc0.w - Pfx) * __builtin_amdgcn_rcpf(one_m);
            T *= one_m;
            const float Gd = valid ? G * dL_dalpha : 0.f;
            const float tt = cb.y * Gd;  // G * dL/dG, dL/dG = opacity * dL/dalpha
            const float u = 2.f * ca.z * dx + ca.w * dy;
            const float v = 2.f * cb.x * dy + ca.w * dx;
            const float tdx = tt * dx;
            // sums (constant factors applied when the row is stored):
            //  0: t*u -> dL/dmean.x * log2e   1: t*v   2: t dx^2  3: t dx dy  4: t dy^2 (-> -2 dL/dconic)
            //  5: G dL/dalpha = dL/dopacity   6..8: w g_c = dL/dcolor
            acc[0] += tt * u;
            acc[1] += tt * v;
            acc[2] += tdx * dx;
            acc[3] += tdx * dy;
            acc[4] += tt * dy * dy;
            acc[5] += Gd;
            acc[6] += wgt * pc0.x;
            acc[7] += wgt * pc0.y;
            acc[8] += wgt * pc0.z;
        }
        // the pixel moves on to the next entry = the next lane
        T = wave_ror1(T);
        Pfx = wave_ror1(Pfx);
        pidx = (pidx + 1) & 63;
    }
    if (has) {
        const size_t row = (size_t)(qbase + myk) * 3;
        qrows[row] = make_float4(acc[0] * il2, acc[1] * il2, -0.5f * acc[2], -0.5f * acc[3]);
        qrows[row + 1] = make_float4(-0.5f * acc[4], acc[5], acc[6], acc[7]);
        qrows[row + 2] = make_float4(acc[8], 0.f, 0.f, 0.f);
    }
}

int launch_render_backward(const float* rec, const uint32_t* ranges, const uint32_t* order, int W, int H,
                           const QuadLists& ql, const float* out_color, const float* dL_dpix, float* qrows,
                           hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy * 4), dim3(64), 0, s, reinterpret_cast<const float4*>(rec),
                       reinterpret_cast<const uint2*>(ranges), order, W, H, gx, ql.qlist, ql.ncon_c, ql.qcount,
                       ql.qstaged, out_color, dL_dpix, reinterpret_cast<float4*>(qrows));
    GS_LAUNCH_CHECK("render_backward", 0, s);
    return GS_OK;
}
