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
//    in registers over its 64 pixels.  The pipeline is skewed, so it never drains between chunks of
//    64 entries: only the first 63 steps of a quadrant run partly empty.
//    (The pixel-parallel formulation needs a 64-lane reduction of nine values per list entry; DPP
//    adds issue at half rate on gfx950 (tools/dpp_rate.hip), which made that reduction ~2/3 of the
//    kernel.)
//  * Entries never touch LDS: every lane gathers the record of its entry of chunk c+2 while chunk
//    c runs (two register sets), converts it at the round boundary, and at its own switch step
//    (lane == step mod 64) selects it into the working set.  Per-pixel constants (dL/dpixel, Gtot,
//    position, last contributor) sit in LDS and are read at the lane's current pixel index, one step
//    ahead.  A lane that has seen all 64 pixels parks its nine sums in an LDS row; rows go to HBM
//    64 at a time at the round boundary.
//  * FRONT-to-back recurrence.  With g = dL/dpixel, Gtot = out_color . g (out_color already holds
//    T_final * bg) and the running inclusive prefix Pfx_i = sum_{j<=i} (c_j . g) alpha_j T_j,
//        dL/dalpha_i = T_i (c_i . g) - (Gtot - Pfx_i) / (1 - alpha_i)
//    which is the reference's back-to-front recurrence (accum_rec / T division) rewritten so that T is
//    rebuilt by the same multiplications the forward did.
//  * Output: the nine raw sums of (pair, quadrant) at row 4 pair + quadrant, pairs in EMISSION order
//    (Gaussian-major), so the per-Gaussian kernel reads one contiguous span per Gaussian: eight sums in
//    a 32-byte row, the ninth in a dense word array the caller pre-fills with ROW_UNWRITTEN, so it also
//    tells which rows were written.
#include "common.h"
#include "blend.h"

__device__ __forceinline__ float wave_ror1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, true));
}

// the nine per-entry values the inner loop reads
struct Entry {
    float x, y, A2, B2, C2, o, r, g, b;
};

__global__ __launch_bounds__(64) void render_bwd_kernel(const float4* __restrict__ rec,
                                                        const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ order, int W, int H, int gx,
                                                        const uint32_t* __restrict__ qlist,
                                                        const uint32_t* __restrict__ ncon_c,
                                                        const uint32_t* __restrict__ qcount,
                                                        const float* __restrict__ out_color,
                                                        const float* __restrict__ dL_dpix,
                                                        float4* __restrict__ qrows, uint32_t* __restrict__ q8) {
    // per-pixel constants, one array per component (adjacent lanes read adjacent words: no bank
    // conflicts; the 32-byte records this replaces cost 8-way conflicts on every step), each stored twice
    // so a round's reads never wrap:  g0, g1, g2, x, y, lim
    __shared__ float pix[6][128];
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
    if (m == 0) return;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float gtot0;  // Gtot of pixel `lane`
    {
        // (g0, g1, g2, Gtot) and (x, y, lim): pixel p meets compacted entry k at step s = k + p, and the
        // pair counts only while k < (its last contributor) <=> s < lim = ncon + p
        float4 c0 = zero4, c1 = make_float4((float)px, (float)py, __uint_as_float((uint32_t)lane), 0.f);
        if (px < W && py < H) {
            const size_t HW = (size_t)H * W;
            const size_t pid = (size_t)py * W + px;
            const float g0 = dL_dpix[pid], g1 = dL_dpix[HW + pid], g2 = dL_dpix[2 * HW + pid];
            c0 = make_float4(g0, g1, g2, out_color[pid] * g0 + out_color[HW + pid] * g1 + out_color[2 * HW + pid] * g2);
            c1.z = __uint_as_float(ncon_c[pid] + (uint32_t)lane);
        }
        pix[0][lane] = pix[0][64 + lane] = c0.x;
        pix[1][lane] = pix[1][64 + lane] = c0.y;
        pix[2][lane] = pix[2][64 + lane] = c0.z;
        pix[3][lane] = pix[3][64 + lane] = c1.x;
        pix[4][lane] = pix[4][64 + lane] = c1.y;
        pix[5][lane] = pix[5][64 + lane] = c1.z;
        gtot0 = c0.w;
    }

    // record (p0,p1,p2) -> the loop's entry form + the row index 4 pair + quadrant
    auto convert = [&](const float4 p0, const float4 p1, const float4 p2, Entry& e, uint32_t& row) {
        e.x = p0.x; e.y = p0.y;
        e.A2 = (-0.5f * LOG2E_F) * p0.z;
        e.B2 = -LOG2E_F * p0.w;
        e.C2 = (-0.5f * LOG2E_F) * p1.x;
        e.o = p1.y; e.r = p1.z; e.g = p1.w; e.b = p2.x;
        const uint32_t off = __float_as_uint(p2.y), rmin = __float_as_uint(p2.z), rsz = __float_as_uint(p2.w);
        const uint32_t minx = rmin & 0xFFFFu, miny = rmin >> 16, w = rsz & 0xFFFFu;
        row = (off + ((uint32_t)ty - miny) * w + ((uint32_t)tx - minx)) * 4u + (uint32_t)q;
    };
    auto gather = [&](int k, float4& p0, float4& p1, float4& p2) {
        if (k < m) {
            const uint32_t id = qlist[qbase + k];
            p0 = rec[(size_t)id * 3];
            p1 = rec[(size_t)id * 3 + 1];
            p2 = rec[(size_t)id * 3 + 2];
        }
    };

    float4 p0 = zero4, p1 = zero4, p2 = zero4;
    Entry cur = {0, 0, 0, 0, 0, 0, 0, 0, 0}, nxt = cur;
    uint32_t cur_row = 0, nxt_row = 0;  // gradient row (4 pair + quadrant) of the working entry / of `nxt`
    gather(lane, p0, p1, p2);
    convert(p0, p1, p2, nxt, nxt_row);  // chunk 0, taken by lane t at step t
    gather(64 + lane, p0, p1, p2);      // chunk 1 in flight during round 0
    __syncthreads();

    float acc[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) acc[c9] = 0.f;
    // index into pix[c][] of the pixel at this lane: (s - lane) mod 64, + 64 within a round
    uint32_t pidx = (uint32_t)((64 - lane) & 63);
    float pc[6];
#pragma unroll
    for (int c6 = 0; c6 < 6; c6++) pc[c6] = pix[c6][pidx];
    // state of the pixel currently at this lane: transmittance and the part of Gtot not yet composited
    // (pixel p starts at lane (64 - p) mod 64; fetch its Gtot from the lane that loaded it)
    float T = 1.0f, Rem = __shfl(gtot0, (int)pidx, 64);

    // A finished entry's nine RAW sums go straight to its row in HBM, stored by the one lane that owns
    // them (global stores count on vmcnt, so they never hold up the LDS waits of the step loop).  What
    // is constant per Gaussian -- opacity, the conic combination of the two first moments, the -1/2 and
    // 1/log2(e) factors -- is applied once per Gaussian by segment_reduce_kernel.
    auto write_row = [&](size_t row) {
        qrows[row * 2] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        qrows[row * 2 + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        q8[row] = __float_as_uint(acc[8]);  // the ninth sum doubles as the "row written" mark
    };

    // An entry k lives in its lane for the steps k .. k+63 and is stored at step k+64 when the lane
    // takes entry k+64; the loop runs s = 0 .. m+62, so exactly the entries k <= m-2 get stored there and
    // entry m-1 is still in its lane's registers at the end.
    const int total = m + 63;
    for (int s0 = 0; s0 < total; s0 += 64) {
        if (s0 > 0) {
            // round start: every lane took its entry of the previous chunk out of `nxt` during the
            // previous round.  The chunk that was in flight -> `nxt`; the chunk after it -> in flight.
            convert(p0, p1, p2, nxt, nxt_row);
            gather(s0 + 64 + lane, p0, p1, p2);
            pidx -= 64u;
        }
        const int tend = min(64, total - s0);
        auto step = [&](const int t) {
            const uint32_t s = (uint32_t)(s0 + t);
            if (lane == t) {
                if (s0 > 0) write_row(cur_row);  // this lane has seen all 64 pixels with its entry
                // ... and take the next entry
                cur = nxt;
                cur_row = nxt_row;
#pragma unroll
                for (int c9 = 0; c9 < 9; c9++) acc[c9] = 0.f;
            }
            {
                const float3 g = make_float3(pc[0], pc[1], pc[2]);  // dL/dpixel of the pixel at this lane
                const float pxf = pc[3], pyf = pc[4];
                const uint32_t lim = __float_as_uint(pc[5]);
                // next step's pixel constants, fetched now
                pidx += 1u;
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) pc[c6] = pix[c6][pidx];
                const float dx = cur.x - pxf, dy = cur.y - pyf;
                // A2 dx^2 + B2 dx dy + C2 dy^2 in five operations
                const float power2 = __builtin_fmaf(cur.A2 * dx, dx, __builtin_fmaf(cur.B2, dx, cur.C2 * dy) * dy);
                const float G = __builtin_amdgcn_exp2f(power2);
                const float al = fminf(0.99f, cur.o * G);
                // validity as VALU compare + select chains (no scalar mask arithmetic): the pair counts iff
                // power <= 0, the entry lies before the pixel's last contributor (s < lim; a lane that has no
                // entry yet holds zeros, i.e. alpha = 0), and alpha >= 1/255 (which implies the forward's
                // relaxed power2 >= thr pre-test).  Rejected pairs carry alpha = 0.
                const float a1 = (power2 <= 0.0f) ? al : 0.f;
                const float a2 = (s < lim) ? a1 : 0.f;
                const bool valid = a2 >= (1.0f / 255.0f);
                const float alpha = valid ? a2 : 0.f;
                const float Gv = valid ? G : 0.f;
                const float wgt = alpha * T;
                const float cg = cur.r * g.x + cur.g * g.y + cur.b * g.z;
                Rem = __builtin_fmaf(-cg, wgt, Rem);
                const float one_m = 1.f - alpha;
                const float dL_dalpha = T * cg - Rem * __builtin_amdgcn_rcpf(one_m);
                T *= one_m;
                const float Gd = Gv * dL_dalpha;  // G dL/dalpha; times opacity it is G dL/dG (applied when parked)
                const float tdx = Gd * dx, tdy = Gd * dy;
                // sums:  0: t dx   1: t dy   2: t dx^2   3: t dx dy   4: t dy^2   5: G dL/dalpha = dL/dopacity
                //        6..8: w g_c = dL/dcolor
                acc[0] += tdx;
                acc[1] += tdy;
                acc[2] += tdx * dx;
                acc[3] += tdx * dy;
                acc[4] += tdy * dy;
                acc[5] += Gd;
                acc[6] += wgt * g.x;
                acc[7] += wgt * g.y;
                acc[8] += wgt * g.z;
            }
            // the pixel moves on to the next entry = the next lane
            T = wave_ror1(T);
            Rem = wave_ror1(Rem);
        };
        // two steps per trip: the one-step-ahead pixel constants alternate between two register sets
        int t = 0;
        for (; t + 1 < tend; t += 2) {
            step(t);
            step(t + 1);
        }
        if (t < tend) step(t);
    }
    // entry m-1 was never stored
    if (lane == ((m - 1) & 63)) write_row(cur_row);
}

int launch_render_backward(const float* rec, const uint32_t* ranges, const uint32_t* order, int W, int H,
                           const QuadLists& ql, const float* out_color, const float* dL_dpix, float* qrows,
                           uint32_t* q8, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy * 4), dim3(64), 0, s, reinterpret_cast<const float4*>(rec),
                       reinterpret_cast<const uint2*>(ranges), order, W, H, gx, ql.qlist, ql.ncon_c, ql.qcount,
                       out_color, dL_dpix, reinterpret_cast<float4*>(qrows), q8);
    GS_LAUNCH_CHECK("render_backward", 0, s);
    return GS_OK;
}
