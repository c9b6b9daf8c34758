// preprocess.hip -- per-Gaussian forward preprocess and mark_visible (A10).
// Replaces upstream preprocessCUDA / checkFrustum.  One thread per Gaussian, streaming.
// Built with -ffp-contract=off: the integer outputs (radii, tile rectangle, tiles_touched, depth
// key) must be bit-identical to the CPU oracle, so every fp32 operation is individually rounded
// and evaluated in the order written here.
#include "common.h"
#include "gs_math.h"

// SH -> RGB of Gaussian i.  The (M,3) block of a Gaussian is 3M contiguous floats read by ONE lane; for
// the full 16-coefficient layout it is 192 bytes at a 16-byte aligned offset, fetched as twelve dwordx4
// loads instead of 48 dword loads (a quarter of the load instructions, each lane touching whole 16-byte
// pieces of its sectors).  Same arithmetic either way.
__device__ __forceinline__ float3 load_and_eval_sh(int deg, int M, const float3 p, const float3 campos,
                                                   const float* __restrict__ shs, int i, uint32_t* clamped) {
    const float* g = shs + (size_t)i * M * 3;
    if (M == 16 && ((uintptr_t)shs & 15u) == 0) {  // wave-uniform
        float l[48];
        const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float4 v = g4[k];
            l[4 * k] = v.x; l[4 * k + 1] = v.y; l[4 * k + 2] = v.z; l[4 * k + 3] = v.w;
        }
        return sh_to_rgb(deg, p, campos, l, clamped);
    }
    return sh_to_rgb(deg, p, campos, g, clamped);
}

__global__ __launch_bounds__(256) void preprocess_kernel(
    int P, int deg, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
    float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
    int W, int H, float tanfovx, float tanfovy, float focal_x, float focal_y, int gx, int gy, int tile_rect,
    float* __restrict__ rec, float* __restrict__ depths, uint32_t* __restrict__ tiles, uint32_t* __restrict__ clamped,
    uint32_t* __restrict__ sort_keys, uint32_t* __restrict__ sort_vals, int32_t* __restrict__ radii,
    uint32_t* __restrict__ wave_tiles, uint32_t* __restrict__ wave_kmin, uint32_t* __restrict__ wave_kmax, ZeroJob zero,
    ZeroJob zero2) {
    zero_job(zero);   // the depth sort's digit totals (saves a fill launch)
    zero_job(zero2);  // the tile binning's per-tile pair totals (its counting pass adds into them)
    // threads past the end (last workgroup only) redo Gaussian P - 1 and store nothing: every lane of every wave reaches
    // the wave-level sum of tiles touched at the end
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = gi < P;
    const int i = live ? gi : P - 1;

    float V[16], PV[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { V[k] = viewmatrix[k]; PV[k] = projmatrix[k]; }

    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    float r2x = 0.f;
    uint32_t off_bits = 0, rmin_bits = 0, rsize_bits = 0, cl = 0, tt = 0;
    float depth = 0.f;
    int radius = 0;

    // The kernel is one generation of waves: its time is the number of DEPENDENT memory round trips a thread makes.  What a
    // thread needs is therefore requested up front (position, shape, opacity: 44 bytes, also for the few Gaussians the near
    // plane culls), the 192 bytes of SH coefficients right behind the near-plane test -- in flight while the covariance is
    // projected -- instead of one trip each behind the tests that decide whether they are needed (four trips).
    const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    float c6in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float3 sc_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6in[k] = cov3D_precomp[6 * i + k];
    } else {
        sc_in = make_float3(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]);
        q_in = reinterpret_cast<const float4*>(rotations)[i];
    }
    const float opacity_in = opacities[i];
    const float3 pv = xform4x3(p, V);
    bool ok = pv.z > 0.2f;  // near cull (p_view.z <= 0.2 is culled)
    const bool sh_early = ok && !colors_precomp && M == 16 && ((uintptr_t)shs & 15u) == 0;
    float shl[48];
    if (sh_early) {
        const float4* g4 = reinterpret_cast<const float4*>(shs + (size_t)i * 48);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float4 v = g4[k];
            shl[4 * k] = v.x; shl[4 * k + 1] = v.y; shl[4 * k + 2] = v.z; shl[4 * k + 3] = v.w;
        }
    }
    if (ok) {
        const float4 ph = xform4x4(p, PV);
        const float pw = 1.0f / (ph.w + 0.0000001f);
        const float ppx = ph.x * pw, ppy = ph.y * pw;
        float c6[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = c6in[k];
        } else {
            cov3d_from_scale_rot(sc_in, scale_modifier, q_in, c6);
        }
        float cov[3], Mx[2][3], tcl[3], tt2[2];
        cov2d(p, focal_x, focal_y, tanfovx, tanfovy, c6, V, cov, Mx, tcl, tt2);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float cA = cov[2] * det_inv, cB = -cov[1] * det_inv, cC = cov[0] * det_inv;
            const float mid = 0.5f * (cov[0] + cov[2]);
            const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float l1 = mid + sq, l2 = mid - sq;
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
            // ndc2Pix in double, as upstream: ((v + 1.0) * S - 1.0) * 0.5
            const float px = (float)((((double)ppx + 1.0) * (double)W - 1.0) * 0.5);
            const float py = (float)((((double)ppy + 1.0) * (double)H - 1.0) * 0.5);
            const int r = (int)my_radius;
            int minx = (int)((px - r) / TILE), miny = (int)((py - r) / TILE);
            int maxx = (int)((px + r + TILE - 1) / TILE), maxy = (int)((py + r + TILE - 1) / TILE);
            minx = min(gx, max(0, minx));
            miny = min(gy, max(0, miny));
            maxx = min(gx, max(0, maxx));
            maxy = min(gy, max(0, maxy));
            int w = maxx - minx, h = maxy - miny;
            if (w * h != 0) {
                if (tile_rect) {
                    // bin into the bounding box of the alpha >= 1/255 region only (GsFwdArgs.tile_rect = 1); the
                    // Gaussian stays "visible" (radii, colour, depth) exactly as with the square
                    float hx, hy;
                    if (snug_half_widths(opacity_in, cov[0], cov[2], det, &hx, &hy)) {
                        minx = max(minx, (int)((px - hx) / TILE));
                        miny = max(miny, (int)((py - hy) / TILE));
                        maxx = min(maxx, (int)((px + hx) / TILE) + 1);
                        maxy = min(maxy, (int)((py + hy) / TILE) + 1);
                        if (maxx < minx) maxx = minx;
                        if (maxy < miny) maxy = miny;
                    } else {
                        maxx = minx;
                        maxy = miny;
                    }
                    w = maxx - minx;
                    h = maxy - miny;
                }
                float3 col;
                if (colors_precomp) {
                    col = make_float3(colors_precomp[3 * i], colors_precomp[3 * i + 1], colors_precomp[3 * i + 2]);
                } else if (sh_early) {
                    col = sh_to_rgb(deg, p, make_float3(campos[0], campos[1], campos[2]), shl, &cl);
                } else {
                    col = load_and_eval_sh(deg, M, p, make_float3(campos[0], campos[1], campos[2]), shs, i, &cl);
                }
                depth = pv.z;
                radius = r;
                tt = (uint32_t)(w * h);
                r0 = make_float4(px, py, cA, cB);
                r1 = make_float4(cC, opacity_in, col.x, col.y);
                r2x = col.z;
                rmin_bits = (uint32_t)minx | ((uint32_t)miny << 16);
                rsize_bits = (uint32_t)w | ((uint32_t)h << 16);
            }
        }
    }
    if (live) {
        float4* R = reinterpret_cast<float4*>(rec) + (size_t)i * 3;
        R[0] = r0;
        R[1] = r1;
        R[2] = make_float4(r2x, __uint_as_float(off_bits), __uint_as_float(rmin_bits), __uint_as_float(rsize_bits));
        depths[i] = depth;
        tiles[i] = tt;
        clamped[i] = cl;
        radii[i] = radius;
        // depth-sort key: positive float bits are monotone as unsigned; culled Gaussians go last
        sort_keys[i] = tt ? __float_as_uint(depth) : 0xFFFFFFFFu;
        sort_vals[i] = (uint32_t)i;
    }
    // tiles touched by this wave's 64 Gaussians: the first level of the prefix sum that numbers the (tile, Gaussian)
    // pairs (first_pair_kernel), saving its reduction launch
    // ... and the range of this wave's depth keys (Gaussians that touch a tile only): the bucket depth sort maps keys to
    // buckets linearly between the frame's smallest and largest key
    uint32_t wsum = live ? tt : 0u;
    const bool keyed = live && tt != 0u;
    uint32_t kmin = keyed ? __float_as_uint(depth) : 0xFFFFFFFFu, kmax = keyed ? __float_as_uint(depth) : 0u;
    wsum = wave_sum(wsum);
    kmin = wave_min(kmin);
    kmax = wave_max(kmax);
    if ((threadIdx.x & 63) == 0) {
        wave_tiles[gi >> 6] = wsum;
        wave_kmin[gi >> 6] = kmin;
        wave_kmax[gi >> 6] = kmax;
    }
}

int launch_preprocess(const GsFwdArgs& a, float* rec, float* depths, uint32_t* tiles, uint32_t* clamped,
                      uint32_t* sort_keys, uint32_t* sort_vals, int32_t* radii, uint32_t* wave_tiles, uint32_t* wave_kmin,
                      uint32_t* wave_kmax, ZeroJob zero, ZeroJob zero2, hipStream_t s) {
    const int gx = (a.W + TILE - 1) / TILE, gy = (a.H + TILE - 1) / TILE;
    const float focal_y = a.H / (2.0f * a.tanfovy), focal_x = a.W / (2.0f * a.tanfovx);
    const int blocks = (a.P + 255) / 256;
    // (staging the SH rows through LDS for coalescing measured slower here: the kernel is latency-bound)
    hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(256), 0, s, a.P, a.sh_degree, a.M, a.means3D, a.scales,
                       a.scale_modifier, a.rotations, a.opacities, a.shs, a.colors_precomp, a.cov3D_precomp,
                       a.viewmatrix, a.projmatrix, a.campos, a.W, a.H, a.tanfovx, a.tanfovy, focal_x, focal_y, gx, gy,
                       a.tile_rect,
                       rec, depths, tiles, clamped, sort_keys, sort_vals, radii, wave_tiles, wave_kmin, wave_kmax, zero, zero2);
    GS_LAUNCH_CHECK("preprocess", a.debug, s);
    return GS_OK;
}

// Shared-geometry second render: same Gaussians, camera and covariances as a
// previous call, new colours.  Copies the per-Gaussian state the later stages read (splat record,
// tiles touched) and replaces the colour (and the SH clamp mask).
__global__ __launch_bounds__(256) void recolor_kernel(int P, int deg, int M, const float* __restrict__ means3D,
                                                      const float* __restrict__ shs,
                                                      const float* __restrict__ colors_precomp,
                                                      const float* __restrict__ campos,
                                                      const float4* __restrict__ rec_src,
                                                      const uint32_t* __restrict__ tiles_src,
                                                      float4* __restrict__ rec_dst, uint32_t* __restrict__ tiles_dst,
                                                      uint32_t* __restrict__ clamped_dst,
                                                      unsigned long long* __restrict__ not_ones, const CopyJob c0,
                                                      const CopyJob c1, const ZeroJob z0, const int recolor_blocks, const SecondOnes ones,
                                                      const int pixel_blocks) {
    if ((int)blockIdx.x >= recolor_blocks) {  // (workgroup-uniform) the side job: the all-ones image, speculatively
        second_ones_body(ones, (int)blockIdx.x - recolor_blocks, pixel_blocks);
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the new image state's own copy of the tile ranges and the launch order (its backward reads them): two copy launches less
    for (int k = i; k < c0.words; k += recolor_blocks * (int)blockDim.x) c0.dst[k] = c0.src[k];
    for (int k = i; k < c1.words; k += recolor_blocks * (int)blockDim.x) c1.dst[k] = c1.src[k];
    for (int k = i; k < z0.words; k += recolor_blocks * (int)blockDim.x) z0.ptr[k] = 0u;  // (the chunk-parallel forward's hand-off words)
    if (i >= P) return;
    float4 r0 = rec_src[(size_t)i * 3], r1 = rec_src[(size_t)i * 3 + 1], r2 = rec_src[(size_t)i * 3 + 2];
    const uint32_t tt = tiles_src[i];
    uint32_t cl = 0;
    bool bad = false;
    if (tt) {
        float3 col;
        if (colors_precomp) {
            col = make_float3(colors_precomp[3 * i], colors_precomp[3 * i + 1], colors_precomp[3 * i + 2]);
        } else {
            const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
            col = load_and_eval_sh(deg, M, p, make_float3(campos[0], campos[1], campos[2]), shs, i, &cl);
        }
        r1.z = col.x;
        r1.w = col.y;
        r2.x = col.z;
        bad = !(col.x == 1.0f && col.y == 1.0f && col.z == 1.0f);
    }
    // `not_ones` (zero since the first render of this geometry): set unless every Gaussian that touches a tile is given
    // the colour (1, 1, 1) -- the reference's opacity pass -- in which case the image is 1 - T (render_fwd.hip)
    if (not_ones) {
        const unsigned long long b = __ballot(bad);  // (one atomic per wave at most, by its first such lane)
        if (b != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)b) - 1) atomicOr(not_ones, 1ull);
    }
    rec_dst[(size_t)i * 3] = r0;
    rec_dst[(size_t)i * 3 + 1] = r1;
    rec_dst[(size_t)i * 3 + 2] = r2;
    tiles_dst[i] = tt;
    clamped_dst[i] = cl;
}

int launch_recolor(const GsFwdArgs& a, const float* rec_src, const uint32_t* tiles_src, float* rec_dst,
                   uint32_t* tiles_dst, uint32_t* clamped_dst, unsigned long long* not_ones, CopyJob c0, CopyJob c1,
                   ZeroJob z0, const SecondOnes* ones, hipStream_t s) {
    const int recolor_blocks = (a.P + 255) / 256;
    const int pixel_blocks = ones ? (int)(((size_t)ones->W * ones->H + 255) / 256) : 0;
    const int side = ones ? pixel_blocks + ones->ntiles : 0;
    const SecondOnes none{};
    hipLaunchKernelGGL(recolor_kernel, dim3((unsigned)(recolor_blocks + side)), dim3(256), 0, s, a.P, a.sh_degree, a.M, a.means3D,
                       a.shs, a.colors_precomp, a.campos, reinterpret_cast<const float4*>(rec_src), tiles_src,
                       reinterpret_cast<float4*>(rec_dst), tiles_dst, clamped_dst, not_ones, c0, c1, z0, recolor_blocks,
                       ones ? *ones : none, pixel_blocks);
    GS_LAUNCH_CHECK("recolor", a.debug, s);
    return GS_OK;
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ viewmatrix,
                                                           uint8_t* __restrict__ present) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float V[16];
#pragma unroll
    for (int k = 0; k < 16; k++) V[k] = viewmatrix[k];
    const float3 pv = xform4x3(make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]), V);
    present[i] = pv.z > 0.2f ? 1 : 0;
}

int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    GS_LAUNCH_CHECK("mark_visible", 0, s);
    return GS_OK;
}
