// blend.h -- the per-(pixel, Gaussian) evaluation shared by the forward and backward render
// kernels (they must take identical skip decisions), per-(tile, Gaussian) staging, and wave64
// reduction helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LOG2E_F 1.4426950408889634f

// ---------------------------------------------------------------------------------------------
// Tile ownership: one wave64 per 16x16 tile; the tile is four 8x8 QUADRANTS, lane l owns pixel
// (l & 7, l >> 3) of every quadrant, quadrant q sits at (8 * (q & 1), 8 * (q >> 1)).
// ---------------------------------------------------------------------------------------------

// A staged (tile, Gaussian) entry: 48 bytes in LDS, read at a wave-uniform address.
//   a = (x, y, A2, B2)   centre in pixels; conic pre-scaled into the log2 domain:
//                        power2 = A2 dx^2 + C2 dy^2 + B2 dx dy = log2(e) * power
//   b = (C2, opacity, thr, quadrant mask bits)
//   c = (r, g, b, row index of the pair in emission order (backward only))
struct Staged {
    float4 a, b, c;
};

// Builds the staged entry from a splat record (r0, r1, r2 as written by the preprocess kernel).
// `thr`: alpha >= 1/255  <=>  opacity * 2^power2 >= 1/255  <=>  power2 >= -log2(255 * opacity); kept
// slightly relaxed (-1e-3) and used only as a cheap early reject -- the decision itself is still
// taken on alpha.  Quadrant mask: bit q set iff the axis-aligned bounding box of the ellipse
// {power2 >= thr} (half extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy) with cov = conic^-1 and
// 2 tau = -2 thr / log2 e) reaches a pixel centre of quadrant q.  Conservative by a 1e-4 relative +
// 0.01 px margin, so culled quadrants hold only pairs the per-pixel test would reject anyway.
__device__ __forceinline__ Staged stage_entry(const float4 r0, const float4 r1, const float4 r2, int X0, int Y0) {
    Staged s;
    const float gx = r0.x, gy = r0.y, A = r0.z, B = r0.w, C = r1.x, o = r1.y;
    const float thr = -__log2f(255.f * o) - 1e-3f;
    uint32_t mask = 0;
    if (thr <= 0.f) {
        const float det = A * C - B * B;
        if (det > 0.f) {
            const float two_tau = (-2.f / LOG2E_F) * thr;
            const float k = two_tau / det;
            const float ex = sqrtf(k * C) * 1.0001f + 0.01f;
            const float ey = sqrtf(k * A) * 1.0001f + 0.01f;
            const float x0 = gx - ex - (float)X0, x1 = gx + ex - (float)X0;
            const float y0 = gy - ey - (float)Y0, y1 = gy + ey - (float)Y0;
            const bool cx0 = (x1 >= 0.f) && (x0 <= 7.f), cx1 = (x1 >= 8.f) && (x0 <= 15.f);
            const bool cy0 = (y1 >= 0.f) && (y0 <= 7.f), cy1 = (y1 >= 8.f) && (y0 <= 15.f);
            mask = (cx0 && cy0 ? 1u : 0u) | (cx1 && cy0 ? 2u : 0u) | (cx0 && cy1 ? 4u : 0u) | (cx1 && cy1 ? 8u : 0u);
        } else {
            mask = 0xFu;
        }
    }
    s.a = make_float4(gx, gy, (-0.5f * LOG2E_F) * A, -LOG2E_F * B);
    s.b = make_float4((-0.5f * LOG2E_F) * C, o, thr, __uint_as_float(mask));
    s.c = make_float4(r1.z, r1.w, r2.x, 0.f);
    return s;
}

// The same for ONE 8x8 quadrant with origin (QX0, QY0): returns whether the Gaussian can reach the
// quadrant at all (the forward kernel runs one wave per quadrant and compacts on this flag).
// EXACT ellipse-vs-rectangle test: the minimum over the quadrant's pixel-centre rectangle of the
// quadratic form q(d) = A dx^2 + 2 B dx dy + C dy^2 (convex, so either the centre lies inside, or the
// minimum sits on one of the four edges, where it is a clamped 1-D parabola) against
// 2 tau = 2 ln(255 opacity) (+ slack).  alpha >= 1/255 <=> q <= 2 ln(255 opacity); the test keeps a
// 1e-3 (log2) slack plus a relative 1e-4, so every pair it drops would fail the per-pixel alpha test.
__device__ __forceinline__ float edge_min(float a, float b2, float c, float xf, float y0, float y1) {
    // min over y in [y0, y1] of a xf^2 + b2 xf y + c y^2   (b2 = 2 B)
    const float ys = fminf(fmaxf(-0.5f * b2 * xf / c, y0), y1);
    return a * xf * xf + (b2 * xf + c * ys) * ys;
}
__device__ __forceinline__ bool stage_entry_quad(const float4 r0, const float4 r1, const float4 r2, int QX0, int QY0,
                                                 Staged& s) {
    const float gx = r0.x, gy = r0.y, A = r0.z, B = r0.w, C = r1.x, o = r1.y;
    const float thr = -__log2f(255.f * o) - 1e-3f;
    bool hit = false;
    if (thr <= 0.f) {
        hit = true;
        if (A > 0.f && C > 0.f) {
            const float two_tau = (-2.f / LOG2E_F) * thr;
            // rectangle of pixel centres relative to the Gaussian centre
            const float x0 = (float)QX0 - gx, x1 = x0 + 7.f;
            const float y0 = (float)QY0 - gy, y1 = y0 + 7.f;
            const bool inside = (x0 <= 0.f) && (x1 >= 0.f) && (y0 <= 0.f) && (y1 >= 0.f);
            const float b2 = 2.f * B;
            const float qmin = fminf(fminf(edge_min(A, b2, C, x0, y0, y1), edge_min(A, b2, C, x1, y0, y1)),
                                     fminf(edge_min(C, b2, A, y0, x0, x1), edge_min(C, b2, A, y1, x0, x1)));
            hit = inside || (qmin * 0.9999f - 1e-3f <= two_tau);
        }
    }
    s.a = make_float4(gx, gy, (-0.5f * LOG2E_F) * A, -LOG2E_F * B);
    s.b = make_float4((-0.5f * LOG2E_F) * C, o, thr, 0.f);
    s.c = make_float4(r1.z, r1.w, r2.x, 0.f);
    return hit;
}

// alpha of one Gaussian at one pixel in the log2 domain.  Rejected if
// power > 0 or alpha < 1/255.  G = exp(power) is returned for the backward pass.
__device__ __forceinline__ bool splat_alpha2(float power2, float o, float thr, float& alpha, float& G) {
    if (power2 > 0.0f || power2 < thr) return false;
    G = __builtin_amdgcn_exp2f(power2);
    alpha = fminf(0.99f, o * G);
    return alpha >= (1.0f / 255.0f);
}

// ---- wave64 sum via DPP (no LDS): after the call lane 63 holds the sum over all 64 lanes ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_get<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xF>(v);  // row_mirror  -> every lane holds its row-of-16 sum
    v += dpp_get<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_get<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3 -> lane 63 = total
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
