// blend.h -- the per-(pixel, Gaussian) evaluation shared by the forward and backward render
// kernels (they must take identical skip decisions), per-(tile, Gaussian) staging, and wave64
// reduction helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LOG2E_F 1.4426950408889634f

// ---------------------------------------------------------------------------------------------
// Ownership: a 16x16 tile is four 8x8 QUADRANTS, quadrant q sits at (8 * (q & 1), 8 * (q >> 1)); one
// wave64 per quadrant, lane l <-> pixel (l & 7, l >> 3) in the forward.
// ---------------------------------------------------------------------------------------------

// A staged (quadrant, Gaussian) entry: 48 bytes in LDS, read at a wave-uniform address.
//   a = (x, y, A2, B2)   centre in pixels; conic pre-scaled into the log2 domain:
//                        power2 = A2 dx^2 + C2 dy^2 + B2 dx dy = log2(e) * power
//   b = (C2, opacity, thr, compacted index + 1)
//   c = (r, g, b, position in the tile's list (1-based))
struct Staged {
    float4 a, b, c;
};

// Builds the staged entry of a splat record (r0, r1, r2 as written by the preprocess kernel) for ONE
// 8x8 quadrant with origin (QX0, QY0) and returns whether the Gaussian can reach the quadrant at
// all (the forward kernel runs one wave per quadrant and compacts on this flag).  `thr`: alpha >= 1/255
// <=> opacity * 2^power2 >= 1/255 <=> power2 >= -log2(255 opacity), kept 1e-3 relaxed.
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
            // relative slack: 1e-4, plus the rounding of a form whose terms cancel (long thin Gaussians): ~2^-24 A C / det
            // per operation (gs_math.h: snug_half_widths); when the bound says nothing the entry is kept
            const float detc = A * C - B * B;
            const float rel = 1e-4f + (A * C) / detc * 1.9073486e-6f;
            hit = inside || !(detc > 0.f) || !(rel < 0.5f) || (qmin * (1.0f - rel) - 1e-3f <= two_tau);
        }
    }
    s.a = make_float4(gx, gy, (-0.5f * LOG2E_F) * A, -LOG2E_F * B);
    s.b = make_float4((-0.5f * LOG2E_F) * C, o, thr, 0.f);
    s.c = make_float4(r1.z, r1.w, r2.x, 0.f);
    return hit;
}

// the staged form alone (no footprint test): for a walk of a quadrant's recorded compacted list, whose entries are hits
__device__ __forceinline__ void stage_entry_convert(const float4 r0, const float4 r1, const float4 r2, Staged& s) {
    s.a = make_float4(r0.x, r0.y, (-0.5f * LOG2E_F) * r0.z, -LOG2E_F * r0.w);
    s.b = make_float4((-0.5f * LOG2E_F) * r1.x, r1.y, 0.f, 0.f);
    s.c = make_float4(r1.z, r1.w, r2.x, 0.f);
}

// ---- DPP helpers (cross-lane moves without LDS) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return wave_max(v); }  // (common.h: DPP ladder)
