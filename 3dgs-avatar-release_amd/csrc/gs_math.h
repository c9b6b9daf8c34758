// gs_math.h -- per-Gaussian device math shared by the forward preprocess and the per-Gaussian
// backward kernel.  Conventions: row-vector matrices with flat index
// row*4+col (scene/cameras.py:35-39 stores W2C transposed), quaternion (w,x,y,z) as
// utils/general_utils.py:87-108, covariance six-vector [xx,xy,xz,yy,yz,zz] (:73-85),
// Sigma = (R S)(R S)^T (scene/gaussian_model.py:28-32).
// Translation units that need bit-exact integer outputs include this with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__device__ __constant__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                                       -0.4570457994644658f, 0.3731763325901154f,
                                                       -0.4570457994644658f, 1.445305721320277f,
                                                       -0.5900435899266435f};

__device__ __forceinline__ float3 xform4x3(const float3 p, const float* m) {
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float* m) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

__device__ __forceinline__ void quat_to_R(const float4 q, float R[3][3]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;  // (w,x,y,z) stored in that order
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, float mod, const float4 q, float* c6) {
    float R[3][3];
    quat_to_R(q, R);
    const float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
    float L[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * s[j];
    c6[0] = L[0][0] * L[0][0] + L[0][1] * L[0][1] + L[0][2] * L[0][2];
    c6[1] = L[0][0] * L[1][0] + L[0][1] * L[1][1] + L[0][2] * L[1][2];
    c6[2] = L[0][0] * L[2][0] + L[0][1] * L[2][1] + L[0][2] * L[2][2];
    c6[3] = L[1][0] * L[1][0] + L[1][1] * L[1][1] + L[1][2] * L[1][2];
    c6[4] = L[1][0] * L[2][0] + L[1][1] * L[2][1] + L[1][2] * L[2][2];
    c6[5] = L[2][0] * L[2][0] + L[2][1] * L[2][1] + L[2][2] * L[2][2];
}

// EWA projection: cov2D = (J Rv) Sigma (J Rv)^T + 0.3 I.  Outputs cov = (a, b, c), M = J Rv, the
// clamped view-space point t and the unclamped ratios (t.x/t.z, t.y/t.z) for the backward mask.
__device__ __forceinline__ void cov2d(const float3 mean, float fx, float fy, float tanfovx, float tanfovy,
                                      const float* c6, const float* V, float* cov, float M[2][3], float* t_out,
                                      float* ratios) {
    float3 t = xform4x3(mean, V);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        M[0][k] = J00 * V[4 * k + 0] + J02 * V[4 * k + 2];
        M[1][k] = J11 * V[4 * k + 1] + J12 * V[4 * k + 2];
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float MS[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) MS[i][j] = M[i][0] * S[0][j] + M[i][1] * S[1][j] + M[i][2] * S[2][j];
    cov[0] = MS[0][0] * M[0][0] + MS[0][1] * M[0][1] + MS[0][2] * M[0][2] + 0.3f;
    cov[1] = MS[0][0] * M[1][0] + MS[0][1] * M[1][1] + MS[0][2] * M[1][2];
    cov[2] = MS[1][0] * M[1][0] + MS[1][1] * M[1][1] + MS[1][2] * M[1][2] + 0.3f;
    t_out[0] = t.x; t_out[1] = t.y; t_out[2] = t.z;
    ratios[0] = txtz; ratios[1] = tytz;
}

// An upper bound of ln(x) for x >= 1 from exactly rounded IEEE operations only (no libm: the tile rectangle it
// feeds must be bit-identical on the CPU oracle): x = m 2^e with m in [1,2); ln m <= m - 1 (tangent at 1) and
// ln m <= ln 1.5 + (m - 1.5) / 1.5 (tangent at 1.5); at most 0.095 above ln x.
__device__ __forceinline__ float ln_upper_bound(float x) {
    const uint32_t u = __float_as_uint(x);
    const int e = (int)(u >> 23) - 127;
    const float m = __uint_as_float((u & 0x007FFFFFu) | 0x3F800000u);
    const float lm = (m < 1.5f) ? (m - 1.0f) : (0.405465126f + (m - 1.5f) * 0.666666687f);
    return (float)e * 0.693147182f + lm;
}

// Half-widths of the axis-aligned bounding box of { alpha >= 1/255 } = { d^T Sigma^-1 d <= 2 ln(255 opacity) } for
// the 2-D covariance (a, b, c) with determinant det: sqrt(thr a), sqrt(thr c).  Returns false when alpha stays below
// 1/255 everywhere.
// The per-pixel test is evaluated with the ROUNDED fp32 conic (c/det, -b/det, a/det) and a handful of rounded
// products, and for a long thin Gaussian det = a c - b^2 is a difference of nearly equal numbers: the quadratic form
// the pixels see is off by a relative ~2^-24 (a c / det) per rounding, i.e. its level set { <= thr } is the exact
// ellipse scaled by up to sqrt(1 + k 2^-24 a c / det).  The threshold is widened by that factor (k = 32: conic
// entries, log2 scaling, the five operations of the power, with room to spare): 1 + 2e-6 for round Gaussians, tens of
// percent for needles hundreds of pixels long; when a c / det is so large that the bound says nothing (or det <= 0)
// the half-widths are unbounded and the caller's 3-sigma square stands.
__device__ __forceinline__ bool snug_half_widths(float opacity, float cov_a, float cov_c, float det, float* hx, float* hy) {
    const float x = 255.0f * opacity;
    if (!(x >= 1.0f)) return false;
    const float thr = 2.0f * ln_upper_bound(x) + 0.002f;  // + margin for the rounding of the per-pixel test
    const float cond = (cov_a * cov_c) / det;              // >= 1 for a positive definite covariance
    const float widen = 1.0f + cond * 1.9073486e-6f;       // 32 * 2^-24
    if (!(det > 0.0f) || !(widen <= 2.0f)) {
        *hx = *hy = 1.0e7f;
        return true;
    }
    const float thr_w = thr * widen;
    *hx = fminf(sqrtf(thr_w * cov_a), 1.0e7f);
    *hy = fminf(sqrtf(thr_w * cov_c), 1.0e7f);
    return true;
}

// SH -> RGB for a UNIT direction (x, y, z) (utils/sh_utils.py:58-101 polynomial; +0.5; clamp at 0 recorded
// as bit c of *clamped).  sh points at this Gaussian's (M,3) block.
__device__ __forceinline__ float3 sh_eval_dir(int deg, const float x, const float y, const float z,
                                              const float* __restrict__ sh, uint32_t* clamped) {
    float out[3];
    uint32_t cl = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        float r = SH_C0 * SHC(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SHC(4) + SH_C2[1] * yz * SHC(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SHC(6) +
                    SH_C2[3] * xz * SHC(7) + SH_C2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SHC(9) + SH_C3[1] * xy * z * SHC(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SHC(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SHC(13) + SH_C3[5] * z * (xx - yy) * SHC(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += 0.5f;
        if (r < 0.f) cl |= (1u << c);
        out[c] = fmaxf(r, 0.f);
    }
    *clamped = cl;
    return make_float3(out[0], out[1], out[2]);
}

// SH -> RGB seen from campos (the rasterizer's own conversion: direction = (mean - campos) / |mean - campos|)
__device__ __forceinline__ float3 sh_to_rgb(int deg, const float3 mean, const float3 campos, const float* __restrict__ sh,
                                            uint32_t* clamped) {
    const float dx = mean.x - campos.x, dy = mean.y - campos.y, dz = mean.z - campos.z;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    return sh_eval_dir(deg, dx / len, dy / len, dz / len, sh, clamped);
}

// ---- SH tiles: the (M,3) coefficient block of Gaussian i is 3M contiguous floats, so a thread-per-
// Gaussian kernel reading (or writing) it directly touches a different cache line per lane on every
// access.  These helpers move a whole workgroup's rows between HBM and LDS with coalesced dword
// accesses; in LDS a row has an ODD stride `ld`, so lanes walking the same column hit distinct banks.
__device__ __forceinline__ int sh_tile_ld(int M) { return (3 * M) | 1; }
// (rows of a multiple of four floats at a 16-byte aligned base move as dwordx4: a quarter of the global
// memory instructions)
__device__ __forceinline__ void sh_tile_load(const float* __restrict__ g, float* __restrict__ lds, int nrows, int rf,
                                             int ld) {
    if ((rf & 3) == 0 && ((uintptr_t)g & 15u) == 0) {
        const int rf4 = rf >> 2, total4 = nrows * rf4;
        int r = (int)threadIdx.x / rf4, c = (int)threadIdx.x - r * rf4;
        const int dr = (int)blockDim.x / rf4, dc = (int)blockDim.x - dr * rf4;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        // four loads in flight per thread and trip (as a plain loop every load is waited for before its LDS stores: twelve
        // dependent round trips for a full SH tile)
        for (int e0 = threadIdx.x; e0 < total4; e0 += 4 * (int)blockDim.x) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = e0 + u * (int)blockDim.x < total4 ? g4[e0 + u * (int)blockDim.x] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (e0 + u * (int)blockDim.x < total4) {
                    float* d = lds + r * ld + 4 * c;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
                r += dr;
                c += dc;
                if (c >= rf4) { c -= rf4; r++; }
            }
        }
        return;
    }
    const int total = nrows * rf;
    int r = (int)threadIdx.x / rf, c = (int)threadIdx.x - r * rf;
    const int dr = (int)blockDim.x / rf, dc = (int)blockDim.x - dr * rf;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        lds[r * ld + c] = g[e];
        r += dr;
        c += dc;
        if (c >= rf) { c -= rf; r++; }
    }
}
__device__ __forceinline__ void sh_tile_store(float* __restrict__ g, const float* __restrict__ lds, int nrows, int rf,
                                              int ld) {
    if ((rf & 3) == 0 && ((uintptr_t)g & 15u) == 0) {
        const int rf4 = rf >> 2, total4 = nrows * rf4;
        int r = (int)threadIdx.x / rf4, c = (int)threadIdx.x - r * rf4;
        const int dr = (int)blockDim.x / rf4, dc = (int)blockDim.x - dr * rf4;
        float4* g4 = reinterpret_cast<float4*>(g);
        for (int e = threadIdx.x; e < total4; e += blockDim.x) {
            const float* d = lds + r * ld + 4 * c;
            g4[e] = make_float4(d[0], d[1], d[2], d[3]);
            r += dr;
            c += dc;
            if (c >= rf4) { c -= rf4; r++; }
        }
        return;
    }
    const int total = nrows * rf;
    int r = (int)threadIdx.x / rf, c = (int)threadIdx.x - r * rf;
    const int dr = (int)blockDim.x / rf, dc = (int)blockDim.x - dr * rf;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        g[e] = lds[r * ld + c];
        r += dr;
        c += dc;
        if (c >= rf) { c -= rf; r++; }
    }
}
