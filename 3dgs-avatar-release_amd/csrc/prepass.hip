// prepass.hip -- the two per-Gaussian torch chains that sit right in front of the rasterizer in the
// reference, each as one forward and one backward kernel, thread per Gaussian:
//   * covariance from scaling and rotation: scene/gaussian_model.py:28-32 + utils/general_utils.py:73-108,
//     194-207 (build_scaling_rotation, L L^T, strip_symmetric); the rotation is either the quaternion
//     (normalised by build_rotation) or the 3x3 `rotation_precomp` the rigid deformer attaches
//     (models/deformer/rigid.py:229-231);
//   * SH -> RGB colours: models/texture/texture.py:21-38 (view direction from the camera centre,
//     optionally rotated into the canonical frame by the transpose of the forward bone rotation and by a
//     view-noise matrix, normalised with +1e-12, eval_sh, +0.5, clamp at 0).
// The reference spends ~10 small torch launches on each and as many again in autograd.
#include "common.h"
#include "gs_math.h"

// normalised quaternion -> R, as utils/general_utils.py:87-108
__device__ __forceinline__ void rot_from_input(const float* __restrict__ rot, int is_matrix, int i, float R[3][3], float4* qn,
                                               float* norm) {
    if (is_matrix) {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) R[a][b] = rot[9 * (size_t)i + 3 * a + b];
        *qn = make_float4(1.f, 0.f, 0.f, 0.f);
        *norm = 1.f;
    } else {
        const float4 r = reinterpret_cast<const float4*>(rot)[i];
        const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
        *qn = make_float4(r.x / n, r.y / n, r.z / n, r.w / n);
        *norm = n;
        quat_to_R(*qn, R);
    }
}

__global__ __launch_bounds__(256) void build_cov_kernel(int N, const float* __restrict__ scales, float mod,
                                                        const float* __restrict__ rot, int is_matrix,
                                                        float* __restrict__ cov6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float R[3][3], n;
    float4 q;
    rot_from_input(rot, is_matrix, i, R, &q, &n);
    const float s[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
    float L[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) L[a][b] = R[a][b] * s[b];
    float* o = cov6 + 6 * (size_t)i;
    o[0] = L[0][0] * L[0][0] + L[0][1] * L[0][1] + L[0][2] * L[0][2];
    o[1] = L[0][0] * L[1][0] + L[0][1] * L[1][1] + L[0][2] * L[1][2];
    o[2] = L[0][0] * L[2][0] + L[0][1] * L[2][1] + L[0][2] * L[2][2];
    o[3] = L[1][0] * L[1][0] + L[1][1] * L[1][1] + L[1][2] * L[1][2];
    o[4] = L[1][0] * L[2][0] + L[1][1] * L[2][1] + L[1][2] * L[2][2];
    o[5] = L[2][0] * L[2][0] + L[2][1] * L[2][1] + L[2][2] * L[2][2];
}

__global__ __launch_bounds__(256) void build_cov_bwd_kernel(int N, const float* __restrict__ scales, float mod,
                                                            const float* __restrict__ rot, int is_matrix,
                                                            const float* __restrict__ dL_dcov6,
                                                            float* __restrict__ dL_dscales, float* __restrict__ dL_drot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float R[3][3], n;
    float4 q;
    rot_from_input(rot, is_matrix, i, R, &q, &n);
    const float s[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
    const float* g = dL_dcov6 + 6 * (size_t)i;
    // strip_symmetric reads the upper triangle of Sigma = L L^T: dL/dSigma = G (upper only), dL/dL = (G + G^T) L
    const float S[3][3] = {{2.f * g[0], g[1], g[2]}, {g[1], 2.f * g[3], g[4]}, {g[2], g[4], 2.f * g[5]}};
    float dLm[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
            dLm[a][b] = (S[a][0] * R[0][b] + S[a][1] * R[1][b] + S[a][2] * R[2][b]) * s[b];  // ((G+G^T) L)_{ab}, L = R diag(s)
    // L_ab = R_ab s_b:  dL/ds_b = sum_a dLm_ab R_ab (times mod for the input scale),  dL/dR_ab = dLm_ab s_b
#pragma unroll
    for (int b = 0; b < 3; b++)
        dL_dscales[3 * (size_t)i + b] = mod * (dLm[0][b] * R[0][b] + dLm[1][b] * R[1][b] + dLm[2][b] * R[2][b]);
    float dR[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) dR[a][b] = dLm[a][b] * s[b];
    if (is_matrix) {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) dL_drot[9 * (size_t)i + 3 * a + b] = dR[a][b];
    } else {
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        float gq[4];
        gq[0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
        gq[1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) - 4 * x * (dR[2][2] + dR[1][1]);
        gq[2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[2][1] + dR[1][2]) - 4 * y * (dR[2][2] + dR[0][0]);
        gq[3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[2][1] + dR[1][2]) - 4 * z * (dR[1][1] + dR[0][0]);
        // through q = r_in / |r_in|:  d/dr_in = (gq - q (q . gq)) / |r_in|
        const float dot = r * gq[0] + x * gq[1] + y * gq[2] + z * gq[3];
        reinterpret_cast<float4*>(dL_drot)[i] =
            make_float4((gq[0] - r * dot) / n, (gq[1] - x * dot) / n, (gq[2] - y * dot) / n, (gq[3] - z * dot) / n);
    }
}

struct Mat3 { float m[9]; };  // row-major, by value (the optional view-noise matrix)

// direction of texture.py:23-35: d = xyz - campos; cano: d = R_fwd^T d; noise: d = d @ noise; unit = d / (|d| + 1e-12)
__device__ __forceinline__ void view_dir(const float* __restrict__ xyz, const float* __restrict__ campos,
                                         const float* __restrict__ R_fwd, int use_noise, const Mat3& noise, int i, float d[3],
                                         float* len) {
    float v[3] = {xyz[3 * (size_t)i] - campos[0], xyz[3 * (size_t)i + 1] - campos[1], xyz[3 * (size_t)i + 2] - campos[2]};
    if (R_fwd) {
        const float* R = R_fwd + 9 * (size_t)i;
        const float w[3] = {R[0] * v[0] + R[3] * v[1] + R[6] * v[2], R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
                            R[2] * v[0] + R[5] * v[1] + R[8] * v[2]};  // R^T v
        v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
    }
    if (use_noise) {
        const float w[3] = {v[0] * noise.m[0] + v[1] * noise.m[3] + v[2] * noise.m[6],
                            v[0] * noise.m[1] + v[1] * noise.m[4] + v[2] * noise.m[7],
                            v[0] * noise.m[2] + v[1] * noise.m[5] + v[2] * noise.m[8]};  // v @ noise
        v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
    }
    const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2];
    *len = l;
}

__device__ __forceinline__ void load_sh48(const float* __restrict__ g, int M, float* l, bool vec) {
    if (vec) {
        const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float4 v = g4[k];
            l[4 * k] = v.x; l[4 * k + 1] = v.y; l[4 * k + 2] = v.z; l[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 48; k++) l[k] = (k < 3 * M) ? g[k] : 0.f;
    }
}

__global__ __launch_bounds__(256) void sh2rgb_kernel(int N, int deg, int M, const float* __restrict__ shs,
                                                     const float* __restrict__ xyz, const float* __restrict__ campos,
                                                     const float* __restrict__ R_fwd, int use_noise, Mat3 noise,
                                                     float* __restrict__ colors, uint8_t* __restrict__ clamped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float d[3], len;
    view_dir(xyz, campos, R_fwd, use_noise, noise, i, d, &len);
    const float inv = 1.0f / (len + 1e-12f);
    float l[48];
    load_sh48(shs + (size_t)i * M * 3, M, l, M == 16 && ((uintptr_t)shs & 15u) == 0);
    uint32_t cl;
    const float3 c = sh_eval_dir(deg, d[0] * inv, d[1] * inv, d[2] * inv, l, &cl);
    colors[3 * (size_t)i] = c.x;
    colors[3 * (size_t)i + 1] = c.y;
    colors[3 * (size_t)i + 2] = c.z;
    clamped[i] = (uint8_t)cl;
}

__global__ __launch_bounds__(256) void sh2rgb_bwd_kernel(int N, int deg, int M, const float* __restrict__ shs,
                                                         const float* __restrict__ xyz, const float* __restrict__ campos,
                                                         const float* __restrict__ R_fwd, int use_noise, Mat3 noise,
                                                         const uint8_t* __restrict__ clamped,
                                                         const float* __restrict__ dL_dcolors, float* __restrict__ dL_dshs,
                                                         float* __restrict__ dL_dxyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float d[3], len;
    view_dir(xyz, campos, R_fwd, use_noise, noise, i, d, &len);
    const float inv = 1.0f / (len + 1e-12f);
    const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
    const bool vec = M == 16 && (((uintptr_t)shs | (uintptr_t)dL_dshs) & 15u) == 0;
    float sh[48], gsh[48];
    load_sh48(shs + (size_t)i * M * 3, M, sh, vec);
    const uint32_t cl = clamped[i];
    float ddir[3] = {0.f, 0.f, 0.f};
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float g = ((cl >> c) & 1u) ? 0.f : dL_dcolors[3 * (size_t)i + c];
#define SHC(k) sh[(k) * 3 + c]
#define GSH(k) gsh[(k) * 3 + c]
#pragma unroll
        for (int k = 0; k < 16; k++) GSH(k) = 0.f;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        GSH(0) = SH_C0 * g;
        if (deg > 0) {
            GSH(1) = -SH_C1 * y * g;
            GSH(2) = SH_C1 * z * g;
            GSH(3) = -SH_C1 * x * g;
            dx = -SH_C1 * SHC(3);
            dy = -SH_C1 * SHC(1);
            dz = SH_C1 * SHC(2);
            if (deg > 1) {
                GSH(4) = SH_C2[0] * xy * g;
                GSH(5) = SH_C2[1] * yz * g;
                GSH(6) = SH_C2[2] * (2.f * zz - xx - yy) * g;
                GSH(7) = SH_C2[3] * xz * g;
                GSH(8) = SH_C2[4] * (xx - yy) * g;
                dx += SH_C2[0] * y * SHC(4) + SH_C2[2] * 2.f * -x * SHC(6) + SH_C2[3] * z * SHC(7) + SH_C2[4] * 2.f * x * SHC(8);
                dy += SH_C2[0] * x * SHC(4) + SH_C2[1] * z * SHC(5) + SH_C2[2] * 2.f * -y * SHC(6) + SH_C2[4] * 2.f * -y * SHC(8);
                dz += SH_C2[1] * y * SHC(5) + SH_C2[2] * 2.f * 2.f * z * SHC(6) + SH_C2[3] * x * SHC(7);
                if (deg > 2) {
                    GSH(9) = SH_C3[0] * y * (3.f * xx - yy) * g;
                    GSH(10) = SH_C3[1] * xy * z * g;
                    GSH(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * g;
                    GSH(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                    GSH(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * g;
                    GSH(14) = SH_C3[5] * z * (xx - yy) * g;
                    GSH(15) = SH_C3[6] * x * (xx - 3.f * yy) * g;
                    dx += SH_C3[0] * SHC(9) * 3.f * 2.f * xy + SH_C3[1] * SHC(10) * yz + SH_C3[2] * SHC(11) * -2.f * xy +
                          SH_C3[3] * SHC(12) * -3.f * 2.f * xz + SH_C3[4] * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                          SH_C3[5] * SHC(14) * 2.f * xz + SH_C3[6] * SHC(15) * 3.f * (xx - yy);
                    dy += SH_C3[0] * SHC(9) * 3.f * (xx - yy) + SH_C3[1] * SHC(10) * xz + SH_C3[2] * SHC(11) * (-3.f * yy + 4.f * zz - xx) +
                          SH_C3[3] * SHC(12) * -3.f * 2.f * yz + SH_C3[4] * SHC(13) * -2.f * xy + SH_C3[5] * SHC(14) * -2.f * yz +
                          SH_C3[6] * SHC(15) * -3.f * 2.f * xy;
                    dz += SH_C3[1] * SHC(10) * xy + SH_C3[2] * SHC(11) * 4.f * 2.f * yz + SH_C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) +
                          SH_C3[4] * SHC(13) * 4.f * 2.f * xz + SH_C3[5] * SHC(14) * (xx - yy);
                }
            }
        }
#undef SHC
#undef GSH
        ddir[0] += dx * g;
        ddir[1] += dy * g;
        ddir[2] += dz * g;
    }
    // coefficient gradients (coefficients above the active degree get zero)
    float* go = dL_dshs + (size_t)i * M * 3;
    if (vec) {
        float4* g4 = reinterpret_cast<float4*>(go);
#pragma unroll
        for (int k = 0; k < 12; k++) g4[k] = make_float4(gsh[4 * k], gsh[4 * k + 1], gsh[4 * k + 2], gsh[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < 48; k++)
            if (k < 3 * M) go[k] = gsh[k];
    }
    // unit = v / (|v| + eps):  d unit / d v = I / (l + eps) - v v^T / (l (l + eps)^2)
    const float dotv = d[0] * ddir[0] + d[1] * ddir[1] + d[2] * ddir[2];
    const float k2 = (len > 0.f) ? dotv * inv * inv / len : 0.f;
    float gv[3] = {ddir[0] * inv - d[0] * k2, ddir[1] * inv - d[1] * k2, ddir[2] * inv - d[2] * k2};
    if (use_noise) {  // v_out = v_in @ noise  ->  g_in = noise g_out
        const float w[3] = {noise.m[0] * gv[0] + noise.m[1] * gv[1] + noise.m[2] * gv[2],
                            noise.m[3] * gv[0] + noise.m[4] * gv[1] + noise.m[5] * gv[2],
                            noise.m[6] * gv[0] + noise.m[7] * gv[1] + noise.m[8] * gv[2]};
        gv[0] = w[0]; gv[1] = w[1]; gv[2] = w[2];
    }
    if (R_fwd) {  // v_out = R^T v_in  ->  g_in = R g_out
        const float* R = R_fwd + 9 * (size_t)i;
        const float w[3] = {R[0] * gv[0] + R[1] * gv[1] + R[2] * gv[2], R[3] * gv[0] + R[4] * gv[1] + R[5] * gv[2],
                            R[6] * gv[0] + R[7] * gv[1] + R[8] * gv[2]};
        gv[0] = w[0]; gv[1] = w[1]; gv[2] = w[2];
    }
    dL_dxyz[3 * (size_t)i] = gv[0];
    dL_dxyz[3 * (size_t)i + 1] = gv[1];
    dL_dxyz[3 * (size_t)i + 2] = gv[2];
}

static Mat3 mat3_of(const float* m) {
    Mat3 r;
    for (int k = 0; k < 9; k++) r.m[k] = m ? m[k] : (k % 4 == 0 ? 1.f : 0.f);
    return r;
}

int launch_build_cov(int N, const float* scales, float mod, const float* rot, int is_matrix, float* cov6, hipStream_t s) {
    StageScope st("build_cov", s);
    hipLaunchKernelGGL(build_cov_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, scales, mod, rot, is_matrix, cov6);
    GS_LAUNCH_CHECK("build_cov", 0, s);
    return GS_OK;
}
int launch_build_cov_bwd(int N, const float* scales, float mod, const float* rot, int is_matrix, const float* dL_dcov6,
                         float* dL_dscales, float* dL_drot, hipStream_t s) {
    StageScope st("build_cov_bwd", s);
    hipLaunchKernelGGL(build_cov_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, scales, mod, rot, is_matrix, dL_dcov6,
                       dL_dscales, dL_drot);
    GS_LAUNCH_CHECK("build_cov_bwd", 0, s);
    return GS_OK;
}
int launch_sh2rgb(int N, int deg, int M, const float* shs, const float* xyz, const float* campos, const float* R_fwd,
                  const float* noise_host, float* colors, uint8_t* clamped, hipStream_t s) {
    StageScope st("sh2rgb", s);
    hipLaunchKernelGGL(sh2rgb_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, deg, M, shs, xyz, campos, R_fwd,
                       noise_host ? 1 : 0, mat3_of(noise_host), colors, clamped);
    GS_LAUNCH_CHECK("sh2rgb", 0, s);
    return GS_OK;
}
int launch_sh2rgb_bwd(int N, int deg, int M, const float* shs, const float* xyz, const float* campos, const float* R_fwd,
                      const float* noise_host, const uint8_t* clamped, const float* dL_dcolors, float* dL_dshs, float* dL_dxyz,
                      hipStream_t s) {
    StageScope st("sh2rgb_bwd", s);
    hipLaunchKernelGGL(sh2rgb_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, deg, M, shs, xyz, campos, R_fwd,
                       noise_host ? 1 : 0, mat3_of(noise_host), clamped, dL_dcolors, dL_dshs, dL_dxyz);
    GS_LAUNCH_CHECK("sh2rgb_bwd", 0, s);
    return GS_OK;
}
