// gaussian_bwd.hip -- per-Gaussian backward (replaces upstream
// computeCov2DCUDA + preprocessCUDA backward), fused with the segmented sum of the per-pair
// gradient rows the tile kernel wrote (render_bwd.hip): each Gaussian's rows are one contiguous
// segment [first_pair, first_pair + tiles_touched).  One thread per Gaussian; every output row
// is written in full (zeros for culled Gaussians), so the caller passes un-initialised tensors.
#include "common.h"
#include "gs_math.h"
#include "blend.h"

#ifndef GB_THREADS
#define GB_THREADS 256  // Gaussians per workgroup of the per-Gaussian backward (its SH tile: 49 floats per Gaussian in LDS)
#endif

// Segmented sum of the per-(pair, quadrant) gradient rows: 16 lanes per Gaussian walk its contiguous span of 4 tt rows
// (common.h: gradient_row), EIGHT ROWS PER STEP -- lane j of the group reads 16 bytes: row j >> 1 of the step, half j & 1
// of the 32-byte row -- so the lanes of a group read 256 contiguous bytes (a Gaussian-per-lane-group walk in which every
// lane fetched whole rows touched 64 different lines per load instruction and ran at half the rate random lines can be
// read at).  The backward tile kernel wrote only the rows whose ninth sum (the dense word array q8) is not
// ROW_UNWRITTEN.  Fold with DPP adds.
// sums[i] = 12 floats.
__global__ __launch_bounds__(256) void segment_reduce_kernel(int P, const int32_t* __restrict__ radii,
                                                             const float4* __restrict__ rec,
                                                             const uint32_t* __restrict__ tiles,
                                                             const uint32_t* __restrict__ q8,
                                                             const float4* __restrict__ qrows,
                                                             float4* __restrict__ sums, uint32_t* __restrict__ marks_flag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && marks_flag) *marks_flag = 0u;  // the marks are no longer as the forward left them (common.h: MARKS_CLEAN)
    const int i = t >> 4, j = t & 15;
    const int rs = j >> 1, h = j & 1;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a8 = 0.f;
    const bool in = i < P;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    if (in && radii[i] > 0) {
        const uint32_t off = __float_as_uint(rec[(size_t)i * 3 + 2].y);
        const uint32_t nrows = 4u * tiles[i];
        if (j < 2) {  // (what the last step needs of the record: requested now, not behind the rows)
            r0 = rec[(size_t)i * 3];
            r1 = rec[(size_t)i * 3 + 1];
        }
        const size_t row0 = (size_t)off * 4;
        // The kernel is a chain of dependent load latencies (record -> marks -> rows) times the number of wave generations,
        // not bandwidth: a trip covers 64 rows of the Gaussian (8 per lane: most Gaussians need one or two trips), all marks
        // of a trip are requested together, then all its rows, and the next trip's marks are requested before this trip's
        // rows are added up.
        constexpr int U = 8;
        uint32_t mark[U], mark_n[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = (uint32_t)rs + 8u * u;
            mark[u] = k < nrows ? q8[row0 + k] : ROW_UNWRITTEN;
        }
        for (uint32_t k0 = (uint32_t)rs; k0 < nrows; k0 += 8u * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t k = k0 + 8u * u;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mark[u] != ROW_UNWRITTEN) v[u] = qrows[(row0 + k) * 2 + h];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t k = k0 + 8u * U + 8u * u;
                mark_n[u] = k < nrows ? q8[row0 + k] : ROW_UNWRITTEN;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w;
                if (h == 0 && mark[u] != ROW_UNWRITTEN) a8 += __uint_as_float(mark[u]);
                mark[u] = mark_n[u];
            }
        }
    }
    // fold the eight row slots (lane ^ 2, lane ^ 4, lane ^ 8): lanes with h = 0 end up with the sums 0..3 and 8, lanes
    // with h = 1 with the sums 4..7
    // (DPP: the other pair of the quad, then the quads 4 and 8 lanes down the row of 16 -- every lane ends up with the
    // sum over the eight lanes of its parity; three VALU instructions per sum instead of three trips through the LDS unit)
    float f[5] = {a0, a1, a2, a3, a8};
#pragma unroll
    for (int c = 0; c < 5; c++) {
        float v = f[c];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm:[2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
        f[c] = v;
    }
    // The rows hold raw sums over pixels: (t dx, t dy, t dx^2, t dx dy, t dy^2, G dL/dalpha, w g_rgb) with
    // t = G dL/dalpha.  Apply what is constant per Gaussian: opacity (dL/dG = opacity dL/dalpha), the
    // conic combination giving dL/dmean2D (in the log2 domain the tile kernels work in), and the -1/2 of
    // dL/dconic.
    if (in && j < 2) {
        const float A2 = (-0.5f * LOG2E_F) * r0.z, B2 = -LOG2E_F * r0.w, C2 = (-0.5f * LOG2E_F) * r1.x, op = r1.y;
        const float il2 = 1.0f / LOG2E_F;
        if (j == 0) {
            const float oa0 = op * f[0], oa1 = op * f[1];
            sums[(size_t)i * 3] = make_float4((2.f * A2 * oa0 + B2 * oa1) * il2, (2.f * C2 * oa1 + B2 * oa0) * il2,
                                              f[2] * (-0.5f * op), f[3] * (-0.5f * op));
            sums[(size_t)i * 3 + 2] = make_float4(f[4], 0.f, 0.f, 0.f);
        } else {
            sums[(size_t)i * 3 + 1] = make_float4(f[0] * (-0.5f * op), f[1], f[2], f[3]);
        }
    }
}

__global__ __launch_bounds__(256) void gaussian_bwd_kernel(
    int P, int deg, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos, int W,
    int H, float tanfovx, float tanfovy, float fx, float fy, const int32_t* __restrict__ radii,
    const uint32_t* __restrict__ clamped, const float4* __restrict__ sums, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dsh, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dscales, float* __restrict__ dL_drotations, float* __restrict__ dL_dcov3D) {
    extern __shared__ float sh_tile[];  // SH rows of this workgroup: coefficients in, gradients out (in place)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ld = sh_tile_ld(M);
    const int row0 = blockIdx.x * blockDim.x;
    const int nrows = min((int)blockDim.x, P - row0);
    if (shs) {
        sh_tile_load(shs + (size_t)row0 * M * 3, sh_tile, nrows, M * 3, ld);
        __syncthreads();
    }
    const bool inb = i < P;
    const bool live = inb && radii[i] > 0;
    if (inb) {

    // ---- the nine 2-D gradient sums of this Gaussian (segment_reduce_kernel) ----
    float s[9];
    {
        const float4 a0 = sums[(size_t)i * 3], a1 = sums[(size_t)i * 3 + 1];
        const float a2 = sums[(size_t)i * 3 + 2].x;
        s[0] = a0.x; s[1] = a0.y; s[2] = a0.z; s[3] = a0.w;
        s[4] = a1.x; s[5] = a1.y; s[6] = a1.z; s[7] = a1.w;
        s[8] = a2;
    }
    const float g2x = s[0] * (0.5f * W), g2y = s[1] * (0.5f * H);  // d/d(NDC): pixel gradient * 0.5 * (W, H)
    const float gA = s[2], gB = s[3], gC = s[4], gO = s[5];
    const float gR[3] = {s[6], s[7], s[8]};

    dL_dmeans2D[3 * i] = g2x;
    dL_dmeans2D[3 * i + 1] = g2y;
    dL_dmeans2D[3 * i + 2] = 0.f;
    dL_dopacity[i] = gO;
    dL_dcolors[3 * i] = gR[0];
    dL_dcolors[3 * i + 1] = gR[1];
    dL_dcolors[3 * i + 2] = gR[2];

    float gm[3] = {0.f, 0.f, 0.f};
    float gc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gs[3] = {0.f, 0.f, 0.f};
    float gq[4] = {0.f, 0.f, 0.f, 0.f};

    if (live) {
        float V[16], PV[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { V[k] = viewmatrix[k]; PV[k] = projmatrix[k]; }
        const float3 mean = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
        float c6[6];
        float3 sc = make_float3(0, 0, 0);
        float4 q = make_float4(1, 0, 0, 0);
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * i + k];
        } else {
            sc = make_float3(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]);
            q = reinterpret_cast<const float4*>(rotations)[i];
            cov3d_from_scale_rot(sc, scale_modifier, q, c6);
        }
        // (i) conic -> cov2D, (ii) cov2D -> cov3D, (iii) cov2D -> mean through J
        float cov[3], Mx[2][3], t[3], tt2[2];
        cov2d(mean, fx, fy, tanfovx, tanfovy, c6, V, cov, Mx, t, tt2);
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float x_grad_mul = (tt2[0] < -limx || tt2[0] > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tt2[1] < -limy || tt2[1] > limy) ? 0.f : 1.f;
        const float ca = cov[0], cb = cov[1], cc = cov[2];
        const float denom = ca * cc - cb * cb;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * gA + 2 * cb * cc * gB + (denom - ca * cc) * gC);
            dL_dc = denom2inv * (-ca * ca * gC + 2 * ca * cb * gB + (denom - ca * cc) * gA);
            dL_db = denom2inv * 2 * (cb * cc * gA - (denom + 2 * cb * cb) * gB + ca * cb * gC);
            const float* m0 = Mx[0];
            const float* m1 = Mx[1];
            gc[0] = (m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc);
            gc[3] = (m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc);
            gc[5] = (m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc);
            gc[1] = 2 * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2 * m1[0] * m1[1] * dL_dc;
            gc[2] = 2 * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2 * m1[0] * m1[2] * dL_dc;
            gc[4] = 2 * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2 * m1[1] * m1[2] * dL_dc;
        }
        const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float dM[2][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float m0S = Mx[0][0] * S[k][0] + Mx[0][1] * S[k][1] + Mx[0][2] * S[k][2];
            const float m1S = Mx[1][0] * S[k][0] + Mx[1][1] * S[k][1] + Mx[1][2] * S[k][2];
            dM[0][k] = 2 * m0S * dL_da + m1S * dL_db;
            dM[1][k] = 2 * m1S * dL_dc + m0S * dL_db;
        }
        const float dJ00 = V[0] * dM[0][0] + V[4] * dM[0][1] + V[8] * dM[0][2];
        const float dJ02 = V[2] * dM[0][0] + V[6] * dM[0][1] + V[10] * dM[0][2];
        const float dJ11 = V[1] * dM[1][0] + V[5] * dM[1][1] + V[9] * dM[1][2];
        const float dJ12 = V[2] * dM[1][0] + V[6] * dM[1][1] + V[10] * dM[1][2];
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = x_grad_mul * -fx * tz2 * dJ02;
        const float dty = y_grad_mul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
        gm[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        gm[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        gm[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
        // (iv) mean2D -> mean3D through the perspective divide
        const float4 mh = xform4x4(mean, PV);
        const float mw = 1.0f / (mh.w + 0.0000001f);
        const float mul1 = mh.x * mw * mw, mul2 = mh.y * mw * mw;
        gm[0] += (PV[0] * mw - PV[3] * mul1) * g2x + (PV[1] * mw - PV[3] * mul2) * g2y;
        gm[1] += (PV[4] * mw - PV[7] * mul1) * g2x + (PV[5] * mw - PV[7] * mul2) * g2y;
        gm[2] += (PV[8] * mw - PV[11] * mul1) * g2x + (PV[9] * mw - PV[11] * mul2) * g2y;
        // (v) SH backward
        if (shs) {
            float* sh = sh_tile + threadIdx.x * ld;  // coefficients, overwritten by their gradients
            float* gsh = sh;
            const uint32_t cl = clamped[i];
            const float dox = mean.x - campos[0], doy = mean.y - campos[1], doz = mean.z - campos[2];
            const float len = sqrtf(dox * dox + doy * doy + doz * doz);
            const float x = dox / len, y = doy / len, z = doz / len;
            float ddir[3] = {0.f, 0.f, 0.f};
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float g = ((cl >> c) & 1u) ? 0.f : gR[c];
                float shr[16];
#pragma unroll
                for (int k = 0; k < 16; k++) shr[k] = (k < (deg + 1) * (deg + 1)) ? sh[k * 3 + c] : 0.f;
#define SHC(k) shr[k]
#define GSH(k) gsh[(k) * 3 + c]
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
                                  SH_C3[3] * SHC(12) * -3.f * 2.f * yz + SH_C3[4] * SHC(13) * -2.f * xy +
                                  SH_C3[5] * SHC(14) * -2.f * yz + SH_C3[6] * SHC(15) * -3.f * 2.f * xy;
                            dz += SH_C3[1] * SHC(10) * xy + SH_C3[2] * SHC(11) * 4.f * 2.f * yz + SH_C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * SHC(13) * 4.f * 2.f * xz + SH_C3[5] * SHC(14) * (xx - yy);
                        }
                    }
                }
                // coefficients above the active degree get zero gradient
                for (int k = (deg + 1) * (deg + 1); k < M; k++) GSH(k) = 0.f;
#undef SHC
#undef GSH
                ddir[0] += dx * g;
                ddir[1] += dy * g;
                ddir[2] += dz * g;
            }
            const float sum2 = dox * dox + doy * doy + doz * doz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            gm[0] += ((+sum2 - dox * dox) * ddir[0] - doy * dox * ddir[1] - doz * dox * ddir[2]) * invsum32;
            gm[1] += (-dox * doy * ddir[0] + (sum2 - doy * doy) * ddir[1] - doz * doy * ddir[2]) * invsum32;
            gm[2] += (-dox * doz * ddir[0] - doy * doz * ddir[1] + (sum2 - doz * doz) * ddir[2]) * invsum32;
        }
        // (vi) cov3D -> scale, rotation (scale gradient w.r.t. mod*scale, factor mod NOT applied: A8 vi)
        if (!cov3D_precomp) {
            float R[3][3];
            quat_to_R(q, R);
            const float sv[3] = {scale_modifier * sc.x, scale_modifier * sc.y, scale_modifier * sc.z};
            const float dS[3][3] = {{gc[0], 0.5f * gc[1], 0.5f * gc[2]},
                                    {0.5f * gc[1], gc[3], 0.5f * gc[4]},
                                    {0.5f * gc[2], 0.5f * gc[4], gc[5]}};
            float L[3][3], dL[3][3], dR[3][3];
#pragma unroll
            for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
                for (int c_ = 0; c_ < 3; c_++) L[r_][c_] = R[r_][c_] * sv[c_];
#pragma unroll
            for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
                for (int c_ = 0; c_ < 3; c_++)
                    dL[r_][c_] = 2.0f * (dS[r_][0] * L[0][c_] + dS[r_][1] * L[1][c_] + dS[r_][2] * L[2][c_]);
#pragma unroll
            for (int c_ = 0; c_ < 3; c_++) {
                gs[c_] = R[0][c_] * dL[0][c_] + R[1][c_] * dL[1][c_] + R[2][c_] * dL[2][c_];
#pragma unroll
                for (int r_ = 0; r_ < 3; r_++) dR[r_][c_] = dL[r_][c_] * sv[c_];
            }
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            gq[0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
            gq[1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) - 4 * x * (dR[2][2] + dR[1][1]);
            gq[2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[2][1] + dR[1][2]) - 4 * y * (dR[2][2] + dR[0][0]);
            gq[3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[2][1] + dR[1][2]) - 4 * z * (dR[1][1] + dR[0][0]);
        }
    } else if (shs) {
        float* gsh = sh_tile + threadIdx.x * ld;
        for (int k = 0; k < M * 3; k++) gsh[k] = 0.f;
    }
    dL_dmeans3D[3 * i] = gm[0];
    dL_dmeans3D[3 * i + 1] = gm[1];
    dL_dmeans3D[3 * i + 2] = gm[2];
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = gc[k];
    if (dL_dscales) {
        dL_dscales[3 * i] = gs[0];
        dL_dscales[3 * i + 1] = gs[1];
        dL_dscales[3 * i + 2] = gs[2];
    }
    if (dL_drotations) reinterpret_cast<float4*>(dL_drotations)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }  // inb
    if (shs) {
        __syncthreads();
        sh_tile_store(dL_dsh + (size_t)row0 * M * 3, sh_tile, nrows, M * 3, ld);
    }
}

int launch_gaussian_backward(const GsFwdArgs& a, const int32_t* radii, const float* rec, const uint32_t* tiles,
                             const uint32_t* clamped, const uint32_t* q8, const float* qrows, float* sums,
                             uint32_t* marks_flag, const GsGrads& g, hipStream_t s) {
    const float fy = a.H / (2.0f * a.tanfovy), fx = a.W / (2.0f * a.tanfovx);
    hipLaunchKernelGGL(segment_reduce_kernel, dim3((unsigned)(((size_t)a.P * 16 + 255) / 256)), dim3(256), 0, s, a.P, radii,
                       reinterpret_cast<const float4*>(rec), tiles, q8,
                       reinterpret_cast<const float4*>(qrows), reinterpret_cast<float4*>(sums), marks_flag);
    GS_LAUNCH_CHECK("segment_reduce", a.debug, s);
    const size_t lds = a.shs ? (size_t)GB_THREADS * ((3 * a.M) | 1) * sizeof(float) : 0;
    hipLaunchKernelGGL(gaussian_bwd_kernel, dim3((a.P + GB_THREADS - 1) / GB_THREADS), dim3(GB_THREADS), lds, s, a.P, a.sh_degree, a.M, a.means3D,
                       a.scales, a.scale_modifier, a.rotations, a.shs, a.cov3D_precomp, a.viewmatrix, a.projmatrix,
                       a.campos, a.W, a.H, a.tanfovx, a.tanfovy, fx, fy, radii, clamped,
                       reinterpret_cast<const float4*>(sums), g.dL_dmeans3D, g.dL_dmeans2D,
                       g.dL_dsh, g.dL_dcolors, g.dL_dopacity, g.dL_dscales, g.dL_drotations, g.dL_dcov3D);
    GS_LAUNCH_CHECK("gaussian_backward", a.debug, s);
    return GS_OK;
}
