// blend.h -- the per-(pixel, Gaussian) evaluation shared by the forward and backward render
// kernels (they must take identical skip decisions), and wave64 reduction helpers.
#pragma once
#include <hip/hip_runtime.h>

// alpha of one Gaussian at one pixel: d = centre - pixel,
// power = -0.5 (A dx^2 + C dy^2) - B dx dy; rejected if power > 0 or alpha < 1/255.
// G = exp(power) is returned for the backward pass.
__device__ __forceinline__ bool splat_alpha(float dx, float dy, float A, float B, float C, float o, float& alpha,
                                            float& G) {
    const float power = -0.5f * (A * dx * dx + C * dy * dy) - B * dx * dy;
    if (power > 0.0f) return false;
    G = __expf(power);
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
