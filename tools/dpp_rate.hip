// Microbenchmark: issue rate of cross-lane ops on gfx950 (wave64), 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 1) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
                if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 6) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    int iters = 2048;
    int nb = 256 * blocks_per_cu;
    float* d; (void)hipMalloc(&d, (size_t)nb * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, 16, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)nb * 4 * iters * 32.0 / 1024.0;
    printf("%-22s blocks/CU=%d  %.3f ms  cycles/instr/SIMD @2.4GHz: %.2f\n", name, blocks_per_cu, ms, ms * 1e6 / per_simd * 2.4);
    (void)hipFree(d);
}
int main() {
    for (int b : {2, 8}) {
        run<3>("v_add_f32", b);
        run<0>("add_dpp quad_perm", b);
        run<1>("add_dpp row_mirror", b);
        run<2>("add_dpp row_bcast15", b);
        run<4>("mov_dpp quad_perm", b);
        run<5>("add_dpp row_shr1", b);
        run<6>("add_dpp wave_shr1", b);
    }
    return 0;
}
