// Microbenchmark: VALU issue rate on gfx950 (wave64): independent fma chains, dpp adds, exp.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long* clk = nullptr) {
    // shader clock under this load: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), first wave of a block
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3f;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
                if (MODE == 1) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, false));
                if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]) ;
                if (MODE == 3) a[i] = (a[i] > c) ? a[i] * m : a[i] + c;  // cmp + cndmask-ish
                if (MODE == 4 && (i & 1) == 0) {  // packed fp32 fma: two floats per instruction
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 v = {a[i], a[i + 1]}, mm = {m, m}, cc = {c, c};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(mm), "v"(cc));
                    a[i] = v.x; a[i + 1] = v.y;
                }
                if (MODE == 5 && (i & 1) == 0) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 v = {a[i], a[i + 1]}, mm = {m, m};
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(mm));
                    a[i] = v.x; a[i + 1] = v.y;
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&clk[0], t1 - t0);
        atomicAdd(&clk[1], r1 - r0);
    }
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    int iters = 4096;
    int nb = 256 * blocks_per_cu;
    float* d; hipMalloc(&d, (size_t)nb * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    unsigned long long* clk; hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, iters, 1.0f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); hipFree(clk);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)nb * 4 /*waves*/ * iters * 32.0;  // wave-instructions of the measured op
    double per_simd = winstr / 1024.0;
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    printf("%-8s blocks/CU=%d  %.3f ms  wave-instr/SIMD=%.3g  ns/instr/SIMD=%.3f  (cycles @2.4GHz: %.2f)  shader clock %.2f GHz -> %.2f cycles\n", name, blocks_per_cu, ms,
           per_simd, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, ghz, ms * 1e6 / per_simd * ghz);
    hipFree(d);
}
int main() {
    for (int b : {1, 2, 4, 8}) { run<0>("fma", b); }
    for (int b : {1, 2, 8}) { run<1>("dpp_add", b); }
    for (int b : {1, 2, 8}) { run<2>("exp2", b); }
    for (int b : {1, 2, 8}) { run<3>("cmp_sel", b); }
    for (int b : {1, 2, 8}) { run<4>("pk_fma(x0.5 instr)", b); }
    for (int b : {1, 2, 8}) { run<5>("pk_mul(x0.5 instr)", b); }
    return 0;
}
