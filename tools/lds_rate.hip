// Microbenchmark: what an LDS instruction costs on gfx950 when only four lanes of the wave are active (the backward's
// entry switch and a candidate accumulator hand-off run that way), alone and beside a VALU stream.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_rate.bin tools/lds_rate.hip && tools/lds_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// OP: 0 ds_read_b128, 1 ds_read_b64, 2 ds_read_b32, 3 ds_wrxchg2_rtn_b64, 4 ds_wrxchg_rtn_b32, 5 ds_write_b128
// FEW: 0 all 64 lanes, 1 one lane per row of 16 (four lanes)
// NV: v_fma_f32 per group of four LDS instructions
template <int OP, int FEW, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ f4 mem[256 * 3];
    for (int i = threadIdx.x; i < 256 * 3; i += 256) mem[i] = f4{0, 0, 0, 0};
    __syncthreads();
    unsigned addr = threadIdx.x * 48;
    f4 a = {seed, seed, seed, seed}, b = a, c = a, d = a;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i;
    const unsigned long long few = 0x0001000100010001ull;
    unsigned long long saved;
    for (int it = 0; it < iters; it++) {
        if (FEW) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(few));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (OP == 0)
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4\n"
                             : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr));
            if (OP == 1)
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:16\n ds_read_b64 %2, %4 offset:32\n ds_read_b64 %3, %4 offset:8\n"
                             : "=v"(*(f2*)&a), "=v"(*(f2*)&b), "=v"(*(f2*)&c), "=v"(*(f2*)&d) : "v"(addr));
            if (OP == 2)
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:16\n ds_read_b32 %2, %4 offset:32\n ds_read_b32 %3, %4 offset:8\n"
                             : "=v"(a.x), "=v"(b.x), "=v"(c.x), "=v"(d.x) : "v"(addr));
            if (OP == 3) {
                f2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
                asm volatile("ds_wrxchg2_rtn_b64 %0, %2, %3, %4 offset1:1\n ds_wrxchg2_rtn_b64 %1, %2, %5, %6 offset0:2 offset1:3\n"
                             : "=&v"(a), "=&v"(b) : "v"(addr), "v"(a0), "v"(a1), "v"(b0), "v"(b1));
                f2 c0 = {c.x, c.y}, c1 = {c.z, c.w}, d0 = {d.x, d.y}, d1 = {d.z, d.w};
                asm volatile("ds_wrxchg2_rtn_b64 %0, %2, %3, %4 offset0:4 offset1:5\n ds_wrxchg2_rtn_b64 %1, %2, %5, %6 offset1:1\n"
                             : "=&v"(c), "=&v"(d) : "v"(addr), "v"(c0), "v"(c1), "v"(d0), "v"(d1));
            }
            if (OP == 4)
                asm volatile("ds_wrxchg_rtn_b32 %0, %4, %0\n ds_wrxchg_rtn_b32 %1, %4, %1 offset:16\n ds_wrxchg_rtn_b32 %2, %4, %2 offset:32\n"
                             "ds_wrxchg_rtn_b32 %3, %4, %3 offset:8\n"
                             : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(addr));
            if (OP == 5)
                asm volatile("ds_write_b128 %4, %0\n ds_write_b128 %4, %1 offset:16\n ds_write_b128 %4, %2 offset:32\n ds_write_b128 %4, %3\n"
                             : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(addr));
        }
        if (FEW) asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
#pragma unroll
        for (int u = 0; u < 2 * NV; u++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(seed), "v"(v[(u + 3) & 7]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = a.x + b.y + c.z + d.w;
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int FEW, int NV>
void run(const char* name, int blocks_per_cu) {
    int iters = 4096;
    int nb = 256 * blocks_per_cu;
    float* d; (void)hipMalloc(&d, (size_t)nb * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, FEW, NV>), dim3(nb), dim3(256), 0, 0, d, 16, 0.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, FEW, NV>), dim3(nb), dim3(256), 0, 0, d, iters, 0.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double lds_per_cu = (double)blocks_per_cu * 4 * iters * 8.0;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s lanes=%-2d fma/8lds=%-3d blocks/CU=%d  %.3f ms  cycles per LDS instr per CU: %.2f   (per SIMD per iteration: %.1f)\n", name,
           FEW ? 4 : 64, 2 * NV, blocks_per_cu, ms, cyc / lds_per_cu, cyc / (blocks_per_cu * (double)iters));
    (void)hipFree(d);
}
int main() {
    for (int b : {2, 6}) {
        run<0, 0, 0>("ds_read_b128", b);  run<0, 1, 0>("ds_read_b128", b);
        run<1, 0, 0>("ds_read_b64", b);   run<1, 1, 0>("ds_read_b64", b);
        run<2, 0, 0>("ds_read_b32", b);   run<2, 1, 0>("ds_read_b32", b);
        run<3, 0, 0>("ds_wrxchg2_rtn_b64", b); run<3, 1, 0>("ds_wrxchg2_rtn_b64", b);
        run<4, 0, 0>("ds_wrxchg_rtn_b32", b);  run<4, 1, 0>("ds_wrxchg_rtn_b32", b);
        run<5, 0, 0>("ds_write_b128", b); run<5, 1, 0>("ds_write_b128", b);
    }
    // beside a VALU stream: 48 fma per 8 LDS instructions is about the backward's ratio (≈ 50 VALU per 6-9 LDS instructions)
    run<0, 1, 24>("ds_read_b128 + fma", 6); run<3, 1, 24>("ds_wrxchg2_rtn_b64 + fma", 6); run<1, 0, 24>("ds_read_b64 + fma", 6);
    run<0, 1, 12>("ds_read_b128 + fma", 6); run<3, 1, 12>("ds_wrxchg2_rtn_b64 + fma", 6); run<1, 0, 12>("ds_read_b64 + fma", 6);
    return 0;
}
