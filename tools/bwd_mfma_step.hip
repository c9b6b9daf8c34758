// Microbenchmark for DESIGN.md 5.5 "MFMA, considered": the per-step instruction mix of an UN-SKEWED backward round
// (16 entries of a ring meet the same pixel at once; T and Rem by 16-lane prefix product / prefix sum with DPP; the
// nine sums accumulated by two v_mfma_f32_16x16x4_f32 per step; per-pixel carry through LDS), timed on synthetic
// data, to compare with the shipped kernel's ~150 SIMD cycles per step (0.29 ms for 4.7 M steps on 1024 SIMDs).
// Not a correct backward -- only its instruction stream.   hipcc --offload-arch=gfx950 -O3 -o bwd_mfma_step.bin ...
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MUL_DPP(x, ctrl) asm volatile("v_mul_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(x))
#define ADD_DPP(x, ctrl) asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(x))

// MODE 0: everything; 1: the two MFMAs replaced by two plain FMAs (what do the matrix instructions cost?);
// 2: no prefix scans (what do the 9 dependent DPP operations cost?)
template <int MODE>
__global__ __launch_bounds__(64) void step_kernel(const float4* __restrict__ rec, float4* __restrict__ out, int rounds,
                                                  int nrec) {
    __shared__ float pix[7][64];
    __shared__ float atab[2][64][16];
    __shared__ float carry[2][64];
    const int lane = threadIdx.x, j = lane & 15, ring = lane >> 4;
    for (int c = 0; c < 7; c++) pix[c][lane] = 0.001f * (float)(lane + c) + (c == 3 ? 100.f : 0.f);
    for (int n = 0; n < 16; n++) { atab[0][lane][n] = n < 6 ? 0.5f + n : 0.f; atab[1][lane][n] = (n >= 6 && n < 9) ? 0.25f : 0.f; }
    carry[0][lane] = 1.0f;
    carry[1][lane] = 0.5f;
    __syncthreads();
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 p0 = rec[((size_t)blockIdx.x * 16 + j) % nrec * 3], p1 = rec[((size_t)blockIdx.x * 16 + j) % nrec * 3 + 1],
           p2 = rec[((size_t)blockIdx.x * 16 + j) % nrec * 3 + 2];
    for (int r = 0; r < rounds; r++) {
        // the round's 16 entries: one per ring lane, taken by all lanes at once
        const float ex = p0.x, ey = p0.y, A2 = -0.72f * p0.z, B2 = -1.44f * p0.w, C2 = -0.72f * p1.x, eo = p1.y, er = p1.z,
                    eg = p1.w, eb = p2.x;
        const size_t nx = ((size_t)blockIdx.x * 16 + (size_t)(r + 1) * 16 + j) % nrec;
        p0 = rec[nx * 3]; p1 = rec[nx * 3 + 1]; p2 = rec[nx * 3 + 2];  // next round's entries in flight
#pragma unroll 2
        for (int t = 0; t < 16; t++) {
            const int pidx = ring * 16 + t;  // the ring's pixel of this step (uniform inside the ring)
            const float g0 = pix[0][pidx], g1 = pix[1][pidx], g2 = pix[2][pidx], pxf = pix[3][pidx], pyf = pix[4][pidx];
            const uint32_t lim = __float_as_uint(pix[5][pidx]);
            const float Tin = carry[0][pidx], Rin = carry[1][pidx];
            const float dx = ex - pxf, dy = ey - pyf;
            const float power2 = __builtin_fmaf(A2 * dx, dx, __builtin_fmaf(B2, dx, C2 * dy) * dy);
            const float G = __builtin_amdgcn_exp2f(power2);
            const float al = fminf(0.99f, eo * G);
            const float a1 = (power2 <= 0.0f) ? al : 0.f;
            const float a2 = ((uint32_t)(r * 16 + j) < lim + 0x3F000000u) ? a1 : 0.f;
            const bool valid = a2 >= (1.0f / 255.0f);
            const float alpha = valid ? a2 : 0.f;
            const float Gv = valid ? G : 0.f;
            const float one_m = 1.f - alpha;
            // inclusive prefix product over the ring's 16 entries, then the exclusive one
            float P = one_m;
            float excl = 1.0f;
            if (MODE != 2) {
                MUL_DPP(P, "row_shr:1");
                MUL_DPP(P, "row_shr:2");
                MUL_DPP(P, "row_shr:4");
                MUL_DPP(P, "row_shr:8");
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(excl) : "v"(P));
            } else {
                excl = P + 0.5f;
            }
            const float T = Tin * excl;
            const float wgt = alpha * T;
            const float cg = er * g0 + eg * g1 + eb * g2;
            float S = cg * wgt;
            if (MODE != 2) {
                ADD_DPP(S, "row_shr:1");
                ADD_DPP(S, "row_shr:2");
                ADD_DPP(S, "row_shr:4");
                ADD_DPP(S, "row_shr:8");
            }
            const float Rem = Rin - S;
            const float dL_dalpha = T * cg - Rem * __builtin_amdgcn_rcpf(one_m);
            const float Gd = Gv * dL_dalpha;
            if (j == 15) { carry[0][pidx] = Tin * P; carry[1][pidx] = Rem; }
            const float m1 = atab[0][pidx][j], m2 = atab[1][pidx][j];
            if (MODE != 1) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m1, Gd, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m2, wgt, acc, 0, 0, 0);
            } else {
                acc[0] = __builtin_fmaf(m1, Gd, acc[0]);
                acc[1] = __builtin_fmaf(m2, wgt, acc[1]);
            }
        }
        // the round's sums: three partial-wave stores, as in the shipped kernel
        const size_t row = ((size_t)blockIdx.x * rounds + r) * 16 + j;
        if (ring < 3) out[row * 3 + ring] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        acc = floatx4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int MODE>
static void run(const char* name, float4* rec, float4* out, int nrec, int blocks, int rounds) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(step_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, rec, out, rounds, nrec);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(step_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, rec, out, rounds, nrec);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double steps = (double)blocks * rounds * 16;
    printf("%-34s %.3f ms for %.2f M steps -> %.1f SIMD cycles per step @2.4 GHz on 1024 SIMDs\n", name, best, steps / 1e6,
           best * 1e-3 * 2.4e9 * 1024.0 / steps);
}

int main() {
    const int nrec = 200000, blocks = 16384, rounds = 18;  // 18 rounds x 16 steps = 288 steps per quadrant
    float4 *rec, *out;
    (void)hipMalloc(&rec, (size_t)nrec * 48);
    (void)hipMalloc(&out, (size_t)blocks * rounds * 16 * 48);
    (void)hipMemset(rec, 0x3c, (size_t)nrec * 48);  // small positive floats
    run<0>("un-skewed, scans + 2 MFMA:", rec, out, nrec, blocks, rounds);
    run<1>("  ... MFMAs replaced by 2 FMAs:", rec, out, nrec, blocks, rounds);
    run<2>("  ... without the prefix scans:", rec, out, nrec, blocks, rounds);
    printf("(shipped kernel: 0.29 ms for 4.7 M steps -> ~150 cycles per step)\n");
    return 0;
}
