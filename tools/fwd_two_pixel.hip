// Microbenchmark for VERDICT r02 item 3 / DESIGN.md 7.3: the forward's inner loop with ONE pixel per lane (the shipped
// render_quadrant_1: a wave per 8x8 quadrant, three wave-uniform LDS reads of the staged entry per step) against a
// version with TWO pixels per lane (a wave per 8x16 pair of quadrants: the same three reads serve two pixels; dx and
// A2 dx are shared, the list a wave walks is the UNION of the two quadrants' lists).  Instruction streams only, on
// synthetic staged entries: the staging pass (one footprint test per list entry and lane), the LDS traffic, the blend
// chain with its compares / selects and the last-contributor mark are the product's; records, qlist, checkpoints and the
// early exit are left out of both.  Mode 0: 16384 waves x E entries, 64 pixels each; mode 1: 8192 waves x UNION x E
// entries, 128 pixels each -- the same image.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o fwd_two_pixel.bin ...
// Attribution builds (mode 0 line only): -DEXP_NOLDS = the loop without its LDS reads (every entry value made opaque per
// iteration, or the compiler hoists half the blend chain and the run reads 91 us): 145 us against 136 -- the reads cost
// nothing; -DEXP_NOB / -DEXP_NOC = without the second / third read: 133 / 140 us; -DEXP_NOEXP = without v_exp_f32: 125 us.
// The loop is bound by its own instruction stream (24 VALU + 1 transcendental per entry: 56 issue cycles, 85 measured).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// a cheap stand-in for stage_entry_quad's ~45 VALU instructions per staged list entry
__device__ __forceinline__ float stage_cost(float a, float b) {
    float x = a;
#pragma unroll
    for (int k = 0; k < 20; k++) x = __builtin_fmaf(x, b, a) * 0.999f;
    return x;
}

template <int PIX>
__global__ __launch_bounds__(64) void fwd_kernel(const float4* __restrict__ rec, float* __restrict__ out, int entries, int nrec,
                                                 unsigned long long* clk) {
    __shared__ float4 srec[66 * 3];
    const int lane = threadIdx.x;
    // shader clock under this load: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz)
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    const float pxf = (float)(blockIdx.x % 128 * 8 + (lane & 7)), pyf = (float)(blockIdx.x / 128 * 8 * PIX + (lane >> 3));
    uint32_t vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    float T[PIX], C0[PIX], C1[PIX], C2[PIX];
    uint32_t mark[PIX];
#pragma unroll
    for (int p = 0; p < PIX; p++) { T[p] = 1.0f; C0[p] = C1[p] = C2[p] = 0.f; mark[p] = 0xFFFFFFFFu; }
    size_t ri = ((size_t)blockIdx.x * 64 + lane) % nrec;
    float4 p0 = rec[ri * 3], p1 = rec[ri * 3 + 1], p2 = rec[ri * 3 + 2];
    for (int base = 0; base < entries; base += 64) {
        // staging: every lane converts (and tests) one list entry, the hits are compacted into LDS
        const float t = stage_cost(p0.x, p0.y);
        const bool hit = t != 12345.f;  // (always: the compaction's ballot / popcount are kept, the list is all hits)
        const unsigned long long bal = __ballot(hit);
        const int cnt = min(__popcll(bal), entries - base);
        wave_lds_sync();
        const int slot = __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        srec[slot * 3] = make_float4(p0.x, p0.y, -0.72f * p0.z, -1.44f * p0.w);
        srec[slot * 3 + 1] = make_float4(-0.72f * p1.x, p1.y, t * 0.f, 0.f);
        srec[slot * 3 + 2] = make_float4(p1.z, p1.w, p2.x, 0.f);
        wave_lds_sync();
        ri = (ri + 64) % nrec;
        p0 = rec[ri * 3]; p1 = rec[ri * 3 + 1]; p2 = rec[ri * 3 + 2];  // next batch in flight
        const char* sp = reinterpret_cast<const char*>(srec) + vzero;
        auto blend = [&](const float4 a, const float2 b, const float4 c) {
            const float dx = a.x - pxf, adx = a.z * dx;
#pragma unroll
            for (int p = 0; p < PIX; p++) {
                const float dy = a.y - (pyf + 8.f * p);
                const float power2 = __builtin_fmaf(adx, dx, __builtin_fmaf(a.w, dx, b.x * dy) * dy);
#ifdef EXP_NOEXP
                const float G = power2 * 0.001f + 0.01f;
#else
                const float G = __builtin_amdgcn_exp2f(power2);
#endif
                const float al = fminf(0.99f, b.y * G);
                const bool valid = power2 <= 0.0f && al >= (1.0f / 255.0f);
                const float a2 = valid ? al : 0.f;
                const float test_T = T[p] * (1.f - a2);
                const bool pass = test_T >= 0.0001f;
                const float wT = a2 * T[p];
                const float w = pass ? wT : 0.f;
                T[p] = pass ? test_T : -fabsf(T[p]);
                C0[p] += c.x * w;
                C1[p] += c.y * w;
                C2[p] += c.z * w;
                mark[p] = (valid && pass) ? (uint32_t)(uintptr_t)sp : mark[p];
            }
        };
        auto ld_a = [&](int o) { return *reinterpret_cast<const float4*>(sp + o); };
#ifdef EXP_NOB
        auto ld_b = [&](int o) { return make_float2(-0.01f, 0.05f); };
#else
        auto ld_b = [&](int o) { return *reinterpret_cast<const float2*>(sp + o + 16); };
#endif
#ifdef EXP_NOC
        auto ld_c = [&](int o) { return make_float4(0.5f, 0.4f, 0.3f, 0.f); };
#else
        auto ld_c = [&](int o) {
            const float4 c = *reinterpret_cast<const float4*>(sp + o + 32);
            asm volatile("" ::"v"(c.w));
            return c;
        };
#endif
        float4 a0 = ld_a(0), c0 = ld_c(0);
        float2 b0 = ld_b(0);
        int j = 0;
#ifdef EXP_NOLDS
        const float4 a1 = ld_a(48), c1 = ld_c(48);
        const float2 b1 = ld_b(48);
        for (; j + 1 < cnt; j += 2) {  // (attribution run: the loop without its LDS reads)
            // (every value opaque per iteration: nothing of the blend chain may be hoisted out of the loop)
            asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z), "+v"(a0.w), "+v"(b0.x), "+v"(b0.y), "+v"(c0.x), "+v"(c0.y), "+v"(c0.z), "+v"(sp));
            blend(a0, b0, c0);
            sp += 96;
            float4 a1v = a1, c1v = c1;
            float2 b1v = b1;
            asm volatile("" : "+v"(a1v.x), "+v"(a1v.y), "+v"(a1v.z), "+v"(a1v.w), "+v"(b1v.x), "+v"(b1v.y), "+v"(c1v.x), "+v"(c1v.y), "+v"(c1v.z));
            blend(a1v, b1v, c1v);
        }
#else
        for (; j + 1 < cnt; j += 2) {
            const float4 a1 = ld_a(48), c1 = ld_c(48);
            const float2 b1 = ld_b(48);
            blend(a0, b0, c0);
            a0 = ld_a(96); c0 = ld_c(96); b0 = ld_b(96);
            sp += 96;
            blend(a1, b1, c1);
        }
#endif
        if (j < cnt) blend(a0, b0, c0);
    }
#pragma unroll
    for (int p = 0; p < PIX; p++)
        out[((size_t)blockIdx.x * PIX + p) * 64 + lane] = C0[p] + C1[p] + C2[p] + T[p] + __uint_as_float(mark[p] & 1u);
    if (clk && lane == 0 && (blockIdx.x & 63) == 0) {
        const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&clk[0], t1 - t0);
        atomicAdd(&clk[1], r1 - r0);
    }
}

// Mode 2: the staged entry is not written to LDS at all: it stays in the staging lane's registers and entry j is
// broadcast with nine v_readlane_b32 (SGPR operands in the blend chain): no LDS traffic, nine VALU issue slots more.
__global__ __launch_bounds__(64) void fwd_readlane_kernel(const float4* __restrict__ rec, float* __restrict__ out, int entries,
                                                           int nrec) {
    const int lane = threadIdx.x;
    const float pxf = (float)(blockIdx.x % 128 * 8 + (lane & 7)), pyf = (float)(blockIdx.x / 128 * 8 + (lane >> 3));
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    int mark = -1;
    size_t ri = ((size_t)blockIdx.x * 64 + lane) % nrec;
    float4 p0 = rec[ri * 3], p1 = rec[ri * 3 + 1], p2 = rec[ri * 3 + 2];
    for (int base = 0; base < entries; base += 64) {
        const float t = stage_cost(p0.x, p0.y);
        const unsigned long long bal = __ballot(t != 12345.f);
        const int cnt = min(__popcll(bal), entries - base);
        const float ex = p0.x, ey = p0.y, A2 = -0.72f * p0.z, B2 = -1.44f * p0.w, C2c = -0.72f * p1.x, eo = p1.y + t * 0.f, er = p1.z,
                    eg = p1.w, eb = p2.x;
        ri = (ri + 64) % nrec;
        p0 = rec[ri * 3]; p1 = rec[ri * 3 + 1]; p2 = rec[ri * 3 + 2];
        for (int j = 0; j < cnt; j++) {
            auto rl = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
            const float sx = rl(ex), sy = rl(ey), sA = rl(A2), sB = rl(B2), sC = rl(C2c), so = rl(eo), sr = rl(er), sg = rl(eg), sb = rl(eb);
            const float dx = sx - pxf, dy = sy - pyf;
            const float power2 = __builtin_fmaf(sA * dx, dx, __builtin_fmaf(sB, dx, sC * dy) * dy);
            const float G = __builtin_amdgcn_exp2f(power2);
            const float al = fminf(0.99f, so * G);
            const bool valid = power2 <= 0.0f && al >= (1.0f / 255.0f);
            const float a2 = valid ? al : 0.f;
            const float test_T = T * (1.f - a2);
            const bool pass = test_T >= 0.0001f;
            const float wT = a2 * T;
            const float w = pass ? wT : 0.f;
            T = pass ? test_T : -fabsf(T);
            C0 += sr * w;
            C1 += sg * w;
            C2 += sb * w;
            mark = (valid && pass) ? j : mark;
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = C0 + C1 + C2 + T + (float)(mark & 1);
}

int main(int argc, char** argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 241;        // entries a quadrant wave walks (config 3: 3.95 M / 16384)
    const float uni = argc > 2 ? atof(argv[2]) : 1.15f;  // |union of two adjacent quadrants' lists| / |one list|
    const int nrec = 200000, quads = 16384;
    float4* rec;
    float* out;
    CHECK(hipMalloc(&rec, (size_t)nrec * 48));
    CHECK(hipMalloc(&out, (size_t)quads * 64 * 4));
    float4* h = (float4*)malloc((size_t)nrec * 48);
    srand(1);
    for (int i = 0; i < nrec; i++) {
        const float x = rand() % 1024, y = rand() % 1024, s = 40.f + rand() % 60;
        h[i * 3] = make_float4(x, y, 1.f / s, 0.1f / s);
        h[i * 3 + 1] = make_float4(1.f / s, 0.02f + 0.0001f * (rand() % 100), 0.5f, 0.4f);  // small alphas: no pixel saturates
        h[i * 3 + 2] = make_float4(0.3f, 0.f, 0.f, 0.f);
    }
    CHECK(hipMemcpy(rec, h, (size_t)nrec * 48, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    unsigned long long* clk;
    CHECK(hipMalloc(&clk, 16));
    for (int mode = 0; mode < 3; mode++) {
        CHECK(hipMemset(clk, 0, 16));
        const int waves = mode == 1 ? quads / 2 : quads, ent = mode == 1 ? (int)(E * uni + 0.5f) : E;
        float best = 1e9f;
        for (int it = 0; it < 40; it++) {
            CHECK(hipEventRecord(e0));
            if (mode == 1) hipLaunchKernelGGL(fwd_kernel<2>, dim3(waves), dim3(64), 0, 0, rec, out, ent, nrec, clk);
            else if (mode == 0) hipLaunchKernelGGL(fwd_kernel<1>, dim3(waves), dim3(64), 0, 0, rec, out, ent, nrec, clk);
            else hipLaunchKernelGGL(fwd_readlane_kernel, dim3(waves), dim3(64), 0, 0, rec, out, ent, nrec);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 5 && ms < best) best = ms;
        }
        printf("%s: %5d waves x %3d entries: %.1f us\n", mode == 0 ? "1 pixel per lane, LDS-staged entries (shipped loop)" :
               (mode == 1 ? "2 pixels per lane, LDS-staged entries" : "1 pixel per lane, entries by v_readlane"), waves, ent, best * 1e3f);
        unsigned long long hc[2];
        CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        if (hc[1]) printf("    shader clock while it runs: %.2f GHz (s_memtime / s_memrealtime over the waves' lifetimes)\n", (double)hc[0] / (double)hc[1] * 0.1);
    }
    return 0;
}
