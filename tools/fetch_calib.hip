// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access shapes of this
// library, against byte counts known from the access pattern and against the time the pattern takes.
// MI355X_MICROARCH.md (HBM section) establishes FETCH_SIZE = 1/2 of the bytes for wide coalesced streaming
// reads and says every other shape must be calibrated before an absolute figure is trusted.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
//   tools/fetch_calib.bin                                   (times, GB/s by the pattern's own byte count)
//   rocprofv3 --pmc FETCH_SIZE -d out --output-format csv -- tools/fetch_calib.bin   (then WRITE_SIZE)
//
// Patterns (each touches >= 1 GiB of a 2 GiB table, far beyond the 256 MiB Infinity Cache, every byte once):
//   stream16     lane i loads 16 B at base + 16 i                      (the guide's calibrated case)
//   stream4      lane i loads 4 B at base + 4 i
//   rec48_seq    lane i loads the 48-byte record i (3 x 16 B), records packed back to back, in order
//   rec48_rand   the same records in a random order (a permutation: every record once)
//   rec48_slot   48-byte records in 128-byte slots, random order          (one 64-B half line per record)
//   row32_rand   32-byte rows (2 x 16 B), random order                    (the backward's gradient rows, read side)
//   wrow32_rand  32-byte rows written (2 x 16 B stores), random order     (the backward's gradient rows, write side)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void stream16(const float4* __restrict__ t, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = t[i];
        a += v.x + v.y + v.z + v.w;
    }
    if (a == 123.456f) sink[0] = a;
}
__global__ void stream4(const float* __restrict__ t, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += t[i];
    if (a == 123.456f) sink[0] = a;
}
// record r lives at byte r * stride; ids == nullptr: in order
__global__ void rec48(const char* __restrict__ t, const uint32_t* __restrict__ ids, size_t n, size_t stride, float* sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = ids ? ids[i] : i;
        const float4* p = reinterpret_cast<const float4*>(t + r * stride);
        const float4 v0 = p[0], v1 = p[1], v2 = p[2];
        a += v0.x + v1.y + v2.z;
    }
    if (a == 123.456f) sink[0] = a;
}
__global__ void row32(const char* __restrict__ t, const uint32_t* __restrict__ ids, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* p = reinterpret_cast<const float4*>(t + (size_t)ids[i] * 32);
        const float4 v0 = p[0], v1 = p[1];
        a += v0.x + v1.y;
    }
    if (a == 123.456f) sink[0] = a;
}
__global__ void wrow32(char* __restrict__ t, const uint32_t* __restrict__ ids, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4* p = reinterpret_cast<float4*>(t + (size_t)ids[i] * 32);
        p[0] = make_float4(1.f, 2.f, 3.f, 4.f);
        p[1] = make_float4(5.f, 6.f, 7.f, 8.f);
    }
}

static std::vector<uint32_t> permutation(size_t n, uint64_t seed) {
    std::vector<uint32_t> p(n);
    for (size_t i = 0; i < n; i++) p[i] = (uint32_t)i;
    uint64_t s = seed;
    for (size_t i = n - 1; i > 0; i--) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const size_t j = (size_t)((s >> 17) % (i + 1));
        const uint32_t x = p[i]; p[i] = p[j]; p[j] = x;
    }
    return p;
}

template <typename F>
static void timed(const char* name, double bytes, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-12s pattern bytes %8.1f MiB  %7.3f ms  %7.1f GB/s\n", name, bytes / 1048576.0, ms, bytes / ms / 1e6);
}

int main() {
    const size_t TABLE = (size_t)2 << 30;
    char* t = nullptr;
    float* sink = nullptr;
    CK(hipMalloc(&t, TABLE));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(t, 0, TABLE));
    const size_t nrec = TABLE / 48 / 2;      // 1 GiB of packed 48-byte records
    const size_t nslot = TABLE / 128 / 2;    // 8 M records in 128-byte slots (1 GiB of slots, 384 MiB of records)
    const size_t nrow = TABLE / 32 / 2;      // 1 GiB of 32-byte rows
    std::vector<uint32_t> pr = permutation(nrec, 1), ps = permutation(nslot, 2), pw = permutation(nrow, 3);
    uint32_t *dr, *ds, *dw;
    CK(hipMalloc(&dr, nrec * 4));
    CK(hipMalloc(&ds, nslot * 4));
    CK(hipMalloc(&dw, nrow * 4));
    CK(hipMemcpy(dr, pr.data(), nrec * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, ps.data(), nslot * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, pw.data(), nrow * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    const size_t G = (size_t)1 << 30;
    for (int rep = 0; rep < 2; rep++) {  // the second round is the one to read (first touch of the pages in the first)
        printf("-- round %d\n", rep);
        timed("stream16", (double)G, [&] { hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const float4*)t, G / 16, sink); });
        timed("stream4", (double)G, [&] { hipLaunchKernelGGL(stream4, grid, block, 0, 0, (const float*)(t + G), G / 4, sink); });
        timed("rec48_seq", (double)nrec * 48, [&] { hipLaunchKernelGGL(rec48, grid, block, 0, 0, t, (const uint32_t*)nullptr, nrec, (size_t)48, sink); });
        timed("rec48_rand", (double)nrec * (48 + 4), [&] { hipLaunchKernelGGL(rec48, grid, block, 0, 0, t + G, dr, nrec, (size_t)48, sink); });
        timed("rec48_slot", (double)nslot * (48 + 4), [&] { hipLaunchKernelGGL(rec48, grid, block, 0, 0, t, ds, nslot, (size_t)128, sink); });
        timed("row32_rand", (double)nrow * (32 + 4), [&] { hipLaunchKernelGGL(row32, grid, block, 0, 0, t + G, dw, nrow, sink); });
        timed("wrow32_rand", (double)nrow * 32, [&] { hipLaunchKernelGGL(wrow32, grid, block, 0, 0, t, dw, nrow); });
    }
    CK(hipDeviceSynchronize());
    printf("records: rec48 %zu, slot %zu, row32 %zu\n", nrec, nslot, nrow);
    return 0;
}
