// Which XCD does a workgroup run on?  Prints, for a launch of 4096 workgroups of 64 threads, the raw HW_REG_XCC_ID of every
// workgroup against blockIdx % 8 (the dispatcher's round-robin, as the XCD-aware block maps assume it), and HW_REG_HW_ID.
//   hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
__global__ void probe(uint32_t* out) {
    uint32_t x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = h; }
}
int main() {
    const int n = 4096;
    uint32_t* d;
    hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, d);
    uint32_t* h = (uint32_t*)malloc(n * 8);
    hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s CUs %d\n", p.name, p.multiProcessorCount);
    std::map<uint32_t, int> raw;
    int table[8][16] = {};
    for (int b = 0; b < n; b++) { raw[h[2 * b]]++; table[b & 7][h[2 * b] & 15]++; }
    for (auto& kv : raw) printf("raw XCC_ID 0x%08x : %d workgroups\n", kv.first, kv.second);
    for (int r = 0; r < 8; r++) {
        printf("blockIdx %% 8 = %d :", r);
        for (int x = 0; x < 16; x++) if (table[r][x]) printf("  xcc %d x %d", x, table[r][x]);
        printf("\n");
    }
    printf("first 16 HW_ID: ");
    for (int b = 0; b < 16; b++) printf("%08x ", h[2 * b + 1]);
    printf("\n");
    return 0;
}
