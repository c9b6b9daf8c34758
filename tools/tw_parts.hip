// Where the time of the tile-list launches goes: the product kernels (csrc/binning.hip, included as they are) alone on the
// chip on a synthetic ranking shaped like config 3's (200k Gaussians in random depth order, rectangles of ~17 tiles on a
// 64 x 64 tile grid).  Built once per stop point (-DTW_STOP_AFTER=1 | 3 | 99: tile_write_kernel with its later parts
// left out); tools/tw_parts.sh builds and runs all four.
#include "../3dgs-avatar-release_amd/csrc/binning.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

void gs_set_error(int, const char*) {}
int gs_tune_get(int) { return 0; }
void gs_prof_begin(const char*, hipStream_t) {}
void gs_prof_end(hipStream_t) {}

template <class F>
static float time_us(F launch, int reps = 200) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int P = 200000, gx = 64, gy = 64, ntiles = gx * gy;
    std::vector<uint4> rank(P);
    std::vector<uint32_t> chunk((P + 255) / 256, 0u);
    srand(3);
    unsigned long long D = 0;
    for (int r = 0; r < P; r++) {
        // nearer Gaussians (low ranks) are larger on screen, as in the benchmark scene: depth 2 .. 4 over the ranking
        const float depth = 2.0f + 2.0f * (float)r / (float)P, base = 4.2f * 3.0f / depth;
        const int w = (int)(base + (rand() % 3 - 1) + 0.5f) < 1 ? 1 : (int)(base + (rand() % 3 - 1) + 0.5f);
        const int h = (int)(base + (rand() % 3 - 1) + 0.5f) < 1 ? 1 : (int)(base + (rand() % 3 - 1) + 0.5f);
        int x0 = rand() % (gx + w) - w, y0 = rand() % (gy + h) - h;
        int x1 = x0 + w, y1 = y0 + h;
        x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; x1 = x1 > gx ? gx : x1; y1 = y1 > gy ? gy : y1;
        const int tt = (x1 - x0) * (y1 - y0);
        rank[r] = make_uint4((uint32_t)r, (uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16), (uint32_t)tt);
        chunk[r >> 8] += (uint32_t)tt;
        D += (unsigned long long)tt;
    }
    const BinGrid G = bin_grid(gx, gy);
    const int nseg = bin_segments(G, P);
    uint4* d_rank; uint32_t *d_chunk, *d_seg, *d_tot, *d_ranges, *d_order, *d_list, *d_start; unsigned long long* d_count;
    (void)hipMalloc(&d_rank, (size_t)P * 16); (void)hipMalloc(&d_chunk, chunk.size() * 4); (void)hipMalloc(&d_seg, (size_t)ntiles * 4 * G.nseg_max);
    (void)hipMalloc(&d_tot, (size_t)ntiles * 4 + 256); (void)hipMalloc(&d_ranges, (size_t)ntiles * 8); (void)hipMalloc(&d_order, (size_t)ntiles * 4);
    (void)hipMalloc(&d_list, (size_t)D * 4 + 1024); (void)hipMalloc(&d_count, 64); (void)hipMalloc(&d_start, (TB_MAX_SEG + 2) * 4);
    (void)hipMemcpy(d_rank, rank.data(), (size_t)P * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_chunk, chunk.data(), chunk.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_count, &D, 8, hipMemcpyHostToDevice);
    const PairCount pc{d_count, (uint32_t)D};
    const LongLists ll{0, nullptr};
    int band_rows = TC_CELLS / (gx + 1) - 1;
    if (band_rows > gy) band_rows = gy;
    const int nbands = (gy + band_rows - 1) / band_rows;
    auto zero = [&] { (void)hipMemsetAsync(d_tot, 0, (size_t)ntiles * 4, 0); };
    auto count = [&] { hipLaunchKernelGGL(tile_count_kernel, dim3((unsigned)(nbands * nseg)), dim3(TC_THREADS), 0, 0, d_rank, d_chunk, d_start, P, gx, gy, band_rows, nbands, nseg, ntiles, d_seg, d_tot); };
    auto write = [&] { hipLaunchKernelGGL(tile_write_kernel<false>, dim3((unsigned)(1 + G.nblocks * nseg)), dim3(TBK_THREADS), 0, 0, d_rank, d_start, P, gx, gy, G.nbx, G.nblocks, nseg, ntiles, d_seg, d_tot, reinterpret_cast<uint2*>(d_ranges), d_order, d_list, pc, ll); };
    zero(); count(); (void)hipDeviceSynchronize();
    if (TW_STOP_AFTER == 99) {
        printf("D = %llu pairs, %d segments x %d blocks\n", D, nseg, G.nblocks);
        printf("tile_count (+ the clearing of the totals)      %6.2f us\n", time_us([&] { zero(); count(); }));
    }
    zero(); count(); (void)hipDeviceSynchronize();
    printf("tile_write, TW_STOP_AFTER = %-2d                  %6.2f us\n", TW_STOP_AFTER, time_us(write));
    return 0;
}
