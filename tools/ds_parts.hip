// Where the time of the depth ranking's launches goes: the product kernels (csrc/depth_sort.hip, included as they are) run
// alone on synthetic keys, with inputs shaped so that one part of the work at a time disappears.
//   hipcc --offload-arch=gfx950 -O3 -I3dgs-avatar-release_amd/csrc -o tools/ds_parts.bin tools/ds_parts.hip && tools/ds_parts.bin
#include "../3dgs-avatar-release_amd/csrc/depth_sort.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// stand-ins for what the library's other translation units provide
void gs_set_error(int, const char*) {}
int gs_tune_get(int) { return 0; }

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
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 0; }

// the wave-per-bucket kernel as it was before the ranking by counting (a bitonic network for every bucket), with parts left
// out.  PART: 1 = no sorting network, 2 = no rank records, 3 = neither, + 4 = no chunk sums, + 8 = no gathers
template <int PART>
__global__ __launch_bounds__(64) void wave_parts_kernel(const unsigned long long* __restrict__ tmp, const uint32_t* __restrict__ tot,
                                                        const uint32_t* __restrict__ loc, const uint32_t* __restrict__ grp, int nb,
                                                        const RankOut ro) {
    __shared__ unsigned long long s[DS_WAVE_CAP];
    const int lane = threadIdx.x;
    const int b = min((int)blockIdx.x, nb);
    const int n = (int)tot[b];
    const uint32_t in_group = loc[b];
    uint32_t part = lane < (b >> 6) ? grp[lane] : 0u;
    if (n == 0 || n > DS_WAVE_CAP || b == nb) return;
    const uint32_t start = wave_sum(part) + in_group;
    const unsigned long long* seg = tmp + start;
    ChunkAcc acc;
    auto step_done = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int i = lane; i < n; i += 64) s[i] = seg[i];
    step_done();
    if (!(PART & 1)) {
        int n_pad = 2;
        while (n_pad < n) n_pad <<= 1;
        const int half = n_pad >> 1;
        for (int k = 2; k <= n_pad; k <<= 1) {
            for (int t = lane; t < half; t += 64) {
                const int blk = t / (k >> 1), off = t % (k >> 1);
                const int i = blk * k + off, p = blk * k + (k - 1 - off);
                if (p < n) {
                    const unsigned long long a = s[i], c = s[p];
                    if (a > c) { s[i] = c; s[p] = a; }
                }
            }
            step_done();
            for (int j = k >> 2; j > 0; j >>= 1) {
                for (int t = lane; t < half; t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i + j;
                    if (p < n) {
                        const unsigned long long a = s[i], c = s[p];
                        if (a > c) { s[i] = c; s[p] = a; }
                    }
                }
                step_done();
            }
        }
    }
    if (PART & 12) {  // 4: rank records without the chunk sums; 8: without the gathers; 12: stores only
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            if (i < n) {
                const uint32_t id = (uint32_t)s[i];
                RankRec q;
                q.c = make_float4(0.f, 0.f, 1.f, 2.f);
                q.tt = id & 15u;
                if (!(PART & 8)) q = rank_fetch(ro, true, id);
                rank_store(ro, start + i, id, q);
                if (!(PART & 4)) atomicAdd(&ro.chunk_pairs[(start + i) >> 8], q.tt);
            }
        }
    } else if (!(PART & 2)) {
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            rank_emit(ro, start + i, i < n, i < n ? (uint32_t)s[i] : 0u, true, acc);
        }
        chunk_flush(ro, acc);
    } else if (lane == 0) {
        ro.sorted_idx[start] = (uint32_t)s[0];
    }
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 200000;
    const int nwaves = (P + 63) / 64, nb = ds_buckets(P), nbp = nb + 1, blocks = (P + DS_ITEMS - 1) / DS_ITEMS;
    std::vector<uint32_t> keys(P), tiles(P), wsum(nwaves, 0), wmin(nwaves, 0xFFFFFFFFu), wmax(nwaves, 0);
    srand(1);
    for (int i = 0; i < P; i++) {
        const float d = 2.0f + 8.0f * (rand() / (float)RAND_MAX);
        keys[i] = *(const uint32_t*)&d;
        tiles[i] = 1 + rand() % 30;
        wsum[i / 64] += tiles[i];
        wmin[i / 64] = keys[i] < wmin[i / 64] ? keys[i] : wmin[i / 64];
        wmax[i / 64] = keys[i] > wmax[i / 64] ? keys[i] : wmax[i / 64];
    }
    uint32_t *d_keys, *d_tiles, *d_wsum, *d_wmin, *d_wmax, *d_cnt, *d_pre, *d_tot, *d_loc, *d_grp, *d_range, *d_big, *d_chunk, *d_sorted;
    float* d_rec; unsigned long long *d_count, *d_tmp, *d_tmp2; uint4* d_rank;
    (void)hipMalloc(&d_keys, P * 4); (void)hipMalloc(&d_tiles, P * 4); (void)hipMalloc(&d_wsum, nwaves * 4);
    (void)hipMalloc(&d_wmin, nwaves * 4); (void)hipMalloc(&d_wmax, nwaves * 4);
    (void)hipMalloc(&d_cnt, (size_t)blocks * nbp * 4); (void)hipMalloc(&d_pre, (size_t)blocks * nbp * 4);
    (void)hipMalloc(&d_tot, nbp * 4); (void)hipMalloc(&d_loc, nbp * 4); (void)hipMalloc(&d_grp, 66 * 4); (void)hipMalloc(&d_range, 16);
    (void)hipMalloc(&d_big, nbp * 4); (void)hipMalloc(&d_chunk, (P / 256 + 2) * 4); (void)hipMalloc(&d_sorted, P * 4);
    (void)hipMalloc(&d_rec, (size_t)P * REC_F * 4); (void)hipMalloc(&d_count, 64); (void)hipMalloc(&d_tmp, (size_t)P * 8); (void)hipMalloc(&d_tmp2, (size_t)P * 8);
    (void)hipMalloc(&d_rank, (size_t)P * 16);
    (void)hipMemset(d_rec, 0, (size_t)P * REC_F * 4);
    (void)hipMemcpy(d_keys, keys.data(), P * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_tiles, tiles.data(), P * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_wsum, wsum.data(), nwaves * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_wmin, wmin.data(), nwaves * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_wmax, wmax.data(), nwaves * 4, hipMemcpyHostToDevice);
    const PairNumbering pn{d_tiles, d_wsum, d_rec, d_count, nullptr, d_chunk, (P + 255) / 256};
    const PairNumbering pn_nochunks{d_tiles, d_wsum, d_rec, d_count, nullptr, d_chunk, 0};
    const RankOut ro{d_rec, d_tiles, d_sorted, d_rank, d_chunk};
    const size_t lds_count = (size_t)nbp * 4, lds_scatter = (size_t)(nbp + 66) * 4;
    printf("P = %d: %d counting workgroups, %d buckets\n", P, blocks, nb);
    printf("empty kernel, 1 workgroup                      %6.2f us\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, nullptr); }));
    printf("empty kernel, %4d workgroups of 256           %6.2f us\n", blocks, time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, 0, nullptr); }));
    printf("empty kernel, %4d workgroups of 64            %6.2f us\n", nbp, time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(nbp), dim3(64), 0, 0, nullptr); }));
    auto count = [&](int nw, int nbk, const PairNumbering& q, int p) {
        hipLaunchKernelGGL(ds_count_kernel, dim3((p + DS_ITEMS - 1) / DS_ITEMS), dim3(DS_THREADS), (size_t)(nbk + 1) * 4, 0, d_keys, d_wmin, d_wmax, nw, p, nbk, d_cnt, d_range, q);
    };
    printf("ds_count as shipped                            %6.2f us\n", time_us([&] { count(nwaves, nb, pn, P); }));
    printf("ds_count without the chunk clears              %6.2f us\n", time_us([&] { count(nwaves, nb, pn_nochunks, P); }));
    printf("ds_count, 256 buckets (small count rows)       %6.2f us\n", time_us([&] { count(nwaves, 256, pn, P); }));
    printf("ds_count, one workgroup's keys only (P = 2048) %6.2f us\n", time_us([&] { count(32, nb, pn, 2048); }));
    // the whole chain, launch after launch as the library issues it
    auto prefix = [&] { hipLaunchKernelGGL(ds_prefix_kernel, dim3((nbp + 63) / 64), dim3(256), 0, 0, d_cnt, d_pre, d_tot, d_loc, d_grp, nbp, blocks); };
    auto scatter = [&] { hipLaunchKernelGGL(ds_scatter_kernel, dim3(blocks), dim3(DS_THREADS), lds_scatter, 0, d_keys, d_range, P, nb, d_pre, d_loc, d_grp, d_tmp); };
    const int helpers = P / 4096 < 1 ? 1 : (P / 4096 > 128 ? 128 : P / 4096);
    auto waves = [&] { hipLaunchKernelGGL(ds_bucket_sort_kernel, dim3((nb + DS_WG_BUCKETS - 1) / DS_WG_BUCKETS + helpers), dim3(DS_THREADS), 0, 0, d_tmp, d_tmp2, d_tot, d_loc, d_grp, nb, ro); };
    count(nwaves, nb, pn, P); prefix(); scatter(); (void)hipDeviceSynchronize();
    printf("ds_prefix                                      %6.2f us\n", time_us(prefix));
    printf("ds_scatter                                     %6.2f us\n", time_us(scatter));
    printf("ds_bucket_sort (a wave per bucket)             %6.2f us\n", time_us(waves));
    printf("  ... without the sorting network             %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<1>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("  ... without the rank records                 %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<2>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("  ... with neither                             %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<3>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("  ... no network, records without chunk sums   %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<5>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("  ... no network, records without gathers      %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<9>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("  ... no network, stores only                  %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<13>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("  ... all of it (copy of the product kernel)   %6.2f us\n", time_us([&] { hipLaunchKernelGGL(wave_parts_kernel<0>, dim3(nb), dim3(64), 0, 0, d_tmp, d_tot, d_loc, d_grp, nb, ro); }));
    printf("the four launches back to back                 %6.2f us\n", time_us([&] { count(nwaves, nb, pn, P); prefix(); scatter(); waves(); }));
    (void)lds_count;
    return 0;
}
