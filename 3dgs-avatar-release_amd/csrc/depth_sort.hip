// depth_sort.hip -- the Gaussians in ascending (depth key, index) order: the first half of the reference's
// "stable sort of (tile << 32 | depth_bits) keys" (binning.hip is the second half).
//
// A bucket sort, four launches instead of the twelve of a 4 x 8-bit LSD radix sort over the same 200k keys (which is
// nothing but launch latency: 1.6 MB of keys, ~4.7 us per launch).  Keys are positive float bits, so they order like the
// depths.  (1) count: every key goes to bucket floor(NB (key - kmin) / (kmax - kmin + 1)) -- kmin / kmax of the frame from
// the per-wave ranges the preprocess kernel left -- per-workgroup counts in LDS; Gaussians that touch no tile go to an
// extra bucket behind all others (nothing downstream looks at their order).  (2) prefix: per bucket over the workgroups,
// and over the buckets.  (3) scatter: (key << 32 | index) to the bucket's slice, any order inside it.  (4) one workgroup
// per bucket sorts its slice (bitonic network on the 64-bit composites in LDS): equal keys end up in ascending index
// order, i.e. the order of a STABLE sort on the key.  NB ~ P / 64 buckets (at most 4096): a bucket holds ~64 keys for a uniform
// spread of depths; one that holds more than the LDS takes (a scene with most Gaussians at one depth) is sorted in
// global memory by the same network -- slow (milliseconds for 200k keys in one bucket) but exact.
#include "common.h"

#define DS_THREADS 256
#define DS_CAP_BIG 16384   // composites a 256-thread workgroup sorts in LDS (128 KB): second launch, for unevenly spread depths

__device__ __forceinline__ uint32_t ds_bucket_of(uint32_t key, uint32_t kmin, unsigned long long span, int nb) {
    // span = kmax - kmin + 1 (>= 1); keys outside [kmin, kmax] only for Gaussians that touch no tile (0xFFFFFFFF)
    if (key == 0xFFFFFFFFu || key < kmin) return (uint32_t)nb;
    const unsigned long long b = (unsigned long long)(key - kmin) * (unsigned long long)nb / span;
    return b < (unsigned long long)nb ? (uint32_t)b : (uint32_t)(nb - 1);
}

// the frame's key range from the preprocess waves' partial ranges (every workgroup reduces them itself: a few KB)
__device__ __forceinline__ void ds_key_range(const uint32_t* __restrict__ wave_kmin, const uint32_t* __restrict__ wave_kmax,
                                             int nwaves, uint32_t* s_red /* LDS: 8 words */, uint32_t* kmin,
                                             unsigned long long* span) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int w = tid; w < nwaves; w += DS_THREADS) {
        lo = min(lo, wave_kmin[w]);
        hi = max(hi, wave_kmax[w]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
    }
    if (lane == 0) { s_red[wid] = lo; s_red[4 + wid] = hi; }
    __syncthreads();
    lo = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    hi = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    __syncthreads();
    *kmin = lo;
    *span = hi >= lo ? (unsigned long long)(hi - lo) + 1ull : 1ull;  // (no keyed Gaussian at all: everything goes to bucket nb)
}

// The same launch numbers the pairs (binning.hip, "Pair numbering": first_pair = exclusive prefix sum of tiles_touched in
// INDEX order, from the per-wave sums the preprocess kernel left; a workgroup's 2048 Gaussians are 32 of those waves) and
// delivers the frame's pair count, and clears the per-chunk pair sums the bucket kernels add to.
__global__ __launch_bounds__(DS_THREADS) void ds_count_kernel(const uint32_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ wave_kmin,
                                                              const uint32_t* __restrict__ wave_kmax, int nwaves, int P,
                                                              int nb, uint32_t* __restrict__ cnt,
                                                              uint32_t* __restrict__ krange, const PairNumbering pn) {
    extern __shared__ uint32_t s_hist[];  // nb + 1 counters + 8 words
    uint32_t* s_red = s_hist + nb + 1;
    const int tid = threadIdx.x;
    {
        __shared__ unsigned long long s_before[DS_THREADS / 64];
        __shared__ uint32_t s_wave[DS_ITEMS / 64 + 1];  // exclusive prefix of this workgroup's wave sums, + their total
        const int lane = tid & 63, wid = tid >> 6;
        constexpr int WPB = DS_ITEMS / 64;  // preprocess waves per workgroup of this launch
        const int w0 = WPB * (int)blockIdx.x;
        for (int c = blockIdx.x * DS_THREADS + tid; c < pn.nchunks; c += gridDim.x * DS_THREADS) pn.chunk_pairs[c] = 0u;
        unsigned long long before = 0;  // 64-bit: overflow of the 32-bit index space stays detectable in the count
        for (int w = tid; w < w0; w += DS_THREADS) before += pn.wave_tiles[w];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
        if (lane == 0) s_before[wid] = before;
        if (wid == 0) {
            static_assert(WPB <= 64, "one wave scans the workgroup's wave sums");
            const uint32_t v = (lane < WPB && w0 + lane < nwaves) ? pn.wave_tiles[w0 + lane] : 0u;
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            if (lane < WPB) s_wave[lane] = x - v;
            if (lane == 63) s_wave[WPB] = x;
        }
        __syncthreads();
        const unsigned long long block_base = (s_before[0] + s_before[1]) + (s_before[2] + s_before[3]);
        static_assert(DS_THREADS == 256, "four partial sums");
#pragma unroll
        for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
            const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
            const uint32_t v = i < P ? pn.tiles[i] : 0u;
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            if (i < P) pn.rec[(size_t)i * REC_F + 9] = __uint_as_float((uint32_t)block_base + s_wave[u * (DS_THREADS / 64) + wid] + x - v);
        }
        if (blockIdx.x == gridDim.x - 1 && tid == 0) {
            const unsigned long long total = block_base + s_wave[WPB];
            pn.count[0] = total;
            pn.count[2] = 0ull;  // "a second render's colours are not all ones" (recolor_kernel), for that render's use
            // ... and straight into the caller's pinned host word, which the host is polling: the pair count
            // reaches the CPU a PCIe write after it exists instead of after a copy + stream-sync wake-up
            if (pn.host_count) __hip_atomic_store(pn.host_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the host reads this word only: no release -- a system-scope release writes the L2 back)
        }
    }
    uint32_t k[DS_ITEMS / DS_THREADS];
#pragma unroll
    for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        k[u] = i < P ? keys[i] : 0u;
    }
    for (int b = tid; b <= nb; b += DS_THREADS) s_hist[b] = 0u;
    uint32_t kmin;
    unsigned long long span;
    ds_key_range(wave_kmin, wave_kmax, nwaves, s_red, &kmin, &span);  // (its barriers also cover the zeroing)
    if (blockIdx.x == 0 && tid == 0) {  // for the scattering pass: it need not reduce the waves' ranges again
        krange[0] = kmin;
        krange[1] = (uint32_t)span;
        krange[2] = (uint32_t)(span >> 32);
    }
#pragma unroll
    for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        if (i < P) atomicAdd(&s_hist[ds_bucket_of(k[u], kmin, span, nb)], 1u);
    }
    __syncthreads();
    for (int b = tid; b <= nb; b += DS_THREADS) cnt[(size_t)blockIdx.x * (nb + 1) + b] = s_hist[b];
}

// One wave per group of 64 buckets.  Per bucket: the workgroups' counts -> their exclusive prefix (in place) and the
// bucket's total; per group: the buckets' exclusive prefix inside the group and the group's total.
__global__ __launch_bounds__(64) void ds_prefix_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ pre,
                                                       uint32_t* __restrict__ tot, uint32_t* __restrict__ loc,
                                                       uint32_t* __restrict__ grp, int nbp, int blocks) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    if (b < nbp) {
        // (input and output are different arrays: the loads of all trips can be in flight together)
        for (int g0 = 0; g0 < blocks; g0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = (g0 + u < blocks) ? cnt[(size_t)(g0 + u) * nbp + b] : 0u;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (g0 + u < blocks) pre[(size_t)(g0 + u) * nbp + b] = acc;
                acc += v[u];
            }
        }
        tot[b] = acc;
    }
    uint32_t x = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (b < nbp) loc[b] = x - acc;
    if (lane == 63) grp[blockIdx.x] = x;
}

// first slot of every bucket into LDS (s_base[0 .. nbp]): group sums scanned by the first wave, + the in-group prefix
__device__ __forceinline__ void ds_bucket_bases(const uint32_t* __restrict__ loc, const uint32_t* __restrict__ grp, int nbp,
                                                uint32_t* s_grp /* LDS: 65 words */, uint32_t* s_base) {
    const int tid = threadIdx.x;
    const int ngrp = (nbp + 63) / 64;  // <= 65 (nb <= 4096)
    if (tid < 64) {
        uint32_t x = 0, carry = 0;
        for (int g0 = 0; g0 < ngrp; g0 += 64) {  // at most two trips
            const uint32_t v = g0 + tid < ngrp ? grp[g0 + tid] : 0u;
            x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if (tid >= d) x += y;
            }
            if (g0 + tid < ngrp) s_grp[g0 + tid] = carry + x - v;
            carry += __shfl(x, 63, 64);
        }
    }
    __syncthreads();
    for (int b = tid; b < nbp; b += DS_THREADS) s_base[b] = s_grp[b >> 6] + loc[b];
    __syncthreads();
}

__global__ __launch_bounds__(DS_THREADS) void ds_scatter_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ krange, int P,
                                                                int nb, const uint32_t* __restrict__ cnt,
                                                                const uint32_t* __restrict__ loc,
                                                                const uint32_t* __restrict__ grp,
                                                                unsigned long long* __restrict__ tmp) {
    extern __shared__ uint32_t s_mem[];  // (nb + 1) slots: next free slot of every bucket for this workgroup; 65 + 8 words
    const int nbp = nb + 1;
    uint32_t* s_next = s_mem;
    uint32_t* s_grp = s_mem + nbp;
    const int tid = threadIdx.x;
    uint32_t k[DS_ITEMS / DS_THREADS];
#pragma unroll
    for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        k[u] = i < P ? keys[i] : 0u;
    }
    const uint32_t kmin = krange[0];
    const unsigned long long span = (unsigned long long)krange[1] | ((unsigned long long)krange[2] << 32);
    ds_bucket_bases(loc, grp, nbp, s_grp, s_next);
    for (int b = tid; b < nbp; b += DS_THREADS) s_next[b] += cnt[(size_t)blockIdx.x * nbp + b];  // + the workgroups before this one
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        if (i < P) {
            const uint32_t pos = atomicAdd(&s_next[ds_bucket_of(k[u], kmin, span, nb)], 1u);
            tmp[pos] = ((unsigned long long)k[u] << 32) | (unsigned long long)(uint32_t)i;
        }
    }
}

// Ascending-only bitonic network over n_pad = 2^m >= n elements: the first step of every merge pairs element i with its
// mirror image inside the block (i ^ (k - 1)), the others pair i with i + j, and EVERY compare-exchange puts the smaller
// element at the lower index.  Elements at or beyond n therefore behave as +infinity that never moves: pairs that reach
// beyond n are simply skipped (no padding is stored, n need not be a power of two).  `LOAD` / `STORE` abstract the memory.
template <typename LOAD, typename STORE, typename SYNC>
__device__ __forceinline__ void ds_bitonic(const long long n, LOAD load, STORE store, SYNC sync) {
    long long n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    const long long half = n_pad >> 1;
    for (long long k = 2; k <= n_pad; k <<= 1) {
        for (long long t = threadIdx.x; t < half; t += DS_THREADS) {  // flip step
            const long long blk = t / (k >> 1), off = t % (k >> 1);
            const long long i = blk * k + off, p = blk * k + (k - 1 - off);
            if (p < n) {
                const unsigned long long a = load(i), c = load(p);
                if (a > c) { store(i, c); store(p, a); }
            }
        }
        sync();
        for (long long j = k >> 2; j > 0; j >>= 1) {
            for (long long t = threadIdx.x; t < half; t += DS_THREADS) {
                const long long i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i + j;
                if (p < n) {
                    const unsigned long long a = load(i), c = load(p);
                    if (a > c) { store(i, c); store(p, a); }
                }
            }
            sync();
        }
    }
}

// What binning needs of the Gaussian at every depth rank, written by the bucket kernels as they emit the ranking:
// (index, rect min x | y << 16, rect size w | h << 16, tiles touched) -- 16 bytes streamed by the binning workgroups
// instead of a 48-byte record gathered per rank -- and the tiles touched per chunk of 256 consecutive ranks (the binning
// workgroups cut the ranking into segments of equal WORK with it).  One call per wave and trip: lane l holds rank
// r0 + l (r0 wave-uniform), which spans at most two chunks; `acc` keeps the running sum of the chunk the wave is in
// and is added to chunk_pairs (cleared by ds_count_kernel) when the wave moves on: integer adds, order-independent.
struct ChunkAcc { uint32_t cur = 0xFFFFFFFFu, sum = 0u; };
__device__ __forceinline__ void chunk_flush(const RankOut& ro, ChunkAcc& acc) {
    if (acc.cur != 0xFFFFFFFFu && acc.sum != 0u && (threadIdx.x & 63) == 0) atomicAdd(&ro.chunk_pairs[acc.cur], acc.sum);
}
__device__ __forceinline__ void rank_emit(const RankOut& ro, uint32_t r, bool valid, uint32_t id, bool keyed, ChunkAcc& acc) {
    uint32_t tt = 0;
    if (valid) {
        const float4 c = reinterpret_cast<const float4*>(ro.rec)[(size_t)id * 3 + 2];
        tt = ro.tiles[id];
        ro.sorted_idx[r] = id;
        ro.ranklist[r] = make_uint4(id, __float_as_uint(c.z), __float_as_uint(c.w), tt);
    }
    if (!keyed) return;  // the Gaussians that touch no tile: nothing to add
    const uint32_t cA = (uint32_t)__builtin_amdgcn_readfirstlane((int)r) >> 8;  // (lane 0 holds the trip's first rank)
    uint32_t sA = (valid && (r >> 8) == cA) ? tt : 0u, sB = (valid && (r >> 8) != cA) ? tt : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sA += __shfl_xor(sA, d, 64);
        sB += __shfl_xor(sB, d, 64);
    }
    if (acc.cur != cA) { chunk_flush(ro, acc); acc.cur = cA; acc.sum = 0u; }
    acc.sum += sA;
    if (sB != 0u) { chunk_flush(ro, acc); acc.cur = cA + 1u; acc.sum = sB; }
}

// One WAVE per bucket for the buckets of up to DS_WAVE_CAP composites (all of them when the depths are evenly spread: ~64
// per bucket): the network runs in LDS without workgroup barriers -- a wave's LDS operations execute in order.
#define DS_WAVE_CAP 1024
__global__ __launch_bounds__(64) void ds_bucket_sort_wave_kernel(const unsigned long long* __restrict__ tmp,
                                                                 const uint32_t* __restrict__ tot,
                                                                 const uint32_t* __restrict__ loc,
                                                                 const uint32_t* __restrict__ grp, int nb,
                                                                 const RankOut ro) {
    __shared__ unsigned long long s[DS_WAVE_CAP];
    const int lane = threadIdx.x;
    // bucket; bucket nb holds the Gaussians that touch no tile and is shared by the workgroups nb, nb + 1, ...
    const int b = min((int)blockIdx.x, nb);
    const int n = (int)tot[b];
    if (n == 0 || (n > DS_WAVE_CAP && b != nb)) return;  // (larger buckets: ds_bucket_sort_kernel)
    uint32_t part = 0;
    for (int q = lane; q < (b >> 6); q += 64) part += grp[q];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    const uint32_t start = part + loc[b];
    const unsigned long long* seg = tmp + start;
    ChunkAcc acc;
    if (b == nb) {
        // no order needed among the Gaussians that touch no tile (every later stage skips them)
        for (int i0 = ((int)blockIdx.x - nb) * 64; i0 < n; i0 += ((int)gridDim.x - nb) * 64) {
            const int i = i0 + lane;
            rank_emit(ro, start + i, i < n, i < n ? (uint32_t)seg[i] : 0u, false, acc);
        }
        return;
    }
    // between two steps: nothing may be kept in registers or moved across (the LDS itself keeps a wave's accesses in order)
    auto step_done = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int i = lane; i < n; i += 64) s[i] = seg[i];
    step_done();
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    const int half = n_pad >> 1;
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int t = lane; t < half; t += 64) {  // flip step (see ds_bitonic)
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
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        rank_emit(ro, start + i, i < n, i < n ? (uint32_t)s[i] : 0u, true, acc);
    }
    chunk_flush(ro, acc);
}

// One workgroup per bucket.  Buckets of [n_lo, n_hi] composites are handled by this launch (the others by the launch of
// the other LDS size); at most CAP of them are sorted in LDS, more in global memory.
template <int CAP>
__global__ __launch_bounds__(DS_THREADS) void ds_bucket_sort_kernel(unsigned long long* __restrict__ tmp,
                                                                    const uint32_t* __restrict__ tot,
                                                                    const uint32_t* __restrict__ loc,
                                                                    const uint32_t* __restrict__ grp, int nb, int n_lo,
                                                                    const RankOut ro) {
    __shared__ unsigned long long s[CAP];
    __shared__ uint32_t s_grp[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;  // bucket; bucket nb holds the Gaussians that touch no tile
    const int n = (int)tot[b];
    if (n < n_lo || b == nb) return;  // (workgroup-uniform) not this launch's bucket
    // first slot of this bucket: the groups before its own + its prefix inside the group
    {
        const int g = b >> 6;
        uint32_t part = 0;
        for (int q = tid; q < g; q += DS_THREADS) part += grp[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if ((tid & 63) == 0) s_grp[tid >> 6] = part;
        __syncthreads();
    }
    const uint32_t start = s_grp[0] + s_grp[1] + s_grp[2] + s_grp[3] + loc[b];
    unsigned long long* seg = tmp + start;
    ChunkAcc acc;
    if (n <= CAP) {
        for (int i = tid; i < n; i += DS_THREADS) s[i] = seg[i];
        __syncthreads();
        ds_bitonic(n, [&](long long i) { return s[i]; }, [&](long long i, unsigned long long v) { s[i] = v; },
                   [&]() { __syncthreads(); });
        for (int i0 = 0; i0 < n; i0 += DS_THREADS) {
            const int i = i0 + tid;
            rank_emit(ro, start + i, i < n, i < n ? (uint32_t)s[i] : 0u, true, acc);
        }
        chunk_flush(ro, acc);
        return;
    }
    // A bucket larger than the LDS takes (most of the scene at one depth): the same network in global memory, by this
    // workgroup alone.  The elements go through agent-scope (L2) accesses and every stage ends with a device fence, so
    // that a wave reads what another wave of the workgroup stored in the stage before.  Slow, exact.
    ds_bitonic(n,
               [&](long long i) { return __hip_atomic_load(&seg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
               [&](long long i, unsigned long long v) { __hip_atomic_store(&seg[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
               [&]() { __threadfence(); __syncthreads(); });
    for (int i0 = 0; i0 < n; i0 += DS_THREADS) {
        const int i = i0 + tid;
        rank_emit(ro, start + i, i < n,
                  i < n ? (uint32_t)__hip_atomic_load(&seg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u, true, acc);
    }
    chunk_flush(ro, acc);
}

int launch_depth_sort(const uint32_t* keys, const uint32_t* wave_kmin, const uint32_t* wave_kmax, int nwaves, int P,
                      DepthSortState st, PairNumbering pn, RankOut ro, int debug, hipStream_t s) {
    const int nbp = st.nb + 1;
    const size_t lds_count = (size_t)(nbp + 8) * 4, lds_scatter = (size_t)(nbp + 66) * 4;
    hipLaunchKernelGGL(ds_count_kernel, dim3(st.blocks), dim3(DS_THREADS), lds_count, s, keys, wave_kmin, wave_kmax, nwaves, P,
                       st.nb, st.cnt, st.range, pn);
    GS_LAUNCH_CHECK("depth_sort.count", debug, s);
    hipLaunchKernelGGL(ds_prefix_kernel, dim3((nbp + 63) / 64), dim3(64), 0, s, st.cnt, st.pre, st.tot, st.loc, st.grp, nbp,
                       st.blocks);
    GS_LAUNCH_CHECK("depth_sort.prefix", debug, s);
    hipLaunchKernelGGL(ds_scatter_kernel, dim3(st.blocks), dim3(DS_THREADS), lds_scatter, s, keys, st.range, P, st.nb, st.pre,
                       st.loc, st.grp, st.tmp);
    GS_LAUNCH_CHECK("depth_sort.scatter", debug, s);
    // a wave per bucket for the buckets of up to DS_WAVE_CAP composites (all of them unless the depths are very unevenly
    // spread), then the larger ones with 128 KB of LDS per workgroup (its workgroups leave at once when there is none)
    const int helpers = P / 1024 < 1 ? 1 : (P / 1024 > 512 ? 512 : P / 1024);  // waves sharing the no-tile bucket
    hipLaunchKernelGGL(ds_bucket_sort_wave_kernel, dim3(st.nb + helpers), dim3(64), 0, s, st.tmp, st.tot, st.loc, st.grp, st.nb, ro);
    GS_LAUNCH_CHECK("depth_sort.buckets", debug, s);
    hipLaunchKernelGGL(ds_bucket_sort_kernel<DS_CAP_BIG>, dim3(nbp), dim3(DS_THREADS), 0, s, st.tmp, st.tot, st.loc, st.grp,
                       st.nb, DS_WAVE_CAP + 1, ro);
    GS_LAUNCH_CHECK("depth_sort.big_buckets", debug, s);
    return GS_OK;
}
