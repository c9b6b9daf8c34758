// depth_sort.hip -- the Gaussians in ascending (depth key, index) order: the first half of the reference's
// "stable sort of (tile << 32 | depth_bits) keys" (binning.hip is the second half).
//
// A bucket sort, four launches instead of the twelve of a 4 x 8-bit LSD radix sort over the same 200k keys (which is
// nothing but launch latency: 1.6 MB of keys, ~4.7 us per launch in a frame's trace).  Keys are positive float bits, so
// they order like the depths.  (1) count: every key goes to a bucket by a linear map of [kmin, kmax] -- the frame's range,
// from the per-wave ranges the preprocess kernel left -- per-workgroup counts in LDS; Gaussians that touch no tile go to an
// extra bucket behind all others (nothing downstream looks at their order).  (2) prefix: per bucket over the workgroups,
// and over the buckets.  (3) scatter: (key << 32 | index) to the bucket's slice, any order inside it.  (4) a WAVE per
// bucket puts its slice in order -- the composites are distinct and equal keys end up in ascending index order, i.e. the
// order of a STABLE sort on the key -- and emits the rank records.  NB ~ P / 64 buckets (at most 4096): a bucket holds ~64
// keys for a uniform spread of depths (ranked by counting, no network); unevenly spread depths go down the other paths
// of the same launch: a wave's bitonic network (up to 1024), the workgroup's (up to 4096), sub-buckets by sampling
// (beyond).  All of these kernels are chains of dependent round trips, not bandwidth: see DESIGN.md 5.2.
#include "common.h"

#define DS_THREADS 256
#define DS_WAVE_CAP 1024   // composites one wave sorts in its quarter of the workgroup's LDS
#define DS_WG_BUCKETS 4    // buckets (waves) per workgroup of the bucket launch
#define DS_CAP_BIG (DS_WG_BUCKETS * DS_WAVE_CAP)  // composites the whole workgroup sorts in LDS (32 KB): a bucket too large for a wave

// Bucket of a key: floor((key - kmin) * scale / 2^32) with scale = floor(nb * 2^32 / span), span = kmax - kmin + 1 -- a
// non-decreasing map of [kmin, kmax] onto [0, nb) that costs a 32 x 64-bit multiply per key (the exact quotient
// (key - kmin) * nb / span is a 64-bit division per key: ~3 us of the counting and of the scattering pass each).
// (key - kmin) < span, so the product stays below nb * 2^32.  Keys outside [kmin, kmax] only for Gaussians that touch
// no tile (0xFFFFFFFF): bucket nb.
__device__ __forceinline__ unsigned long long ds_bucket_scale(unsigned long long span, int nb) {
    return ((unsigned long long)nb << 32) / span;  // span >= 1
}
__device__ __forceinline__ uint32_t ds_bucket_of(uint32_t key, uint32_t kmin, unsigned long long scale, int nb) {
    if (key == 0xFFFFFFFFu || key < kmin) return (uint32_t)nb;
    const uint32_t d = key - kmin;
    const uint32_t lo = (uint32_t)scale, hi = (uint32_t)(scale >> 32);
    const uint32_t b = __umulhi(d, lo) + d * hi;  // (d * scale) >> 32 < nb: no carry out of the low word is lost
    return b < (uint32_t)nb ? b : (uint32_t)(nb - 1);
}

// The counting pass.  The same launch numbers the pairs (binning.hip, "Pair numbering": first_pair = exclusive prefix sum of
// tiles_touched in INDEX order, from the per-wave sums the preprocess kernel left; a workgroup's 2048 Gaussians are 32 of
// those waves), delivers the frame's pair count, and clears the per-chunk pair sums the bucket kernels add to.
// Every workgroup reduces the preprocess waves' key ranges and the pair sums of the waves before it by itself (a few KB).
// The kernel is a chain of memory round trips and nothing else, so everything a thread reads is requested before anything
// is waited for: its keys and tile counts, then the per-wave words in trips of 16 x 3 loads.
#define DS_WAVE_WORDS 16
__global__ __launch_bounds__(DS_THREADS) void ds_count_kernel(const uint32_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ wave_kmin,
                                                              const uint32_t* __restrict__ wave_kmax, int nwaves, int P,
                                                              int nb, uint32_t* __restrict__ cnt,
                                                              uint32_t* __restrict__ krange, const PairNumbering pn) {
    extern __shared__ uint32_t s_hist[];  // nb + 1 counters
    __shared__ uint32_t s_red[8];
    __shared__ unsigned long long s_before[DS_THREADS / 64];
    __shared__ uint32_t s_wave[DS_ITEMS / 64 + 1];  // exclusive prefix of this workgroup's wave sums, + their total
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int WPB = DS_ITEMS / 64;  // preprocess waves per workgroup of this launch
    constexpr int NU = DS_ITEMS / DS_THREADS;
    static_assert(WPB <= 64, "one wave scans the workgroup's wave sums");
    static_assert(DS_THREADS == 256, "four partial sums");
    const int w0 = WPB * (int)blockIdx.x;
    const uint32_t* __restrict__ tiles = pn.tiles;
    const uint32_t* __restrict__ wave_tiles = pn.wave_tiles;
    uint32_t k[NU], tv[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        k[u] = i < P ? keys[i] : 0u;
        tv[u] = i < P ? tiles[i] : 0u;
    }
    const uint32_t own = (wid == 0 && lane < WPB && w0 + lane < nwaves) ? wave_tiles[w0 + lane] : 0u;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    unsigned long long before = 0;  // 64-bit: overflow of the 32-bit index space stays detectable in the count
    for (int base = 0; base < nwaves; base += DS_THREADS * DS_WAVE_WORDS) {
        uint32_t a[DS_WAVE_WORDS], c[DS_WAVE_WORDS], t[DS_WAVE_WORDS];
#pragma unroll
        for (int u = 0; u < DS_WAVE_WORDS; u++) {
            const int w = base + u * DS_THREADS + tid;
            a[u] = w < nwaves ? wave_kmin[w] : 0xFFFFFFFFu;
            c[u] = w < nwaves ? wave_kmax[w] : 0u;
            t[u] = w < w0 ? wave_tiles[w] : 0u;  // (w0 <= nwaves)
        }
#pragma unroll
        for (int u = 0; u < DS_WAVE_WORDS; u++) {
            lo = min(lo, a[u]);
            hi = max(hi, c[u]);
            before += t[u];
        }
    }
    for (int c = blockIdx.x * DS_THREADS + tid; c < pn.nchunks; c += gridDim.x * DS_THREADS) pn.chunk_pairs[c] = 0u;
    for (int b = tid; b <= nb; b += DS_THREADS) s_hist[b] = 0u;
    lo = wave_min(lo);
    hi = wave_max(hi);
    before = wave_sum(before);
    if (lane == 0) { s_red[wid] = lo; s_red[4 + wid] = hi; s_before[wid] = before; }
    if (wid == 0) {
        const uint32_t x = wave_scan_incl(own);
        if (lane < WPB) s_wave[lane] = x - own;
        if (lane == 63) s_wave[WPB] = x;
    }
    __syncthreads();
    // the frame's key range; keys outside it only for Gaussians that touch no tile
    const uint32_t kmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    const uint32_t kmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const unsigned long long span = kmax >= kmin ? (unsigned long long)(kmax - kmin) + 1ull : 1ull;  // (no keyed Gaussian at all: everything goes to bucket nb)
    const unsigned long long block_base = (s_before[0] + s_before[1]) + (s_before[2] + s_before[3]);
    const unsigned long long scale = ds_bucket_scale(span, nb);  // (one division per thread instead of one per key)
    if (blockIdx.x == 0 && tid == 0) {  // for the scattering pass: it need not reduce the waves' ranges again
        krange[0] = kmin;
        krange[1] = (uint32_t)scale;
        krange[2] = (uint32_t)(scale >> 32);
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const unsigned long long total = block_base + s_wave[WPB];
        pn.count[0] = total;
        pn.count[2] = 0ull;  // "a second render's colours are not all ones" (recolor_kernel), for that render's use
        // ... and straight into the caller's pinned host word, which the host is polling: the pair count
        // reaches the CPU a PCIe write after it exists instead of after a copy + stream-sync wake-up
        if (pn.host_count) __hip_atomic_store(pn.host_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the host reads this word only: no release -- a system-scope release writes the L2 back)
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        const uint32_t x = wave_scan_incl(tv[u]);
        if (i < P) {
            pn.rec[(size_t)i * REC_F + 9] = __uint_as_float((uint32_t)block_base + s_wave[u * (DS_THREADS / 64) + wid] + x - tv[u]);
            atomicAdd(&s_hist[ds_bucket_of(k[u], kmin, scale, nb)], 1u);
        }
    }
    __syncthreads();
    for (int b = tid; b <= nb; b += DS_THREADS) cnt[(size_t)blockIdx.x * (nb + 1) + b] = s_hist[b];
}

// One workgroup per group of 64 buckets (lane = bucket), its four waves a quarter of the counting workgroups each.  Per
// bucket: the counting workgroups' counts -> their exclusive prefix and the bucket's total; per group: the buckets'
// exclusive prefix inside the group and the group's total.  A wave has up to DSP_ROWS loads in flight; with at most
// 4 x DSP_ROWS counting workgroups (P <= 262144) every count is loaded once, in one trip.
#define DSP_ROWS 32
__global__ __launch_bounds__(256) void ds_prefix_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ pre,
                                                        uint32_t* __restrict__ tot, uint32_t* __restrict__ loc,
                                                        uint32_t* __restrict__ grp, int nbp, int blocks) {
    __shared__ uint32_t s_part[4][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const bool live = b < nbp;
    const int per = (blocks + 3) >> 2, r0 = wid * per, r1 = min(blocks, r0 + per);
    const bool one_trip = per <= DSP_ROWS;  // (workgroup-uniform)
    uint32_t v[DSP_ROWS];
    uint32_t sum = 0;
    for (int g0 = r0; g0 < r1; g0 += DSP_ROWS) {
#pragma unroll
        for (int u = 0; u < DSP_ROWS; u++) v[u] = (live && g0 + u < r1) ? cnt[(size_t)(g0 + u) * nbp + b] : 0u;
#pragma unroll
        for (int u = 0; u < DSP_ROWS; u++) sum += v[u];
    }
    s_part[wid][lane] = sum;
    __syncthreads();
    uint32_t acc = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t q = s_part[w][lane];
        if (w < wid) acc += q;
        total += q;
    }
    for (int g0 = r0; g0 < r1; g0 += DSP_ROWS) {
        if (!one_trip) {
#pragma unroll
            for (int u = 0; u < DSP_ROWS; u++) v[u] = (live && g0 + u < r1) ? cnt[(size_t)(g0 + u) * nbp + b] : 0u;
        }
#pragma unroll
        for (int u = 0; u < DSP_ROWS; u++) {
            if (live && g0 + u < r1) pre[(size_t)(g0 + u) * nbp + b] = acc;
            acc += v[u];
        }
    }
    if (wid == 0) {
        if (live) tot[b] = total;
        const uint32_t x = wave_scan_incl(live ? total : 0u);
        if (live) loc[b] = x - total;
        if (lane == 63) grp[blockIdx.x] = x;
    }
}

// The scattering pass.  First slot of every bucket for this workgroup in LDS (s_next[0 .. nbp]): the group sums scanned by
// the first wave + the in-group prefix + the counting workgroups before this one.  As in the counting pass, everything a
// thread reads is requested up front (keys, its share of the bucket words, the group sums).
#define DS_MAX_NBP 4097  // ds_buckets() <= 4096, + the bucket of the Gaussians that touch no tile
__global__ __launch_bounds__(DS_THREADS) void ds_scatter_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ krange, int P,
                                                                int nb, const uint32_t* __restrict__ cnt,
                                                                const uint32_t* __restrict__ loc,
                                                                const uint32_t* __restrict__ grp,
                                                                unsigned long long* __restrict__ tmp) {
    extern __shared__ uint32_t s_mem[];  // (nb + 1) slots: next free slot of every bucket for this workgroup; 65 words
    const int nbp = nb + 1;
    uint32_t* s_next = s_mem;
    uint32_t* s_grp = s_mem + nbp;
    const int tid = threadIdx.x;
    constexpr int NB_T = (DS_MAX_NBP + DS_THREADS - 1) / DS_THREADS;
    uint32_t k[DS_ITEMS / DS_THREADS], first[NB_T];
#pragma unroll
    for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        k[u] = i < P ? keys[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < NB_T; u++) {
        const int b = u * DS_THREADS + tid;
        first[u] = b < nbp ? loc[b] + cnt[(size_t)blockIdx.x * nbp + b] : 0u;
    }
    const int ngrp = (nbp + 63) / 64;  // <= 65
    if (tid < 64) {
        uint32_t x = 0, carry = 0;
        for (int g0 = 0; g0 < ngrp; g0 += 64) {  // at most two trips
            const uint32_t v = g0 + tid < ngrp ? grp[g0 + tid] : 0u;
            x = wave_scan_incl(v);
            if (g0 + tid < ngrp) s_grp[g0 + tid] = carry + x - v;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        }
    }
    const uint32_t kmin = krange[0];
    const unsigned long long scale = (unsigned long long)krange[1] | ((unsigned long long)krange[2] << 32);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NB_T; u++) {
        const int b = u * DS_THREADS + tid;
        if (b < nbp) s_next[b] = s_grp[b >> 6] + first[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DS_ITEMS / DS_THREADS; u++) {
        const int i = blockIdx.x * DS_ITEMS + u * DS_THREADS + tid;
        if (i < P) {
            const uint32_t pos = atomicAdd(&s_next[ds_bucket_of(k[u], kmin, scale, nb)], 1u);
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
    for (long long k = 2, lk = 0; k <= n_pad; k <<= 1, lk++) {  // lk = log2(k / 2)
        for (long long t = threadIdx.x; t < half; t += DS_THREADS) {  // flip step
            const long long blk = t >> lk, off = t & ((k >> 1) - 1);
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
struct RankRec { float4 c; uint32_t tt; };
__device__ __forceinline__ RankRec rank_fetch(const RankOut& ro, bool valid, uint32_t id) {
    RankRec q;
    q.c = make_float4(0.f, 0.f, 0.f, 0.f);
    q.tt = 0u;
    if (valid) {
        q.c = reinterpret_cast<const float4*>(ro.rec)[(size_t)id * 3 + 2];
        q.tt = ro.tiles[id];
    }
    return q;
}
__device__ __forceinline__ void rank_store(const RankOut& ro, uint32_t r, uint32_t id, const RankRec& q) {
    ro.sorted_idx[r] = id;
    ro.ranklist[r] = make_uint4(id, __float_as_uint(q.c.z), __float_as_uint(q.c.w), q.tt);
}
__device__ __forceinline__ void rank_emit(const RankOut& ro, uint32_t r, bool valid, uint32_t id, bool keyed, ChunkAcc& acc) {
    const RankRec q = rank_fetch(ro, valid, id);
    const uint32_t tt = q.tt;
    if (valid) rank_store(ro, r, id, q);
    if (!keyed) return;  // the Gaussians that touch no tile: nothing to add
    const uint32_t cA = (uint32_t)__builtin_amdgcn_readfirstlane((int)r) >> 8;  // (lane 0 holds the trip's first rank)
    uint32_t sA = (valid && (r >> 8) == cA) ? tt : 0u, sB = (valid && (r >> 8) != cA) ? tt : 0u;
    sA = wave_sum(sA);
    sB = wave_sum(sB);
    if (acc.cur != cA) { chunk_flush(ro, acc); acc.cur = cA; acc.sum = 0u; }
    acc.sum += sA;
    if (sB != 0u) { chunk_flush(ro, acc); acc.cur = cA + 1u; acc.sum = sB; }
}

// A bucket of up to 64 EPL composites without a sorting network: the composites are distinct, so an element's place is the
// number of smaller ones, counted against broadcast LDS reads (n reads and n EPL compares per lane; the network is
// ~log^2 n dependent LDS round trips of ~80 instructions each, 9 of this kernel's 13 us at ~64 per bucket).  The owner lane
// writes the element's rank record straight to its place; what it needs of the Gaussian is requested before the counting
// starts.  At most 256 composites: they span at most two chunks of 256 ranks.
template <int EPL>
__device__ __forceinline__ void ds_small_bucket(const unsigned long long* __restrict__ seg, unsigned long long* s, int n,
                                                uint32_t start, const RankOut& ro) {
    const int lane = threadIdx.x & 63;
    unsigned long long a[EPL];
    RankRec q[EPL];
    uint32_t rk[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) a[e] = e * 64 + lane < n ? seg[e * 64 + lane] : ~0ull;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        if (e * 64 + lane < n) s[e * 64 + lane] = a[e];
        q[e] = rank_fetch(ro, e * 64 + lane < n, (uint32_t)a[e]);
        rk[e] = 0u;
    }
    if (lane < 8) s[n + lane] = ~0ull;  // the reads below come in eights: the slack counts for nothing
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j0 = 0; j0 < n; j0 += 8) {
        unsigned long long v[8];  // (the same words for every lane: broadcasts, all eight in flight)
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = s[j0 + u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int e = 0; e < EPL; e++) rk[e] += v[u] < a[e] ? 1u : 0u;
        }
    }
    const uint32_t cA = start >> 8;
    uint32_t sA = 0u, sB = 0u;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        if (e * 64 + lane < n) {
            const uint32_t r = start + rk[e];
            rank_store(ro, r, (uint32_t)a[e], q[e]);
            if ((r >> 8) == cA) sA += q[e].tt; else sB += q[e].tt;
        }
    }
    sA = wave_sum(sA);
    sB = wave_sum(sB);
    if (lane == 0) {
        if (sA) atomicAdd(&ro.chunk_pairs[cA], sA);
        if (sB) atomicAdd(&ro.chunk_pairs[cA + 1u], sB);
    }
}

// The bucket launch: one WAVE per bucket, four buckets per workgroup.  A bucket of up to DS_WAVE_CAP composites (all of them
// when the depths are evenly spread: ~64 per bucket) is the wave's own business, in its quarter of the workgroup's LDS and
// without workgroup barriers -- a wave's LDS operations execute in order: up to 256 composites by counting
// (ds_small_bucket), more by a bitonic network.  A larger bucket is left for the whole workgroup, behind the one barrier
// of the common case: up to DS_CAP_BIG composites in the four quarters together, more in global memory (slow, exact).
// (Until round 3 the large buckets had a launch of their own with 128 KB of LDS per workgroup: empty for evenly spread
// depths, and 4.7 us in the trace of every frame.)
__device__ __forceinline__ void ds_wave_bucket(const unsigned long long* __restrict__ seg, unsigned long long* s, int n,
                                               uint32_t start, const RankOut& ro) {
    const int lane = threadIdx.x & 63;
    if (n <= 64) { ds_small_bucket<1>(seg, s, n, start, ro); return; }  // (wave-uniform)
    if (n <= 128) { ds_small_bucket<2>(seg, s, n, start, ro); return; }
    if (n <= 256) { ds_small_bucket<4>(seg, s, n, start, ro); return; }
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
    for (int k = 2, lk = 0; k <= n_pad; k <<= 1, lk++) {  // lk = log2(k / 2)
        for (int t = lane; t < half; t += 64) {  // flip step (see ds_bitonic)
            const int blk = t >> lk, off = t & ((k >> 1) - 1);
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
    ChunkAcc acc;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        rank_emit(ro, start + i, i < n, i < n ? (uint32_t)s[i] : 0u, true, acc);
    }
    chunk_flush(ro, acc);
}

// (the whole workgroup; every thread calls it)
__device__ __forceinline__ void ds_large_bucket(unsigned long long* __restrict__ seg, unsigned long long* s, int n,
                                                uint32_t start, const RankOut& ro) {
    const int tid = threadIdx.x;
    ChunkAcc acc;
    if (n <= DS_CAP_BIG) {
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
    // A bucket larger than the LDS takes (much of the scene at one depth): the same network in global memory, by this
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

// A bucket beyond DS_CAP_BIG composites (a key range stretched by far outliers, or much of the scene at one depth): the
// workgroup buckets it once more, by SAMPLING -- 1024 of its composites (key << 32 | index: all distinct), sorted in LDS,
// give the splitters of ~64-composite sub-buckets whose sizes are balanced whatever the distribution (a linear map of
// the composites, tried first, puts 12 000 equal keys among a few hundred distinct ones into ONE sub-bucket); every
// composite finds its sub-bucket by a binary search over the splitters, is counted and scattered through LDS counters
// into a second array, and the four waves then take the sub-buckets like buckets of the launch.  A sub-bucket still too
// large for a wave is sorted by the whole workgroup (ds_large_bucket).  (Until round 3: a bitonic network in global
// memory for the whole bucket -- 1.0 ms for 12 000 composites; ds_large_bucket still ends that way beyond 4096.)
#define DS_SUB_MAX 1024  // (4 KB of counters: with the 32 KB of composites four workgroups still fit a CU)
#define DS_SAMPLE 1024
__device__ __forceinline__ void ds_giant_bucket(unsigned long long* __restrict__ seg, unsigned long long* __restrict__ seg2,
                                                unsigned long long* s, uint32_t* s_sub, int n, uint32_t start,
                                                const RankOut& ro) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nsub = min(DS_SUB_MAX, max(2, n / 64));
    static_assert(DS_SAMPLE <= DS_CAP_BIG / 2, "the sample sits in the front of s, the scan's words behind it");
    for (int i = tid; i < DS_SAMPLE; i += DS_THREADS) s[i] = seg[(long long)i * n / DS_SAMPLE];  // (n > DS_CAP_BIG >= DS_SAMPLE)
    for (int k = tid; k <= nsub; k += DS_THREADS) s_sub[k] = 0u;
    __syncthreads();
    ds_bitonic(DS_SAMPLE, [&](long long i) { return s[i]; }, [&](long long i, unsigned long long v) { s[i] = v; },
               [&]() { __syncthreads(); });
    // splitter j (j = 0 .. nsub - 2) = sample[(j + 1) DS_SAMPLE / nsub - 1]; sub-bucket of c = the number of splitters below c
    auto sub_of = [&](unsigned long long c) {
        int lo = 0, hi = nsub - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s[(mid + 1) * DS_SAMPLE / nsub - 1] < c) lo = mid + 1; else hi = mid;
        }
        return lo;  // (non-decreasing in c)
    };
    for (int i = tid; i < n; i += DS_THREADS) atomicAdd(&s_sub[sub_of(seg[i])], 1u);
    __syncthreads();
    // exclusive prefix of the counts: DS_SUB_MAX / DS_THREADS consecutive counters per thread, a wave scan, the waves' sums
    {
        constexpr int PER = DS_SUB_MAX / DS_THREADS;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) { c[u] = tid * PER + u < nsub ? s_sub[tid * PER + u] : 0u; sum += c[u]; }
        const uint32_t incl = wave_scan_incl(sum);
        __syncthreads();  // (all counts are in registers)
        uint32_t* s_w = reinterpret_cast<uint32_t*>(s + DS_SAMPLE);  // (behind the sample, which the scatter still reads)
        if (lane == 63) s_w[wid] = incl;
        __syncthreads();
        uint32_t before = incl - sum;
        for (int w = 0; w < wid; w++) before += s_w[w];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (tid * PER + u < nsub) s_sub[tid * PER + u] = before;
            before += c[u];
        }
        __syncthreads();
    }
    // scatter: afterwards s_sub[k] is the END of sub-bucket k (its start is the end of k - 1)
    for (int i = tid; i < n; i += DS_THREADS) {
        const unsigned long long c = seg[i];
        seg2[atomicAdd(&s_sub[sub_of(c)], 1u)] = c;
    }
    __threadfence();  // (waves of this workgroup read what other waves of it stored)
    __syncthreads();
    for (int k = wid; k < nsub; k += DS_WG_BUCKETS) {  // (wave-uniform)
        const uint32_t b0 = k ? s_sub[k - 1] : 0u, m = s_sub[k] - b0;
        if (m > 0u && m <= (uint32_t)DS_WAVE_CAP) ds_wave_bucket(seg2 + b0, s + wid * DS_WAVE_CAP, (int)m, start + b0, ro);
    }
    __syncthreads();
    for (int k = 0; k < nsub; k++) {  // (workgroup-uniform) the sub-buckets left over: normally none
        const uint32_t b0 = k ? s_sub[k - 1] : 0u, m = s_sub[k] - b0;
        if (m > (uint32_t)DS_WAVE_CAP) {
            ds_large_bucket(seg2 + b0, s, (int)m, start + b0, ro);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(DS_THREADS) void ds_bucket_sort_kernel(unsigned long long* __restrict__ tmp,
                                                                    unsigned long long* __restrict__ tmp2,
                                                                    const uint32_t* __restrict__ tot,
                                                                    const uint32_t* __restrict__ loc,
                                                                    const uint32_t* __restrict__ grp, int nb,
                                                                    const RankOut ro) {
    __shared__ unsigned long long s[DS_CAP_BIG];
    __shared__ uint32_t s_sub[DS_SUB_MAX + 1];      // (ds_giant_bucket)
    __shared__ uint32_t s_large[DS_WG_BUCKETS][2];  // per wave: size and first slot of a bucket left for the workgroup (0: none)
    static_assert(DS_THREADS == 64 * DS_WG_BUCKETS && DS_CAP_BIG == DS_WG_BUCKETS * DS_WAVE_CAP, "a wave and a quarter of the LDS per bucket");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int bucket_wgs = (nb + DS_WG_BUCKETS - 1) / DS_WG_BUCKETS;
    if ((int)blockIdx.x >= bucket_wgs) {  // (workgroup-uniform)
        // Bucket nb: the Gaussians that touch no tile, shared by the waves of the workgroups behind the buckets' own.  No
        // order needed among them (every later stage skips them).
        const int n = (int)tot[nb];
        const uint32_t in_group = loc[nb];
        const uint32_t part = lane < (nb >> 6) ? grp[lane] : 0u;
        const uint32_t start = wave_sum(part) + in_group;
        const unsigned long long* seg = tmp + start;
        const int w = ((int)blockIdx.x - bucket_wgs) * DS_WG_BUCKETS + wid, nw = ((int)gridDim.x - bucket_wgs) * DS_WG_BUCKETS;
        ChunkAcc acc;
        for (int i0 = w * 64; i0 < n; i0 += nw * 64) {
            const int i = i0 + lane;
            rank_emit(ro, start + i, i < n, i < n ? (uint32_t)seg[i] : 0u, false, acc);
        }
        return;
    }
    const int b = (int)blockIdx.x * DS_WG_BUCKETS + wid;  // this wave's bucket
    int n = 0;
    uint32_t start = 0;
    if (b < nb) {  // (wave-uniform)
        // (all three requested before the first is looked at; nb <= 4096: at most 64 groups before a bucket's own)
        n = (int)tot[b];
        const uint32_t in_group = loc[b];
        const uint32_t part = lane < (b >> 6) ? grp[lane] : 0u;
        start = wave_sum(part) + in_group;
    }
    const bool large = n > DS_WAVE_CAP;
    if (lane == 0) {
        s_large[wid][0] = large ? (uint32_t)n : 0u;
        s_large[wid][1] = start;
    }
    if (n > 0 && !large) ds_wave_bucket(tmp + start, s + wid * DS_WAVE_CAP, n, start, ro);
    __syncthreads();
    for (int w = 0; w < DS_WG_BUCKETS; w++) {
        const int nl = (int)s_large[w][0];  // (workgroup-uniform)
        if (nl) {
            if (nl > DS_CAP_BIG) ds_giant_bucket(tmp + s_large[w][1], tmp2 + s_large[w][1], s, s_sub, nl, s_large[w][1], ro);
            else ds_large_bucket(tmp + s_large[w][1], s, nl, s_large[w][1], ro);
            __syncthreads();  // (s is the next one's)
        }
    }
}

int launch_depth_sort(const uint32_t* keys, const uint32_t* wave_kmin, const uint32_t* wave_kmax, int nwaves, int P,
                      DepthSortState st, PairNumbering pn, RankOut ro, int debug, hipStream_t s) {
    const int nbp = st.nb + 1;
    if (nbp > DS_MAX_NBP) return GS_E_BAD_ARG;  // (ds_buckets() never asks for more)
    const size_t lds_count = (size_t)nbp * 4, lds_scatter = (size_t)(nbp + 66) * 4;
    hipLaunchKernelGGL(ds_count_kernel, dim3(st.blocks), dim3(DS_THREADS), lds_count, s, keys, wave_kmin, wave_kmax, nwaves, P,
                       st.nb, st.cnt, st.range, pn);
    GS_LAUNCH_CHECK("depth_sort.count", debug, s);
    hipLaunchKernelGGL(ds_prefix_kernel, dim3((nbp + 63) / 64), dim3(256), 0, s, st.cnt, st.pre, st.tot, st.loc, st.grp, nbp,
                       st.blocks);
    GS_LAUNCH_CHECK("depth_sort.prefix", debug, s);
    hipLaunchKernelGGL(ds_scatter_kernel, dim3(st.blocks), dim3(DS_THREADS), lds_scatter, s, keys, st.range, P, st.nb, st.pre,
                       st.loc, st.grp, st.tmp);
    GS_LAUNCH_CHECK("depth_sort.scatter", debug, s);
    // a wave per bucket, four per workgroup (which takes the buckets too large for a wave together); behind them the
    // workgroups whose waves share the bucket of the Gaussians that touch no tile
    const int helpers = P / 4096 < 1 ? 1 : (P / 4096 > 128 ? 128 : P / 4096);
    hipLaunchKernelGGL(ds_bucket_sort_kernel, dim3((st.nb + DS_WG_BUCKETS - 1) / DS_WG_BUCKETS + helpers), dim3(DS_THREADS), 0, s,
                       st.tmp, st.tmp2, st.tot, st.loc, st.grp, st.nb, ro);
    GS_LAUNCH_CHECK("depth_sort.buckets", debug, s);
    return GS_OK;
}
