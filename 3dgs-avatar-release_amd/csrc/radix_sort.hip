// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass.
// Replaces the two cub::DeviceRadixSort::SortPairs calls of the upstream path.
//
// Per pass: (1) per-block digit histogram (+ global digit totals), (2) exclusive scan of the
// digit-major [256][nblk] table, one workgroup per digit, (3) stable scatter (see rs_scatter_kernel).
#include "common.h"

#define RS_THREADS 256
#define RS_ROUNDS_BIG (SORT_ITEMS / RS_THREADS)  // 16 keys per thread: large sorts
#define RS_REPL SORT_TOTALS_REPL
#define RS_ROUNDS_SMALL 4                      // 4 keys per thread: small sorts finish sooner on more CUs

// wave-level digit match: lanes holding the same 8-bit digit find each other with 8 ballots.
// Returns the peer mask (lanes with my digit among the `valid` ones).
__device__ __forceinline__ unsigned long long match_digit(uint32_t digit, bool valid, int dbits) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        if (b >= dbits) break;  // wave-uniform
        const bool bit = (digit >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

// Per-block digit histogram: LDS atomic adds into a histogram private to each wave (the LDS unit
// resolves equal addresses inside one instruction at a word per clock, which beats matching equal
// digits with ballots first), summed over the four waves at the end.
template <int RS_ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                             const unsigned long long* __restrict__ n_dev, int shift,
                                                             int dbits, uint32_t* __restrict__ hist, int nblk,
                                                             uint32_t* __restrict__ totals) {
    __shared__ uint32_t h[4][256];
    if (n_dev) { const unsigned long long d = *n_dev; n = d <= (unsigned long long)n ? (int64_t)d : 0; }  // count on the device, n = capacity
    const int tid = threadIdx.x, wid = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) h[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (RS_ROUNDS * RS_THREADS);
    const uint32_t dmask = (1u << dbits) - 1u;
    uint32_t key[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t i = base + r * RS_THREADS + tid;
        key[r] = i < n ? keys[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t i = base + r * RS_THREADS + tid;
        if (i < n) atomicAdd(&h[wid][(key[r] >> shift) & dmask], 1u);
    }
    __syncthreads();
    const uint32_t c = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    hist[(size_t)tid * nblk + blockIdx.x] = c;
    // digit totals, spread over RS_REPL replicas: hundreds of blocks adding to the same 256 words
    // serialise in the L2 atomic units (that, not the counting, dominated this kernel)
    if (c) atomicAdd(&totals[(blockIdx.x % RS_REPL) * 256 + tid], c);
}

// Exclusive scan of the digit-major [256][nblk] table, one workgroup per digit: block d adds the
// total of every smaller digit (from `totals`, accumulated by the histogram kernel) to the running
// prefix of its own row.
__global__ __launch_bounds__(256) void rs_scan_kernel(uint32_t* __restrict__ hist, int nblk,
                                                      const uint32_t* __restrict__ totals) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int d = blockIdx.x;
    uint32_t v = 0u;
    if (tid < d)
        for (int r = 0; r < RS_REPL; r++) v += totals[r * 256 + tid];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k, 64);
    if (lane == 0) ws[wid] = v;
    __syncthreads();
    if (tid == 0) carry_s = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    uint32_t* row = hist + (size_t)d * nblk;
    for (int base = 0; base < nblk; base += 256) {
        const int i = base + tid;
        const uint32_t x0 = i < nblk ? row[i] : 0u;
        uint32_t x = x0;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            uint32_t y = __shfl_up(x, k, 64);
            if (lane >= k) x += y;
        }
        if (lane == 63) ws[wid] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += ws[w];
        const uint32_t total = ws[0] + ws[1] + ws[2] + ws[3];
        const uint32_t carry = carry_s;
        if (i < nblk) row[i] = carry + woff + x - x0;
        __syncthreads();
        if (tid == 0) carry_s = carry + total;
        __syncthreads();
    }
}

// Stable scatter.  Each wave owns a CONTIGUOUS quarter (1024 keys) of the block's 4096, held in
// registers (16 per lane, wave-coalesced loads).  Phase 1: one ballot-match pass gives every key its
// rank among the equal digits of its wave's chunk and leaves the wave-private digit counts in LDS.
// Phase 2: a 256-wide exclusive scan turns them into BLOCK-LOCAL start offsets of every (wave, digit)
// and a per-digit delta = global start - local start.  Phase 3: the pairs are parked in LDS in
// block-sorted order (start offset + rank).  Phase 4:
// the block writes the parked pairs out linearly, so every digit's run leaves as one contiguous,
// coalesced burst (position = local index + delta[digit]).
template <int RS_ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t* __restrict__ kin,
                                                                const uint32_t* __restrict__ vin,
                                                                uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                int64_t n, const unsigned long long* __restrict__ n_dev,
                                                                int shift, int dbits,
                                                                const uint32_t* __restrict__ hist, int nblk) {
    const uint32_t dmask = (1u << dbits) - 1u;
    if (n_dev) { const unsigned long long d = *n_dev; n = d <= (unsigned long long)n ? (int64_t)d : 0; }
    if ((int64_t)blockIdx.x * (RS_ROUNDS * RS_THREADS) >= n) return;  // (workgroup-uniform) the grid covers the capacity
    __shared__ uint32_t whist[4][256];
    __shared__ uint32_t delta[256];
    __shared__ uint32_t wsum[4];
    constexpr int ITEMS = RS_ROUNDS * RS_THREADS;
    __shared__ uint32_t skey[ITEMS];
    __shared__ uint32_t sval[ITEMS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) whist[w][tid] = 0;
    const int64_t bbase = (int64_t)blockIdx.x * ITEMS;
    const int64_t wbase = bbase + wid * (ITEMS / 4);
    const int nvalid = (int)((n - bbase) < (int64_t)ITEMS ? (n - bbase) : (int64_t)ITEMS);
    uint32_t key[RS_ROUNDS], val[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t i = wbase + r * 64 + lane;
        key[r] = 0xFFFFFFFFu;
        val[r] = 0;
        if (i < n) { key[r] = kin[i]; val[r] = vin[i]; }
    }
    __syncthreads();
    // one match pass: rank of every key among the equal digits of its wave's chunk (earlier rounds +
    // lower lanes).  Only this wave touches whist[wid], and a wave's LDS operations execute in
    // program order, so plain read / leader write replaces atomics.
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t lrank[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t digit = (key[r] >> shift) & dmask;
        const unsigned long long peers = match_digit(digit, valid, dbits);
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t seen = whist[wid][digit];
        lrank[r] = seen + rank;
        if (valid && rank == 0) whist[wid][digit] = seen + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    {
        // thread d: counts of digit d per wave -> block-exclusive scan over digits
        const uint32_t c0 = whist[0][tid], c1 = whist[1][tid], c2 = whist[2][tid], c3 = whist[3][tid];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t x = tot;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            uint32_t y = __shfl_up(x, k, 64);
            if (lane >= k) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        const uint32_t lstart = woff + x - tot;
        whist[0][tid] = lstart;
        whist[1][tid] = lstart + c0;
        whist[2][tid] = lstart + c0 + c1;
        whist[3][tid] = lstart + c0 + c1 + c2;
        delta[tid] = hist[(size_t)tid * nblk + blockIdx.x] - lstart;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t lpos = whist[wid][(key[r] >> shift) & dmask] + lrank[r];
            skey[lpos] = key[r];
            sval[lpos] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = r * RS_THREADS + tid;
        if (i < nvalid) {
            const uint32_t k = skey[i];
            const uint32_t pos = (uint32_t)i + delta[(k >> shift) & dmask];
            kout[pos] = k;
            vout[pos] = sval[i];
        }
    }
}

template <int RS_ROUNDS>
static int sort_pairs_impl(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t* hist, int64_t n, int bits,
                           bool totals_zeroed, int debug, hipStream_t s, const unsigned long long* n_dev) {
    const int items = RS_ROUNDS * RS_THREADS;
    const int nblk = (int)((n + items - 1) / items);
    const int passes = radix_passes(bits);
    const int dbits = 8;  // (an equal 6 + 6 split of 12 tile bits measured no faster than 8 + 4)
    // per-pass digit totals live behind the [256][nblk] table (the layouts reserve room for them)
    uint32_t* totals = hist + (size_t)256 * nblk;
    if (!totals_zeroed) {
        hipError_t me = hipMemsetAsync(totals, 0, (size_t)passes * RS_REPL * 256 * 4, s);
        if (me != hipSuccess) { gs_set_error((int)me, "sort.memset"); return GS_E_HIP; }
    }
    uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    for (int p = 0; p < passes; p++) {
        const int shift = dbits * p;
        hipLaunchKernelGGL(rs_hist_kernel<RS_ROUNDS>, dim3(nblk), dim3(RS_THREADS), 0, s, ki, n, n_dev, shift, dbits, hist, nblk,
                           totals + RS_REPL * 256 * p);
        GS_LAUNCH_CHECK("sort.hist", debug, s);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1 << dbits), dim3(256), 0, s, hist, nblk, totals + RS_REPL * 256 * p);
        GS_LAUNCH_CHECK("sort.scan", debug, s);
        hipLaunchKernelGGL(rs_scatter_kernel<RS_ROUNDS>, dim3(nblk), dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, n_dev, shift,
                           dbits, hist, nblk);
        GS_LAUNCH_CHECK("sort.scatter", debug, s);
        uint32_t* t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    return GS_OK;
}

int launch_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t* hist, int64_t n, int bits,
                      bool totals_zeroed, int debug, hipStream_t s, const unsigned long long* n_dev) {
    if (n <= 0) return GS_OK;
    // SORT_SMALL_N and the table sizes in the layouts (common.h: sort_table_words, sort_totals_region) go together
    if (n <= SORT_SMALL_N) return sort_pairs_impl<RS_ROUNDS_SMALL>(k0, v0, k1, v1, hist, n, bits, totals_zeroed, debug, s, n_dev);
    return sort_pairs_impl<RS_ROUNDS_BIG>(k0, v0, k1, v1, hist, n, bits, totals_zeroed, debug, s, n_dev);
}
