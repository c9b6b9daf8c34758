// binning.hip -- tile binning: prefix sum of tiles touched, emission of the
// (tile, Gaussian) pairs, tile ranges.  Replaces upstream InclusiveSum / duplicateWithKeys /
// identifyTileRanges.
//
// MI355X-first formulation of the reference's "stable sort of (tile << 32 | depth_bits) keys": the
// Gaussians are first put in (depth, index) order (a stable 32-bit sort over P items), the pairs are
// then emitted Gaussian-major IN THAT ORDER, and a stable sort on the tile id alone (12-16 bits
// over D items) yields exactly the list a stable 64-bit sort of the upstream keys yields -- with a
// third of the bytes per pass and a third of the passes over the D-sized arrays.
#include "common.h"

#define SC_THREADS 256
#define SC_PER_THREAD (SCAN_ITEMS / SC_THREADS)

// pass 1: gather tiles_touched into depth-rank order and reduce per block
__global__ __launch_bounds__(SC_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ sorted_idx,
                                                                 const uint32_t* __restrict__ tiles,
                                                                 uint32_t* __restrict__ tt_rank,
                                                                 uint32_t* __restrict__ bsum, int P) {
    __shared__ uint32_t ws[4];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SC_PER_THREAD; k++) {
        const int r = base + k * SC_THREADS + tid;
        if (r < P) {
            const uint32_t t = tiles[sorted_idx[r]];
            tt_rank[r] = t;
            sum += t;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((tid & 63) == 0) ws[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// pass 2: per-block exclusive scan; every block adds up the sums of the blocks before it itself (they
// are few: P / 4096), the last block also publishes the grand total = the pair count
__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(const uint32_t* __restrict__ tt_rank,
                                                                const uint32_t* __restrict__ bsum,
                                                                uint32_t* __restrict__ offs, int P,
                                                                unsigned long long* __restrict__ count,
                                                                unsigned long long* host_count) {
    __shared__ uint32_t ws[4];
    __shared__ unsigned long long wb[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned long long before = 0;  // 64-bit: overflow of the 32-bit index space stays detectable in the count
    for (int b = tid; b < (int)blockIdx.x; b += SC_THREADS) before += bsum[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    if (lane == 0) wb[wid] = before;
    const int base = blockIdx.x * SCAN_ITEMS + tid * SC_PER_THREAD;  // consecutive items per thread
    uint32_t v[SC_PER_THREAD];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SC_PER_THREAD; k++) {
        v[k] = (base + k < P) ? tt_rank[base + k] : 0u;
        tsum += v[k];
    }
    uint32_t x = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) ws[wid] = x;
    __syncthreads();
    const unsigned long long block_base = (wb[0] + wb[1]) + (wb[2] + wb[3]);
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += ws[w];
    uint32_t excl = (uint32_t)block_base + woff + x - tsum;
#pragma unroll
    for (int k = 0; k < SC_PER_THREAD; k++) {
        if (base + k < P) offs[base + k] = excl;
        excl += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const unsigned long long total = block_base + ws[0] + ws[1] + ws[2] + ws[3];
        count[0] = total;
        // ... and straight into the caller's pinned host word, which the host is polling: the pair count
        // reaches the CPU a PCIe write after it exists instead of after a copy + stream-sync wake-up
        if (host_count) __hip_atomic_store(host_count, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_scan_tiles(const uint32_t* sorted_idx, const uint32_t* tiles, uint32_t* tt_rank, uint32_t* offs,
                      uint32_t* bsum, unsigned long long* count, unsigned long long* host_count, int P, int debug,
                      hipStream_t s) {
    const int nblk = (P + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk), dim3(SC_THREADS), 0, s, sorted_idx, tiles, tt_rank, bsum, P);
    GS_LAUNCH_CHECK("scan.reduce", debug, s);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk), dim3(SC_THREADS), 0, s, tt_rank, bsum, offs, P, count, host_count);
    GS_LAUNCH_CHECK("scan.apply", debug, s);
    return GS_OK;
}

// Emission of the (tile id, Gaussian index) pairs in depth-rank order, tiles y-outer / x-inner
// (the upstream duplicateWithKeys order).  Also records each Gaussian's first pair index in its
// splat record (slot 9): the backward pass addresses its per-pair gradient rows through it.
//
// Partitioned by OUTPUT, not by Gaussian: the nearest Gaussians come first in depth order and cover
// hundreds of tiles each, so a wave that owns 64 consecutive ranks can have 100x the pairs of
// another.  A workgroup owns EMIT_CHUNK consecutive pairs; emit_owner_kernel has recorded which rank
// owns the first pair of every chunk, so the group stages the (at most EMIT_CHUNK + 1) ranks that
// overlap its chunk in LDS and every pair finds its owner with a binary search over their offsets.
// Stores are fully coalesced.
__global__ __launch_bounds__(256) void emit_owner_kernel(const uint32_t* __restrict__ tt_rank,
                                                         const uint32_t* __restrict__ offs, int P, const PairCount pc,
                                                         uint32_t* __restrict__ owner, ZeroJob zero_a,
                                                         ZeroJob zero_b) {
    zero_job(zero_a);  // the tile sort's digit totals and the tile ranges (saves two fill launches)
    zero_job(zero_b);
    const uint32_t D = pair_count(pc);
    const uint32_t nchunks = (D + EMIT_CHUNK - 1) / EMIT_CHUNK;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P || D == 0) return;
    const uint32_t tt = tt_rank[r];
    if (!tt) return;
    const uint32_t off = offs[r], last = off + tt - 1;
    for (uint32_t c = (off + EMIT_CHUNK - 1) / EMIT_CHUNK; c <= last / EMIT_CHUNK; c++) owner[c] = (uint32_t)r;
    if (last == D - 1) owner[nchunks] = (uint32_t)r;  // the rank that owns the last pair
}

__global__ __launch_bounds__(256) void emit_kernel(const uint32_t* __restrict__ sorted_idx,
                                                   const uint32_t* __restrict__ offs,
                                                   const uint32_t* __restrict__ owner, float* __restrict__ rec,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                   const PairCount pc, int gx) {
    __shared__ uint32_t s_off[EMIT_CHUNK + 1], s_idx[EMIT_CHUNK + 1], s_rmin[EMIT_CHUNK + 1], s_rsz[EMIT_CHUNK + 1];
    const uint32_t D = pair_count(pc);
    const uint32_t nchunks = (D + EMIT_CHUNK - 1) / EMIT_CHUNK;
    const uint32_t c = blockIdx.x;
    if (c >= nchunks) return;  // the grid covers the capacity
    const uint32_t o0 = c * EMIT_CHUNK, o1 = min(D, o0 + EMIT_CHUNK);
    const uint32_t r0 = owner[c], r1 = owner[c + 1 < nchunks ? c + 1 : nchunks];
    // every rank in [r0, r1] has at least one pair (Gaussians without tiles sort behind all others), so
    // they are at most EMIT_CHUNK + 1
    const int cnt = (int)min(r1 - r0 + 1, (uint32_t)(EMIT_CHUNK + 1));
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const uint32_t off = offs[r0 + i], idx = sorted_idx[r0 + i];
        float* R = rec + (size_t)idx * REC_F;
        s_off[i] = off;
        s_idx[i] = idx;
        s_rmin[i] = __float_as_uint(R[10]);
        s_rsz[i] = __float_as_uint(R[11]);
        if (off >= o0) R[9] = __uint_as_float(off);  // the chunk in which the rank starts records its offset
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < EMIT_CHUNK / 256; k++) {
        const uint32_t o = o0 + k * 256 + threadIdx.x;
        if (o >= o1) break;
        // owner: the largest i with s_off[i] <= o
        int lo = 0, hi = cnt - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= o) lo = mid; else hi = mid - 1;
        }
        const uint32_t o_off = s_off[lo], o_rmin = s_rmin[lo], o_rsz = s_rsz[lo];
        const uint32_t w = o_rsz & 0xFFFFu;
        const uint32_t li = o - o_off;
        uint32_t y = (uint32_t)(((float)li + 0.5f) / (float)w);
        if (y * w > li) y--;
        if ((y + 1) * w <= li) y++;
        const uint32_t x = li - y * w;
        keys[o] = ((o_rmin >> 16) + y) * (uint32_t)gx + (o_rmin & 0xFFFFu) + x;
        vals[o] = s_idx[lo];
    }
}

int launch_emit(const uint32_t* sorted_idx, const uint32_t* tt_rank, const uint32_t* offs, float* rec, uint32_t* keys,
                uint32_t* vals, uint32_t* owner, ZeroJob zero_a, ZeroJob zero_b, int P, PairCount pc, int gx, int debug,
                hipStream_t s) {
    if (pc.cap == 0) return GS_OK;
    const uint32_t nchunks = (uint32_t)(((uint64_t)pc.cap + EMIT_CHUNK - 1) / EMIT_CHUNK);
    hipLaunchKernelGGL(emit_owner_kernel, dim3((P + 255) / 256), dim3(256), 0, s, tt_rank, offs, P, pc, owner, zero_a,
                       zero_b);
    GS_LAUNCH_CHECK("emit.owner", debug, s);
    hipLaunchKernelGGL(emit_kernel, dim3(nchunks), dim3(256), 0, s, sorted_idx, offs, owner, rec, keys, vals, pc, gx);
    GS_LAUNCH_CHECK("emit", debug, s);
    return GS_OK;
}

// ranges[tile] = [first, end) in the sorted list (upstream identifyTileRanges); ranges pre-zeroed.
__global__ __launch_bounds__(256) void ranges_kernel(const uint32_t* __restrict__ tile_sorted,
                                                     uint32_t* __restrict__ ranges, const PairCount pc) {
    const int64_t D = (int64_t)pair_count(pc);
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const uint32_t cur = tile_sorted[j];
    if (j == 0) {
        ranges[2 * cur] = 0;
    } else {
        const uint32_t prev = tile_sorted[j - 1];
        if (cur != prev) {
            ranges[2 * prev + 1] = (uint32_t)j;
            ranges[2 * cur] = (uint32_t)j;
        }
    }
    if (j == D - 1) ranges[2 * cur + 1] = (uint32_t)D;
}

int launch_ranges(const uint32_t* tile_sorted, uint32_t* ranges, PairCount pc, int ntiles, bool ranges_zeroed, int debug,
                  hipStream_t s) {
    if (!ranges_zeroed) {
        hipError_t e = hipMemsetAsync(ranges, 0, (size_t)ntiles * 8, s);
        if (e != hipSuccess) { gs_set_error((int)e, "ranges.memset"); return GS_E_HIP; }
    }
    if (pc.cap > 0) {
        hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)(((uint64_t)pc.cap + 255) / 256)), dim3(256), 0, s, tile_sorted,
                           ranges, pc);
        GS_LAUNCH_CHECK("ranges", debug, s);
    }
    return GS_OK;
}

// Launch order of the per-tile render waves: tiles sorted by DESCENDING work estimate (a counting
// sort into 1024 bins of work / max_work; ties in any order).  All tile waves of a frame are
// resident at once and the hardware deals workgroups breadth-first over the SIMDs, so handing out
// the tiles heaviest-first gives every SIMD one tile from each work quantile -- the kernel then
// ends with its SIMDs finishing together instead of on the few that drew several centre tiles.
// mode 0: work = list length (ranges), mode 1: work = sum of keys[4 tile .. 4 tile + 3] (the forward's
// per-quadrant last contributor).
__device__ __forceinline__ uint32_t tile_work(const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, int mode,
                                              int t) {
    return mode ? (keys[4 * t] + keys[4 * t + 1] + keys[4 * t + 2] + keys[4 * t + 3]) : (ranges[t].y - ranges[t].x);
}
// HELD = true: every thread keeps the work of its (up to 32) tiles in registers, so the inputs are loaded once, all
// loads in flight together (ntiles <= 32 * 1024); otherwise the three phases re-read them.
template <bool HELD>
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ keys, int mode, int ntiles,
                                                          uint32_t* __restrict__ order, const FillJob fill) {
    if (blockIdx.x > 0) {  // the side job (see FillJob); workgroup 0 does the ordering
        const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        for (size_t k = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; k < fill.quads; k += (size_t)(gridDim.x - 1) * 1024)
            fill.ptr[k] = ones;
        return;
    }
    constexpr int PER = 32;
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wmax[16];
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    hist[tid] = 0;
    uint32_t held[HELD ? PER : 1];
    uint32_t mx = 0;
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;  // workgroup-uniform
            const int t = i * 1024 + tid;
            held[i] = t < ntiles ? tile_work(ranges, keys, mode, t) : 0u;
            mx = max(mx, held[i]);
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024) mx = max(mx, tile_work(ranges, keys, mode, t));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 0) wmax[wid] = mx;
    __syncthreads();
    mx = 0;
    for (int w = 0; w < 16; w++) mx = max(mx, wmax[w]);
    const float scale = mx ? 1023.0f / (float)mx : 0.f;
    auto bin_of = [&](uint32_t w) { return 1023u - min(1023u, (uint32_t)((float)w * scale)); };
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;
            if (i * 1024 + tid < ntiles) atomicAdd(&hist[bin_of(held[i])], 1u);
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024) atomicAdd(&hist[bin_of(tile_work(ranges, keys, mode, t))], 1u);
    }
    __syncthreads();
    // exclusive scan of the 1024 bins
    const uint32_t v = hist[tid];
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    hist[tid] = woff + x - v;
    __syncthreads();
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;
            const int t = i * 1024 + tid;
            if (t < ntiles) order[atomicAdd(&hist[bin_of(held[i])], 1u)] = (uint32_t)t;
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024)
            order[atomicAdd(&hist[bin_of(tile_work(ranges, keys, mode, t))], 1u)] = (uint32_t)t;
    }
}

int launch_tile_order(const uint32_t* ranges, const uint32_t* keys, int mode, int ntiles, uint32_t* order, FillJob fill,
                      int debug, hipStream_t s) {
    // enough side workgroups to fill at HBM rate, no more than the job has 16 KB pieces
    const size_t pieces = (fill.quads + 1023) / 1024;
    const int side = fill.ptr ? (int)(pieces < 1024 ? pieces : 1024) : 0;
    if (ntiles <= 32 * 1024)
        hipLaunchKernelGGL(tile_order_kernel<true>, dim3(1 + side), dim3(1024), 0, s, reinterpret_cast<const uint2*>(ranges),
                           keys, mode, ntiles, order, fill);
    else
        hipLaunchKernelGGL(tile_order_kernel<false>, dim3(1 + side), dim3(1024), 0, s, reinterpret_cast<const uint2*>(ranges),
                           keys, mode, ntiles, order, fill);
    GS_LAUNCH_CHECK("tile_order", debug, s);
    return GS_OK;
}
