// binning.hip -- tile binning: numbering of the (tile, Gaussian) pairs, the per-tile
// depth-ordered Gaussian lists and the tile ranges.  Replaces upstream InclusiveSum / duplicateWithKeys /
// cub::DeviceRadixSort over 64-bit keys / identifyTileRanges.
//
// MI355X-first formulation of the reference's "stable sort of (tile << 32 | depth_bits) keys".  The Gaussians are
// first put in (depth, index) order (a stable 32-bit sort over P items, radix_sort.hip).  The list of tile t is then
// simply the sub-sequence of that ranking whose tile rectangles cover t -- no pair is ever sorted.  The tile grid is cut
// into BLOCKS of 64 x 4 tiles and the ranking into SEGMENTS; workgroup (block, segment) streams its segment, keeps the
// Gaussians whose rectangle meets its block (ballot compaction, rank order preserved) and, per batch of up to 1024 of
// them, builds in LDS one bit per (tile, Gaussian) pair: bit m of tile t's bitmap = "Gaussian m of the batch covers t".
// The position of a pair in its tile's list is then a popcount prefix.  Two passes of the same kernel: pass 1 stores
// the per-(segment, tile) pair counts; a small kernel turns them into prefixes over the segments and per-tile totals;
// one workgroup scans the totals into the tile ranges (and orders the tiles by work for the render launch); pass 2
// writes every pair's Gaussian index straight to its final slot.  Five launches and ~8 bytes moved per pair, where the
// pair sort took ten launches and four passes over 8-byte pairs; the result is bit for bit the list a stable sort of
// the upstream keys yields.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// Pair numbering.  Pair slots (the backward's gradient rows) are numbered Gaussian-major in INDEX order: Gaussian i owns
// [first_pair[i], first_pair[i] + tiles_touched[i]).  first_pair = exclusive prefix sum of tiles_touched, whose first
// level (one sum per 64-Gaussian wave) the preprocess kernel has already produced; the grand total is the frame's pair
// count (num_rendered), stored in the geom state and straight into the caller's pinned host word.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void first_pair_kernel(const uint32_t* __restrict__ tiles,
                                                         const uint32_t* __restrict__ wave_tiles, float* __restrict__ rec,
                                                         int P, unsigned long long* __restrict__ count,
                                                         unsigned long long* host_count) {
    __shared__ unsigned long long wb[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int w0 = 4 * (int)blockIdx.x;  // first preprocess wave of this workgroup's 256 Gaussians
    unsigned long long before = 0;       // 64-bit: overflow of the 32-bit index space stays detectable in the count
    for (int w = tid; w < w0; w += 256) before += wave_tiles[w];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    if (lane == 0) wb[wid] = before;
    const int i = blockIdx.x * 256 + tid;
    const uint32_t v = i < P ? tiles[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    const unsigned long long block_base = (wb[0] + wb[1]) + (wb[2] + wb[3]);
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wave_tiles[w0 + w];
    if (i < P) rec[(size_t)i * REC_F + 9] = __uint_as_float((uint32_t)block_base + woff + x - v);
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const unsigned long long total =
            block_base + wave_tiles[w0] + wave_tiles[w0 + 1] + wave_tiles[w0 + 2] + wave_tiles[w0 + 3];
        count[0] = total;
        count[2] = 0ull;  // (see ds_count_kernel)
        // ... and straight into the caller's pinned host word, which the host is polling: the pair count
        // reaches the CPU a PCIe write after it exists instead of after a copy + stream-sync wake-up
        if (host_count) __hip_atomic_store(host_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_first_pair(const uint32_t* tiles, const uint32_t* wave_tiles, float* rec, unsigned long long* count,
                      unsigned long long* host_count, int P, int debug, hipStream_t s) {
    hipLaunchKernelGGL(first_pair_kernel, dim3((P + 255) / 256), dim3(256), 0, s, tiles, wave_tiles, rec, P, count,
                       host_count);
    GS_LAUNCH_CHECK("first_pair", debug, s);
    return GS_OK;
}

// The Gaussians in depth-rank order with what binning needs of each: (index, rect min x | y << 16, rect size w | h << 16,
// tiles touched) -- 16 bytes streamed by the binning workgroups instead of a 48-byte record gathered per rank.
__global__ __launch_bounds__(256) void rank_list_kernel(const uint32_t* __restrict__ sorted_idx,
                                                        const float4* __restrict__ rec, const uint32_t* __restrict__ tiles,
                                                        uint4* __restrict__ ranklist, uint32_t* __restrict__ chunk_pairs,
                                                        int P) {
    __shared__ uint32_t ws[4];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tt = 0;
    if (r < P) {
        const uint32_t id = sorted_idx[r];
        const float4 c = rec[(size_t)id * 3 + 2];
        tt = tiles[id];
        ranklist[r] = make_uint4(id, __float_as_uint(c.z), __float_as_uint(c.w), tt);
    }
    // pairs of this chunk of 256 ranks: the binning workgroups cut the ranking into segments of equal WORK with it
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tt += __shfl_xor(tt, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = tt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_pairs[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

int launch_rank_list(const uint32_t* sorted_idx, const float* rec, const uint32_t* tiles, uint4* ranklist,
                     uint32_t* chunk_pairs, int P, int debug, hipStream_t s) {
    hipLaunchKernelGGL(rank_list_kernel, dim3((P + 255) / 256), dim3(256), 0, s, sorted_idx,
                       reinterpret_cast<const float4*>(rec), tiles, ranklist, chunk_pairs, P);
    GS_LAUNCH_CHECK("rank_list", debug, s);
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Tile lists.  Workgroup (block b of 64 x 4 tiles, segment g of the ranking), 1024 threads.
// ---------------------------------------------------------------------------------------------------------------
#define TC_THREADS 1024   // counting pass
#ifndef TBK_THREADS
#define TBK_THREADS 1024  // writing pass
#endif
#define TBK_WAVES (TBK_THREADS / 64)
#define TBK_BATCH 1024   // Gaussians per bitmap batch (32 words per tile)
#ifndef TBK_LPL
#define TBK_LPL 2        // rank-list loads per lane and filter trip
#endif
#define TBK_CHUNK (TBK_THREADS * TBK_LPL)  // ranks filtered per trip (64 * TBK_LPL per wave)
#define TBK_BUF (TBK_BATCH + TBK_CHUNK)    // compacted Gaussians waiting for a batch: < 1024 carried + <= TBK_CHUNK new

// a match's rectangle clipped to the block, in block-local tile coordinates: lx (6 bits) | ly (2) | w - 1 (6) | h - 1 (2)
__device__ __forceinline__ uint32_t pack_local_rect(int lx, int ly, int lw, int lh) {
    return (uint32_t)lx | ((uint32_t)ly << 6) | ((uint32_t)(lw - 1) << 8) | ((uint32_t)(lh - 1) << 14);
}

// Does rank-list entry e meet the tile block [bx0, bx1) x [by0, by1)?  If so *rc = its clipped rectangle (packed).
__device__ __forceinline__ bool block_hit(const uint4 e, int bx0, int by0, int bx1, int by1, uint32_t* rc) {
    const int x0 = (int)(e.y & 0xFFFFu), y0 = (int)(e.y >> 16);
    const int x1 = x0 + (int)(e.z & 0xFFFFu), y1 = y0 + (int)(e.z >> 16);
    const int cx0 = max(x0, bx0), cx1 = min(x1, bx1), cy0 = max(y0, by0), cy1 = min(y1, by1);
    const bool hit = e.w != 0u && cx1 > cx0 && cy1 > cy0;
    *rc = hit ? pack_local_rect(cx0 - bx0, cy0 - by0, cx1 - cx0, cy1 - cy0) : 0u;
    return hit;
}

// ---- counting pass.  How many Gaussians of rank segment g cover tile t?  A rectangle adds +1 / -1 at its four
// corners of a (rows + 1) x (gx + 1) grid in LDS and a 2-D prefix sum turns the corners into coverage counts: four LDS
// atomics per Gaussian instead of one per pair.  Workgroup (band of tile rows, segment); the band is the whole grid
// unless the grid is too large for LDS.
#define TC_CELLS 12288  // grid cells (4-byte) a workgroup holds
__global__ __launch_bounds__(TC_THREADS) void tile_count_kernel(const uint4* __restrict__ ranklist,
                                                                 const uint32_t* __restrict__ chunk_pairs,
                                                                 uint32_t* __restrict__ seg_start, int P, int gx,
                                                                 int gy, int band_rows, int nbands, int nseg, int ntiles,
                                                                 uint32_t* __restrict__ seg_cnt,
                                                                 uint32_t* __restrict__ tile_tot,
                                                                 uint32_t* __restrict__ xcc_mask) {
    __shared__ int grid[TC_CELLS];
    // (the chunk-parallel forward hands its work out per XCD: which XCDs this frame's launches run on)
    if (xcc_mask && threadIdx.x == 0) atomicOr(xcc_mask, 1u << xcc_id());
    __shared__ unsigned long long sb_scratch[16];
    __shared__ uint32_t s_seg[TB_MAX_SEG + 2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int band = (int)blockIdx.x % nbands, sg = (int)blockIdx.x / nbands;
    const int y0b = band * band_rows, y1b = min(gy, y0b + band_rows);
    const int rows = y1b - y0b, ld = gx + 1;
    const int cells = (rows + 1) * ld;
    for (int k = tid; k < cells; k += TC_THREADS) grid[k] = 0;
    // the rank segments (common.h: segment_starts), derived in LDS by every workgroup of this launch, stored by the first
    // for the writing pass
    segment_starts(chunk_pairs, P, nseg, s_seg, sb_scratch);
    __syncthreads();  // (also covers the zeroing above)
    const int r0 = min(P, (int)s_seg[sg] * 256), r1 = min(P, (int)s_seg[sg + 1] * 256);  // this segment's ranks
    if (blockIdx.x == 0 && tid <= nseg) seg_start[tid] = s_seg[tid];
    for (int rb = r0 + tid; rb < r1; rb += 4 * TC_THREADS) {
        uint4 e[4];  // four loads in flight per thread
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = rb + k * TC_THREADS;
            e[k] = r < r1 ? ranklist[r] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (e[k].w == 0u) continue;
            const int x0 = (int)(e[k].y & 0xFFFFu), x1 = x0 + (int)(e[k].z & 0xFFFFu);
            const int cy0 = max((int)(e[k].y >> 16), y0b) - y0b;
            const int cy1 = min((int)(e[k].y >> 16) + (int)(e[k].z >> 16), y1b) - y0b;
            if (cy1 <= cy0) continue;
            atomicAdd(&grid[cy0 * ld + x0], 1);
            atomicAdd(&grid[cy0 * ld + x1], -1);
            atomicAdd(&grid[cy1 * ld + x0], -1);
            atomicAdd(&grid[cy1 * ld + x1], 1);
        }
    }
    __syncthreads();
    // prefix down the columns (a wave per column, lanes = rows: one scan per 64 rows instead of a thread walking the
    // column cell after cell), then along the rows (a wave per row, 64 cells per trip)
    for (int x = wid; x < ld; x += (TC_THREADS / 64)) {
        int carry = 0;
        for (int y0 = 0; y0 < rows; y0 += 64) {
            const int y = y0 + lane;
            const int v = y < rows ? grid[y * ld + x] : 0;
            const int sc = (int)wave_scan_incl((uint32_t)v);
            if (y < rows) grid[y * ld + x] = carry + sc;
            carry += __builtin_amdgcn_readlane(sc, 63);
        }
    }
    __syncthreads();
    for (int y = wid; y < rows; y += (TC_THREADS / 64)) {
        int carry = 0;
        for (int x0 = 0; x0 < gx; x0 += 64) {
            const int x = x0 + lane;
            const int v = x < gx ? grid[y * ld + x] : 0;
            const int s = (int)wave_scan_incl((uint32_t)v);
            const uint32_t c = x < gx ? (uint32_t)(carry + s) : 0u;
            const int t0 = (y0b + y) * gx + x0;  // lane 0's tile (wave-uniform)
            if (x < gx) seg_cnt[(size_t)sg * ntiles + (size_t)(t0 + lane)] = c;
            // the tile's pairs over all segments: integer atomics into words an earlier kernel of the frame cleared (one
            // 256-byte atomic instruction per wave and trip; measured free next to the kernel's 12 us.  Sums per group of
            // tiles added the same way by one lane cost 6 us: 48 workgroups hitting the same word at the same moment)
            if (c) atomicAdd(&tile_tot[t0 + lane], c);
            carry += __builtin_amdgcn_readlane(s, 63);
        }
    }
}

// Launch order of the per-tile render waves: tiles sorted by DESCENDING work estimate (a counting
// sort into 1024 bins of work / max_work; ties in any order).  All tile waves of a frame are
// resident at once and the hardware deals workgroups breadth-first over the SIMDs, so handing out
// the tiles heaviest-first gives every SIMD one tile from each work quantile -- the kernel then
// ends with its SIMDs finishing together instead of on the few that drew several centre tiles.
// mode 0: work = list length (ranges), mode 1: work = sum of keys[4 tile .. 4 tile + 3] (the forward's
// per-quadrant last contributor), mode 2: work = keys[tile] = the tile's pair count (tile_count_kernel).
__device__ __forceinline__ uint32_t tile_work(const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, int mode,
                                              int t) {
    if (mode == 2) return keys[t];
    if (mode) {  // (one 16-byte load: the four quadrant counts of a tile are adjacent and the array is 256-byte aligned)
        const uint4 q = reinterpret_cast<const uint4*>(keys)[t];
        return (q.x + q.y) + (q.z + q.w);
    }
    return ranges[t].y - ranges[t].x;
}
// One workgroup of 1024 threads.  HELD = true: every thread keeps the work of its (up to 32) tiles in registers, so the
// inputs are loaded once, all loads in flight together (ntiles <= 32 * 1024); otherwise the phases re-read them.
// LDS: hist[1024], wmax[16], wsum[16].
// (The ordering without the chunk work list: the code every frame but a chunk-parallel one runs.  Kept apart from
// tile_order_body below -- with the work list's code in the same function the list-writing launch took 41.8 us instead of
// 34.5 at config 3, where not one line of it runs.)
template <bool HELD>
__device__ __forceinline__ void tile_order_plain(const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, int mode,
                                                int ntiles, uint32_t* __restrict__ order, const PairCount pc,
                                                const LongLists ll, uint32_t* hist, uint32_t* wmax, uint32_t* wsum) {
    constexpr int PER = 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    hist[tid] = 0;
    uint32_t held[HELD ? PER : 1];
    uint32_t mx = 0;
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {  // (the loads alone first: all of them in flight before the first is looked at)
            if (i * 1024 >= ntiles) break;  // workgroup-uniform
            const int t = i * 1024 + tid;
            held[i] = t < ntiles ? tile_work(ranges, keys, mode, t) : 0u;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;
            mx = max(mx, held[i]);
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024) mx = max(mx, tile_work(ranges, keys, mode, t));
    }
    mx = wave_max(mx);
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
    const uint32_t x = wave_scan_incl(v);
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    hist[tid] = woff + x - v;
    __syncthreads();
    // Small images (mark_wide, forward only): a tile whose list is long against the frame's total keeps a quadrant wave
    // busy for longer than the rest of the frame takes -- one wave walks n entries in ~60 n cycles, the whole frame is
    // ~0.08 cycles per pair on 1024 SIMDs -- and is rendered by four waves per quadrant instead (render_fwd.hip): bit 31.
    uint32_t wide_from = 0xFFFFFFFFu;
    if (mode == 2) {
        const unsigned long long D = *pc.dev;
        wide_from = max(FWD4_MIN_LIST, (uint32_t)min(D / FWD4_TOTAL_DIV, 0x7FFFFFFFull));
        if (ll.stats) {
            // how many such tiles, and the longest list (GsFwdArgs.frame_stats: diagnostics)
            uint32_t nlong = 0;
            if (HELD) {
#pragma unroll
                for (int i = 0; i < PER; i++) {
                    if (i * 1024 >= ntiles) break;
                    nlong += (i * 1024 + tid < ntiles && held[i] > wide_from) ? 1u : 0u;
                }
            } else {
                for (int t = tid; t < ntiles; t += 1024) nlong += tile_work(ranges, keys, mode, t) > wide_from ? 1u : 0u;
            }
            nlong = wave_sum(nlong);
            __syncthreads();  // (wsum is free again: the scan above has read it)
            if (lane == 0) wsum[wid] = nlong;
            __syncthreads();
            if (tid == 0) {
                uint32_t tot = 0;
                for (int w = 0; w < 16; w++) tot += wsum[w];
                __hip_atomic_store(&ll.stats[0], (long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&ll.stats[1], (long long)mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (!ll.mark) wide_from = 0xFFFFFFFFu;
    }
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;
            const int t = i * 1024 + tid;
            if (t < ntiles) order[atomicAdd(&hist[bin_of(held[i])], 1u)] = (uint32_t)t | (held[i] > wide_from ? 0x80000000u : 0u);
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024) {
            const uint32_t wk = tile_work(ranges, keys, mode, t);
            order[atomicAdd(&hist[bin_of(wk)], 1u)] = (uint32_t)t | (wk > wide_from ? 0x80000000u : 0u);
        }
    }
}

template <bool HELD>
__device__ __forceinline__ void tile_order_body(const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, int mode,
                                                int ntiles, uint32_t* __restrict__ order, const PairCount pc,
                                                const LongLists ll, uint32_t* hist, uint32_t* wmax, uint32_t* wsum) {
    constexpr int PER = 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    hist[tid] = 0;
    uint32_t held[HELD ? PER : 1];
    uint32_t mx = 0;
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {  // (the loads alone first: all of them in flight before the first is looked at)
            if (i * 1024 >= ntiles) break;  // workgroup-uniform
            const int t = i * 1024 + tid;
            held[i] = t < ntiles ? tile_work(ranges, keys, mode, t) : 0u;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;
            mx = max(mx, held[i]);
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024) mx = max(mx, tile_work(ranges, keys, mode, t));
    }
    mx = wave_max(mx);
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
    const uint32_t x = wave_scan_incl(v);
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    hist[tid] = woff + x - v;
    __syncthreads();
    // Small images (mark_wide, forward only): a tile whose list is long against the frame's total keeps a quadrant wave
    // busy for longer than the rest of the frame takes -- one wave walks n entries in ~60 n cycles, the whole frame is
    // ~0.08 cycles per pair on 1024 SIMDs -- and is rendered by four waves per quadrant instead (render_fwd.hip): bit 31.
    uint32_t wide_from = 0xFFFFFFFFu;
    uint32_t ch = 0;  // (mark == 2) entries per chunk of the chunk-parallel forward; 0: no tile is cut this frame
    uint32_t xmask = 0;
    if (mode == 2) {
        const unsigned long long D = *pc.dev;
        wide_from = max(FWD4_MIN_LIST, (uint32_t)min(D / (unsigned long long)ll.total_div, 0x7FFFFFFFull));
        if (ll.mark == 2 && D <= FWDC_SPARSE_PAIRS) wide_from = min(FWD4_MIN_LIST, ll.ch_min);  // (every tile of two chunks or more)
        if (ll.stats || ll.mark == 2) {
            // how many such tiles, and the longest list (GsFwdArgs.frame_stats: diagnostics; the chunk size below)
            uint32_t nlong = 0;
            if (HELD) {
#pragma unroll
                for (int i = 0; i < PER; i++) {
                    if (i * 1024 >= ntiles) break;
                    nlong += (i * 1024 + tid < ntiles && held[i] > wide_from) ? 1u : 0u;
                }
            } else {
                for (int t = tid; t < ntiles; t += 1024) nlong += tile_work(ranges, keys, mode, t) > wide_from ? 1u : 0u;
            }
            nlong = wave_sum(nlong);
            __syncthreads();  // (wsum is free again: the scan above has read it)
            if (lane == 0) wsum[wid] = nlong;
            __syncthreads();
            uint32_t tot = 0;
            for (int w = 0; w < 16; w++) tot += wsum[w];
            if (tid == 0 && ll.stats) {
                __hip_atomic_store(&ll.stats[0], (long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&ll.stats[1], (long long)mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            xmask = (ll.mark == 2 && ll.xcc_mask) ? (*ll.xcc_mask & 0xFFu) : 0u;
            if (xmask && tot > 0 && tot <= FWDC_MAX_UNITS / 2 && D <= (unsigned long long)pc.cap) {
                // The chunk size: sum over the marked tiles of ceil(n / ch) <= D / ch + (their number) -- the smallest
                // power-of-two multiple of FWDC_CH_MIN for which that bound fits the unit table.  A function of the
                // frame's counts alone (not of the order the threads arrive in): the cut, and with it the association of
                // every sum, is the same on every run.
                ch = ll.ch_min;
                while (D / ch + tot > (unsigned long long)FWDC_MAX_UNITS && ch < 0x40000000u) ch <<= 1;
            }
            __syncthreads();  // (wmax has been read by every thread long ago; wsum by all of them just now)
            if (tid == 0) wmax[0] = 0u;  // units handed out
            if (tid < 16) wsum[tid] = 0u;   // [0..7] chunks given to every XCD, [8..15] items given to it
            __syncthreads();
        }
        if (!ll.mark) wide_from = 0xFFFFFFFFu;
        if (ll.mark == 2 && ch == 0) wide_from = 0xFFFFFFFFu;
    }
    // the launch order; (mark == 2) only the tiles of at least two chunks keep their mark
    auto place = [&](const int t, const uint32_t wk) {
        const bool marked = wk > wide_from && (ch == 0u || wk > ch);
        order[atomicAdd(&hist[bin_of(wk)], 1u)] = (uint32_t)t | (marked ? 0x80000000u : 0u);
    };
    if (HELD) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (i * 1024 >= ntiles) break;
            const int t = i * 1024 + tid;
            if (t < ntiles) place(t, held[i]);
        }
    } else {
        for (int t = tid; t < ntiles; t += 1024) place(t, tile_work(ranges, keys, mode, t));
    }
    if (ch) {
        // (mark == 2) The chunk work list, in a pass of its own over the launch order just written (inside the unrolled
        // loop above its code was there 32 times).  A marked tile takes the next ceil(n / ch) units -- consecutive, chunk
        // c at unit u0 + c -- and 4 nch items of one XCD's list.  Two steps: every thread looks at one place of the order
        // and, for a marked tile, takes the units and the items (two atomics) and notes {tile, first unit, first item |
        // XCD << 28, chunks} in LDS; then every WAVE writes the units and items of one noted tile at a time, a lane each
        // (one thread per tile wrote up to 116 items one after the other: + 14 us on this launch at the avatar frame).
        // `hist` is the list-writing kernel's bitmap here (8448 words; the ordering is done with its first 1024).
        __syncthreads();
        uint32_t* __restrict__ stash = hist;       // [FWDC_MAX_TILES][4]
        uint32_t* __restrict__ nstash = &wmax[1];
        if (tid == 0) *nstash = 0u;
        __syncthreads();
        const uint32_t nx = (uint32_t)__popc(xmask);
        for (int pos = tid; pos < ntiles; pos += 1024) {
            const uint32_t ov = order[pos];
            if (!(ov >> 31)) continue;
            const int t = (int)(ov & 0x7FFFFFFFu);
            const uint32_t nch = (tile_work(ranges, keys, mode, t) + ch - 1u) / ch;
            const uint32_t u0 = atomicAdd(&wmax[0], nch);
            // the tile's waves all run on ONE XCD (they hand values to each other through its L2).  Which one: the
            // marked tiles are the first of the launch order, heaviest first, and are dealt out over the XCDs the
            // frame's launches run on in a boustrophedon (0 1 .. 7 7 .. 1 0): every XCD gets one tile of every size
            // class, so the chunks spread about evenly.  (A look at per-XCD load counters does not: the threads of
            // this workgroup place their tiles in the same instant and all see the same minimum -- measured, every
            // long tile of the avatar frame on one XCD; round-robin over the units: 384 to 872 items per XCD.)  Any
            // choice gives the same image.  The tile's 4 nch items are appended to that XCD's list in chunk order: a
            // wave only ever waits for items in front of its own in the list
            const uint32_t ph = (uint32_t)pos % (2u * nx);
            uint32_t pick = ph < nx ? ph : 2u * nx - 1u - ph, best = 0;
            for (uint32_t x = 0; x < 8; x++) {
                if ((xmask >> x) & 1u) {
                    if (pick == 0u) { best = x; break; }
                    pick--;
                }
            }
            const uint32_t i0 = atomicAdd(&wsum[8 + best], 4u * nch);
            const uint32_t k = atomicAdd(nstash, 1u);
            stash[4 * k] = (uint32_t)t;
            stash[4 * k + 1] = u0;
            stash[4 * k + 2] = i0 | (best << 28);
            stash[4 * k + 3] = nch;
        }
        __syncthreads();
        const uint32_t nk = *nstash;
        for (uint32_t k = (uint32_t)wid; k < nk; k += 16) {
            const uint32_t t = stash[4 * k], u0 = stash[4 * k + 1], ib = stash[4 * k + 2], nch = stash[4 * k + 3];
            uint32_t* __restrict__ items = ll.cw_items + (size_t)(ib >> 28) * (FWDC_MAX_UNITS * 4) + (ib & 0x0FFFFFFFu);
            for (uint32_t c = (uint32_t)lane; c < nch; c += 64) ll.cw_units[u0 + c] = make_uint2(t, c | (nch << 16));
            for (uint32_t i = (uint32_t)lane; i < 4u * nch; i += 64) items[i] = (u0 + (i >> 2)) * 4u + (i & 3u);
        }
    }
    if (ll.cw_hdr) {  // (workgroup-uniform) the work list's header: units in use, entries per chunk
        __syncthreads();
        if (tid == 0) {
            ll.cw_hdr[0] = (mode == 2 && ch) ? wmax[0] : 0u;
            ll.cw_hdr[1] = ch;
        }
        if (tid < 8) ll.cw_hdr[4 + tid] = (mode == 2 && ch) ? wsum[8 + tid] : 0u;
    }
}

// The ordering as a launch of its own (the backward: mode 1; an empty frame: mode 0).  Its other workgroups do a side
// job (see FillJob).
template <bool HELD>
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ keys, int mode, int ntiles,
                                                          uint32_t* __restrict__ order, const PairCount pc,
                                                          const FillJob fill, const LongLists ll) {
    if (blockIdx.x > 0) {  // the side job; workgroup 0 does the ordering
        if (fill.clean && *fill.clean == MARKS_CLEAN) return;  // (workgroup-uniform) the forward has done it
        const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        for (size_t k = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; k < fill.quads; k += (size_t)(gridDim.x - 1) * 1024)
            if (fill.stream) store_stream(&fill.ptr[k], ones); else fill.ptr[k] = ones;
        return;
    }
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wmax[16];
    __shared__ uint32_t wsum[16];
    if (mode < 0) return;  // (the fill only: the caller uses an order it already has)
    tile_order_body<HELD>(ranges, keys, mode, ntiles, order, pc, ll, hist, wmax, wsum);
}

// ---- writing pass.  Workgroup (block of 64 x 4 tiles, segment).  The kernel's arithmetic is trivial; what it is built
// around is the number of DEPENDENT steps (global-load latencies and workgroup barriers) a workgroup goes through: all
// first loads are issued together, the filter takes two trips of 4096 ranks with one barrier pair each, and a bitmap
// batch needs two barriers.
#define BM_LD 33          // words per tile in the bitmap (32 + 1: tile-major rows fall into different banks)
#ifndef TW_STOP_AFTER
#define TW_STOP_AFTER 99  // tools/tw_parts.hip builds the kernel with parts left out: 1 prologue, 2 + bitmap clear, 3 + filter
#endif
// CW: workgroup 0 also builds the chunk-parallel forward's work list (a kernel of its own: the other one stays what it was)
template <bool CW>
__global__ __launch_bounds__(TBK_THREADS) void tile_write_kernel(const uint4* __restrict__ ranklist,
                                                                 const uint32_t* __restrict__ seg_start, int P, int gx,
                                                                 int gy, int nbx, int nblocks, int nseg, int ntiles,
                                                                 const uint32_t* __restrict__ seg_cnt,
                                                                 const uint32_t* __restrict__ tile_tot,
                                                                 uint2* __restrict__ ranges, uint32_t* __restrict__ order,
                                                                 uint32_t* __restrict__ point_list, const PairCount pc,
                                                                 const LongLists ll) {
    __shared__ uint32_t m_id[TBK_BUF];
    __shared__ unsigned short m_rc[TBK_BUF];
    __shared__ uint32_t bitmap[TB_TILES * BM_LD];  // [local tile][word]: bit m = "Gaussian m of the batch covers the tile"
    __shared__ uint32_t dst[TB_TILES];             // next free slot of every tile's list for this workgroup
    __shared__ uint32_t wcnt[TBK_WAVES];
    __shared__ unsigned long long sb_scratch[20];
    static_assert(TBK_THREADS == 1024 && TBK_WAVES == 16 && TB_H <= 4 && TB_TILES * BM_LD >= 1024 && TB_TILES >= 32, "the ordering job borrows bitmap / dst");
    if (blockIdx.x == 0) {
        // Workgroup 0: the launch order of the render kernel's tile waves (heaviest first) from the tiles' pair counts --
        // nothing in this launch needs it, so it rides along instead of being a launch of one workgroup
        if (CW) {
            if (ntiles <= 32 * 1024) tile_order_body<true>(nullptr, tile_tot, 2, ntiles, order, pc, ll, bitmap, dst, dst + 16);
            else tile_order_body<false>(nullptr, tile_tot, 2, ntiles, order, pc, ll, bitmap, dst, dst + 16);
        } else if (ntiles <= 32 * 1024) {
            tile_order_plain<true>(nullptr, tile_tot, 2, ntiles, order, pc, ll, bitmap, dst, dst + 16);
        } else {
            tile_order_plain<false>(nullptr, tile_tot, 2, ntiles, order, pc, ll, bitmap, dst, dst + 16);
        }
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = ((int)blockIdx.x - 1) % nblocks, sg = ((int)blockIdx.x - 1) / nblocks;
    const int r0 = min(P, (int)seg_start[sg] * 256), r1 = min(P, (int)seg_start[sg + 1] * 256);  // this segment's ranks
    const int bx0 = (b % nbx) * TB_W, by0 = (b / nbx) * TB_H;
    const int bx1 = min(gx, bx0 + TB_W), by1 = min(gy, by0 + TB_H);
    const unsigned long long frame_pairs = *pc.dev;
    // If the frame's pair count does not fit the state the lists were carved for, every range is left empty: the render
    // that follows then draws an empty frame and touches nothing out of bounds (the host runs the phase again).
    const bool fits = frame_pairs <= (unsigned long long)pc.cap;
    // Tile ranges (upstream identifyTileRanges) and this workgroup's first slot in every list, from the counting pass:
    // first pair of tile t = pairs of all tiles before it (tile_tot) + the pairs the earlier segments put into t (seg_cnt).
    // All threads add up the tiles before the block's first row; wave r takes row r of the block -- 64 consecutive tiles,
    // whose prefix is one wave scan -- and the tiles from its row's start to the next row's.  All loads are independent:
    // one trip to the L2, one barrier.
    uint32_t my_dst = 0;
    {
        const int t_blk0 = by0 * gx + bx0;
        uint32_t part = 0;
        for (int t0 = 0; t0 < t_blk0; t0 += 4 * TBK_THREADS) {  // (four loads in flight)
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t0 + k * TBK_THREADS + tid < t_blk0 ? tile_tot[t0 + k * TBK_THREADS + tid] : 0u;
            part += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const int ty = by0 + wid, tx = bx0 + lane;
        const bool row_ok = tid < TB_TILES && ty < by1;  // (wave-uniform)
        const int t_row0 = ty * gx + bx0, t = t_row0 + lane;
        const bool tile_ok = row_ok && tx < bx1;
        uint32_t tot = 0, pre = 0, rowsum = 0;
        if (row_ok) {
            // (linear tile order: wraps into the next tile row; only rows that have a row behind them in the block are used)
            for (int k = lane; k < gx; k += 64) rowsum += t_row0 + k < ntiles ? tile_tot[t_row0 + k] : 0u;
            if (tile_ok) {
                tot = tile_tot[t];
                for (int s0 = 0; s0 < sg; s0 += 8) {
                    uint32_t v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = (s0 + k < sg) ? seg_cnt[(size_t)(s0 + k) * ntiles + t] : 0u;
#pragma unroll
                    for (int k = 0; k < 8; k++) pre += v[k];
                }
            }
        }
        part = wave_sum(part);
        rowsum = wave_sum(rowsum);
        uint32_t* s_part = reinterpret_cast<uint32_t*>(sb_scratch);  // 16 partial sums + 4 row sums
        if (lane == 0) {
            s_part[wid] = part;
            if (wid < TB_H) s_part[16 + wid] = rowsum;
        }
        __syncthreads();
        if (row_ok) {
            uint32_t before = 0;
            for (int w = 0; w < TBK_WAVES; w++) before += s_part[w];
            for (int r = 0; r < wid; r++) before += s_part[16 + r];
            const uint32_t x = wave_scan_incl(tot);
            const uint32_t first = before + x - tot;
            if (sg == 0 && tile_ok) ranges[t] = fits ? make_uint2(first, first + tot) : make_uint2(0u, 0u);
            my_dst = first + pre;
        }
        __syncthreads();
    }
    if (TW_STOP_AFTER <= 1) return;  // (tools/tw_parts.hip: where the kernel's time goes; 99 in the product)
    if (!fits || !point_list) return;  // (workgroup-uniform) the host sees the count and runs the phase again, larger state
    for (int k = tid; k < TB_TILES * BM_LD; k += TBK_THREADS) bitmap[k] = 0u;
    uint4 e[TBK_LPL];
    auto load_trip = [&](const int rb) {
#pragma unroll
        for (int k = 0; k < TBK_LPL; k++) {
            const int r = rb + wid * (64 * TBK_LPL) + k * 64 + lane;  // a wave owns 64 * TBK_LPL consecutive ranks of the trip
            e[k] = r < r1 ? ranklist[r] : make_uint4(0, 0, 0, 0);  // (tiles touched = 0: never a match)
        }
    };
    if (TW_STOP_AFTER <= 2) return;
    load_trip(r0);
    if (tid < TB_TILES) dst[tid] = my_dst;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int count = 0;  // compacted Gaussians in the buffer (workgroup-uniform)
    const int grp = lane >> 4, gl = lane & 15;

    // One bitmap batch over the first n buffered Gaussians (the bitmap is all zero on entry and on exit).
    auto process_batch = [&](const int n) {
        const int nw = (n + 31) >> 5;
        // Gaussian-major: a group of 16 lanes per Gaussian (four per wave and trip), lane j takes the tiles j, j + 16, ...
        // of its clipped rectangle (the average rectangle has ~12 tiles in a block)
        for (int m0 = wid * 4; m0 < n; m0 += TBK_WAVES * 4) {
            const int m = m0 + grp;
            if (m < n) {
                const uint32_t rc = m_rc[m];
                const uint32_t lw = ((rc >> 8) & 63u) + 1u, lh = ((rc >> 14) & 3u) + 1u;
                const uint32_t area = lw * lh, lx = rc & 63u, ly = (rc >> 6) & 3u;
                for (uint32_t j = (uint32_t)gl; j < area; j += 16u) {
                    const uint32_t dy = (j >= lw ? 1u : 0u) + (j >= 2u * lw ? 1u : 0u) + (j >= 3u * lw ? 1u : 0u);
                    const uint32_t lt = (ly + dy) * TB_W + lx + (j - dy * lw);
                    atomicOr(&bitmap[lt * BM_LD + (m >> 5)], 1u << (m & 31));
                }
            }
        }
        __syncthreads();
        // Tile-major: a group of 16 lanes per tile expands the tile's bitmap words into its list (lane j: words j and
        // j + 16; the words' offsets inside the tile's run are a 16-lane prefix sum of their popcounts).  The lanes of a
        // group write into one contiguous run, so a store instruction touches a few lines -- a Gaussian-major write-out
        // scatters 64 lanes over 64 lists (measured 6x slower on the whole kernel).
        for (int lt0 = wid * 4; lt0 < TB_TILES; lt0 += TBK_WAVES * 4) {
            const int lt = lt0 + grp;
            uint32_t bits0 = 0, bits1 = 0;
            if (gl < nw) { bits0 = bitmap[lt * BM_LD + gl]; bitmap[lt * BM_LD + gl] = 0u; }
            if (gl + 16 < nw) { bits1 = bitmap[lt * BM_LD + gl + 16]; bitmap[lt * BM_LD + gl + 16] = 0u; }
            const uint32_t c0 = (uint32_t)__popc(bits0), c1 = (uint32_t)__popc(bits1);
            // both prefix sums in one 16-lane DPP ladder (c0 | c1 << 16: a group's 16 words hold at most 512 bits), the
            // groups' totals by four lane reads -- no trip through the LDS unit's cross-lane path
            const uint32_t xs = row_scan_incl(c0 | (c1 << 16));
            const uint32_t x0 = xs & 0xFFFFu, x1 = xs >> 16;
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)xs, 15), e1 = (uint32_t)__builtin_amdgcn_readlane((int)xs, 31);
            const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)xs, 47), e3 = (uint32_t)__builtin_amdgcn_readlane((int)xs, 63);
            const uint32_t ends = grp == 0 ? e0 : (grp == 1 ? e1 : (grp == 2 ? e2 : e3));
            const uint32_t tot0 = ends & 0xFFFFu, tot1 = ends >> 16;
            const uint32_t d0 = dst[lt];
            uint32_t pos = d0 + x0 - c0;
            while (bits0) {
                const int bp = __ffs((int)bits0) - 1;
                bits0 &= bits0 - 1u;
                point_list[pos++] = m_id[(gl << 5) + bp];
            }
            pos = d0 + tot0 + x1 - c1;
            while (bits1) {
                const int bp = __ffs((int)bits1) - 1;
                bits1 &= bits1 - 1u;
                point_list[pos++] = m_id[((gl + 16) << 5) + bp];
            }
            if (gl == 0) dst[lt] = d0 + tot0 + tot1;  // (only this group touches tile lt)
        }
        __syncthreads();
    };

    // Filter: the wave's 256 ranks are compacted with four ballots; one barrier for the waves' counts, then every wave
    // writes its matches (rank order) at its offset.  The next trip's entries are loaded before this trip's are used.
    __syncthreads();
    for (int rb = r0; rb < r1 || count > 0;) {
        if (rb < r1) {
            uint32_t rc[TBK_LPL], id[TBK_LPL];
            unsigned long long bal[TBK_LPL];
            int c[TBK_LPL], csum = 0;
#pragma unroll
            for (int k = 0; k < TBK_LPL; k++) {
                const bool h = block_hit(e[k], bx0, by0, bx1, by1, &rc[k]);
                id[k] = e[k].x;
                bal[k] = __ballot(h);
                c[k] = __popcll(bal[k]);
                csum += c[k];
            }
            if (lane == 0) wcnt[wid] = (uint32_t)csum;
            if (rb + TBK_CHUNK < r1) load_trip(rb + TBK_CHUNK);
            __syncthreads();
            uint32_t woff = 0, tot = 0;
            for (int w = 0; w < TBK_WAVES; w++) {
                const uint32_t cw = wcnt[w];
                woff += w < wid ? cw : 0u;
                tot += cw;
            }
            int slot0 = count + (int)woff;
#pragma unroll
            for (int k = 0; k < TBK_LPL; k++) {
                if ((bal[k] >> lane) & 1ull) {
                    const int slot = slot0 + __popcll(bal[k] & lt_mask);
                    m_id[slot] = id[k];
                    m_rc[slot] = (unsigned short)rc[k];
                }
                slot0 += c[k];
            }
            count += (int)tot;
            rb += TBK_CHUNK;
            __syncthreads();
        }
        const bool last = rb >= r1;
        while (count >= TBK_BATCH || (last && count > 0)) {
            const int n = min(count, TBK_BATCH);
            if (TW_STOP_AFTER > 3) process_batch(n);
            const int rem = count - n;  // < TBK_CHUNK: at most TBK_LPL elements per thread move to the front
            uint32_t cid[TBK_LPL];
            unsigned short crc[TBK_LPL];
#pragma unroll
            for (int k = 0; k < TBK_LPL; k++) {
                const int i = tid + k * TBK_THREADS;
                cid[k] = i < rem ? m_id[n + i] : 0u;
                crc[k] = i < rem ? m_rc[n + i] : (unsigned short)0;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < TBK_LPL; k++) {
                const int i = tid + k * TBK_THREADS;
                if (i < rem) { m_id[i] = cid[k]; m_rc[i] = crc[k]; }
            }
            count = rem;
            __syncthreads();
        }
    }
}

int launch_tile_order(const uint32_t* ranges, const uint32_t* keys, int mode, int ntiles, uint32_t* order, PairCount pc,
                      FillJob fill, LongLists ll, int debug, hipStream_t s) {
    // side workgroups: enough to fill at HBM rate (4 MB in flight per sweep), no more than the job has 16 KB pieces -- and
    // few enough to cost nothing when they find the job done (FillJob::clean) and leave
    const size_t pieces = (fill.quads + 1023) / 1024;
    const int side = fill.ptr ? (int)(pieces < 255 ? pieces : 255) : 0;
    if (ntiles <= 32 * 1024)
        hipLaunchKernelGGL(tile_order_kernel<true>, dim3(1 + side), dim3(1024), 0, s, reinterpret_cast<const uint2*>(ranges),
                           keys, mode, ntiles, order, pc, fill, ll);
    else
        hipLaunchKernelGGL(tile_order_kernel<false>, dim3(1 + side), dim3(1024), 0, s, reinterpret_cast<const uint2*>(ranges),
                           keys, mode, ntiles, order, pc, fill, ll);
    GS_LAUNCH_CHECK("tile_order", debug, s);
    return GS_OK;
}

// The whole tile binning of one frame: counting pass (per-(segment, tile) counts, per-tile totals), then the writing pass,
// whose workgroups derive the tile ranges and their list slots from the counts themselves and whose first workgroup
// orders the tiles for the render launch.  `totals_zeroed`: tc.tile_tot .. (tc.zero_bytes) were cleared by an earlier
// kernel of this frame (the preprocess kernel); otherwise they are cleared here.
int launch_tile_lists(const uint4* ranklist, const uint32_t* chunk_pairs, uint32_t* seg_start, int P, int gx, int gy, TileCounts tc, uint32_t* ranges,
                      uint32_t* order, uint32_t* point_list, PairCount pc, LongLists ll, bool totals_zeroed, int debug,
                      hipStream_t s) {
    uint32_t* seg_cnt = tc.seg_cnt;
    const BinGrid G = bin_grid(gx, gy);
    const int nseg = bin_segments(G, P);  // segments of about equal work (segment_bounds)
    const int ntiles = gx * gy;
    if (!totals_zeroed) {
        const int zrc = gs_zero_async(tc.tile_tot, tc.zero_bytes, "tile_totals.zero", s);  // (a kernel node: graph-capturable)
        if (zrc != GS_OK) return zrc;
    }
    {
        // counting pass: bands of tile rows that fit the LDS grid (the whole tile grid up to ~110 x 110 tiles)
        int band_rows = TC_CELLS / (gx + 1) - 1;
        if (band_rows < 1) return GS_E_TOO_LARGE;  // (more than 12k tile columns)
        if (band_rows > gy) band_rows = gy;
        const int nbands = (gy + band_rows - 1) / band_rows;
        StageScope sc_("tile_count", s);
        hipLaunchKernelGGL(tile_count_kernel, dim3((unsigned)(nbands * nseg)), dim3(TC_THREADS), 0, s, ranklist, chunk_pairs, seg_start, P,
                           gx, gy, band_rows, nbands, nseg, ntiles, seg_cnt, tc.tile_tot, tc.xcc_mask);
        GS_LAUNCH_CHECK("tile_count", debug, s);
    }
    {
        StageScope sc_("tile_write", s);
        if (ll.mark == 2)
            hipLaunchKernelGGL(tile_write_kernel<true>, dim3((unsigned)(1 + G.nblocks * nseg)), dim3(TBK_THREADS), 0, s, ranklist,
                               seg_start, P, gx, gy, G.nbx, G.nblocks, nseg, ntiles, seg_cnt, tc.tile_tot,
                               reinterpret_cast<uint2*>(ranges), order, pc.cap > 0 ? point_list : nullptr, pc, ll);
        else
            hipLaunchKernelGGL(tile_write_kernel<false>, dim3((unsigned)(1 + G.nblocks * nseg)), dim3(TBK_THREADS), 0, s, ranklist,
                               seg_start, P, gx, gy, G.nbx, G.nblocks, nseg, ntiles, seg_cnt, tc.tile_tot,
                               reinterpret_cast<uint2*>(ranges), order, pc.cap > 0 ? point_list : nullptr, pc, ll);
        GS_LAUNCH_CHECK("tile_write", debug, s);
    }
    return GS_OK;
}
