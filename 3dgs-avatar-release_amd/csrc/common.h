// common.h -- shared declarations of the gfx950 rasterizer library (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gsplat_mi355.h"

#define TILE 16                 // tile edge in pixels (parity contract: 16x16)
#define WAVE 64                 // CDNA wavefront
#define REC_F 12                // floats per splat record (48 B = 3 x 16 B)
#define GS_MAX_PAIRS 0x3FFFFFFFll  // (tile, Gaussian) pairs per frame: 4 * pair + quadrant must fit 32 bits

// ---------------------------------------------------------------------------------------------
// State-buffer layouts.  Pure functions of (P) / (D, W, H) / (W, H): every call re-derives the
// same carve, so the library keeps no pointers between calls (re-entrant, graph-capturable).
// ---------------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define SORT_ITEMS 4096          // elements per radix-sort block (256 threads x 16)
#define SORT_SMALL_N (1 << 20)   // sorts up to this size use 1024-element blocks instead
#define SORT_TOTALS_REPL 16       // replicas of the per-pass digit totals (spreads the atomic adds)
// words of the radix sort's tables for n elements (either block size): [256][nblk] + 4 passes of totals
static inline size_t sort_table_words(size_t n) {
    const size_t items = n <= (size_t)SORT_SMALL_N ? 1024 : SORT_ITEMS;
    return (size_t)256 * ((n + items - 1) / items + 4 * SORT_TOTALS_REPL);
}
#define SCAN_ITEMS 2048          // elements per scan block (256 threads x 8)

#ifndef DS_ITEMS
#define DS_ITEMS 2048  // keys per workgroup in the counting / scattering passes of the bucket depth sort
#endif
#ifndef DS_PER_BUCKET
#define DS_PER_BUCKET 64  // (128: config 3 - 0.5 %, a thin-shell cloud -- depths bunched at two surfaces -- - 2.4 %)
#endif
// depth buckets for P Gaussians: ~64 per bucket for a uniform spread, a multiple of 64, 256 ... 4096
static inline int ds_buckets(int P) {
    int nb = (P / DS_PER_BUCKET + 63) / 64 * 64;
    return nb < 256 ? 256 : (nb > 4096 ? 4096 : nb);
}

#define TB_MAX_SEG 64  // rank segments of the tile-list launches, at most (binning.hip)
struct GeomLayout {
    size_t rec, depths, tiles, clamped, key0, key1, val0, val1, ranklist, chunk_pairs, wsum, wkmin, wkmax, hist, count,
        ds_tmp, ds_tmp2, ds_cnt, ds_pre, ds_tot, ds_loc, ds_grp, ds_range, seg_start, total;
    int nblk_sort, nwaves, ds_nb, ds_blocks;
};
static inline GeomLayout geom_layout(int P) {
    GeomLayout L;
    size_t o = 0;
    size_t n = (size_t)(P > 0 ? P : 1);
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.nblk_sort = (int)((n + SORT_ITEMS - 1) / SORT_ITEMS);
    L.nwaves = (int)((n + 255) / 256) * 4;  // waves of the preprocess launch (256-thread workgroups)
    L.rec = take(n * REC_F * 4);
    L.depths = take(n * 4);
    L.tiles = take(n * 4);
    L.clamped = take(n * 4);
    L.key0 = take(n * 4);
    L.key1 = take(n * 4);
    L.val0 = take(n * 4);
    L.val1 = take(n * 4);
    L.ranklist = take(n * 16);               // (index, rect min, rect size, tiles touched) in depth-rank order
    L.chunk_pairs = take(((n + 255) / 256) * 4);  // tiles touched per chunk of 256 consecutive ranks
    L.wsum = take((size_t)L.nwaves * 4);     // tiles touched per preprocess wave
    L.wkmin = take((size_t)L.nwaves * 4);    // smallest / largest depth key of the wave's Gaussians that touch a tile
    L.wkmax = take((size_t)L.nwaves * 4);
    L.hist = take(sort_table_words(n) * 4);
    L.count = take(64);
    // bucket depth sort (depth_sort.hip): DS_NB(P) depth buckets + one for the Gaussians that touch no tile
    L.ds_nb = ds_buckets(P);
    L.ds_blocks = (int)((n + DS_ITEMS - 1) / DS_ITEMS);
    L.ds_tmp = take(n * 8);                                        // (key << 32 | index), bucket after bucket
    L.ds_tmp2 = take(n * 8);                                       // ... and sub-bucket after sub-bucket (a bucket beyond 4096)
    L.ds_cnt = take((size_t)L.ds_blocks * (L.ds_nb + 1) * 4);      // [block][bucket] counts ...
    L.ds_pre = take((size_t)L.ds_blocks * (L.ds_nb + 1) * 4);      // ... and their exclusive prefix over the blocks
    L.ds_tot = take((size_t)(L.ds_nb + 1) * 4);
    L.ds_loc = take((size_t)(L.ds_nb + 1) * 4);
    L.ds_grp = take((size_t)((L.ds_nb + 1 + 63) / 64) * 4);
    L.ds_range = take(16);
    L.seg_start = take((size_t)(TB_MAX_SEG + 2) * 4);              // first chunk of 256 ranks of every rank segment (binning.hip)
    L.total = o;
    return L;
}

struct BinLayout {
    size_t point_list, qlist, marks, marks_flag, total;
};
static inline BinLayout bin_layout(int64_t D) {
    BinLayout L;
    size_t o = 0;
    size_t n = (size_t)(D > 0 ? D : 1);
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.point_list = take(n * 4);  // Gaussian indices, tile-major, (depth, index) order inside a tile
    L.qlist = take(n * 16);      // per quadrant: compacted Gaussian indices the forward visited
    // The backward's gradient-row marks (one word per row, four rows per pair; render_bwd.hip): ROW_UNWRITTEN in every word
    // is what the backward starts from.  They live HERE, not in the backward's scratch, so that the forward can write them
    // in the shadow of its VALU-bound render kernel; marks_flag says whether they still are as the forward left them.
    L.marks = take(n * 16);
    L.marks_flag = take(4);
    L.total = o;
    return L;
}

// Tile binning works on BLOCKS of TB_W x TB_H tiles (256 tiles: one workgroup holds a bitmap of them in LDS) and on
// SEGMENTS of the depth-ranked Gaussian list; workgroup (block, segment) counts / writes the pairs of its segment's
// Gaussians with its block's tiles (binning.hip).
#define TB_W 64
#ifndef TB_H
#define TB_H 4   // 1, 2 or 4 (the clipped rectangle's rows are packed in 2 bits)
#endif
#define TB_TILES (TB_W * TB_H)
struct BinGrid { int nbx, nby, nblocks, nseg_max; };
static inline BinGrid bin_grid(int gx, int gy) {
    BinGrid G;
    G.nbx = (gx + TB_W - 1) / TB_W;
    G.nby = (gy + TB_H - 1) / TB_H;
    G.nblocks = G.nbx * G.nby;
#ifndef TB_TARGET_WGS
#define TB_TARGET_WGS 512  // about two workgroups per CU over the chip
#endif
    int s = (TB_TARGET_WGS + G.nblocks - 1) / G.nblocks;
    G.nseg_max = s < 1 ? 1 : (s > TB_MAX_SEG ? TB_MAX_SEG : s);
    return G;
}
// segments actually used for P Gaussians: at least 1024 of them per segment
static inline int bin_segments(const BinGrid& G, int P) {
    int s = (P + 1023) / 1024;
    if (s < 1) s = 1;
    return s < G.nseg_max ? s : G.nseg_max;
}

// process-wide tuning switches (gs_tuning)
enum { GS_TUNE_XCD_MAP = 0, GS_TUNE_DEPTH_SORT = 1, GS_TUNE_NT_STORES = 2, GS_TUNE_BWD_CHUNKS = 3, GS_TUNE_FWD4 = 4, GS_TUNE_SMALL_TILES = 5, GS_TUNE_SHARED_QLIST = 6, GS_TUNE_ONES_FAST = 7, GS_TUNE_BWD_ORDER = 8, GS_TUNE_FWD_MARKS = 9, GS_TUNE_FWDC_CH = 10, GS_TUNE_FWDC_DIV = 11, GS_TUNE_COUNT = 12 };
int gs_tune_get(int key);

// Backward in chunks (small images): a frame of few, long lists -- a trained avatar at 512 x 512 has ~150 tiles of 2000-7000
// entries -- leaves most SIMDs idle while a handful of waves walk their quadrant's list alone, one dependent step after
// the other.  The forward therefore checkpoints the per-pixel compositing state (T, C) at the first batch boundary at
// least BWD_CH compacted entries after the previous checkpoint, notes the compacted index it stands at (ck_start), and
// the backward runs one wave per (quadrant, chunk), each starting from its checkpoint.  Only for images of up to
// BWD_CHUNK_MAX_TILES tiles: a larger frame fills the chip with one wave per quadrant.
#define BWD_CH 128
#define BWD_KMAX 16  // chunks per quadrant; the last one takes whatever lies beyond the last checkpoint
#define BWD_CHUNK_MAX_TILES 2048
#define FWD4_MAX_TILES 2048   // the forward runs four waves per quadrant up to this many tiles (render_fwd.hip)
#define FWD4_MIN_LIST 512u    // ... all four on the tiles whose list is longer than this and than (frame's pairs) / FWD4_TOTAL_DIV
#ifndef FWD4_TOTAL_DIV
#define FWD4_TOTAL_DIV 320ull  // (160: avatar frame - 1.5 %; 640: configs 2 and 4 - 13 % / - 5 %; 1280: - 18 % / - 9 %)
#endif

// Chunk-parallel forward of the long lists (round 4, render_fwd.hip: render_chunk).  The tile_order job cuts the list of
// every marked tile into CHUNKS of `ch` entries (FWDC_CH_MIN, doubled until the frame's chunks fit FWDC_MAX_UNITS) and
// numbers the (tile, chunk) UNITS; the render launch runs one wave per (unit, quadrant) in front of its tile waves.
#define FWDC_CH_MIN 256u
// A frame of at most this many pairs leaves most of the chip idle whatever is done (a trained avatar at 512 x 512: 0.35 M):
// every tile of more than two chunks is cut.  Above it only the tiles FWD4_TOTAL_DIV marks -- where the chip is busy
// anyway the two passes of a cut tile are work added, not latency removed (configs 2 and 4 with every tile cut: render
// launch 62 -> 112 us, 87 -> 190 us)
#define FWDC_SPARSE_PAIRS 524288ull
#define FWDC_MAX_UNITS 4096
#define FWDC_MAX_TILES (FWDC_MAX_UNITS / 2)  // marked tiles per frame (each has at least two chunks)
#ifdef FWDC_PROF
#define FWDC_SLOTS 9            // (+ a slot of time stamps: tools/fwdc_prof.py)
#else
#define FWDC_SLOTS 8
#endif
                                // per (unit, quadrant): 8 x 64 floats -- P, T at the chunk's start, colour share (3), T after the last blend, last contributor (compacted + 1, position in the tile's list)
#define FWDC_DEAD 0x80000000u   // in a flag word: every pixel of the quadrant was done before the chunk
#define FWDC_SPIN_MAX (1 << 16) // polls of a predecessor's flag before the wave gives up (it never should: see render_chunk)
struct ImgLayout {
    size_t ranges, n_contrib, final_T, ncon_c, tile_nmax, order, seg_cnt, tile_tot, all_ones, cw_flag, cw_done, cw_q, tile_zero_bytes, l1_part, ckpt, ck_start, cw_hdr, cw_units, cw_items, cw_rec, total;
    int gx, gy, bwd_chunks;
};
// `long_lists`: GsFwdArgs.long_lists (the few-long-lists machinery on an image of any size)
static inline bool few_long_lists_mode(int ntiles, int long_lists) {
    return ntiles <= gs_tune_get(GS_TUNE_SMALL_TILES) || long_lists != 0;
}
static inline ImgLayout img_layout(int W, int H, int long_lists = 0) {
    ImgLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.gx = (W + TILE - 1) / TILE;
    L.gy = (H + TILE - 1) / TILE;
    const size_t nt = (size_t)L.gx * L.gy;
    L.ranges = take(nt * 8);
    L.n_contrib = take((size_t)W * H * 4);
    L.final_T = take((size_t)W * H * 4);
    L.ncon_c = take((size_t)W * H * 4);            // per pixel: last contributor in its quadrant's COMPACTED list
    L.tile_nmax = take(nt * 16);                   // per quadrant: compacted entries up to the last contributor
    L.order = take(nt * 4);
    L.seg_cnt = take(nt * 4 * (size_t)bin_grid(L.gx, L.gy).nseg_max);  // [segment][tile] pair counts
    L.tile_tot = take(nt * 4);                     // pairs per tile
    // 1: this image is a second render of another call's geometry whose colours are all (1, 1, 1), written as 1 - T
    // (second_ones_kernel); 0 for every other image (cleared with the totals by the preprocess kernel, or by
    // recolor_kernel): the one-pass backward of both images then needs no colours of the second image at all
    L.all_ones = take(4);
    L.tile_zero_bytes = L.all_ones + 4 - L.tile_tot;
    const bool fl = few_long_lists_mode((int)nt, long_lists);
    const bool cw = fl && gs_tune_get(GS_TUNE_FWD4) == 2;  // (the chunk-parallel forward's state: only where it is switched on)
    L.cw_flag = L.cw_done = L.cw_q = 0;
    if (cw) {
        // the chunk-parallel forward's hand-off words, cleared with the totals: per (unit, quadrant) "hits + 1" once the
        // chunk's transmittance product is published; per quadrant the number of its chunks that are finished
        L.cw_flag = take((size_t)FWDC_MAX_UNITS * 4 * 4);
        L.cw_done = take(nt * 16);
        // [XCD][item of its list]: 1 once a wave has taken the item; behind them one word: the XCDs this frame's launches
        // run on (a bit each)
        L.cw_q = take((size_t)8 * FWDC_MAX_UNITS * 4 * 4 + 64);
        L.tile_zero_bytes = L.cw_q + (size_t)8 * FWDC_MAX_UNITS * 4 * 4 + 64 - L.tile_tot;
    }
    L.l1_part = take(nt * 16);  // per quadrant: sum |out_color - l1_target| over its pixels (GsFwdArgs.l1_target)
    L.bwd_chunks = fl ? BWD_KMAX : 1;
    // [quadrant][chunk 1 .. bwd_chunks - 1][64 pixels] (T, C0, C1, C2) before the chunk's first entry
    L.ckpt = take(nt * 4 * (size_t)(L.bwd_chunks - 1) * 64 * 16);
    L.ck_start = take(nt * 4 * (size_t)L.bwd_chunks * 4);  // [quadrant][chunk]: first compacted entry of the chunk, ~0 = none
    L.cw_hdr = L.cw_units = L.cw_items = L.cw_rec = 0;
    if (cw) {
        L.cw_hdr = take(64);                                   // [0] units in use, [1] entries per chunk, [4..11] items given to every XCD
        L.cw_units = take((size_t)FWDC_MAX_UNITS * 8);         // {tile, chunk | chunks of the tile << 16}
        L.cw_items = take((size_t)8 * FWDC_MAX_UNITS * 4 * 4); // [XCD][item]: unit * 4 + quadrant, a tile's items in chunk order
        L.cw_rec = take((size_t)FWDC_MAX_UNITS * 4 * FWDC_SLOTS * 64 * 4);
    }
    L.total = o;
    return L;
}

static inline int radix_passes(int bits) { return (bits + 7) / 8; }

// ---------------------------------------------------------------------------------------------
// Error plumbing
// ---------------------------------------------------------------------------------------------
void gs_set_error(int hip_err, const char* stage);
#define GS_LAUNCH_CHECK(stage, dbg, stream)                                            \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ == hipSuccess && (dbg)) e__ = hipStreamSynchronize((hipStream_t)(stream)); \
        if (e__ != hipSuccess) { gs_set_error((int)e__, stage); return GS_E_HIP; }     \
    } while (0)

// `bytes` (a multiple of 4) of zeros written by a kernel (no memset node: see "stream capture" in capi.hip)
int gs_zero_async(void* ptr, size_t bytes, const char* stage, hipStream_t s);
// Per-stage hipEvent timing (gs_profile_enable / gs_profile_collect); a no-op unless enabled.
void gs_prof_begin(const char* stage, hipStream_t s);
void gs_prof_end(hipStream_t s);
// a word range some kernel clears on the side (grid-stride), saving a fill launch
struct ZeroJob { uint32_t* ptr; int words; };
// Inclusive prefix sum over the 64 lanes of a wave by DPP row shifts and row broadcasts (six dependent VALU instructions;
// a __shfl_up ladder is six ds_bpermute round trips through the LDS unit, ~10 x the latency -- and the short binning
// kernels are nothing but latency).  Every lane must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += dpp_or_zero<0x111, 0xF>(x);  // row_shr:1
    x += dpp_or_zero<0x112, 0xF>(x);  // row_shr:2
    x += dpp_or_zero<0x114, 0xF>(x);  // row_shr:4
    x += dpp_or_zero<0x118, 0xF>(x);  // row_shr:8
    x += dpp_or_zero<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3
    x += dpp_or_zero<0x143, 0xC>(x);  // row_bcast:31 into rows 2 and 3
    return x;
}
// the same inside every row of 16 lanes
__device__ __forceinline__ uint32_t row_scan_incl(uint32_t x) {
    x += dpp_or_zero<0x111, 0xF>(x);
    x += dpp_or_zero<0x112, 0xF>(x);
    x += dpp_or_zero<0x114, 0xF>(x);
    x += dpp_or_zero<0x118, 0xF>(x);
    return x;
}
__device__ __forceinline__ unsigned long long wave_scan_incl(unsigned long long x) {
    // (two 32-bit ladders would lose the carries: shift the halves, add as 64-bit)
#define GS_SCAN64_STEP(CTRL, MASK)                                                                           \
    x += (unsigned long long)dpp_or_zero<CTRL, MASK>((uint32_t)x) |                                          \
         ((unsigned long long)dpp_or_zero<CTRL, MASK>((uint32_t)(x >> 32)) << 32);
    GS_SCAN64_STEP(0x111, 0xF) GS_SCAN64_STEP(0x112, 0xF) GS_SCAN64_STEP(0x114, 0xF) GS_SCAN64_STEP(0x118, 0xF)
    GS_SCAN64_STEP(0x142, 0xA) GS_SCAN64_STEP(0x143, 0xC)
#undef GS_SCAN64_STEP
    return x;
}

// Wave-wide sum / max / min by the same ladder (the result is wave-uniform: lane 63 of the inclusive scan).
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(x), 63); }
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long x) {
    x = wave_scan_incl(x);
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
    x = max(x, dpp_or_zero<0x111, 0xF>(x));
    x = max(x, dpp_or_zero<0x112, 0xF>(x));
    x = max(x, dpp_or_zero<0x114, 0xF>(x));
    x = max(x, dpp_or_zero<0x118, 0xF>(x));
    x = max(x, dpp_or_zero<0x142, 0xA>(x));
    x = max(x, dpp_or_zero<0x143, 0xC>(x));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_min(uint32_t x) { return ~wave_max(~x); }
// (float: the sum is taken in the ladder's fixed order -- the same bits on every run)
__device__ __forceinline__ float wave_sum(float x) {
#define GS_FSUM_STEP(CTRL, MASK) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, MASK, 0xF, false));
    GS_FSUM_STEP(0x111, 0xF) GS_FSUM_STEP(0x112, 0xF) GS_FSUM_STEP(0x114, 0xF) GS_FSUM_STEP(0x118, 0xF)
    GS_FSUM_STEP(0x142, 0xA) GS_FSUM_STEP(0x143, 0xC)
#undef GS_FSUM_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// ---- rank segments (binning.hip).  The ranking is cut into `nseg` SEGMENTS of about equal work, not of equal length: the
// nearest Gaussians come first and cover hundreds of tiles each, so equal-length segments leave the workgroups of the
// first segments with ten times the pairs of the others (they then ARE the kernel's run time).  Work of a chunk of 256
// ranks = its pairs + SEG_RANK_W (the cost of filtering 256 ranks, in pairs); chunk c belongs to segment
// floor(before * scale / 2^32), scale = nseg 2^32 / total work -- non-decreasing in the work `before` in front of the
// chunk, so the segments are contiguous runs of chunks (possibly empty).  segment_starts stores the first chunk of every
// segment: seg_start[0 .. nseg], seg_start[nseg] = the number of chunks.  Every workgroup of the COUNTING pass derives
// them in LDS (one load of the chunk sums, a workgroup-wide scan) and its first one also stores them; the workgroups of
// the WRITING pass -- sixteen times as many, each spending 5 us on the same derivation until round 3 -- read two words.
// What matters for the lists is only that both passes cut the ranking at the same places.
#ifndef SEG_RANK_W
#define SEG_RANK_W 4096u
#endif
__device__ __forceinline__ int segment_of(unsigned long long before, unsigned long long scale, int nseg) {
    const unsigned long long hi = __umul64hi(before, scale), lo = before * scale;
    return (int)min((unsigned long long)(nseg - 1), (hi << 32) | (lo >> 32));
}
__device__ __forceinline__ void segment_starts(const uint32_t* __restrict__ chunk_pairs, int P, int nseg, uint32_t* seg_start /* LDS */,
                                               unsigned long long* scratch /* LDS: 16 words */) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nthreads = (int)blockDim.x, nwaves = nthreads >> 6;
    const int nchunks = (P + 255) / 256;
    const int per = (nchunks + nthreads - 1) / nthreads;  // consecutive chunks per thread
    const int c0 = tid * per;
    // (all of a thread's loads at once: four in flight, kept when they are all it has -- P <= 1024 * the workgroup's threads)
    auto work = [&](int c) { return chunk_pairs[c] + SEG_RANK_W; };
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    const uint32_t w_prev = (c0 > 0 && c0 < nchunks) ? work(c0 - 1) : 0u;
    unsigned long long mine = 0;
    for (int k0 = 0; k0 < per; k0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = (k0 + k < per && c0 + k0 + k < nchunks) ? work(c0 + k0 + k) : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) mine += w[k];
    }
    const unsigned long long x = wave_scan_incl(mine);
    __syncthreads();  // (scratch may be in use until here)
    if (lane == 63) scratch[wid] = x;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
    for (int wv = 0; wv < nwaves; wv++) {
        const unsigned long long c = scratch[wv];
        woff += wv < wid ? c : 0ull;
        total += c;
    }
    const unsigned long long scale = (unsigned long long)((double)nseg * 4294967296.0 / (double)total);  // (total >= SEG_RANK_W: P > 0)
    unsigned long long before = woff + x - mine;  // work before this thread's first chunk
    int prev_seg = -1;                            // segment of the chunk before it
    if (c0 > 0 && c0 < nchunks) prev_seg = segment_of(before - w_prev, scale, nseg);
    for (int k0 = 0; k0 < per; k0 += 4) {
        if (per > 4) {  // (workgroup-uniform; otherwise the four words are still there)
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = (k0 + k < per && c0 + k0 + k < nchunks) ? work(c0 + k0 + k) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = c0 + k0 + k;
            if (k0 + k < per && c < nchunks) {
                const int sgc = segment_of(before, scale, nseg);
                for (int sg = prev_seg + 1; sg <= sgc; sg++) seg_start[sg] = (uint32_t)c;  // first chunk of segment sg (or of the next non-empty one)
                prev_seg = sgc;
                before += w[k];
                if (c == nchunks - 1)
                    for (int sg = prev_seg + 1; sg <= nseg; sg++) seg_start[sg] = (uint32_t)nchunks;
            }
        }
    }
}

// the XCD (accelerator complex die: its own L2) this wave runs on
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ void zero_job(const ZeroJob z) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < z.words; k += gridDim.x * blockDim.x) z.ptr[k] = 0u;
}

// The frame's pair count (num_rendered) as the binning kernels see it: read from the geom state on the DEVICE, so that
// they can be launched -- with grids sized by the CAPACITY the binning state was carved for -- before the host knows
// the count.  A count beyond the capacity reads as 0: every kernel then does the work of an empty frame and touches
// nothing out of bounds; the host sees the real count in its pinned word and runs the phase again with a larger state.
struct PairCount { const unsigned long long* dev; uint32_t cap; };
__device__ __forceinline__ uint32_t pair_count(const PairCount pc) {
    const unsigned long long d = *pc.dev;
    return d <= (unsigned long long)pc.cap ? (uint32_t)d : 0u;
}

// Process-wide tuning switches (gs_tuning; experiments and A/B runs, not part of the drop-in surface).

// Workgroup -> (tile slot, quadrant) of the render kernels.  Workgroups are dealt round-robin over the 8 XCDs (each
// with its own L2), so with the plain mapping (slot = b / 4, quadrant = b % 4) the four quadrant waves of one tile land
// on four different XCDs and each of them fetches the tile's splat records into its own L2.  xmap = 1: the 32
// workgroups b = 32 g .. 32 g + 31 are 8 tiles x 4 quadrants with slot = 8 g + b % 8 and quadrant = (b / 8) % 4 -- the
// four quadrants of a tile share b % 8, i.e. an XCD (speed only: nothing depends on the placement).
__device__ __forceinline__ void render_block_map(int b, int xmap, int* slot, int* q) {
    if (xmap) {
        *slot = ((b >> 5) << 3) + (b & 7);
        *q = (b >> 3) & 3;
    } else {
        *slot = b >> 2;
        *q = b & 3;
    }
}
// the forward runs four waves per quadrant, and tile_order_kernel marks the tiles that use them all (render_fwd.hip)
static inline bool forward_small_image(int ntiles, int long_lists) { return few_long_lists_mode(ntiles, long_lists) && gs_tune_get(GS_TUNE_FWD4) == 1; }
static inline bool forward_chunked(int ntiles, int long_lists) { return few_long_lists_mode(ntiles, long_lists) && gs_tune_get(GS_TUNE_FWD4) == 2; }
static inline int render_grid_blocks(int ntiles, int xmap) { return xmap ? ((ntiles + 7) / 8) * 32 : ntiles * 4; }

struct StageScope {
    hipStream_t s;
    StageScope(const char* stage, hipStream_t st) : s(st) { gs_prof_begin(stage, st); }
    ~StageScope() { gs_prof_end(s); }
};

// ---------------------------------------------------------------------------------------------
// Stage launchers (each enqueues on `s`, returns GS_OK / GS_E_HIP)
// ---------------------------------------------------------------------------------------------
int launch_preprocess(const GsFwdArgs& a, float* rec, float* depths, uint32_t* tiles, uint32_t* clamped,
                      uint32_t* sort_keys, uint32_t* sort_vals, int32_t* radii, uint32_t* wave_tiles, uint32_t* wave_kmin,
                      uint32_t* wave_kmax, ZeroJob zero, ZeroJob zero2, hipStream_t s);
// bucket depth sort (depth_sort.hip): sorted_idx = the Gaussian indices in ascending (depth key, index) order
struct DepthSortState { unsigned long long *tmp, *tmp2; uint32_t *cnt, *pre, *tot, *loc, *grp, *range; int nb, blocks; };
// pair numbering done by the sort's first launch (see first_pair_kernel, which the radix path uses) and the rank list
// written by its last ones (see rank_list_kernel, likewise)
struct PairNumbering { const uint32_t *tiles, *wave_tiles; float* rec; unsigned long long *count, *host_count; uint32_t* chunk_pairs; int nchunks; };
struct RankOut { const float* rec; const uint32_t* tiles; uint32_t* sorted_idx; uint4* ranklist; uint32_t* chunk_pairs; };
int launch_depth_sort(const uint32_t* keys, const uint32_t* wave_kmin, const uint32_t* wave_kmax, int nwaves, int P,
                      DepthSortState st, PairNumbering pn, RankOut ro, int debug, hipStream_t s);
// word ranges some kernel copies on the side (grid-stride), saving copy launches
struct CopyJob { const uint32_t* src; uint32_t* dst; int words; };
// A second render of the same geometry whose colours are all (1, 1, 1) -- the reference's opacity pass
// (gaussian_renderer/__init__.py:132-142) -- needs no compositing: every channel is sum_k alpha_k T_k = 1 - T_final, plus
// T_final bg, and the first render's per-pixel / per-quadrant records hold for it as they are (same geometry: same T, same
// contributors), checkpoints included (colour composited before a chunk = 1 - T there).  Whether the colours ARE all ones
// is only known when the recolouring launch has looked at them, so that launch's other workgroups write this image
// SPECULATIVELY (second_ones_body; per-pixel part in memory order, a wave per quadrant for the records) and the render
// launch behind it either leaves everything as it is (all ones) or composites over it.  (Until round 3: a launch of its own
// behind the render launch.)
struct SecondOnes {
    const float* src_final_T; const uint32_t* src_n_contrib; const uint32_t* src_ncon_c; const uint32_t* src_qcount;
    const float4* src_ckpt; const uint32_t* src_ck_start;
    float* out_color; float* final_T; uint32_t* n_contrib; uint32_t* ncon_c; uint32_t* qcount; float4* ckpt; uint32_t* ck_start;
    const float* bg; int W, H, gx, ntiles, chunks;
};
__device__ __forceinline__ void second_ones_body(const SecondOnes& A, const int block, const int pixel_blocks) {
    if (block < pixel_blocks) {
        // one pixel per thread, 256 consecutive pixels per workgroup (a wave per quadrant touches eight 32-byte runs per
        // access: 1.4 TB/s)
        const size_t HW = (size_t)A.H * A.W, pid = (size_t)block * 256 + threadIdx.x;
        if (pid < HW) {
            const float Tf = A.src_final_T[pid];
            A.final_T[pid] = Tf;
            A.n_contrib[pid] = A.src_n_contrib[pid];
            A.ncon_c[pid] = A.src_ncon_c[pid];
            A.out_color[pid] = (1.0f - Tf) + Tf * A.bg[0];
            A.out_color[HW + pid] = (1.0f - Tf) + Tf * A.bg[1];
            A.out_color[2 * HW + pid] = (1.0f - Tf) + Tf * A.bg[2];
        }
        return;
    }
    const int quad = (block - pixel_blocks) * 4 + ((int)threadIdx.x >> 6), tile = quad >> 2, lane = threadIdx.x & 63;
    if (tile >= A.ntiles) return;
    if (lane == 0) A.qcount[quad] = A.src_qcount[quad];
    if (A.chunks > 1) {
        if (lane < A.chunks) A.ck_start[(size_t)quad * A.chunks + lane] = A.src_ck_start[(size_t)quad * A.chunks + lane];
        for (int c = 1; c < A.chunks; c++) {
            if (A.src_ck_start[(size_t)quad * A.chunks + c] == 0xFFFFFFFFu) break;  // (in order: none behind it either)
            const size_t ci = ((size_t)quad * (size_t)(A.chunks - 1) + (size_t)(c - 1)) * 64 + lane;
            const float Tc = A.src_ckpt[ci].x;
            A.ckpt[ci] = make_float4(Tc, 1.0f - Tc, 1.0f - Tc, 1.0f - Tc);
        }
    }
}
// `ones`: not null = the launch's other workgroups write the speculative all-ones image
int launch_recolor(const GsFwdArgs& a, const float* rec_src, const uint32_t* tiles_src, float* rec_dst,
                   uint32_t* tiles_dst, uint32_t* clamped_dst, unsigned long long* not_ones, CopyJob c0, CopyJob c1,
                   ZeroJob z0, const SecondOnes* ones, hipStream_t s);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);

// stable LSD radix sort of (u32 key, u32 value) pairs on key bits [0, bits); ping-pongs between
// (k0,v0) and (k1,v1); the result lands in buffer (passes & 1).  `hist` is the table area of the layouts.
// The per-pass digit totals inside it must be zero when the first pass starts: either the sort clears
// them itself (a fill launch), or the caller has a kernel that runs before it clear the region
// sort_totals_region() names (`totals_zeroed`).
// `n_dev` (may be null): the number of elements is read on the device, n is then only their capacity (grid sizes).
int launch_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t* hist, int64_t n, int bits,
                      bool totals_zeroed, int debug, hipStream_t s, const unsigned long long* n_dev = nullptr);
static inline void sort_totals_region(uint32_t* hist, int64_t n, int bits, uint32_t** ptr, int* words) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    const size_t items = nn <= (size_t)SORT_SMALL_N ? 1024 : SORT_ITEMS;
    *ptr = hist + (size_t)256 * ((nn + items - 1) / items);
    *words = ((bits + 7) / 8) * SORT_TOTALS_REPL * 256;
}


// pair numbering: prefix sum of tiles touched in INDEX order -> every Gaussian's first pair (record slot 9) and the
// frame's pair count (geom state + the caller's pinned host word)
int launch_first_pair(const uint32_t* tiles, const uint32_t* wave_tiles, float* rec, unsigned long long* count,
                      unsigned long long* host_count, int P, int debug, hipStream_t s);
// tile binning (binning.hip): rank list -> per-(segment, tile) counts -> ranges + launch order -> tile lists
int launch_rank_list(const uint32_t* sorted_idx, const float* rec, const uint32_t* tiles, uint4* ranklist,
                     uint32_t* chunk_pairs, int P, int debug, hipStream_t s);
// per-(segment, tile) pair counts; per-tile totals (integer atomics of the counting pass into words cleared beforehand:
// `zero_bytes` from tile_tot on)
// (`xcc_mask`, may be null: every workgroup of the counting pass sets the bit of the XCD it runs on)
struct TileCounts { uint32_t *seg_cnt, *tile_tot; size_t zero_bytes; uint32_t* xcc_mask = nullptr; };
// what the tile-order launch of the forward also does: mark the tiles whose list is long against the frame's total in
// the launch order (bit 31; render_fwd.hip) and report how many there are and the longest list (GsFwdArgs.frame_stats)
// `mark`: 0 none, 1 the tiles the four-wave forward takes, 2 the tiles the chunk-parallel forward takes (+ its work list)
struct LongLists { int mark; long long* stats; uint32_t* cw_hdr = nullptr; uint2* cw_units = nullptr; uint32_t ch_min = FWDC_CH_MIN;
                   uint32_t* cw_items = nullptr; const uint32_t* xcc_mask = nullptr;
                   uint32_t total_div = (uint32_t)FWD4_TOTAL_DIV; };
int launch_tile_lists(const uint4* ranklist, const uint32_t* chunk_pairs, uint32_t* seg_start, int P, int gx, int gy, TileCounts tc, uint32_t* ranges,
                      uint32_t* order, uint32_t* point_list, PairCount pc, LongLists ll, bool totals_zeroed, int debug,
                      hipStream_t s);

// training-step bookkeeping (optim.hip, row N4)
int launch_densify_stats(int N, const int32_t* radii, const float* viewspace_grad, float* max_radii2D,
                         float* xyz_gradient_accum, float* denom, hipStream_t s);
int launch_adam(int n, const GsAdamTensor* tensors, double beta1, double beta2, double eps, int64_t step, hipStream_t s);

// K nearest reference points of every query (knn.hip, row N4); workspace = knn_ws_bytes(Nr)
int launch_knn_points(int Nq, const float* queries, int Nr, const float* ref, int K, float* out_d, long long* out_i,
                      void* ws, size_t ws_bytes, hipStream_t s);

// pre-rasterizer per-Gaussian chains (prepass.hip, row N3)
int launch_build_cov(int N, const float* scales, float mod, const float* rot, int is_matrix, float* cov6, hipStream_t s);
int launch_build_cov_bwd(int N, const float* scales, float mod, const float* rot, int is_matrix, const float* dL_dcov6,
                         float* dL_dscales, float* dL_drot, hipStream_t s);
int launch_sh2rgb(int N, int deg, int M, const float* shs, const float* xyz, const float* campos, const float* R_fwd,
                  const float* noise_host, float* colors, uint8_t* clamped, hipStream_t s);
int launch_sh2rgb_bwd(int N, int deg, int M, const float* shs, const float* xyz, const float* campos, const float* R_fwd,
                      const float* noise_host, const uint8_t* clamped, const float* dL_dcolors, float* dL_dshs, float* dL_dxyz,
                      hipStream_t s);

// fused L1 image loss (loss.hip): loss[0] = mean |x - y|, grad = sign(x - y) / n
size_t l1_ws_bytes(int64_t n);
int launch_l1_loss(const float* x, const float* y, int64_t n, float* loss, float* grad, float* partial, hipStream_t s);
// loss[0] = (partial[0] + ... + partial[nparts - 1], in a fixed order) * inv_n  (the second pass of the L1 / BCE losses; also
// the fused L1's, whose partials come from the render launch)
int launch_loss_final(const float* partial, int nparts, float inv_n, float* loss, hipStream_t s);
int launch_bce_loss(const float* x, const float* y, int64_t n, float* loss, float* grad, float* partial, hipStream_t s);
// SSIM of two (C,H,W) images + the three partial-derivative maps its backward filters (loss.hip)
size_t ssim_ws_bytes(int C, int H, int W);
int launch_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_out, float* dm_dmu1,
                        float* dm_ds1, float* dm_ds12, float* partial, hipStream_t s);
int launch_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                         const float* dm_ds1, const float* dm_ds12, const float* dL_dssim, float* dL_dimg1,
                         hipStream_t s);

// `fill`: 16-byte words the kernel's other workgroups set to all-ones while the first one orders the tiles (the
// backward's ROW_UNWRITTEN marks: a fill launch less, and it overlaps the single-workgroup ordering)
#define MARKS_CLEAN 0x600DF111u  // BinLayout::marks_flag: every mark is ROW_UNWRITTEN (the forward's fill, no backward since)
struct FillJob { uint4* ptr; size_t quads; int stream = 0; const uint32_t* clean = nullptr; /* skip the job if *clean == MARKS_CLEAN */ };

// 16-byte store that does not stay in the L2 as a dirty line (streamed output nothing re-reads soon)
typedef uint32_t gs_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(uint4* p, uint4 v) {
    gs_u32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<gs_u32x4*>(p));
}
// mode 0: work = ranges[t].y - ranges[t].x; mode 1: work = keys[4 t .. 4 t + 3] summed (mode 2, work = keys[t] = the
// tile's pair count, runs inside the forward's list-writing launch: binning.hip)
int launch_tile_order(const uint32_t* ranges, const uint32_t* keys, int mode, int ntiles, uint32_t* order, PairCount pc,
                      FillJob fill, LongLists ll, int debug, hipStream_t s);
// per-quadrant compacted lists and their bookkeeping (forward writes, backward reads)
struct QuadLists {
    uint32_t* qlist;    // [4 D]: quadrant (tile t, q) owns [4 ranges[t].x + q n_t, ... + n_t)
    uint32_t* ncon_c;   // [H W]
    uint32_t* qcount;   // [tiles][4]
    int four_waves = 0;      // forward: four waves per quadrant, all used on the tiles marked in the launch order
    // forward: the marked tiles chunk-parallel (render_chunk) -- the work list (this state's, or the first render's), this
    // state's hand-off words and records, and for a second render the first one's
    int chunked = 0;
    const uint32_t* cw_hdr = nullptr;
    const uint2* cw_units = nullptr;
    const uint32_t* cw_items = nullptr;
    uint32_t* cw_q = nullptr;
    uint32_t* cw_flag = nullptr;
    uint32_t* cw_done = nullptr;
    float* cw_rec = nullptr;
    const uint32_t* src_cw_flag = nullptr;
    const float* src_cw_rec = nullptr;
    const float* src_final_T = nullptr;
    // forward of a second render of the same geometry: the first render's per-quadrant counts and n_contrib (or null)
    const uint32_t* src_qcount = nullptr;
    const uint32_t* src_n_contrib = nullptr;
    const unsigned long long* not_ones = nullptr;  // (zero: that render's colours are all ones -- second_ones_kernel renders)
    float4* ckpt = nullptr;  // compositing state at the chunk boundaries (see BWD_CH), or null
    uint32_t* ck_start = nullptr;  // [quadrant][chunks]: compacted index every chunk starts at
    int chunks = 1;          // chunks per quadrant the backward runs (1: one wave per quadrant walks the whole list)
    // forward: the backward's row marks, filled on the side (null: not this render's job), and their state word
    uint4* marks = nullptr;
    size_t mark_quads = 0;
    uint32_t* marks_flag = nullptr;
    uint32_t* all_ones = nullptr;  // (a second render) the image state's "this image is 1 - T of the first render" word
    // fused L1 loss (GsFwdArgs.l1_target): the target image, one partial sum per quadrant, the mean
    const float* l1_target = nullptr;
    float* l1_part = nullptr;
    float* l1_loss = nullptr;
};
int launch_render_forward(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* order,
                          const float* bg, int W, int H, float* out_color, float* final_T, uint32_t* n_contrib,
                          const QuadLists& ql, hipStream_t s);
// a second image rendered from the same geometry with other colours (gs_forward_shared: the reference's opacity pass
// with colours = 1, gaussian_renderer/__init__.py:132-142), differentiated in the SAME pass: alpha and T are shared,
// so dL/dalpha is linear in (c . g) summed over both images -- one more dot product per step and one more term in
// Gtot; the gradient of the second image's colours is not produced (they are constants)
struct SecondImage {
    const float* colors;     // [P,3]
    const float* out_color;  // [3,H,W] the second render
    const float* dL_dpix;    // [3,H,W]
    const float4* ckpt;      // its checkpoints (chunked backward), or null
    const uint32_t* all_ones;  // a word of its image state: 1 = its colours are all (1, 1, 1) (ImgLayout.all_ones)
};
// fused L1 loss in the backward: dL/dpixel += sign(out_color - target) * (grad ? *grad : 1) / (3 H W)  (target null: off)
struct L1Grad { const float* target; const float* grad; };
int launch_render_backward(const float* rec, const uint32_t* ranges, const uint32_t* order, int W, int H,
                           const QuadLists& ql, const float* out_color, const float* dL_dpix, const float* dL_dopa,
                           const float* final_T, const float* bg, float* qrows, uint32_t* q8, const SecondImage* second,
                           L1Grad l1, hipStream_t s);
// opacity render of a finished forward: (1 - final_T) + final_T * bg0 per pixel (render_fwd.hip)
int launch_opacity_image(const float* final_T, const float* bg, int W, int H, float* out, hipStream_t s);
int launch_gaussian_backward(const GsFwdArgs& a, const int32_t* radii, const float* rec, const uint32_t* tiles,
                             const uint32_t* clamped, const uint32_t* q8, const float* qrows, float* sums,
                             uint32_t* marks_flag, const GsGrads& g, hipStream_t s);
// backward scratch: [4 D rows x 32 B: sums 0..7 of (pair, quadrant) at index gradient_row(...) (below), pairs in
// emission order (one 32-byte sector per row) | P rows x 48 B of per-Gaussian sums | launch order (u32 per tile)];
// sum 8 of every row, or ROW_UNWRITTEN: the 4 D mark words of the binning state (BinLayout::marks)
#define ROW_UNWRITTEN 0xFFFFFFFFu  // a NaN pattern no arithmetic produces
// Row of (pair, quadrant): Gaussian i owns the 4 tt rows [4 first_pair, 4 (first_pair + tt)), tt = w h tiles of its
// rectangle, i.e. a grid of 2w x 2h quadrants.  The rows are laid out ROW-MAJOR OVER THAT QUADRANT GRID: row = 4 first_pair
// + (2 ty + (q >> 1)) 2w + 2 tx + (q & 1) for the tile (tx, ty) of the rectangle.  A 128-byte line (four rows) is then four
// horizontally adjacent quadrants -- a strip of 32 x 8 pixels -- which a Gaussian's footprint (about as large as a tile)
// reaches together or not at all far more often than the four quadrants of one tile (most tiles of a rectangle are cut
// by the footprint's boundary): the rows the backward writes sit in fewer lines for segment_reduce_kernel to fetch, and
// two of a line's rows still come from the two quadrant waves of one tile (same XCD: the L2 merges their stores).
__host__ __device__ __forceinline__ uint32_t gradient_row(uint32_t first_pair, uint32_t w, uint32_t tx, uint32_t ty, uint32_t q) {
    return first_pair * 4u + (2u * ty + (q >> 1)) * (2u * w) + 2u * tx + (q & 1u);
}
static inline size_t scratch_rows_bytes(int64_t D) { return align_up((size_t)(D > 0 ? D : 1) * 4 * 32, 256); }
static inline size_t scratch_sums_bytes(int P) { return align_up((size_t)(P > 0 ? P : 1) * REC_F * 4, 256); }
static inline size_t scratch_total_bytes(int64_t D, int P, int ntiles) {
    return scratch_rows_bytes(D) + scratch_sums_bytes(P) + align_up((size_t)ntiles * 4, 256);  // (the row marks: BinLayout)
}

// report counters (debug_stats.hip): out[0] = pairs composited (alpha >= 1/255 before the pixel is done), out[1] = sum of n_contrib
int launch_pair_stats(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* n_contrib, int W, int H,
                      unsigned long long* out, hipStream_t s);
int launch_knn(int P, const float* points, float* out, void* ws, size_t ws_bytes, hipStream_t s);
size_t knn_ws_bytes(int P);
