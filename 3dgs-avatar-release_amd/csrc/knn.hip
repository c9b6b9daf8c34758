// knn.hip -- distCUDA2: mean squared distance of every point to its three
// nearest OTHER points (exact).  Replaces simple_knn's simple_knn.cu (un-vendored; call site
// scene/gaussian_model.py:186).
//
// Morton-order the points (24-bit code, sorted with this library's radix sort), gather them into
// a contiguous float4 array, build one AABB per box of 64 consecutive points, then per point scan
// only the boxes whose AABB is not farther than the current third-best distance.  Pruning is exact
// in fp32: the box distance uses the same subtract / square / add sequence as the point distance, and
// every step of that sequence is monotone, so fl(box distance) <= fl(point distance) for every
// point inside the box.  Built with -ffp-contract=off so distances match the CPU oracle bit for bit.
#include "common.h"
#include <float.h>

#define KNN_BOX 64   // points per box of the Morton-ordered reference set (one point per lane of the search's wave)
#define KNN_WAVE 64  // the search kernels run one wave per workgroup: their barriers and votes are wave-level
#define KNN_MM_BLOCKS 128  // blocks of the bounding-box pass; each leaves its own partial (no atomics, nothing to clear)

struct KnnLayout {
    size_t key0, key1, val0, val1, hist, pts, boxes, minmax, total;
    int nblk_sort, nbox;
};
static KnnLayout knn_layout(int P) {
    KnnLayout L;
    size_t o = 0;
    size_t n = (size_t)(P > 0 ? P : 1);
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.nblk_sort = (int)((n + SORT_ITEMS - 1) / SORT_ITEMS);
    L.nbox = (int)((n + KNN_BOX - 1) / KNN_BOX);
    L.key0 = take(n * 4);
    L.key1 = take(n * 4);
    L.val0 = take(n * 4);
    L.val1 = take(n * 4);
    L.hist = take(sort_table_words(n) * 4);
    L.pts = take(n * 16);
    L.boxes = take((size_t)L.nbox * 32);
    L.minmax = take((size_t)KNN_MM_BLOCKS * 6 * 4);
    L.total = o;
    return L;
}
size_t knn_ws_bytes(int P) { return knn_layout(P).total; }

// partial bounding boxes: mm[6 b .. 6 b + 5] = (min xyz, max xyz) of block b's grid-stride share
// (also clears the radix sort's digit totals on the side: a fill launch less)
__global__ __launch_bounds__(256) void knn_minmax_kernel(int P, const float* __restrict__ pts, float* __restrict__ mm,
                                                         const ZeroJob zt) {
    __shared__ float smin[3][4], smax[3][4];
    zero_job(zt);
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = pts[3 * i + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { smin[c][threadIdx.x >> 6] = mn[c]; smax[c][threadIdx.x >> 6] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        mm[6 * blockIdx.x + c] = fminf(fminf(smin[c][0], smin[c][1]), fminf(smin[c][2], smin[c][3]));
        mm[6 * blockIdx.x + 3 + c] = fmaxf(fmaxf(smax[c][0], smax[c][1]), fmaxf(smax[c][2], smax[c][3]));
    }
}

// 8 bits -> every third bit
__device__ __forceinline__ uint32_t spread8(uint32_t x) {
    x = (x | (x << 8)) & 0x0000F00Fu;
    x = (x | (x << 4)) & 0x000C30C3u;
    x = (x | (x << 2)) & 0x00249249u;
    return x;
}
#define KNN_MORTON_BITS 24  // 8 per axis: three radix passes; the order only decides how well the boxes prune

__global__ __launch_bounds__(256) void knn_morton_kernel(int P, const float* __restrict__ pts,
                                                         const float* __restrict__ mm, int nparts,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    __shared__ float sred[6][4];
    __shared__ float sbox[6];
    {
        // every block folds the partial bounding boxes itself (nparts <= KNN_MM_BLOCKS <= 256)
        const int t = threadIdx.x;
        float v[6];
#pragma unroll
        for (int c = 0; c < 6; c++) v[c] = (c < 3) ? FLT_MAX : -FLT_MAX;
        if (t < nparts)
#pragma unroll
            for (int c = 0; c < 6; c++) v[c] = mm[6 * t + c];
#pragma unroll
        for (int c = 0; c < 6; c++) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float o = __shfl_xor(v[c], d, 64);
                v[c] = (c < 3) ? fminf(v[c], o) : fmaxf(v[c], o);
            }
            if ((t & 63) == 0) sred[c][t >> 6] = v[c];
        }
        __syncthreads();
        if (t < 6)
            sbox[t] = (t < 3) ? fminf(fminf(sred[t][0], sred[t][1]), fminf(sred[t][2], sred[t][3]))
                              : fmaxf(fmaxf(sred[t][0], sred[t][1]), fmaxf(sred[t][2], sred[t][3]));
        __syncthreads();
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = sbox[c], hi = sbox[3 + c];
        const float ext = fmaxf(hi - lo, 1e-30f);
        float u = (pts[3 * i + c] - lo) / ext * 255.0f;
        u = fminf(fmaxf(u, 0.f), 255.f);
        code |= spread8((uint32_t)u) << c;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// gather the points into Morton order (xyz + original index) and leave one AABB per box of KNN_BOX = 64 consecutive
// points: a workgroup of 256 threads is four boxes, one per wave
__global__ __launch_bounds__(256) void knn_gather_box_kernel(int P, const float* __restrict__ pts,
                                                             const uint32_t* __restrict__ order, float4* __restrict__ out,
                                                             float* __restrict__ boxes) {
    static_assert(KNN_BOX == 64, "one box per wave");
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (r < P) {
        const uint32_t i = order[r];
        const float4 p = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __uint_as_float(i));
        out[r] = p;
        mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z;
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
        }
    const int box = r >> 6;
    if ((threadIdx.x & 63) == 0 && box * KNN_BOX < P) {
        *reinterpret_cast<float4*>(boxes + (size_t)box * 8) = make_float4(mn[0], mn[1], mn[2], 0.f);
        *reinterpret_cast<float4*>(boxes + (size_t)box * 8 + 4) = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
}

__device__ __forceinline__ float dist2(const float4 a, const float4 b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

// ---- the search.  One kernel serves distCUDA2 (K = 3, the point itself excluded, mean of the three squared
// distances) and knn_points (pytorch3d.ops.knn_points as the reference calls it,
// utils/loss_utils.py:76-79,92-96 -- K = 5 / 6 self-KNN of the canonical Gaussians every step -- and
// models/deformer/rigid.py:43 -- nearest SMPL vertex, K = 1; squared distances ascending, ties broken by the smaller
// reference index, the query point itself NOT excluded, as pytorch3d).
//
// One wave per 64 queries.  For a self-KNN the queries ARE the Morton-ordered points, so a wave's queries are box
// `blockIdx.x` of the reference set:
//   0. it scans its own box (and for K >= 3 the two Morton neighbours), all lanes inserting in parallel: every lane
//      then holds a K-th best distance that is already close to final;
//   1. boxes are filtered 64 at a time, one box per lane, against the WAVE's query bounding box and the wave's
//      largest K-th best (a lower bound of every lane's own box distance, see below) -- a few dozen coalesced loads
//      instead of one dependent load per box and wave (that walk over all boxes was most of the old kernel's time);
//   2. the survivors get the exact per-query test; a box some lanes still want is loaded one point per lane and
//      measured against those queries one at a time (scan_box_sparse).
// Pruning stays exact in fp32.  Per query: the gap to a box is written as (point - nearest box point) per axis so
// it rounds like dist2(), and every step of subtract / square / add is monotone, so fl(box distance) <=
// fl(point distance) for every point of the box.  Per wave: the gap between the query bounding box and the box is
// no larger than any query's own gap on every axis (p <= qhi => lo - p >= lo - qhi, and rounding is monotone), so
// fl(wave bound) <= fl(query bound); and a lane's K-th best only shrinks, so a box dropped against the wave's
// largest K-th best now could never be wanted later.
#define KNN_MAXK 8

// (K is a template parameter everywhere: a run-time K would index the two lists dynamically, which puts them in
// scratch memory -- measured 30x slower)
// A list entry is one 64-bit key, (squared distance bits << 32) | reference index: squared distances are >= +0, whose
// bit patterns order like the floats, so ONE unsigned 64-bit compare is "nearer, or as near with the smaller index".
typedef unsigned long long knn_key;
__device__ __forceinline__ knn_key make_key(float d, uint32_t id) { return ((knn_key)__float_as_uint(d) << 32) | id; }
__device__ __forceinline__ float key_dist(knn_key k) { return __uint_as_float((uint32_t)(k >> 32)); }
#define KNN_EMPTY_KEY (((knn_key)0x7F7FFFFFu << 32) | 0xFFFFFFFFu)  // (FLT_MAX, no index)

template <int K>
__device__ __forceinline__ void kbest_insert(knn_key c, knn_key* key) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (key[k] > c) {
            const knn_key t = key[k];
            key[k] = c;
            c = t;
        }
    }
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float lane_value(float v, int src_lane) {  // src_lane wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
// squared gap between the interval [qlo, qhi] (a point when qlo == qhi) and the box [lo, hi], per axis as
// (query side - box side) so that a point query rounds exactly like dist2()
__device__ __forceinline__ float axis_gap(float qlo, float qhi, float lo, float hi) {
    return (qhi < lo) ? (qhi - lo) : ((qlo > hi) ? (qlo - hi) : 0.f);
}

// sorted_queries: the queries ARE the Morton-ordered reference points (query r = sp[r], answers go to row sp[r].w);
// otherwise queries[q] is read in the given order.  DIST2: distCUDA2's output (out_d[row] = mean of the K distances,
// the point itself -- by index, not by position -- left out).
// QPW: queries per wave (16 / 32 / 64; the other lanes only carry box points).  A wave's run time is a long chain of
// dependent steps, so a small problem is finished sooner by more, shorter waves: the launcher picks QPW from Nq.
template <int K, bool DIST2, int QPW>
__global__ __launch_bounds__(KNN_WAVE) void knn_scan_kernel(int Nq, const float* __restrict__ queries, int sorted_queries,
                                                            int Nr, int nbox, const float4* __restrict__ sp,
                                                            const float* __restrict__ boxes, float* __restrict__ out_d,
                                                            long long* __restrict__ out_i) {
    static_assert(KNN_BOX == KNN_WAVE && KNN_BOX % QPW == 0, "a wave's sorted queries lie in one box");
    __shared__ float4 tile[KNN_BOX];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * QPW + lane;
    const bool active = lane < QPW && r < Nq;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t row = 0;
    if (active) {
        if (sorted_queries) {
            p = sp[r];
            row = __float_as_uint(p.w);
        } else {
            p = make_float4(queries[3 * (size_t)r], queries[3 * (size_t)r + 1], queries[3 * (size_t)r + 2], 0.f);
            row = (size_t)r;
        }
    }
    const uint32_t self_id = (DIST2 && active) ? __float_as_uint(p.w) : 0xFFFFFFFFu;
    knn_key key[K];
#pragma unroll
    for (int k = 0; k < K; k++) key[k] = KNN_EMPTY_KEY;

    auto scan_box = [&](int b, bool want) {
        __syncthreads();  // (one wave: orders the previous scan before this refill)
        const int base = b * KNN_BOX, cnt = min(Nr - base, KNN_BOX);
        if (lane < cnt) tile[lane] = sp[base + lane];
        __syncthreads();
        if (want)
            for (int i = 0; i < cnt; i++) {
                const float4 o = tile[i];
                const uint32_t id = __float_as_uint(o.w);
                const knn_key c = make_key(dist2(p, o), id);
                // most points are no better than the current K-th best: one compare instead of the K-deep
                // insertion chain (which a wave only enters when one of its lanes has a candidate)
                if (c < key[K - 1] && id != self_id) kbest_insert<K>(c, key);
            }
    };

    // The same for a box only SOME lanes want (every box after step 0): the box's points sit one per lane, and the
    // wanting queries are taken one at a time -- all 64 lanes measure their point against that query, a ballot
    // finds the few points that beat its K-th best, and only those enter its list.  Cost per (query, box) pair
    // instead of per (box, 64 points) whatever the number of lanes that want it.
    auto load_box = [&](int b, float4& o, bool& ovalid) {
        const int base = b * KNN_BOX;
        ovalid = base + lane < Nr;
        o = ovalid ? sp[base + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto scan_box_sparse = [&](const float4 o, const bool ovalid, unsigned long long wantmask) {
        const uint32_t oid = __float_as_uint(o.w);
        // which points of the box beat which query's K-th best: a 64-bit mask per query, parked in that query's lane
        uint32_t mlo = 0u, mhi = 0u;
        auto beat_mask = [&](int q) {  // the points of the box that beat query q's K-th best
            const float4 pq = make_float4(lane_value(p.x, q), lane_value(p.y, q), lane_value(p.z, q), 0.f);
            const knn_key kq = ((knn_key)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key[K - 1] >> 32), q) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key[K - 1], q);
            const uint32_t sid = (uint32_t)__builtin_amdgcn_readlane((int)self_id, q);
            return __ballot(ovalid && make_key(dist2(pq, o), oid) < kq && oid != sid);
        };
        while (wantmask) {
            // two queries per trip: their chains (read lanes -> distance -> compare -> ballot) are independent, and a
            // lone wave otherwise waits out every step of one chain
            const int q0 = __ffsll((long long)wantmask) - 1;
            wantmask &= wantmask - 1;
            const bool two = wantmask != 0ull;
            const int q1 = two ? __ffsll((long long)wantmask) - 1 : q0;
            wantmask &= wantmask - 1;  // (0 stays 0)
            const unsigned long long pm0 = beat_mask(q0);
            const unsigned long long pm1 = beat_mask(q1);
            if (pm0 && lane == q0) { mlo = (uint32_t)pm0; mhi = (uint32_t)(pm0 >> 32); }
            if (two && pm1 && lane == q1) { mlo = (uint32_t)pm1; mhi = (uint32_t)(pm1 >> 32); }
        }
        // ... and all queries insert their candidates together, one candidate each per trip (the insertion chain is
        // K deep: run once per candidate under a one-lane mask it cost more than the search itself)
        unsigned long long mine = ((unsigned long long)mhi << 32) | mlo;
        while (__any(mine != 0ull)) {
            const bool has = mine != 0ull;
            const int j = has ? __ffsll((long long)mine) - 1 : 0;
            mine &= mine - 1;  // (0 stays 0)
            const float4 c = make_float4(__shfl(o.x, j, 64), __shfl(o.y, j, 64), __shfl(o.z, j, 64), 0.f);
            const uint32_t cid = (uint32_t)__shfl((int)oid, j, 64);
            if (has) kbest_insert<K>(make_key(dist2(p, c), cid), key);  // (the same subtraction, so the same bits as above)
        }
    };

    // 0. own box (self-KNN only; -2 = nothing scanned yet)
    const int own = sorted_queries ? (int)(blockIdx.x * QPW) / KNN_BOX : -2;
    const int nb0 = K >= 3 ? 1 : 0;  // for K >= 3 the two Morton neighbours too: measured faster (tighter bounds for step 1)
    if (sorted_queries)
        for (int b = max(own - nb0, 0); b <= min(own + nb0, nbox - 1); b++) scan_box(b, active);

    // the wave's query bounding box (inactive lanes neutral)
    const float BIG = FLT_MAX;
    const float qlx = wave_min_f(active ? p.x : BIG), qhx = wave_max_f(active ? p.x : -BIG);
    const float qly = wave_min_f(active ? p.y : BIG), qhy = wave_max_f(active ? p.y : -BIG);
    const float qlz = wave_min_f(active ? p.z : BIG), qhz = wave_max_f(active ? p.z : -BIG);

    // 1. + 2.  chunks of 64 boxes, outwards from the chunk of the own box; the next chunk's boxes are loaded while
    // this one's survivors are scanned
    const int nchunk = (nbox + 63) / 64;
    const int c0 = sorted_queries ? own / 64 : 0;
    auto chunk_of = [&](int t) {  // t-th chunk of the walk c0, c0+1, c0-1, c0+2, ... (-1: outside)
        const int d = (t + 1) >> 1;
        const int c = (t & 1) ? c0 + d : c0 - d;
        return (c >= 0 && c < nchunk) ? c : -1;
    };
    auto load_chunk = [&](int c, float4& lo, float4& hi) {
        const int bb = c * 64 + lane;
        if (c >= 0 && bb < nbox) {
            lo = *reinterpret_cast<const float4*>(boxes + (size_t)bb * 8);
            hi = *reinterpret_cast<const float4*>(boxes + (size_t)bb * 8 + 4);
        }
    };
    const int tmax = 2 * max(c0, nchunk - 1 - c0);  // last t that can name a chunk
    float4 lo = make_float4(0, 0, 0, 0), hi = lo, nlo = lo, nhi = lo;
    float4 po = lo;  // the wanted box whose points are in flight
    bool pov = false, have = false;
    unsigned long long pmask = 0ull;
    load_chunk(chunk_of(0), lo, hi);
    for (int t = 0; t <= tmax; t++) {
        const int c = chunk_of(t);
        if (t + 1 <= tmax) load_chunk(chunk_of(t + 1), nlo, nhi);
        if (c >= 0) {
            const int bb = c * 64 + lane;
            const float wx = axis_gap(qlx, qhx, lo.x, hi.x), wy = axis_gap(qly, qhy, lo.y, hi.y),
                        wz = axis_gap(qlz, qhz, lo.z, hi.z);
            const float wbound = wx * wx + wy * wy + wz * wz;
            const float rmax = wave_max_f(active ? key_dist(key[K - 1]) : 0.f);
            unsigned long long cand = __ballot(bb < nbox && !(wbound > rmax));
            while (cand) {
                const int j = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int b = c * 64 + j;
                if (b >= own - nb0 && b <= own + nb0) continue;  // scanned in step 0
                const float lx = lane_value(lo.x, j), ly = lane_value(lo.y, j), lz = lane_value(lo.z, j);
                const float hx = lane_value(hi.x, j), hy = lane_value(hi.y, j), hz = lane_value(hi.z, j);
                const float gx = axis_gap(p.x, p.x, lx, hx), gy = axis_gap(p.y, p.y, ly, hy), gz = axis_gap(p.z, p.z, lz, hz);
                const float dbox = gx * gx + gy * gy + gz * gz;
                const unsigned long long wantmask = __ballot(active && !(dbox > key_dist(key[K - 1])));
                if (wantmask) {
                    // the box's points are requested now and measured when the NEXT wanted box has been found, so a
                    // memory latency is never waited for with nothing else in flight (the mask is from now: a
                    // superset of what would be wanted then)
                    float4 o;
                    bool ov;
                    load_box(b, o, ov);
                    if (have) scan_box_sparse(po, pov, pmask);
                    po = o; pov = ov; pmask = wantmask; have = true;
                }
            }
        }
        lo = nlo;
        hi = nhi;
    }
    if (have) scan_box_sparse(po, pov, pmask);
    if (active) {
        if (DIST2) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < K; k++) sum += key_dist(key[k]);
            out_d[row] = sum / (float)K;
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t id = (uint32_t)key[k];
                out_d[row * K + k] = key_dist(key[k]);
                out_i[row * K + k] = (id == 0xFFFFFFFFu) ? -1ll : (long long)id;  // fewer than K reference points: -1
            }
        }
    }
}

int launch_knn_points(int Nq, const float* queries, int Nr, const float* ref, int K, float* out_d, long long* out_i,
                      void* ws, size_t ws_bytes, hipStream_t s) {
    const KnnLayout L = knn_layout(Nr);
    if (ws_bytes < L.total) return GS_E_WORKSPACE;
    char* w = (char*)ws;
    uint32_t* k0 = (uint32_t*)(w + L.key0);
    uint32_t* k1 = (uint32_t*)(w + L.key1);
    uint32_t* v0 = (uint32_t*)(w + L.val0);
    uint32_t* v1 = (uint32_t*)(w + L.val1);
    uint32_t* hist = (uint32_t*)(w + L.hist);
    float4* sp = (float4*)(w + L.pts);
    float* boxes = (float*)(w + L.boxes);
    float* mm = (float*)(w + L.minmax);
    const int nb = (Nr + 255) / 256;
    StageScope st("knn_points", s);
    const int nparts = nb < KNN_MM_BLOCKS ? nb : KNN_MM_BLOCKS;
    ZeroJob zt;
    sort_totals_region(hist, Nr, KNN_MORTON_BITS, &zt.ptr, &zt.words);
    hipLaunchKernelGGL(knn_minmax_kernel, dim3(nparts), dim3(256), 0, s, Nr, ref, mm, zt);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, Nr, ref, mm, nparts, k0, v0);
    GS_LAUNCH_CHECK("knn.morton", 0, s);
    int rc = launch_sort_pairs(k0, v0, k1, v1, hist, Nr, KNN_MORTON_BITS, true, 0, s);
    if (rc != GS_OK) return rc;
    const uint32_t* order = (radix_passes(KNN_MORTON_BITS) & 1) ? v1 : v0;
    hipLaunchKernelGGL(knn_gather_box_kernel, dim3(nb), dim3(256), 0, s, Nr, ref, order, sp, boxes);
    const int self = (queries == ref && Nq == Nr) ? 1 : 0;
    // queries per wave: enough waves to give every SIMD a few (1024 SIMDs)
    const int qpw = Nq >= (1 << 18) ? 64 : (Nq >= (1 << 17) ? 32 : 16);
#define KNN_LAUNCH_Q(KK, QQ)                                                                                             \
    hipLaunchKernelGGL((knn_scan_kernel<KK, false, QQ>), dim3((Nq + QQ - 1) / QQ), dim3(KNN_WAVE), 0, s, Nq, queries, self, \
                       Nr, L.nbox, sp, boxes, out_d, out_i)
#define KNN_LAUNCH(KK)                                                                                                   \
    case KK:                                                                                                             \
        if (qpw == 64) KNN_LAUNCH_Q(KK, 64);                                                                             \
        else if (qpw == 32) KNN_LAUNCH_Q(KK, 32);                                                                        \
        else KNN_LAUNCH_Q(KK, 16);                                                                                       \
        break;
    switch (K) {
        KNN_LAUNCH(1) KNN_LAUNCH(2) KNN_LAUNCH(3) KNN_LAUNCH(4) KNN_LAUNCH(5) KNN_LAUNCH(6) KNN_LAUNCH(7) KNN_LAUNCH(8)
        default: return GS_E_BAD_ARG;
    }
#undef KNN_LAUNCH_Q
#undef KNN_LAUNCH
    GS_LAUNCH_CHECK("knn.points", 0, s);
    return GS_OK;
}

int launch_knn(int P, const float* points, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
    const KnnLayout L = knn_layout(P);
    if (ws_bytes < L.total) return GS_E_WORKSPACE;
    char* w = (char*)ws;
    uint32_t* k0 = (uint32_t*)(w + L.key0);
    uint32_t* k1 = (uint32_t*)(w + L.key1);
    uint32_t* v0 = (uint32_t*)(w + L.val0);
    uint32_t* v1 = (uint32_t*)(w + L.val1);
    uint32_t* hist = (uint32_t*)(w + L.hist);
    float4* sp = (float4*)(w + L.pts);
    float* boxes = (float*)(w + L.boxes);
    float* mm = (float*)(w + L.minmax);
    const int nb = (P + 255) / 256;
    const int nparts = nb < KNN_MM_BLOCKS ? nb : KNN_MM_BLOCKS;
    ZeroJob zt;
    sort_totals_region(hist, P, KNN_MORTON_BITS, &zt.ptr, &zt.words);
    hipLaunchKernelGGL(knn_minmax_kernel, dim3(nparts), dim3(256), 0, s, P, points, mm, zt);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, P, points, mm, nparts, k0, v0);
    GS_LAUNCH_CHECK("knn.morton", 0, s);
    int rc = launch_sort_pairs(k0, v0, k1, v1, hist, P, KNN_MORTON_BITS, true, 0, s);
    if (rc != GS_OK) return rc;
    const uint32_t* order = (radix_passes(KNN_MORTON_BITS) & 1) ? v1 : v0;
    hipLaunchKernelGGL(knn_gather_box_kernel, dim3(nb), dim3(256), 0, s, P, points, order, sp, boxes);
    if (P >= (1 << 18))
        hipLaunchKernelGGL((knn_scan_kernel<3, true, 64>), dim3((P + 63) / 64), dim3(KNN_WAVE), 0, s, P, (const float*)nullptr, 1,
                           P, L.nbox, sp, boxes, out, (long long*)nullptr);
    else if (P >= (1 << 17))
        hipLaunchKernelGGL((knn_scan_kernel<3, true, 32>), dim3((P + 31) / 32), dim3(KNN_WAVE), 0, s, P, (const float*)nullptr, 1,
                           P, L.nbox, sp, boxes, out, (long long*)nullptr);
    else
        hipLaunchKernelGGL((knn_scan_kernel<3, true, 16>), dim3((P + 15) / 16), dim3(KNN_WAVE), 0, s, P, (const float*)nullptr, 1,
                           P, L.nbox, sp, boxes, out, (long long*)nullptr);
    GS_LAUNCH_CHECK("knn.search", 0, s);
    return GS_OK;
}
