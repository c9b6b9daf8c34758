// knn.hip -- distCUDA2: mean squared distance of every point to its three
// nearest OTHER points (exact).  Replaces simple_knn's simple_knn.cu (un-vendored; call site
// scene/gaussian_model.py:186).
//
// Morton-order the points (30-bit code, sorted with this library's radix sort), gather them into
// a contiguous float4 array, build one AABB per box of 256 consecutive points, then per point scan
// only the boxes whose AABB is not farther than the current third-best distance.  Pruning is exact
// in fp32: the box distance uses the same subtract / square / add sequence as the point distance, and
// every step of that sequence is monotone, so fl(box distance) <= fl(point distance) for every
// point inside the box.  Built with -ffp-contract=off so distances match the CPU oracle bit for bit.
#include "common.h"
#include <float.h>

#define KNN_BOX 256
#define KNN_WAVE 64  // the search kernels run one wave per workgroup: their barriers and votes are wave-level

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
    L.minmax = take(64);
    L.total = o;
    return L;
}
size_t knn_ws_bytes(int P) { return knn_layout(P).total; }

// order-preserving float <-> uint map for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void knn_init_kernel(uint32_t* mm) {
    if (threadIdx.x < 3) mm[threadIdx.x] = 0xFFFFFFFFu;       // min
    else if (threadIdx.x < 6) mm[threadIdx.x] = 0u;           // max
}

__global__ __launch_bounds__(256) void knn_minmax_kernel(int P, const float* __restrict__ pts, uint32_t* mm) {
    __shared__ float smin[3][4], smax[3][4];
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
        const float a = fminf(fminf(smin[c][0], smin[c][1]), fminf(smin[c][2], smin[c][3]));
        const float b = fmaxf(fmaxf(smax[c][0], smax[c][1]), fmaxf(smax[c][2], smax[c][3]));
        atomicMin(&mm[c], f2ord(a));
        atomicMax(&mm[3 + c], f2ord(b));
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(int P, const float* __restrict__ pts,
                                                         const uint32_t* __restrict__ mm, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = ord2f(mm[c]), hi = ord2f(mm[3 + c]);
        const float ext = fmaxf(hi - lo, 1e-30f);
        float u = (pts[3 * i + c] - lo) / ext * 1023.0f;
        u = fminf(fmaxf(u, 0.f), 1023.f);
        code |= spread10((uint32_t)u) << c;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void knn_gather_kernel(int P, const float* __restrict__ pts,
                                                         const uint32_t* __restrict__ order, float4* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P) return;
    const uint32_t i = order[r];
    out[r] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __uint_as_float(i));
}

__global__ __launch_bounds__(KNN_BOX) void knn_box_kernel(int P, const float4* __restrict__ sp, float* __restrict__ boxes) {
    __shared__ float smin[3][4], smax[3][4];
    const int r = blockIdx.x * KNN_BOX + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (r < P) {
        const float4 p = sp[r];
        mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z;
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
        boxes[blockIdx.x * 8 + c] = fminf(fminf(smin[c][0], smin[c][1]), fminf(smin[c][2], smin[c][3]));
        boxes[blockIdx.x * 8 + 4 + c] = fmaxf(fmaxf(smax[c][0], smax[c][1]), fmaxf(smax[c][2], smax[c][3]));
    }
}

__device__ __forceinline__ void kbest3(float d, float* best) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (best[k] > d) { const float t = best[k]; best[k] = d; d = t; }
    }
}
__device__ __forceinline__ float dist2(const float4 a, const float4 b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

// The points of the box being scanned are staged in LDS by the whole workgroup: its 256 queries are consecutive in Morton order, so they want nearly the same boxes, and a
// per-thread walk over global memory pays a full memory latency per point (one dependent load per iteration).
// A box is loaded when ANY query of the group wants it; each query still scans only the boxes it wants itself,
// so the results are those of the per-thread walk.

__global__ __launch_bounds__(KNN_WAVE) void knn_search_kernel(int P, int nbox, const float4* __restrict__ sp,
                                                         const float* __restrict__ boxes, float* __restrict__ out) {
    __shared__ float4 tile[KNN_BOX];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < P;
    const float4 p = active ? sp[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    // seed the rejection radius from the Morton neighbours
    if (active)
        for (int i = max(0, r - 3); i <= min(P - 1, r + 3); i++) {
            if (i == r) continue;
            kbest3(dist2(p, sp[i]), best);
        }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    // boxes are visited outwards from the group's own box (its 256 queries ARE box blockIdx.x), so the K-th best
    // distance is tight before the far boxes are tested -- most of them then fail the AABB test
    const int own = (int)(blockIdx.x * KNN_WAVE) / KNN_BOX;  // a wave's 64 queries lie in one box
    for (int d = 0; own + d < nbox || own - d >= 0; d++) {
        for (int side = 0; side < 2; side++) {
            const int b = side == 0 ? own + d : own - d;
            if ((d == 0 && side == 1) || b < 0 || b >= nbox) continue;  // uniform
            const float4 lo = *reinterpret_cast<const float4*>(boxes + (size_t)b * 8);  // wave-uniform address
            const float4 hi = *reinterpret_cast<const float4*>(boxes + (size_t)b * 8 + 4);
            // gap per axis, written as (point - nearest box point) so it rounds like dist2()
            const float gx = (p.x < lo.x) ? (p.x - lo.x) : ((p.x > hi.x) ? (p.x - hi.x) : 0.f);
            const float gy = (p.y < lo.y) ? (p.y - lo.y) : ((p.y > hi.y) ? (p.y - hi.y) : 0.f);
            const float gz = (p.z < lo.z) ? (p.z - lo.z) : ((p.z > hi.z) ? (p.z - hi.z) : 0.f);
            const float dbox = gx * gx + gy * gy + gz * gz;
            const bool want = active && !(dbox > reject || dbox > best[2]);
            if (!__any(want)) continue;  // wave-uniform
            __syncthreads();  // (one wave: orders the previous scan before this refill)
            const int base = b * KNN_BOX, cnt = min(P - base, KNN_BOX);
            for (int e = threadIdx.x; e < cnt; e += KNN_WAVE) tile[e] = sp[base + e];
            __syncthreads();
            if (want)
                for (int i = 0; i < cnt; i++) {
                    const float dd = dist2(p, tile[i]);
                    if (dd < best[2] && base + i != r) kbest3(dd, best);  // (equal to the third best changes nothing)
                }
        }
    }
    if (active) out[__float_as_uint(p.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

// ---- general K nearest neighbours (pytorch3d.ops.knn_points as the reference calls it,
// utils/loss_utils.py:76-79,92-96 -- K = 5 / 6 self-KNN of the canonical Gaussians every step -- and
// models/deformer/rigid.py:43 -- nearest SMPL vertex, K = 1).  Same machinery as distCUDA2: the REFERENCE set is
// Morton-ordered and boxed; every query scans the boxes, skipping those whose AABB is farther than its
// current K-th best (exact in fp32, see the header).  Results: squared distances ascending, ties broken by the
// smaller reference index; the query point itself is NOT excluded (a self-KNN returns itself first, as
// pytorch3d does).
#define KNN_MAXK 8

// (K is a template parameter everywhere: a run-time K would index the two lists dynamically, which puts them in
// scratch memory -- measured 30x slower)
template <int K>
__device__ __forceinline__ void kbest_insert(float d, uint32_t id, float* bd, uint32_t* bi) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (bd[k] > d || (bd[k] == d && bi[k] > id)) {
            const float td = bd[k];
            const uint32_t ti = bi[k];
            bd[k] = d; bi[k] = id;
            d = td; id = ti;
        }
    }
}

// sorted_queries: the queries ARE the Morton-ordered reference points (self-KNN; query r = sp[r], answers go to
// row sp[r].w); otherwise queries[q] is read in the given order.
template <int K>
__global__ __launch_bounds__(KNN_WAVE) void knn_points_kernel(int Nq, const float* __restrict__ queries, int sorted_queries,
                                                         int Nr, int nbox, const float4* __restrict__ sp,
                                                         const float* __restrict__ boxes,
                                                         float* __restrict__ out_d, long long* __restrict__ out_i) {
    __shared__ float4 tile[KNN_BOX];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < Nq;
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
    float bd[K];
    uint32_t bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = 0xFFFFFFFFu; }
    float reject = FLT_MAX;
    if (sorted_queries && active) {
        // seed the rejection radius from the Morton neighbours (they are re-found in the scan below)
        for (int i = max(0, r - K); i <= min(Nr - 1, r + K); i++) kbest_insert<K>(dist2(p, sp[i]), __float_as_uint(sp[i].w), bd, bi);
        reject = bd[K - 1];
#pragma unroll
        for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = 0xFFFFFFFFu; }
    }
    // self-KNN: the group's 256 queries are box blockIdx.x; visit the boxes outwards from it so the K-th best
    // distance is tight before the far boxes are tested.  Foreign queries have no such order: from box 0 upwards.
    const int own = sorted_queries ? (int)(blockIdx.x * KNN_WAVE) / KNN_BOX : 0;  // wave-uniform
    for (int d = 0; own + d < nbox || own - d >= 0; d++) {
        for (int side = 0; side < 2; side++) {
            const int b = side == 0 ? own + d : own - d;
            if ((d == 0 && side == 1) || b < 0 || b >= nbox) continue;  // uniform
            const float4 lo = *reinterpret_cast<const float4*>(boxes + (size_t)b * 8);  // wave-uniform address
            const float4 hi = *reinterpret_cast<const float4*>(boxes + (size_t)b * 8 + 4);
            const float gx = (p.x < lo.x) ? (p.x - lo.x) : ((p.x > hi.x) ? (p.x - hi.x) : 0.f);
            const float gy = (p.y < lo.y) ? (p.y - lo.y) : ((p.y > hi.y) ? (p.y - hi.y) : 0.f);
            const float gz = (p.z < lo.z) ? (p.z - lo.z) : ((p.z > hi.z) ? (p.z - hi.z) : 0.f);
            const float dbox = gx * gx + gy * gy + gz * gz;
            const bool want = active && !(dbox > reject || dbox > bd[K - 1]);
            if (!__any(want)) continue;  // wave-uniform
            __syncthreads();  // (one wave: orders the previous scan before this refill)
            const int base = b * KNN_BOX, cnt = min(Nr - base, KNN_BOX);
            for (int e = threadIdx.x; e < cnt; e += KNN_WAVE) tile[e] = sp[base + e];
            __syncthreads();
            if (want)
                for (int i = 0; i < cnt; i++) {
                    const float4 o = tile[i];
                    const float dd = dist2(p, o);
                    const uint32_t id = __float_as_uint(o.w);
                    // most points are no better than the current K-th best: one compare instead of the K-deep
                    // insertion chain (which a wave only enters when one of its lanes has a candidate)
                    if (dd < bd[K - 1] || (dd == bd[K - 1] && id < bi[K - 1])) kbest_insert<K>(dd, id, bd, bi);
                }
        }
    }
    if (active)
#pragma unroll
        for (int k = 0; k < K; k++) {
            out_d[row * K + k] = bd[k];
            out_i[row * K + k] = (bi[k] == 0xFFFFFFFFu) ? -1ll : (long long)bi[k];  // fewer than K reference points: -1
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
    uint32_t* mm = (uint32_t*)(w + L.minmax);
    const int nb = (Nr + 255) / 256;
    StageScope st("knn_points", s);
    hipLaunchKernelGGL(knn_init_kernel, dim3(1), dim3(64), 0, s, mm);
    hipLaunchKernelGGL(knn_minmax_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, Nr, ref, mm);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, Nr, ref, mm, k0, v0);
    GS_LAUNCH_CHECK("knn.morton", 0, s);
    int rc = launch_sort_pairs(k0, v0, k1, v1, hist, Nr, 30, false, 0, s);
    if (rc != GS_OK) return rc;
    const uint32_t* order = (radix_passes(30) & 1) ? v1 : v0;
    hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(256), 0, s, Nr, ref, order, sp);
    hipLaunchKernelGGL(knn_box_kernel, dim3(L.nbox), dim3(KNN_BOX), 0, s, Nr, sp, boxes);
    const int self = (queries == ref && Nq == Nr) ? 1 : 0;
#define KNN_LAUNCH(KK)                                                                                                   \
    case KK:                                                                                                             \
        hipLaunchKernelGGL(knn_points_kernel<KK>, dim3((Nq + KNN_WAVE - 1) / KNN_WAVE), dim3(KNN_WAVE), 0, s, Nq, queries, self, Nr, L.nbox, sp, \
                           boxes, out_d, out_i);                                                                         \
        break;
    switch (K) {
        KNN_LAUNCH(1) KNN_LAUNCH(2) KNN_LAUNCH(3) KNN_LAUNCH(4) KNN_LAUNCH(5) KNN_LAUNCH(6) KNN_LAUNCH(7) KNN_LAUNCH(8)
        default: return GS_E_BAD_ARG;
    }
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
    uint32_t* mm = (uint32_t*)(w + L.minmax);
    const int nb = (P + 255) / 256;
    hipLaunchKernelGGL(knn_init_kernel, dim3(1), dim3(64), 0, s, mm);
    hipLaunchKernelGGL(knn_minmax_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, P, points, mm);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, P, points, mm, k0, v0);
    GS_LAUNCH_CHECK("knn.morton", 0, s);
    int rc = launch_sort_pairs(k0, v0, k1, v1, hist, P, 30, false, 0, s);
    if (rc != GS_OK) return rc;
    const uint32_t* order = (radix_passes(30) & 1) ? v1 : v0;
    hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(256), 0, s, P, points, order, sp);
    hipLaunchKernelGGL(knn_box_kernel, dim3(L.nbox), dim3(KNN_BOX), 0, s, P, sp, boxes);
    hipLaunchKernelGGL(knn_search_kernel, dim3((P + KNN_WAVE - 1) / KNN_WAVE), dim3(KNN_WAVE), 0, s, P, L.nbox, sp, boxes, out);
    GS_LAUNCH_CHECK("knn.search", 0, s);
    return GS_OK;
}
