// capi.hip -- the extern "C" boundary declared in include/gsplat_mi355.h.  Host-side sequencing of
// the stages; no device allocation, no implicit synchronisation (except in debug mode).
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>

static thread_local int g_last_hip = 0;
static thread_local const char* g_last_stage = "";
void gs_set_error(int hip_err, const char* stage) {
    g_last_hip = hip_err;
    g_last_stage = stage;
}

static std::atomic<int> g_tune[GS_TUNE_COUNT] = {};
static bool g_tune_init = false;
static void tune_defaults() {
    if (g_tune_init) return;
    g_tune_init = true;
    g_tune[GS_TUNE_XCD_MAP].store(1);
    g_tune[GS_TUNE_DEPTH_SORT].store(1);
    g_tune[GS_TUNE_NT_STORES].store(1);
    g_tune[GS_TUNE_BWD_CHUNKS].store(1);
    g_tune[GS_TUNE_FWD4].store(1);
    // (GSPLAT_FWD4=0|1|2: the forward of the tiles marked as long on small images -- one wave per quadrant, four waves x four
    // entries per step, or a wave per chunk of the list (render_fwd.hip: render_chunk); the same as gs_tuning("fwd4", v), read once)
    if (const char* e = getenv("GSPLAT_FWD4")) g_tune[GS_TUNE_FWD4].store(atoi(e));
    g_tune[GS_TUNE_FWDC_CH].store((int)FWDC_CH_MIN);  // entries per chunk of the chunk-parallel forward (a power of two >= 64)
    g_tune[GS_TUNE_FWDC_DIV].store((int)FWD4_TOTAL_DIV);  // ... of the tiles whose list is longer than (frame's pairs) / this (and than FWD4_MIN_LIST)
    if (const char* e = getenv("GSPLAT_FWDC_DIV")) { const int v = atoi(e); if (v >= 1) g_tune[GS_TUNE_FWDC_DIV].store(v); }
    if (const char* e = getenv("GSPLAT_FWDC_CH")) { const int v = atoi(e); if (v >= 64 && (v & (v - 1)) == 0) g_tune[GS_TUNE_FWDC_CH].store(v); }
    g_tune[GS_TUNE_SHARED_QLIST].store(1);
    g_tune[GS_TUNE_ONES_FAST].store(1);
    g_tune[GS_TUNE_SMALL_TILES].store(BWD_CHUNK_MAX_TILES);
    g_tune[GS_TUNE_BWD_ORDER].store(1);
    g_tune[GS_TUNE_FWD_MARKS].store(1);
}
int gs_tune_get(int key) {
    tune_defaults();
    return (key >= 0 && key < GS_TUNE_COUNT) ? g_tune[key].load(std::memory_order_relaxed) : 0;
}

// ---- per-stage event timing -------------------------------------------------------------------
#include <vector>
struct ProfRec { const char* name; hipEvent_t a, b; };
struct ProfState {
    bool on = false;
    bool armed = false;  // the current stage is being timed
    int nested = 0;      // stage scopes opened inside the one being timed (they belong to it: no record of their own)
    char filter[32] = {0};
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
// process-wide (autograd runs the backward on its own host thread), guarded by a mutex
#include <mutex>
#include <chrono>
#include <atomic>
static ProfState g_prof;
static std::mutex g_prof_mu;
static bool profiling_on() { return g_prof.on; }
void gs_prof_begin(const char* stage, hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.armed) { g_prof.nested++; return; }
    g_prof.armed = g_prof.on && (g_prof.filter[0] == 0 || strcmp(g_prof.filter, stage) == 0);
    if (!g_prof.armed) return;
    ProfRec r{stage, g_prof.get(), g_prof.get()};
    (void)hipEventRecord(r.a, s);
    g_prof.recs.push_back(r);
}
void gs_prof_end(hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.armed || g_prof.recs.empty()) return;
    if (g_prof.nested > 0) { g_prof.nested--; return; }
    (void)hipEventRecord(g_prof.recs.back().b, s);
    g_prof.armed = false;
}

static int validate(const GsFwdArgs* a) {
    if (!a) return GS_E_BAD_ARG;
    if (a->P < 0 || a->W <= 0 || a->H <= 0) return GS_E_BAD_ARG;
    if (!a->bg || !a->viewmatrix || !a->projmatrix || !a->campos) return GS_E_BAD_ARG;
    if (a->P > 0 && (!a->means3D || !a->opacities)) return GS_E_BAD_ARG;
    if ((a->shs != nullptr) == (a->colors_precomp != nullptr)) return GS_E_EXCLUSIVE;
    const bool sr = a->scales != nullptr && a->rotations != nullptr;
    if ((a->scales != nullptr) != (a->rotations != nullptr)) return GS_E_EXCLUSIVE;
    if (sr == (a->cov3D_precomp != nullptr)) return GS_E_EXCLUSIVE;
    if (a->shs) {
        if (a->sh_degree < 0 || a->sh_degree > 3) return GS_E_BAD_ARG;
        if (a->M < (a->sh_degree + 1) * (a->sh_degree + 1)) return GS_E_BAD_ARG;
        if (a->M > 16) return GS_E_BAD_ARG;  // degree <= 3: the per-Gaussian backward stages 3 M floats per thread in LDS
    }
    // quaternions are read (and their gradients written) as float4
    if (a->rotations && ((uintptr_t)a->rotations & 15u)) return GS_E_BAD_ARG;
    if ((a->W + TILE - 1) / TILE > 0xFFFF || (a->H + TILE - 1) / TILE > 0xFFFF) return GS_E_TOO_LARGE;
    return GS_OK;
}

// ---- stream capture (hipGraph) -----------------------------------------------------------------
// Round 3 recorded a GPU memory fault ("write access to a read-only page") on the first replay of a captured
// gs_forward_preprocess + gs_forward_render.  What those two calls put into the graph besides kernel nodes: a memset
// node (the per-tile totals), a device-to-host memcpy node into the caller's pinned count word, and a kernel that
// STORES into pinned host memory (GsFwdArgs.frame_stats).  None of that is needed under capture: the library now asks
// hipStreamIsCapturing at every entry point, enqueues KERNEL NODES ONLY on a capturing stream (zero_words_kernel instead
// of memset nodes), and answers GS_E_CAPTURE -- before enqueuing anything -- to every call that would need more.
static bool stream_is_capturing(void* stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const hipError_t e = hipStreamIsCapturing((hipStream_t)stream, &st);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // (asking about the legacy stream while another stream captures in global mode is itself a capture error)
        return e == hipErrorStreamCaptureImplicit || e == hipErrorStreamCaptureUnsupported || e == hipErrorStreamCaptureInvalidated;
    }
    return st != hipStreamCaptureStatusNone;
}
// capture-safe entry points: refuse debug mode (it synchronises) and the stage timer (event nodes) under capture
#define GS_CAPTURE_OK_IF(stream, cond)                                           \
    do {                                                                         \
        if (stream_is_capturing(stream) && (!(cond) || profiling_on())) return GS_E_CAPTURE; \
    } while (0)
#define GS_NO_CAPTURE(stream) GS_CAPTURE_OK_IF(stream, false)

__global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t words) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < words; k += (size_t)gridDim.x * blockDim.x) p[k] = 0u;
}
// `bytes` (a multiple of 4) of zeros as a kernel node (a memset node in eager mode would do: one code path for both)
int gs_zero_async(void* ptr, size_t bytes, const char* stage, hipStream_t s) {
    const size_t words = bytes / 4;
    if (words == 0) return GS_OK;
    const size_t blocks = (words + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, (uint32_t*)ptr, words);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gs_set_error((int)e, stage); return GS_E_HIP; }
    return GS_OK;
}

// shader clock under a VALU-bound load: an FMA stream on every SIMD; the first lane of every workgroup adds its
// s_memtime ticks (shader cycles) and s_memrealtime ticks (100 MHz) to out[0] / out[1]
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, float seed, float* __restrict__ sink) {
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + (float)i + (float)threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) sum += a[i];
    if (sum == 12345.678f) sink[0] = sum;  // (keeps the chains alive; never true in practice)
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&out[0], t1 - t0);
        atomicAdd(&out[1], r1 - r0);
    }
}

__global__ __launch_bounds__(64) void xcc_probe_kernel(uint32_t* __restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

extern "C" {

int gs_xcc_probe(uint32_t* xcc, int32_t n_blocks, void* stream) {
    if (!xcc || n_blocks <= 0) return GS_E_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3((unsigned)n_blocks), dim3(64), 0, s, xcc);
    GS_LAUNCH_CHECK("xcc_probe", 0, s);
    return GS_OK;
}

int gs_clock_probe(uint64_t* ticks, int32_t iters, void* stream) {
    if (!ticks || iters <= 0) return GS_E_BAD_ARG;
    GS_NO_CAPTURE(stream);
    hipStream_t s = (hipStream_t)stream;
    int rc = gs_zero_async(ticks, 32, "clock_probe.zero", s);
    if (rc != GS_OK) return rc;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(2048), dim3(256), 0, s, (unsigned long long*)ticks, iters, 1.0f, (float*)(ticks + 2));
    GS_LAUNCH_CHECK("clock_probe", 0, s);
    return GS_OK;
}

int gs_geom_bytes(int32_t P, size_t* out) {
    if (!out || P < 0) return GS_E_BAD_ARG;
    *out = geom_layout(P).total;
    return GS_OK;
}
int gs_image_bytes(int32_t W, int32_t H, size_t* out) {
    if (!out || W <= 0 || H <= 0) return GS_E_BAD_ARG;
    *out = img_layout(W, H).total;
    return GS_OK;
}
int gs_image_bytes_for(const GsFwdArgs* a, size_t* out) {
    if (!a || !out || a->W <= 0 || a->H <= 0) return GS_E_BAD_ARG;
    *out = img_layout(a->W, a->H, a->long_lists).total;
    return GS_OK;
}
int gs_binning_bytes(int64_t D, int32_t W, int32_t H, size_t* out) {
    if (!out || D < 0 || W <= 0 || H <= 0) return GS_E_BAD_ARG;
    if (D > GS_MAX_PAIRS) return GS_E_TOO_LARGE;
    *out = bin_layout(D).total;
    return GS_OK;
}
int gs_backward_scratch_bytes(int64_t D, int32_t P, int32_t W, int32_t H, size_t* out) {
    if (!out || D < 0 || P < 0 || W <= 0 || H <= 0) return GS_E_BAD_ARG;
    const ImgLayout I = img_layout(W, H);
    *out = scratch_total_bytes(D, P, I.gx * I.gy);
    return GS_OK;
}

// `poll`: a pinned host word the scan kernel writes the count into directly (gs_forward spins on it)
static int forward_phase1(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* img, size_t img_bytes, int32_t* radii,
                          int64_t* count_host_pinned, unsigned long long* poll, void* stream) {
    int rc = validate(a);
    if (rc != GS_OK) return rc;
    if (!geom || !img || (a->P > 0 && !radii)) return GS_E_BAD_ARG;
    const GeomLayout L = geom_layout(a->P);
    const ImgLayout I = img_layout(a->W, a->H, a->long_lists);
    if (geom_bytes < L.total || img_bytes < I.total) return GS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* g = (char*)geom;
    unsigned long long* count = (unsigned long long*)(g + L.count);
    if (a->P == 0) {
        rc = gs_zero_async(count, 24, "count.zero", s);
        if (rc != GS_OK) return rc;
    } else {
        uint32_t* k0 = (uint32_t*)(g + L.key0);
        uint32_t* k1 = (uint32_t*)(g + L.key1);
        uint32_t* v0 = (uint32_t*)(g + L.val0);
        uint32_t* v1 = (uint32_t*)(g + L.val1);
        ZeroJob zt;  // the depth sort's digit totals, cleared by the preprocess kernel
        sort_totals_region((uint32_t*)(g + L.hist), a->P, 32, &zt.ptr, &zt.words);
        // ... and the image state's per-tile pair totals, which phase 2's counting pass adds into
        const ZeroJob zi{(uint32_t*)((char*)img + I.tile_tot), (int)(I.tile_zero_bytes / 4)};
        { StageScope sc_("preprocess", s);
        rc = launch_preprocess(*a, (float*)(g + L.rec), (float*)(g + L.depths), (uint32_t*)(g + L.tiles),
                               (uint32_t*)(g + L.clamped), k0, v0, radii, (uint32_t*)(g + L.wsum), (uint32_t*)(g + L.wkmin),
                               (uint32_t*)(g + L.wkmax), zt, zi, s); }
        if (rc != GS_OK) return rc;
        // pair numbering (Gaussian-major, index order) and the pair count need nothing of the depth sort, so the count is
        // on its way to the host while the sort runs; (depth key, index) order: ties keep ascending Gaussian index (the
        // reference's tie order); the ranking ends in v0 and, with what binning needs of every Gaussian, in the rank list
        if (gs_tune_get(GS_TUNE_DEPTH_SORT)) {
            StageScope sc_("depth_sort", s);  // the numbering rides in its first launch, the rank list in its last ones
            const DepthSortState st{(unsigned long long*)(g + L.ds_tmp), (unsigned long long*)(g + L.ds_tmp2),
                                    (uint32_t*)(g + L.ds_cnt), (uint32_t*)(g + L.ds_pre),
                                    (uint32_t*)(g + L.ds_tot),
                                    (uint32_t*)(g + L.ds_loc), (uint32_t*)(g + L.ds_grp), (uint32_t*)(g + L.ds_range), L.ds_nb,
                                    L.ds_blocks};
            const PairNumbering pn{(const uint32_t*)(g + L.tiles), (const uint32_t*)(g + L.wsum), (float*)(g + L.rec), count, poll,
                                   (uint32_t*)(g + L.chunk_pairs), (a->P + 255) / 256};
            const RankOut ro{(const float*)(g + L.rec), (const uint32_t*)(g + L.tiles), v0, (uint4*)(g + L.ranklist),
                             (uint32_t*)(g + L.chunk_pairs)};
            rc = launch_depth_sort(k0, (const uint32_t*)(g + L.wkmin), (const uint32_t*)(g + L.wkmax), L.nwaves, a->P, st, pn, ro,
                                   a->debug, s);
            if (rc != GS_OK) return rc;
        } else {  // the LSD radix sort (gs_tuning "depth_sort" = 0): 4 passes, ends in (k0, v0)
            { StageScope sc_("pair_scan", s);
            rc = launch_first_pair((const uint32_t*)(g + L.tiles), (const uint32_t*)(g + L.wsum), (float*)(g + L.rec), count, poll,
                                   a->P, a->debug, s); }
            if (rc != GS_OK) return rc;
            { StageScope sc_("depth_sort", s);
            rc = launch_sort_pairs(k0, v0, k1, v1, (uint32_t*)(g + L.hist), a->P, 32, true, a->debug, s); }
            if (rc != GS_OK) return rc;
            { StageScope sc_("rank_list", s);
            rc = launch_rank_list(v0, (const float*)(g + L.rec), (const uint32_t*)(g + L.tiles), (uint4*)(g + L.ranklist),
                                  (uint32_t*)(g + L.chunk_pairs), a->P, a->debug, s); }
            if (rc != GS_OK) return rc;
        }
    }
    if (count_host_pinned && (!poll || a->P == 0)) {
        hipError_t e = hipMemcpyAsync(count_host_pinned, count, 8, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) { gs_set_error((int)e, "count.copy"); return GS_E_HIP; }
    }
    return GS_OK;
}

int gs_forward_preprocess(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* img, size_t img_bytes,
                          int32_t* radii, int64_t* count_host_pinned, void* stream) {
    GS_CAPTURE_OK_IF(stream, a && !a->debug && count_host_pinned == nullptr);
    return forward_phase1(a, geom, geom_bytes, img, img_bytes, radii, count_host_pinned, nullptr, stream);
}

// Phase 2 against a binning state carved for `cap` pairs.  The kernels read the frame's pair count from the geom state
// on the device (PairCount): the phase can be enqueued before the host knows the count; a count beyond `cap` makes every
// kernel do the work of an empty frame (nothing out of bounds) and the caller runs the phase again with a larger state.
// `totals_zeroed`: phase 1 has just run on this image state (its preprocess kernel cleared the per-tile pair totals).
static int forward_phase2(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* img,
                          size_t img_bytes, int64_t cap, float* out_color, void* stream, bool totals_zeroed) {
    if (!geom || !img || !out_color || cap < 0 || (cap > 0 && !binning)) return GS_E_BAD_ARG;
    if (a->l1_target && !a->l1_loss) return GS_E_BAD_ARG;  // the fused L1 loss needs somewhere to put its value
    if (cap > GS_MAX_PAIRS) return GS_E_TOO_LARGE;
    const GeomLayout L = geom_layout(a->P);
    const ImgLayout I = img_layout(a->W, a->H, a->long_lists);
    const BinLayout B = bin_layout(cap);
    if (geom_bytes < L.total || img_bytes < I.total || (cap > 0 && binning_bytes < B.total)) return GS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* g = (char*)geom;
    char* b = (char*)binning;
    char* im = (char*)img;
    const int ntiles = I.gx * I.gy;
    uint32_t* ranges = (uint32_t*)(im + I.ranges);
    const uint32_t* point_list = nullptr;
    const unsigned long long* count_dev = (const unsigned long long*)(g + L.count);
    const PairCount pc{count_dev, (uint32_t)cap};
    const bool chunked = forward_chunked(ntiles, a->long_lists);  // the marked tiles chunk-parallel (render_fwd.hip: render_chunk)
    int rc;
    if (a->P > 0) {
        point_list = cap > 0 ? (const uint32_t*)(b + B.point_list) : nullptr;
        rc = launch_tile_lists((const uint4*)(g + L.ranklist), (const uint32_t*)(g + L.chunk_pairs), (uint32_t*)(g + L.seg_start),
                               a->P, I.gx, I.gy,
                               TileCounts{(uint32_t*)(im + I.seg_cnt), (uint32_t*)(im + I.tile_tot), I.tile_zero_bytes,
                                          chunked ? (uint32_t*)(im + I.cw_q) + 8 * FWDC_MAX_UNITS * 4 : nullptr},
                               ranges, (uint32_t*)(im + I.order),
                               cap > 0 ? (uint32_t*)(b + B.point_list) : nullptr, pc,
                               LongLists{chunked ? 2 : (forward_small_image(ntiles, a->long_lists) ? 1 : 0), (long long*)a->frame_stats,
                                         chunked ? (uint32_t*)(im + I.cw_hdr) : nullptr, chunked ? (uint2*)(im + I.cw_units) : nullptr,
                                         (uint32_t)gs_tune_get(GS_TUNE_FWDC_CH), chunked ? (uint32_t*)(im + I.cw_items) : nullptr,
                                         chunked ? (const uint32_t*)(im + I.cw_q) + 8 * FWDC_MAX_UNITS * 4 : nullptr,
                                         chunked ? (uint32_t)gs_tune_get(GS_TUNE_FWDC_DIV) : (uint32_t)FWD4_TOTAL_DIV},
                               totals_zeroed, a->debug, s);
        if (rc != GS_OK) return rc;
    } else {
        rc = gs_zero_async(ranges, (size_t)ntiles * 8, "ranges.zero", s);
        if (rc != GS_OK) return rc;
        StageScope sc_("ranges_order", s);
        rc = launch_tile_order(ranges, nullptr, 0, ntiles, (uint32_t*)(im + I.order), pc, FillJob{nullptr, 0},
                               LongLists{0, nullptr, chunked ? (uint32_t*)(im + I.cw_hdr) : nullptr, nullptr}, a->debug, s);
        if (rc != GS_OK) return rc;
    }
    QuadLists ql;
    ql.qlist = cap > 0 ? (uint32_t*)(b + B.qlist) : nullptr;
    ql.ncon_c = (uint32_t*)(im + I.ncon_c);
    ql.qcount = (uint32_t*)(im + I.tile_nmax);
    ql.chunks = gs_tune_get(GS_TUNE_BWD_CHUNKS) ? I.bwd_chunks : 1;
    ql.four_waves = forward_small_image(I.gx * I.gy, a->long_lists) ? 1 : 0;
    if (chunked) {
        ql.chunked = 1;
        ql.cw_hdr = (const uint32_t*)(im + I.cw_hdr);
        ql.cw_units = (const uint2*)(im + I.cw_units);
        ql.cw_items = (const uint32_t*)(im + I.cw_items);
        ql.cw_q = (uint32_t*)(im + I.cw_q);
        ql.cw_flag = (uint32_t*)(im + I.cw_flag);
        ql.cw_done = (uint32_t*)(im + I.cw_done);
        ql.cw_rec = (float*)(im + I.cw_rec);
    }
    ql.ckpt = ql.chunks > 1 ? (float4*)(im + I.ckpt) : nullptr;
    ql.ck_start = ql.chunks > 1 ? (uint32_t*)(im + I.ck_start) : nullptr;
    if (cap > 0) {  // the backward's row marks, set on the side by the render launch (BinLayout::marks) -- or declared unset
        // (not for a frame no backward can follow -- GsFwdArgs.forward_only: a frame rendered under no_grad)
        ql.marks = (gs_tune_get(GS_TUNE_FWD_MARKS) && !a->forward_only) ? (uint4*)(b + B.marks) : nullptr;
        ql.mark_quads = (size_t)cap;
        ql.marks_flag = (uint32_t*)(b + B.marks_flag);
    }
    ql.l1_target = a->l1_target;  // the fused L1 loss rides in the render launch (+ its one-workgroup final sum)
    ql.l1_part = (float*)(im + I.l1_part);
    ql.l1_loss = a->l1_loss;
    { StageScope sc_("render_fwd", s);
    rc = launch_render_forward((const float*)(g + L.rec), point_list, ranges, (const uint32_t*)(im + I.order), a->bg,
                               a->W, a->H, out_color, (float*)(im + I.final_T), (uint32_t*)(im + I.n_contrib), ql, s); }
    if (rc != GS_OK) return rc;
    if (a->debug) {
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { gs_set_error((int)e, "render_forward"); return GS_E_HIP; }
    }
    return GS_OK;
}

int gs_forward_render(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                      void* img, size_t img_bytes, int64_t D, float* out_color, void* stream) {
    GS_CAPTURE_OK_IF(stream, a && !a->debug && a->frame_stats == nullptr);
    int rc = validate(a);
    if (rc != GS_OK) return rc;
    return forward_phase2(a, geom, geom_bytes, binning, binning_bytes, img, img_bytes, D, out_color, stream, false);
}

// Both phases in one call, WITHOUT a GPU idle stretch for the pair count.  The binning state the caller passes was
// carved for `capacity` pairs (gs_binning_bytes(capacity): the caller sizes it from the previous frame's count).
// Phase 2 is enqueued right behind phase 1, its kernels reading the count on the device; only then does the host wait
// for the count -- the scan kernel stores it straight into the caller's pinned word -- while the GPU already works on
// phase 2.  If the count turns out larger than the capacity, phase 2 has rendered an empty frame into the buffers
// (nothing out of bounds): GS_E_WORKSPACE is returned with *num_rendered set and the caller runs gs_forward_render with
// a state of the right size.  With capacity 0 (no estimate yet) only phase 1 runs and GS_E_WORKSPACE is returned.
int gs_forward(const GsFwdArgs* a, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, int64_t capacity,
               void* img, size_t img_bytes, int32_t* radii, int64_t* count_host_pinned, float* out_color,
               int64_t* num_rendered, void* stream) {
    if (!count_host_pinned || !num_rendered || capacity < 0) return GS_E_BAD_ARG;
    GS_NO_CAPTURE(stream);  // (the host waits for the pair count: gs_forward_preprocess + gs_forward_render are the capturable form)
    volatile int64_t* word = count_host_pinned;
    const int64_t pending = -1;
    static std::atomic<bool> poll_works{true};  // cleared for the process if a device write to the word is ever not seen
    const bool poll = a && a->P > 0 && poll_works.load(std::memory_order_relaxed);
    if (poll) *word = pending;
    int rc = forward_phase1(a, geom, geom_bytes, img, img_bytes, radii, count_host_pinned,
                            poll ? (unsigned long long*)count_host_pinned : nullptr, stream);
    if (rc != GS_OK) return rc;
    const bool speculate = capacity > 0 && binning && binning_bytes >= bin_layout(capacity).total && capacity <= GS_MAX_PAIRS;
    if (speculate) {
        rc = forward_phase2(a, geom, geom_bytes, binning, binning_bytes, img, img_bytes, capacity, out_color, stream, true);
        if (rc != GS_OK) return rc;
    }
    bool have = false;
    if (poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (long spin = 0;; spin++) {
            if (*word != pending) { have = true; break; }
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
    }
    if (!have) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) { gs_set_error((int)e, "count.sync"); return GS_E_HIP; }
        if (poll && *word == pending) {
            // the host word is not device-visible memory on this system: read the count from the geom state and
            // use the copy + synchronise form from now on
            poll_works.store(false, std::memory_order_relaxed);
            int64_t c = 0;
            e = hipMemcpy(&c, (const char*)geom + geom_layout(a->P).count, 8, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { gs_set_error((int)e, "count.copy"); return GS_E_HIP; }
            *word = c;
        }
    }
    const int64_t D = *word;
    *num_rendered = D;
    if (D > (int64_t)GS_MAX_PAIRS) return GS_E_TOO_LARGE;
    if (!speculate || D > capacity) return GS_E_WORKSPACE;  // caller sizes the state for D, then gs_forward_render
    return GS_OK;
}

int gs_forward_shared(const GsFwdArgs* a, const void* geom_src, const void* img_src, void* geom, size_t geom_bytes,
                      void* binning, size_t binning_bytes, void* img, size_t img_bytes, int64_t D, float* out_color,
                      void* stream) {
    GS_CAPTURE_OK_IF(stream, a && !a->debug && a->P > 0);
    int rc = validate(a);
    if (rc != GS_OK) return rc;
    if (!geom_src || !img_src || !geom || !img || !out_color || D < 0 || (D > 0 && !binning)) return GS_E_BAD_ARG;
    const GeomLayout L = geom_layout(a->P);
    const ImgLayout I = img_layout(a->W, a->H, a->long_lists);
    const BinLayout B = bin_layout(D);
    if (geom_bytes < L.total || img_bytes < I.total || (D > 0 && binning_bytes < B.total)) return GS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const char* gs = (const char*)geom_src;
    const char* is = (const char*)img_src;
    char* g = (char*)geom;
    char* b = (char*)binning;
    char* im = (char*)img;
    const int ntiles = I.gx * I.gy;
    const bool chunked = forward_chunked(ntiles, a->long_lists);
    unsigned long long* not_ones = nullptr;
    if (a->P > 0) {
        StageScope sc_("recolor", s);
        // (the first render's geom state carries the "not all ones" word, zero since that render; the caller passes the
        // SAME long_lists as to the first render, so that both image states have the same layout)
        not_ones = (gs_tune_get(GS_TUNE_ONES_FAST) && gs_tune_get(GS_TUNE_SHARED_QLIST) && a->colors_precomp && D > 0)
                       ? (unsigned long long*)(const_cast<char*>(gs) + L.count) + 2 : nullptr;
        // ... and, when the colours may turn out to be all ones, the launch's other workgroups write the image that is
        // right in that case (1 - T of the first render, with its records): the render launch below then leaves at once
        const int chunks_ = gs_tune_get(GS_TUNE_BWD_CHUNKS) ? I.bwd_chunks : 1;
        const SecondOnes so{(const float*)(is + I.final_T), (const uint32_t*)(is + I.n_contrib), (const uint32_t*)(is + I.ncon_c),
                            (const uint32_t*)(is + I.tile_nmax), (const float4*)(is + I.ckpt), (const uint32_t*)(is + I.ck_start),
                            out_color, (float*)(im + I.final_T), (uint32_t*)(im + I.n_contrib), (uint32_t*)(im + I.ncon_c),
                            (uint32_t*)(im + I.tile_nmax), chunks_ > 1 ? (float4*)(im + I.ckpt) : nullptr,
                            chunks_ > 1 ? (uint32_t*)(im + I.ck_start) : nullptr, a->bg, a->W, a->H, I.gx, ntiles,
                            chunks_ > 1 ? chunks_ : 1};
        rc = launch_recolor(*a, (const float*)(gs + L.rec), (const uint32_t*)(gs + L.tiles), (float*)(g + L.rec),
                            (uint32_t*)(g + L.tiles), (uint32_t*)(g + L.clamped), not_ones,
                            CopyJob{(const uint32_t*)(is + I.ranges), (uint32_t*)(im + I.ranges), ntiles * 2},
                            CopyJob{(const uint32_t*)(is + I.order), (uint32_t*)(im + I.order), ntiles},
                            chunked ? ZeroJob{(uint32_t*)(im + I.cw_flag), (int)((I.cw_q + (size_t)8 * FWDC_MAX_UNITS * 4 * 4 + 64 - I.cw_flag) / 4)} : ZeroJob{nullptr, 0},
                            not_ones ? &so : nullptr, s);
        if (rc != GS_OK) return rc;
    } else {
        // (no Gaussians: no recolouring launch to ride in) the new image state's own copy of the tile ranges and launch order
        hipError_t e = hipMemcpyAsync(im + I.ranges, is + I.ranges, (size_t)ntiles * 8, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(im + I.order, is + I.order, (size_t)ntiles * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { gs_set_error((int)e, "shared.copy"); return GS_E_HIP; }
    }
    hipError_t e = hipSuccess;
    const uint32_t* point_list = D > 0 ? (const uint32_t*)(b + B.point_list) : nullptr;
    QuadLists ql;
    ql.qlist = D > 0 ? (uint32_t*)(b + B.qlist) : nullptr;  // read, not rewritten: the recorded quadrant lists are what is walked
    ql.ncon_c = (uint32_t*)(im + I.ncon_c);
    ql.qcount = (uint32_t*)(im + I.tile_nmax);
    if (gs_tune_get(GS_TUNE_SHARED_QLIST)) {
        ql.src_qcount = (const uint32_t*)(is + I.tile_nmax);  // (the fields before the checkpoints sit at the same offsets
        ql.src_n_contrib = (const uint32_t*)(is + I.n_contrib);  //  whatever long_lists the first render was given)
        ql.not_ones = not_ones;
    }
    ql.chunks = gs_tune_get(GS_TUNE_BWD_CHUNKS) ? I.bwd_chunks : 1;
    ql.four_waves = forward_small_image(I.gx * I.gy, a->long_lists) ? 1 : 0;
    if (chunked && a->P > 0) {
        // the first render's work list (the copied launch order carries its marks); this state's own hand-off words
        // (cleared by the recolouring launch) and records; walking the recorded lists also the first render's records
        ql.chunked = 1;
        ql.cw_hdr = (const uint32_t*)(is + I.cw_hdr);
        ql.cw_units = (const uint2*)(is + I.cw_units);
        ql.cw_items = (const uint32_t*)(is + I.cw_items);
        ql.cw_q = (uint32_t*)(im + I.cw_q);
        ql.cw_flag = (uint32_t*)(im + I.cw_flag);
        ql.cw_done = (uint32_t*)(im + I.cw_done);
        ql.cw_rec = (float*)(im + I.cw_rec);
        if (ql.src_qcount) {
            ql.src_cw_flag = (const uint32_t*)(is + I.cw_flag);
            ql.src_cw_rec = (const float*)(is + I.cw_rec);
            ql.src_final_T = (const float*)(is + I.final_T);
        }
    }
    ql.ckpt = ql.chunks > 1 ? (float4*)(im + I.ckpt) : nullptr;
    ql.ck_start = ql.chunks > 1 ? (uint32_t*)(im + I.ck_start) : nullptr;
    ql.all_ones = (uint32_t*)(im + I.all_ones);  // (set by the render launch: 1 iff it left the speculative image alone)
    { StageScope sc_("render_fwd", s);
    rc = launch_render_forward((const float*)(g + L.rec), point_list, (const uint32_t*)(im + I.ranges),
                               (const uint32_t*)(im + I.order), a->bg, a->W, a->H, out_color, (float*)(im + I.final_T),
                               (uint32_t*)(im + I.n_contrib), ql, s); }
    if (rc != GS_OK) return rc;
    if (a->debug) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { gs_set_error((int)e, "render_forward"); return GS_E_HIP; }
    }
    return GS_OK;
}

int gs_opacity_image(const GsFwdArgs* a, const void* img, size_t img_bytes, float* opacity, void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    int rc = validate(a);
    if (rc != GS_OK) return rc;
    if (!img || !opacity) return GS_E_BAD_ARG;
    const ImgLayout I = img_layout(a->W, a->H, a->long_lists);
    if (img_bytes < I.total) return GS_E_WORKSPACE;
    return launch_opacity_image((const float*)((const char*)img + I.final_T), a->bg, a->W, a->H, opacity, (hipStream_t)stream);
}

static int backward_impl(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes, void* binning,
                         size_t binning_bytes, const void* img, size_t img_bytes, int64_t D, const float* out_color,
                         const float* dL_dpix, const float* dL_dopacity_img, void* scratch, size_t scratch_bytes,
                         const GsGrads* gr, void* stream, const GsSecondImage* second = nullptr) {
    GS_CAPTURE_OK_IF(stream, a && !a->debug);
    int rc = validate(a);
    if (rc != GS_OK) return rc;
    if (!geom || !img || !out_color || (!dL_dpix && !a->l1_target) || !gr || D < 0 || (D > 0 && !binning) || (a->P > 0 && !scratch))
        return GS_E_BAD_ARG;
    if (a->P > 0 && (!radii || !gr->dL_dmeans3D || !gr->dL_dmeans2D || !gr->dL_dcolors || !gr->dL_dopacity || !gr->dL_dcov3D))
        return GS_E_BAD_ARG;
    if (a->P > 0 && a->shs && !gr->dL_dsh) return GS_E_BAD_ARG;
    if (a->P > 0 && a->scales && (!gr->dL_dscales || !gr->dL_drotations)) return GS_E_BAD_ARG;
    if (gr->dL_drotations && ((uintptr_t)gr->dL_drotations & 15u)) return GS_E_BAD_ARG;  // written as float4
    const GeomLayout L = geom_layout(a->P);
    const ImgLayout I = img_layout(a->W, a->H, a->long_lists);
    const BinLayout B = bin_layout(D);
    size_t need = 0;
    gs_backward_scratch_bytes(D, a->P, a->W, a->H, &need);
    if (geom_bytes < L.total || img_bytes < I.total || (D > 0 && binning_bytes < B.total) || (a->P > 0 && scratch_bytes < need))
        return GS_E_WORKSPACE;
    if (a->P == 0) return GS_OK;
    hipStream_t s = (hipStream_t)stream;
    const char* g = (const char*)geom;
    char* b = (char*)binning;
    const char* im = (const char*)img;
    if (D > 0) {
        QuadLists ql;
        ql.qlist = (uint32_t*)(b + B.qlist);
        ql.ncon_c = (uint32_t*)(im + I.ncon_c);
        ql.qcount = (uint32_t*)(im + I.tile_nmax);
        ql.chunks = gs_tune_get(GS_TUNE_BWD_CHUNKS) ? I.bwd_chunks : 1;
        ql.ckpt = ql.chunks > 1 ? (float4*)(im + I.ckpt) : nullptr;
        ql.ck_start = ql.chunks > 1 ? (uint32_t*)(im + I.ck_start) : nullptr;
        // the row marks: in the binning state, where the forward has set every word to ROW_UNWRITTEN beside its render
        // kernel; if a backward has run on this state since (marks_flag), the tile-order launch's other workgroups do it
        uint32_t* q8 = (uint32_t*)(b + B.marks);
        uint32_t* marks_flag = (uint32_t*)(b + B.marks_flag);
        uint32_t* order_b = (uint32_t*)((char*)scratch + scratch_rows_bytes(D) + scratch_sums_bytes(a->P));
        const bool own_order = gs_tune_get(GS_TUNE_BWD_ORDER) != 0;
        { StageScope sc_("tile_order", s);
        rc = launch_tile_order((const uint32_t*)(im + I.ranges), ql.qcount, own_order ? 1 : -1, I.gx * I.gy, order_b, PairCount{nullptr, 0},
                               FillJob{reinterpret_cast<uint4*>(q8), (size_t)D, gs_tune_get(GS_TUNE_NT_STORES) & 1, marks_flag},
                               LongLists{0, nullptr}, a->debug, s); }
        if (rc != GS_OK) return rc;
        if (!own_order) order_b = (uint32_t*)(const_cast<char*>(im) + I.order);
        SecondImage si{nullptr, nullptr, nullptr, nullptr, nullptr};
        if (second) {
            // the second render's own image state: its checkpoints (same chunk boundaries: same geometry, same rule)
            const ImgLayout I2 = img_layout(a->W, a->H, second->long_lists);
            if (second->img_bytes < I2.total || I2.bwd_chunks != I.bwd_chunks) return GS_E_BAD_ARG;
            si = SecondImage{second->colors, second->out_color, second->dL_dpix,
                             ql.chunks > 1 ? (const float4*)((const char*)second->img + I2.ckpt) : nullptr,
                             (const uint32_t*)((const char*)second->img + I2.all_ones)};
        }
        { StageScope sc_("render_bwd", s);
        rc = launch_render_backward((const float*)(g + L.rec), (const uint32_t*)(im + I.ranges), order_b, a->W, a->H, ql,
                                    out_color, dL_dpix, dL_dopacity_img, (const float*)(im + I.final_T), a->bg, (float*)scratch, q8,
                                    second ? &si : nullptr, L1Grad{a->l1_target, a->l1_grad}, s); }
        if (rc != GS_OK) return rc;
        if (a->debug) {
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) { gs_set_error((int)e, "render_backward"); return GS_E_HIP; }
        }
    }
    StageScope sc_("gaussian_bwd", s);
    return launch_gaussian_backward(*a, radii, (const float*)(g + L.rec), (const uint32_t*)(g + L.tiles),
                                    (const uint32_t*)(g + L.clamped), D > 0 ? (const uint32_t*)(b + B.marks) : nullptr,
                                    (const float*)scratch, (float*)((char*)scratch + scratch_rows_bytes(D)),
                                    D > 0 ? (uint32_t*)(b + B.marks_flag) : nullptr, *gr, s);
}

int gs_backward(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes, void* binning,
                size_t binning_bytes, const void* img, size_t img_bytes, int64_t D, const float* out_color,
                const float* dL_dpix, void* scratch, size_t scratch_bytes, const GsGrads* gr, void* stream) {
    return backward_impl(a, radii, geom, geom_bytes, binning, binning_bytes, img, img_bytes, D, out_color, dL_dpix, nullptr,
                         scratch, scratch_bytes, gr, stream);
}

int gs_backward_with_opacity(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes,
                             void* binning, size_t binning_bytes, const void* img, size_t img_bytes, int64_t D,
                             const float* out_color, const float* dL_dpix, const float* dL_dopacity_img, void* scratch,
                             size_t scratch_bytes, const GsGrads* gr, void* stream) {
    if (!dL_dopacity_img) return GS_E_BAD_ARG;
    return backward_impl(a, radii, geom, geom_bytes, binning, binning_bytes, img, img_bytes, D, out_color, dL_dpix,
                         dL_dopacity_img, scratch, scratch_bytes, gr, stream);
}

int gs_backward_with_second(const GsFwdArgs* a, const int32_t* radii, const void* geom, size_t geom_bytes,
                            void* binning, size_t binning_bytes, const void* img, size_t img_bytes, int64_t D,
                            const float* out_color, const float* dL_dpix, const GsSecondImage* second, void* scratch,
                            size_t scratch_bytes, const GsGrads* grads, void* stream) {
    if (!second || !second->colors || !second->out_color || !second->dL_dpix || !second->img) return GS_E_BAD_ARG;
    return backward_impl(a, radii, geom, geom_bytes, binning, binning_bytes, img, img_bytes, D, out_color, dL_dpix, nullptr,
                         scratch, scratch_bytes, grads, stream, second);
}

int gs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    uint8_t* present, void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    (void)projmatrix;
    if (P < 0 || !viewmatrix || (P > 0 && (!means3D || !present))) return GS_E_BAD_ARG;
    if (P == 0) return GS_OK;
    return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

int knn_workspace_bytes(int32_t P, size_t* out) {
    if (!out || P < 0) return GS_E_BAD_ARG;
    *out = knn_ws_bytes(P);
    return GS_OK;
}
int knn_dist2(int32_t P, const float* points, float* mean_d2, void* workspace, size_t workspace_bytes, void* stream) {
    GS_NO_CAPTURE(stream);  // (the sorts clear their tables with memset nodes: untested under replay)
    if (P < 0 || (P > 0 && (!points || !mean_d2 || !workspace))) return GS_E_BAD_ARG;
    if (P == 0) return GS_OK;
    return launch_knn(P, points, mean_d2, workspace, workspace_bytes, (hipStream_t)stream);
}

int gs_build_covariance(int32_t N, const float* scaling, float scaling_modifier, const float* rotation, int32_t rotation_is_matrix,
                        float* cov6, void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (N < 0 || (N > 0 && (!scaling || !rotation || !cov6))) return GS_E_BAD_ARG;
    if (!rotation_is_matrix && ((uintptr_t)rotation & 15u)) return GS_E_BAD_ARG;  // quaternions are read as float4
    if (N == 0) return GS_OK;
    return launch_build_cov(N, scaling, scaling_modifier, rotation, rotation_is_matrix ? 1 : 0, cov6, (hipStream_t)stream);
}
int gs_build_covariance_backward(int32_t N, const float* scaling, float scaling_modifier, const float* rotation,
                                 int32_t rotation_is_matrix, const float* dL_dcov6, float* dL_dscaling, float* dL_drotation,
                                 void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (N < 0 || (N > 0 && (!scaling || !rotation || !dL_dcov6 || !dL_dscaling || !dL_drotation))) return GS_E_BAD_ARG;
    if (!rotation_is_matrix && (((uintptr_t)rotation | (uintptr_t)dL_drotation) & 15u)) return GS_E_BAD_ARG;
    if (N == 0) return GS_OK;
    return launch_build_cov_bwd(N, scaling, scaling_modifier, rotation, rotation_is_matrix ? 1 : 0, dL_dcov6, dL_dscaling,
                                dL_drotation, (hipStream_t)stream);
}
int gs_sh2rgb(int32_t N, int32_t sh_degree, int32_t M, const float* shs, const float* xyz, const float* campos,
              const float* fwd_rotation, const float* view_noise_host, float* colors, uint8_t* clamped, void* stream) {
    GS_CAPTURE_OK_IF(stream, view_noise_host == nullptr);  // (a host matrix would be baked into the graph)
    if (N < 0 || sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16) return GS_E_BAD_ARG;
    if (N > 0 && (!shs || !xyz || !campos || !colors || !clamped)) return GS_E_BAD_ARG;
    if (N == 0) return GS_OK;
    return launch_sh2rgb(N, sh_degree, M, shs, xyz, campos, fwd_rotation, view_noise_host, colors, clamped, (hipStream_t)stream);
}
int gs_sh2rgb_backward(int32_t N, int32_t sh_degree, int32_t M, const float* shs, const float* xyz, const float* campos,
                       const float* fwd_rotation, const float* view_noise_host, const uint8_t* clamped,
                       const float* dL_dcolors, float* dL_dshs, float* dL_dxyz, void* stream) {
    GS_CAPTURE_OK_IF(stream, view_noise_host == nullptr);
    if (N < 0 || sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16) return GS_E_BAD_ARG;
    if (N > 0 && (!shs || !xyz || !campos || !clamped || !dL_dcolors || !dL_dshs || !dL_dxyz)) return GS_E_BAD_ARG;
    if (N == 0) return GS_OK;
    return launch_sh2rgb_bwd(N, sh_degree, M, shs, xyz, campos, fwd_rotation, view_noise_host, clamped, dL_dcolors, dL_dshs,
                             dL_dxyz, (hipStream_t)stream);
}

int gs_l1_loss_workspace_bytes(int64_t n, size_t* out) {
    if (!out || n < 0) return GS_E_BAD_ARG;
    *out = l1_ws_bytes(n);
    return GS_OK;
}
int gs_l1_loss(int64_t n, const float* x, const float* y, float* loss, float* dL_dx, void* workspace, size_t workspace_bytes,
               void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (n <= 0 || !x || !y || !loss || !dL_dx || !workspace) return GS_E_BAD_ARG;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)dL_dx) & 15u) return GS_E_BAD_ARG;  // float4 accesses
    if (workspace_bytes < l1_ws_bytes(n)) return GS_E_WORKSPACE;
    return launch_l1_loss(x, y, n, loss, dL_dx, (float*)workspace, (hipStream_t)stream);
}

int gs_bce_loss(int64_t n, const float* x, const float* y, float* loss, float* dL_dx, void* workspace, size_t workspace_bytes,
                void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (n <= 0 || !x || !y || !loss || !dL_dx || !workspace) return GS_E_BAD_ARG;
    if (workspace_bytes < l1_ws_bytes(n)) return GS_E_WORKSPACE;
    return launch_bce_loss(x, y, n, loss, dL_dx, (float*)workspace, (hipStream_t)stream);
}

int gs_ssim_workspace_bytes(int32_t C, int32_t H, int32_t W, size_t* out) {
    if (!out || C <= 0 || H <= 0 || W <= 0) return GS_E_BAD_ARG;
    *out = ssim_ws_bytes(C, H, W);
    return GS_OK;
}
int gs_ssim_forward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, float* ssim_out, float* dm_dmu1,
                    float* dm_dsigma1_sq, float* dm_dsigma12, void* workspace, size_t workspace_bytes, void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_out || !workspace) return GS_E_BAD_ARG;
    if ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr) || (dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr)) return GS_E_BAD_ARG;
    if (workspace_bytes < ssim_ws_bytes(C, H, W)) return GS_E_WORKSPACE;
    return launch_ssim_forward(C, H, W, img1, img2, ssim_out, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (float*)workspace,
                               (hipStream_t)stream);
}
int gs_ssim_backward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, const float* dm_dmu1,
                     const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dssim, float* dL_dimg1,
                     void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dssim || !dL_dimg1)
        return GS_E_BAD_ARG;
    return launch_ssim_backward(C, H, W, img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dssim, dL_dimg1,
                                (hipStream_t)stream);
}

int gs_densify_stats(int32_t N, const int32_t* radii, const float* viewspace_grad, float* max_radii2D,
                     float* xyz_gradient_accum, float* denom, void* stream) {
    GS_CAPTURE_OK_IF(stream, true);
    if (N < 0 || (N > 0 && (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom))) return GS_E_BAD_ARG;
    if (N == 0) return GS_OK;
    return launch_densify_stats(N, radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom, (hipStream_t)stream);
}
int gs_adam_step(int32_t n_tensors, const GsAdamTensor* tensors, double beta1, double beta2, double eps, int64_t step,
                 void* stream) {
    GS_NO_CAPTURE(stream);  // (the step number is a host scalar: a replay would repeat the captured step's bias correction)
    if (n_tensors < 0 || n_tensors > GS_ADAM_MAX_TENSORS || (n_tensors > 0 && !tensors) || step < 1) return GS_E_BAD_ARG;
    for (int k = 0; k < n_tensors; k++) {
        const GsAdamTensor& t = tensors[k];
        if (t.n < 0 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) return GS_E_BAD_ARG;
    }
    if (n_tensors == 0) return GS_OK;
    return launch_adam(n_tensors, tensors, beta1, beta2, eps, step, (hipStream_t)stream);
}

int knn_points(int32_t Nq, const float* queries, int32_t Nr, const float* ref, int32_t K, float* dists, int64_t* idx,
               void* workspace, size_t workspace_bytes, void* stream) {
    GS_NO_CAPTURE(stream);
    if (Nq < 0 || Nr <= 0 || K < 1 || K > 8 || (Nq > 0 && (!queries || !dists || !idx)) || !ref || !workspace) return GS_E_BAD_ARG;
    if (Nq == 0) return GS_OK;
    return launch_knn_points(Nq, queries, Nr, ref, K, dists, (long long*)idx, workspace, workspace_bytes, (hipStream_t)stream);
}

int gs_pair_stats(const GsFwdArgs* a, const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes, const void* img,
                  size_t img_bytes, int64_t D, uint64_t* counts, void* stream) {
    GS_NO_CAPTURE(stream);
    int rc = validate(a);
    if (rc != GS_OK) return rc;
    if (!geom || !img || !counts || D < 0 || (D > 0 && !binning)) return GS_E_BAD_ARG;
    const GeomLayout L = geom_layout(a->P);
    const ImgLayout I = img_layout(a->W, a->H, a->long_lists);
    const BinLayout B = bin_layout(D);
    if (geom_bytes < L.total || img_bytes < I.total || (D > 0 && binning_bytes < B.total)) return GS_E_WORKSPACE;
    const char* g = (const char*)geom;
    const char* b = (const char*)binning;
    const char* im = (const char*)img;
    return launch_pair_stats((const float*)(g + L.rec), D > 0 ? (const uint32_t*)(b + B.point_list) : nullptr,
                             (const uint32_t*)(im + I.ranges), (const uint32_t*)(im + I.n_contrib), a->W, a->H,
                             (unsigned long long*)counts, (hipStream_t)stream);
}

int gs_geom_field(void* geom, int32_t P, int32_t field, void** out) {
    if (!geom || !out || P < 0) return GS_E_BAD_ARG;
    const GeomLayout L = geom_layout(P);
    char* g = (char*)geom;
    switch (field) {
        case 0: *out = g + L.depths; break;
        case 1: *out = g + L.tiles; break;
        case 2: *out = g + L.rec; break;
        case 3: *out = g + L.clamped; break;
        case 4: *out = g + L.val0; break;
        case 5: *out = g + L.count; break;
        default: return GS_E_BAD_ARG;
    }
    return GS_OK;
}
int gs_binning_field(void* binning, int64_t D, int32_t W, int32_t H, int32_t field, void** out) {
    if (!binning || !out || D < 0) return GS_E_BAD_ARG;
    const BinLayout B = bin_layout(D);
    (void)W; (void)H;
    char* b = (char*)binning;
    switch (field) {
        case 0: *out = b + B.point_list; break;  // (the tile id of every entry follows from the image state's ranges)
        default: return GS_E_BAD_ARG;
    }
    return GS_OK;
}
int gs_image_field(void* img, int32_t W, int32_t H, int32_t field, void** out) {
    if (!img || !out) return GS_E_BAD_ARG;
    const ImgLayout I = img_layout(W, H);
    if (field >= 6 && field <= 9 && !I.cw_rec) return GS_E_BAD_ARG;  // (the chunk-parallel forward's state: small images only)
    char* m = (char*)img;
    switch (field) {
        case 0: *out = m + I.ranges; break;
        case 1: *out = m + I.n_contrib; break;
        case 2: *out = m + I.final_T; break;
        case 3: *out = m + I.tile_nmax; break;
        case 4: *out = m + I.ncon_c; break;
        case 5: *out = m + I.order; break;
        case 6: *out = m + I.cw_hdr; break;    // 16 words: units in use, entries per chunk, .., [4..11] items per XCD
        case 7: *out = m + I.cw_units; break;  // FWDC_MAX_UNITS x {tile, chunk | chunks << 16}
        case 8: *out = m + I.cw_flag; break;   // 4 FWDC_MAX_UNITS words: hits + 1 | dead << 31
        case 9: *out = m + I.cw_rec; break;    // 4 FWDC_MAX_UNITS records of FWDC_SLOTS x 64 floats
        default: return GS_E_BAD_ARG;
    }
    return GS_OK;
}

int gs_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.on = on != 0;
    return GS_OK;
}
int gs_profile_reserve(int n_events) {
    // creating an event and RECORDING it for the first time each cost ~0.1 ms: done here, a timed loop would otherwise
    // show them as slow first steps
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t last = nullptr;
    while ((int)g_prof.pool.size() < n_events) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return GS_E_HIP;
        (void)hipEventRecord(e, nullptr);
        last = e;
        g_prof.pool.push_back(e);
    }
    if (last) (void)hipEventSynchronize(last);
    return GS_OK;
}
int gs_profile_filter(const char* stage) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    memset(g_prof.filter, 0, sizeof(g_prof.filter));
    if (stage) strncpy(g_prof.filter, stage, sizeof(g_prof.filter) - 1);
    return GS_OK;
}
int gs_profile_collect(int max, const char** names, float* ms, int32_t* launches, int32_t* n_out) {
    if (max < 0 || !n_out || (max > 0 && (!names || !ms || !launches))) return GS_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof.recs) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
        if (e != hipSuccess) { gs_set_error((int)e, "profile.collect"); t = 0.f; }
        int k = 0;
        for (; k < n; k++) if (names[k] == r.name) break;
        if (k == n) {
            if (n < max) { names[n] = r.name; ms[n] = 0.f; launches[n] = 0; n++; } else k = -1;
        }
        if (k >= 0) { ms[k] += t; launches[k] += 1; }
        g_prof.pool.push_back(r.a);
        g_prof.pool.push_back(r.b);
    }
    g_prof.recs.clear();
    *n_out = n;
    return GS_OK;
}

int gs_tuning(const char* name, int value) {
    if (!name) return GS_E_BAD_ARG;
    tune_defaults();
    if (strcmp(name, "xcd_map") == 0) { g_tune[GS_TUNE_XCD_MAP].store(value); return GS_OK; }
    if (strcmp(name, "small_tiles") == 0) { g_tune[GS_TUNE_SMALL_TILES].store(value); return GS_OK; }  // changes the image state's size
    if (strcmp(name, "ones_fast") == 0) { g_tune[GS_TUNE_ONES_FAST].store(value); return GS_OK; }
    if (strcmp(name, "shared_qlist") == 0) { g_tune[GS_TUNE_SHARED_QLIST].store(value); return GS_OK; }
    if (strcmp(name, "fwd4") == 0) { g_tune[GS_TUNE_FWD4].store(value); return GS_OK; }
    if (strcmp(name, "fwdc_div") == 0) {
        if (value < 1) return GS_E_BAD_ARG;
        g_tune[GS_TUNE_FWDC_DIV].store(value);
        return GS_OK;
    }
    if (strcmp(name, "fwdc_ch") == 0) {
        if (value < 64 || (value & (value - 1)) != 0) return GS_E_BAD_ARG;
        g_tune[GS_TUNE_FWDC_CH].store(value);
        return GS_OK;
    }
    if (strcmp(name, "bwd_chunks") == 0) { g_tune[GS_TUNE_BWD_CHUNKS].store(value); return GS_OK; }  // flip between frames only
    if (strcmp(name, "nt_stores") == 0) { g_tune[GS_TUNE_NT_STORES].store(value); return GS_OK; }
    if (strcmp(name, "fwd_marks") == 0) { g_tune[GS_TUNE_FWD_MARKS].store(value); return GS_OK; }  // 0: the backward sets its row marks itself (A/B)
    if (strcmp(name, "bwd_order") == 0) { g_tune[GS_TUNE_BWD_ORDER].store(value); return GS_OK; }  // 0: the backward walks the tiles in the forward's launch order (A/B: + 12 us at config 3)
    if (strcmp(name, "depth_sort") == 0) { g_tune[GS_TUNE_DEPTH_SORT].store(value); return GS_OK; }  // 1 bucket sort, 0 LSD radix
    return GS_E_BAD_ARG;
}

const char* gs_status_string(int code) {
    switch (code) {
        case GS_OK: return "ok";
        case GS_E_BAD_ARG: return "bad argument (null required pointer, non-positive size, unsupported SH degree / more than 16 SH coefficients, or rotations not 16-byte aligned)";
        case GS_E_EXCLUSIVE: return "provide exactly one of shs/colors_precomp and exactly one of (scales, rotations)/cov3D_precomp";
        case GS_E_TOO_LARGE: return "num_rendered or tile grid exceeds the supported index space";
        case GS_E_HIP: return "HIP error";
        case GS_E_WORKSPACE: return "state/workspace buffer smaller than gs_*_bytes requires";
        case GS_E_CAPTURE: return "the stream is being captured into a graph and this call is not capture-safe with these arguments (include/gsplat_mi355.h, \"Stream capture\"); nothing was enqueued";
        default: return "unknown status";
    }
}
int gs_last_hip_error(void) { return g_last_hip; }
const char* gs_last_stage(void) { return g_last_stage; }
const char* gs_build_info(void) { return "gsplat_mi355 gfx950 wave64 tile16 rec48"; }

}  // extern "C"
