// render_fwd.hip -- per-tile front-to-back alpha compositing (replaces
// upstream renderCUDA forward).
//
// CDNA4 mapping: ONE wave64 per 8x8 QUADRANT of a 16x16 tile (four waves per tile, one pixel per
// lane; 4 x tiles waves keep every SIMD 8 deep and let the dispatcher balance the heavy centre tiles).
// The tile's depth-ordered list is staged 64 entries at a time: each lane gathers one 48-byte splat
// record (next batch prefetched into registers), converts it (log2-domain conic, alpha threshold) and
// tests whether the Gaussian's alpha >= 1/255 footprint (ellipse bbox) reaches THIS quadrant; the
// entries that do are COMPACTED into LDS with a ballot prefix, so the inner loop carries no mask
// logic at all and reads each staged entry at a wave-uniform address.  The per-pixel body is
// branch-free (compute, then select).  No barrier spans more than this one wave; the early-out is
// a wave ballot.
//
// The compaction is also RECORDED for the backward pass: qlist (the quadrant's compacted Gaussian
// indices, in order), the per-pixel last contributor in compacted coordinates, and the per-quadrant
// count up to the last contributor.  The backward therefore never re-derives relevance.
#include "common.h"
#include "blend.h"

__global__ __launch_bounds__(64) void render_fwd_kernel(const float4* __restrict__ rec,
                                                        const uint32_t* __restrict__ point_list,
                                                        const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ order,
                                                        const float* __restrict__ bg, int W, int H, int gx,
                                                        int ntiles, int xmap,
                                                        float* __restrict__ out_color, float* __restrict__ final_T,
                                                        uint32_t* __restrict__ n_contrib,
                                                        uint32_t* __restrict__ qlist, uint32_t* __restrict__ ncon_c,
                                                        uint32_t* __restrict__ qcount, float4* __restrict__ ckpt,
                                                        uint32_t* __restrict__ ck_start, const int chunks) {
    __shared__ float4 srec[66 * 3];  // 64 staged entries + the two the pipelined loop may read past a batch
    int slot, q;
    render_block_map((int)blockIdx.x, xmap, &slot, &q);
    if (slot >= ntiles) return;
    const int tile = (int)order[slot];  // heaviest tiles first (tile_order_kernel)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int QX0 = tx * TILE + 8 * (q & 1), QY0 = ty * TILE + 8 * (q >> 1);
    const int px = QX0 + (lane & 7), py = QY0 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const bool inside = px < W && py < H;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const uint32_t qbase = 4u * range.x + (uint32_t)q * (uint32_t)n;  // this quadrant's slice of qlist / gradient rows
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

    // A pixel is "alive" while T > 0.  The sign of T doubles as the done flag: when a contribution
    // would push T below 1e-4 the pixel is frozen as T := -T, which keeps final_T and makes every later
    // test fail by itself.  All per-pixel decisions are VALU compares + selects (v_cmp -> vcc ->
    // v_cndmask): no scalar mask arithmetic in the inner loop -- the CU's single scalar unit was the
    // bottleneck of the mask-based version.
    float T = inside ? 1.0f : -1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0, last_k = 0;
    uint32_t kcount = 0;  // wave-uniform: compacted entries staged so far
    int nck = 0;                   // checkpoints written so far (chunks of the backward begun, less one)
    uint32_t next_ck = BWD_CH;     // ... the next one is due at the first batch that starts at or beyond this entry
    bool live = __ballot(T > 0.f) != 0ull;  // wave-uniform

    float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
    uint32_t pid_g = 0;
    if (live && lane < n) {
        pid_g = point_list[range.x + lane];
        p0 = rec[(size_t)pid_g * 3];
        p1 = rec[(size_t)pid_g * 3 + 1];
        p2 = rec[(size_t)pid_g * 3 + 2];
    }
    for (int base = 0; base < n && live; base += 64) {
        Staged s;
        const bool hit = stage_entry_quad(p0, p1, p2, QX0, QY0, s) && (base + lane < n);
        const unsigned long long bal = __ballot(hit);
        const int cnt = __popcll(bal);
        __syncthreads();
        if (hit) {
            const int slot = __popcll(bal & lt_mask);
            const uint32_t k = kcount + (uint32_t)slot;  // compacted index of this entry
            s.c.w = __uint_as_float((uint32_t)(base + lane + 1));  // position in the tile's list (1-based)
            srec[slot * 3] = s.a;
            srec[slot * 3 + 1] = s.b;
            srec[slot * 3 + 2] = s.c;
            qlist[qbase + k] = pid_g;  // record the compaction for the backward
        }
        __syncthreads();
        kcount += (uint32_t)cnt;
        if (base + 64 + lane < n) {
            pid_g = point_list[range.x + base + 64 + lane];
            p0 = rec[(size_t)pid_g * 3];
            p1 = rec[(size_t)pid_g * 3 + 1];
            p2 = rec[(size_t)pid_g * 3 + 2];
        }
        // The staged entries are read at a wave-uniform address.  Keeping that address in a VGPR the
        // compiler cannot prove uniform (vzero) makes it one v_add per entry + immediate offsets; proven
        // uniform, it is rebuilt from SGPRs with a v_mov per dword and a dozen scalar adds per entry, and
        // the CU's single scalar unit becomes the bottleneck.
        const char* sp = reinterpret_cast<const char*>(srec) + vzero;
        const uint32_t kbase = kcount - (uint32_t)cnt;  // compacted index of this batch's first entry
        // a chunk of the backward starts here when the previous one has its BWD_CH entries (common.h, BWD_CH): the state
        // before this batch is checkpointed
        if (nck + 1 < chunks && kbase >= next_ck) {
            ckpt[((size_t)(tile * 4 + q) * (size_t)(chunks - 1) + (size_t)nck) * 64 + lane] = make_float4(fabsf(T), C0, C1, C2);
            nck++;
            if (lane == 0) ck_start[(size_t)(tile * 4 + q) * (size_t)chunks + nck] = kbase;
            next_ck = kbase + BWD_CH;
        }
        // One entry against the 64 pixels.  `a`, `b`, `c` = the three staged quads of the entry.
        auto blend = [&](const float4 a, const float2 b, const float4 c, const uint32_t k1) {
            const float dx = a.x - pxf, dy = a.y - pyf;
            // A2 dx^2 + B2 dx dy + C2 dy^2, evaluated exactly as the backward does
            const float power2 = __builtin_fmaf(a.z * dx, dx, __builtin_fmaf(a.w, dx, b.x * dy) * dy);
            const float G = __builtin_amdgcn_exp2f(power2);
            const float al = fminf(0.99f, b.y * G);
            const float a2 = (power2 <= 0.0f && al >= (1.0f / 255.0f)) ? al : 0.f;  // alpha, or 0 if the pair is rejected
            const float test_T = T * (1.f - a2);                  // == T for a rejected pair, < 0 for a frozen pixel
            const bool pass = test_T >= 0.0001f;
            const float wT = a2 * T;
            const float w = pass ? wT : 0.f;                      // > 0 exactly when the pair is blended
            T = pass ? test_T : -fabsf(T);                        // freeze: keeps |T| = final_T
            C0 += c.x * w;
            C1 += c.y * w;
            C2 += c.z * w;
            last_k = (w > 0.f) ? k1 : last_k;  // (a wave-uniform value: no LDS operand)
        };
        // The loop is software-pipelined by hand, two entries per trip: the LDS reads of the NEXT entry are issued
        // before the current one is evaluated.  With eight waves per SIMD the LDS latency hides behind the other waves
        // anyway; a frame of few, long lists (a trained avatar: 150 tiles of 2000-7000 entries) runs one wave per SIMD
        // and paid the ~100 cycles of every entry's reads in full.  (Reads one entry past the batch: srec has a 65th.)
        auto ld_a = [&](int o) { return *reinterpret_cast<const float4*>(sp + o); };
        auto ld_b = [&](int o) { return *reinterpret_cast<const float2*>(sp + o + 16); };
        auto ld_c = [&](int o) {
            const float4 c = *reinterpret_cast<const float4*>(sp + o + 32);
            // keep the whole 16 bytes of `c` one ds_read_b128 (4 LDS cycles): only r, g, b are used in the loop, and
            // the 12-byte ds_read_b96 the compiler would pick takes 8 -- with it the loop needs 14 LDS cycles per entry
            // and wave, 56 per four SIMDs against 52 cycles of VALU issue: the kernel was LDS-bound
            asm volatile("" ::"v"(c.w));
            return c;
        };
        float4 a0 = ld_a(0), c0 = ld_c(0);
        float2 b0 = ld_b(0);
        int j = 0;
        for (; j + 1 < cnt; j += 2) {
            const float4 a1 = ld_a(48), c1 = ld_c(48);
            const float2 b1 = ld_b(48);
            blend(a0, b0, c0, kbase + (uint32_t)j + 1u);
            a0 = ld_a(96); c0 = ld_c(96); b0 = ld_b(96);
            sp += 96;
            blend(a1, b1, c1, kbase + (uint32_t)j + 2u);
        }
        if (j < cnt) blend(a0, b0, c0, kbase + (uint32_t)j + 1u);
        // the last contributor's position in the TILE's list (n_contrib), looked up once per batch
        if (last_k > kbase) last = __float_as_uint(srec[(last_k - 1u - kbase) * 3 + 2].w);
        live = __ballot(T > 0.f) != 0ull;  // every pixel of the quadrant frozen: stop
    }
    {
        const uint32_t nm = wave_max_u32(last_k);
        if (lane == 0) qcount[tile * 4 + q] = nm;  // the backward's loop bound / work estimate for this quadrant
        // chunks of the backward that never began
        if (chunks > 1 && lane > nck && lane < chunks) ck_start[(size_t)(tile * 4 + q) * (size_t)chunks + lane] = 0xFFFFFFFFu;
    }
    if (inside) {
        const size_t HW = (size_t)H * W;
        const size_t pid = (size_t)py * W + px;
        const float Tf = fabsf(T);
        final_T[pid] = Tf;
        n_contrib[pid] = last;
        ncon_c[pid] = last_k;
        out_color[pid] = C0 + Tf * bg[0];
        out_color[HW + pid] = C1 + Tf * bg[1];
        out_color[2 * HW + pid] = C2 + Tf * bg[2];
    }
}

int launch_render_forward(const float* rec, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* order,
                          const float* bg, int W, int H, float* out_color, float* final_T, uint32_t* n_contrib,
                          const QuadLists& ql, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int xmap = gs_tune_get(GS_TUNE_XCD_MAP);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(render_grid_blocks(gx * gy, xmap)), dim3(64), 0, s,
                       reinterpret_cast<const float4*>(rec), point_list, reinterpret_cast<const uint2*>(ranges), order, bg,
                       W, H, gx, gx * gy, xmap, out_color, final_T, n_contrib, ql.qlist, ql.ncon_c, ql.qcount, ql.ckpt,
                       ql.ck_start, ql.ckpt ? ql.chunks : 1);
    GS_LAUNCH_CHECK("render_forward", 0, s);
    return GS_OK;
}

// The opacity render -- what a second rasterizer call with colours = 1 returns in every channel
// (gaussian_renderer/__init__.py:132-142): sum_i alpha_i T_i + T_final * bg = (1 - T_final) + T_final * bg[0].
// The forward already has T_final per pixel, so no second render is needed.
__global__ __launch_bounds__(256) void opacity_image_kernel(const float* __restrict__ final_T, const float* __restrict__ bg,
                                                            int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float Tf = final_T[i];
    out[i] = (1.0f - Tf) + Tf * bg[0];
}

int launch_opacity_image(const float* final_T, const float* bg, int W, int H, float* out, hipStream_t s) {
    const int n = W * H;
    hipLaunchKernelGGL(opacity_image_kernel, dim3((n + 255) / 256), dim3(256), 0, s, final_T, bg, n, out);
    GS_LAUNCH_CHECK("opacity_image", 0, s);
    return GS_OK;
}
