// render_bwd.hip -- backward of the per-tile compositing (replaces upstream
// renderCUDA backward and its 9 float atomicAdds per (pixel, Gaussian) pair).
//
// CDNA4 formulation -- GAUSSIAN-PARALLEL, no atomics at all:
//  * One wave64 per 8x8 quadrant.  The forward recorded the quadrant's compacted list (qlist) of the
//    Gaussians whose footprint reaches it, up to its last contributor.
//  * LANES ARE LIST ENTRIES, the pixels stream through them.  The wave is FOUR RINGS of 16 lanes (the
//    rows DPP row_ror:1 rotates); ring r owns 16 of the 64 pixels and all four rings hold the same 16
//    entries: entry e sits at position e mod 16 of every ring for the 16 steps e .. e+15, and at step s
//    works on the pixel at ring position s - e.  A pixel's running state (T and the not yet composited
//    part of Gtot) therefore moves one lane per step and visits the entries front to back, exactly like
//    the forward.  The pipeline is skewed, so it never drains between chunks of 16 entries ("rounds"):
//    only the first and last 15 steps of a quadrant run partly empty (a single 64-lane ring wasted 63
//    steps per quadrant, 22 % of all steps on the benchmark scene).
//    (The pixel-parallel formulation needs a 64-lane reduction of nine values per list entry; DPP
//    adds issue at half rate on gfx950 (tools/dpp_rate.hip), which made that reduction ~2/3 of the
//    kernel.)
//  * Entries never touch LDS: every lane gathers the record of its entry two chunks ahead, converts it
//    at the round boundary, and at its own switch step (ring position == step mod 16) moves it into
//    the working set.  Per-pixel constants (dL/dpixel, position, last-contributor limit) sit in LDS, one
//    array per component, and are read at the lane's current pixel index one step ahead.
//  * Each lane keeps TWO sets of nine gradient sums, named by the parity of the chunk they belong to:
//    lanes that have switched in the current round add into one set, the others still into the other
//    (the nine FMAs issued under complementary exec masks).  A set is complete for every lane one
//    round later: it is folded over the four rings, stored by ring 0 and cleared at a round start --
//    nothing is stored or cleared under a one-lane exec mask, where an instruction costs as much as
//    with 64 lanes.
//  * FRONT-to-back recurrence.  With g = dL/dpixel, Gtot = out_color . g (out_color already holds
//    T_final * bg) and Rem_i = Gtot - sum_{j<=i} (c_j . g) alpha_j T_j,
//        dL/dalpha_i = T_i (c_i . g) - Rem_i / (1 - alpha_i)
//    which is the reference's back-to-front recurrence (accum_rec / T division) rewritten so that T is
//    rebuilt by the same multiplications the forward did.
//  * Output: the nine raw sums of (pair, quadrant) at row gradient_row(...) (common.h), pairs in EMISSION order
//    (Gaussian-major), so the per-Gaussian kernel reads one contiguous span per Gaussian: eight sums in
//    a 32-byte row, the ninth in a dense word array the caller pre-fills with ROW_UNWRITTEN, so it also
//    tells which rows were written.
#include "common.h"
#include "blend.h"

// rotate by one lane inside every row of 16 lanes (DPP row_ror:1)
__device__ __forceinline__ float ring_ror1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));
}
#define RING 16  // lanes per ring; a wave holds 64 / RING rings
#define RING_STRIDE 48  // words between the rings' per-pixel constants in LDS

// x[0..8] += (Gd dx, Gd dy, tdx dx, tdx dy, tdy dy, Gd, wgt gx, wgt gy, wgt gz) on the lanes of `mask`,
// y[0..8] += the same on all other lanes: the nine instructions issued twice under complementary exec
// masks.  (Written as if / else the compiler either flattens the two blocks into three selects per sum
// or indexes the sets through scratch memory.)  The caller runs with all 64 lanes on.
//
// ENTRY SWITCH, riding in the same two blocks.  Position t of every ring takes its entry of the round's chunk at step t --
// one lane per ring and step.  Moving nine values under a one-lane exec mask costs nine VALU issue slots per step (an
// instruction costs the same with 1 or 64 lanes on): here the LDS unit does it instead.  The lanes of `m1 & ~mask` (those
// that switch at the NEXT step; the accumulation is the last thing of a step that reads the entry... and it does not)
// read their new entry from its staged copy straight into the working registers (q0, q1, q2): no other lane is written.
// Round 4: the reads are issued AND waited for inside one asm statement -- in front of and behind the second set's nine
// instructions and the two ring rotations that end a step -- so the compiler never sees a point where loads into q0 / q1
// / q2 are in flight.  (Until round 3 they were issued in the first set's statement and waited for in the second's, with
// a disassembly check at build time as the only guard; same-box A/B of the one-statement form: render_bwd 1 us faster.)
typedef float bwd_f4 __attribute__((ext_vector_type(4)));
template <bool SECOND>
__device__ __forceinline__ void split_accumulate(float (&x)[9], float (&y)[9], unsigned long long mask, float Gd, float dx,
                                                 float dy, float tdx, float tdy, float wgt, float gx, float gy,
                                                 float gz, unsigned long long m1, uint32_t lds_addr, bwd_f4& q0, bwd_f4& q1,
                                                 bwd_f4& q2, float& rot0, float& rot1) {
    unsigned long long save;
    // The first statement is the X set alone; the second issues the reads, runs the Y set (and the caller's two ring
    // rotations, `rot0` / `rot1`) in their shadow and waits.  The 30-operand limit of an asm statement is what keeps the X
    // set out of it.  (SECOND: q2 = (b, r2, g2, b2), 16 bytes; otherwise only its first word is used: a 4-byte read.)
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_fmac_f32 %[x0], %[Gd], %[dx]\n\t"
        "v_fmac_f32 %[x1], %[Gd], %[dy]\n\t"
        "v_fmac_f32 %[x2], %[tdx], %[dx]\n\t"
        "v_fmac_f32 %[x3], %[tdx], %[dy]\n\t"
        "v_fmac_f32 %[x4], %[tdy], %[dy]\n\t"
        "v_add_f32 %[x5], %[x5], %[Gd]\n\t"
        "v_fmac_f32 %[x6], %[w], %[gx]\n\t"
        "v_fmac_f32 %[x7], %[w], %[gy]\n\t"
        "v_fmac_f32 %[x8], %[w], %[gz]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]),
          [x7] "+v"(x[7]), [x8] "+v"(x[8]), [sv] "=&s"(save)
        : [m] "s"(mask), [Gd] "v"(Gd), [dx] "v"(dx), [dy] "v"(dy), [tdx] "v"(tdx), [tdy] "v"(tdy), [w] "v"(wgt), [gx] "v"(gx),
          [gy] "v"(gy), [gz] "v"(gz)
        : "scc");
#define GS_ACC_Y                                     \
        "s_andn2_b64 exec, %[sv], %[m]\n\t"          \
        "v_fmac_f32 %[y0], %[Gd], %[dx]\n\t"         \
        "v_fmac_f32 %[y1], %[Gd], %[dy]\n\t"         \
        "v_fmac_f32 %[y2], %[tdx], %[dx]\n\t"        \
        "v_fmac_f32 %[y3], %[tdx], %[dy]\n\t"        \
        "v_fmac_f32 %[y4], %[tdy], %[dy]\n\t"        \
        "v_add_f32 %[y5], %[y5], %[Gd]\n\t"          \
        "v_fmac_f32 %[y6], %[w], %[gx]\n\t"          \
        "v_fmac_f32 %[y7], %[w], %[gy]\n\t"          \
        "v_fmac_f32 %[y8], %[w], %[gz]\n\t"          \
        "s_mov_b64 exec, %[sv]\n\t"                  \
        "v_mov_b32_dpp %[r0], %[r0] row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "v_mov_b32_dpp %[r1], %[r1] row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "s_waitcnt lgkmcnt(0)"
#define GS_ACC_Y_OUT [y0] "+v"(y[0]), [y1] "+v"(y[1]), [y2] "+v"(y[2]), [y3] "+v"(y[3]), [y4] "+v"(y[4]), [y5] "+v"(y[5]), \
                     [y6] "+v"(y[6]), [y7] "+v"(y[7]), [y8] "+v"(y[8]), [q0] "+v"(q0), [q1] "+v"(q1), [r0] "+v"(rot0), [r1] "+v"(rot1), \
                     [sv] "=&s"(save)
#define GS_ACC_Y_IN [m] "s"(mask), [m1] "s"(m1), [a] "v"(lds_addr), [Gd] "v"(Gd), [dx] "v"(dx), [dy] "v"(dy), [tdx] "v"(tdx), \
                    [tdy] "v"(tdy), [w] "v"(wgt), [gx] "v"(gx), [gy] "v"(gy), [gz] "v"(gz)
    if constexpr (SECOND) {
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_andn2_b64 exec, %[m1], %[m]\n\t"
            "ds_read_b128 %[q0], %[a]\n\t"
            "ds_read_b128 %[q1], %[a] offset:16\n\t"
            "ds_read_b128 %[q2], %[a] offset:32\n\t" GS_ACC_Y
            : GS_ACC_Y_OUT, [q2] "+v"(q2)
            : GS_ACC_Y_IN
            : "memory", "scc");
    } else {
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_andn2_b64 exec, %[m1], %[m]\n\t"
            "ds_read_b128 %[q0], %[a]\n\t"
            "ds_read_b128 %[q1], %[a] offset:16\n\t"
            "ds_read_b32 %[q2], %[a] offset:32\n\t" GS_ACC_Y
            : GS_ACC_Y_OUT, [q2] "+v"(q2.x)
            : GS_ACC_Y_IN
            : "memory", "scc");
    }
#undef GS_ACC_Y
#undef GS_ACC_Y_OUT
#undef GS_ACC_Y_IN
}

// The per-entry values the inner loop reads, as the three 16-byte quads they are staged with in LDS:
// q0 = (x, y, A2, B2), q1 = (C2, opacity, r, g), q2 = (b, and the entry's colour in a second image of the same geometry)
struct EntryQ {
    bwd_f4 q0, q1, q2;
};
#define ENT_SLOTS 2  // chunks of 16 staged entries in LDS: the one the lanes are switching to, the one after it

// Entry switch.  Position t of every ring takes its entry of the round's chunk at step t -- one lane per ring and step.
// Moving nine values under a one-lane exec mask costs nine VALU issue slots per step (an instruction costs the same with 1
// or 64 lanes on): here the LDS unit does it instead.  The lanes of `m1 & ~m0` (those that switch at the NEXT step) read
// their new entry from its staged copy straight into the working registers; no other lane is written.  Issue and wait
// sit in one block: between two blocks the compiler would be free to copy registers whose load is still in flight.
template <bool SECOND>
__device__ __forceinline__ void switch_entry(EntryQ& e, uint32_t lds_addr, unsigned long long m0, unsigned long long m1) {
    unsigned long long save;
    if constexpr (SECOND) {
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_andn2_b64 exec, %[m1], %[m0]\n\t"
            "ds_read_b128 %[q0], %[a]\n\t"
            "ds_read_b128 %[q1], %[a] offset:16\n\t"
            "ds_read_b128 %[q2], %[a] offset:32\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_waitcnt lgkmcnt(0)"
            : [q0] "+v"(e.q0), [q1] "+v"(e.q1), [q2] "+v"(e.q2), [sv] "=&s"(save)
            : [a] "v"(lds_addr), [m0] "s"(m0), [m1] "s"(m1)
            : "memory", "scc");
    } else {
        float b = e.q2.x;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_andn2_b64 exec, %[m1], %[m0]\n\t"
            "ds_read_b128 %[q0], %[a]\n\t"
            "ds_read_b128 %[q1], %[a] offset:16\n\t"
            "ds_read_b32 %[b], %[a] offset:32\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_waitcnt lgkmcnt(0)"
            : [q0] "+v"(e.q0), [q1] "+v"(e.q1), [b] "+v"(b), [sv] "=&s"(save)
            : [a] "v"(lds_addr), [m0] "s"(m0), [m1] "s"(m1)
            : "memory", "scc");
        e.q2.x = b;
    }
}

// OPA: the image has a fourth channel whose "colour" is 1 for every Gaussian -- the opacity render the reference
// obtains with a second rasterizer call (gaussian_renderer/__init__.py:132-142) -- and dL_dopa is the gradient of
// that channel: one more term in (c . g) and in Gtot, nothing else changes.
// MODE 0: one image; 1 (OPA): + the opacity channel; 2 (SECOND): + a second image of the same geometry; 3 (SONES): + a
// second image whose colours are all (1, 1, 1) -- the reference's opacity pass (gaussian_renderer/__init__.py:132-142).
// Modes 1 and 3 run MODE 0's loop: an image of colours one enters through the pixel's starting value alone (see the set-up).  Both 2 and 3 are launched for a second image; each
// leaves at once unless the second image's `all_ones` word says it is its case (the host does not know).
#ifndef BWD_MIN_WAVES
#define BWD_MIN_WAVES 1
#endif
template <int MODE>
__global__ __launch_bounds__(64, BWD_MIN_WAVES) void render_bwd_kernel(const float4* __restrict__ rec,
                                                        const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ order, int W, int H, int gx,
                                                        int ntiles, int xmap,
                                                        const uint32_t* __restrict__ qlist,
                                                        const uint32_t* __restrict__ ncon_c,
                                                        const uint32_t* __restrict__ qcount,
                                                        const float* __restrict__ out_color,
                                                        const float* __restrict__ dL_dpix,
                                                        const float* __restrict__ dL_dopa,
                                                        const float* __restrict__ final_T, const float* __restrict__ bg,
                                                        float4* __restrict__ qrows, uint32_t* __restrict__ q8,
                                                        const float4* __restrict__ ckpt,
                                                        const uint32_t* __restrict__ ck_start, const int chunks,
                                                        const int blocks_per_chunk, const SecondImage second,
                                                        const L1Grad l1, const float l1_inv_n) {
    constexpr bool OPA = MODE == 1, SECOND = MODE == 2, SONES = MODE == 3;
    if constexpr (SECOND) { if (*second.all_ones != 0u) return; }
    if constexpr (SONES) { if (*second.all_ones == 0u) return; }
    // per-pixel constants, one array per component (adjacent lanes read adjacent words: no bank
    // conflicts; the 32-byte records this replaces cost 8-way conflicts on every step); ring r owns the
    // 16 pixels 16 r .. 16 r + 15, each ring's 16 values stored twice in a row so a round's reads never
    // wrap, and the rings RING_STRIDE = 48 words apart: the four rings' 16-word windows then fall into four
    // different quarters of the banks (at 32 words apart rings 0 / 2 and 1 / 3 collide):  g0, g1, g2, x, y, lim
    // ... stored as PAIRS (one ds_read_b64 per pair: 2 LDS cycles for 8 bytes per lane where two ds_read_b32 take 4 -- with
    // the entry switch done by LDS reads the kernel sits close to the LDS's cycle budget): (g0, g1) (g2, x) (y, lim)
    // [+ (h0, h1) (h2, -)]
    constexpr int NP2 = SECOND ? 5 : 3;
    __shared__ float2 pix[NP2][4 * RING_STRIDE];
    // the converted entries of a chunk, staged by ring 0 at a round start for the switch reads of the NEXT round
    __shared__ bwd_f4 ent[ENT_SLOTS][RING][3];
    int slot, q;
    // chunk-major: all first chunks, heaviest tiles first, then all second chunks, ... (chunks = 1: one wave per quadrant)
    const int chunk = chunks > 1 ? (int)blockIdx.x / blocks_per_chunk : 0;
    render_block_map((int)blockIdx.x - chunk * blocks_per_chunk, xmap, &slot, &q);
    if (slot >= ntiles) return;
    const int tile = (int)(order[slot] & 0x7FFFFFFFu);  // heaviest tiles first (tile_order_kernel on the forward's counts; or the forward's own order, bit 31 = its mark)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int j = lane & (RING - 1), ring = lane / RING;  // position in the ring / which ring
    const int QX0 = tx * TILE + 8 * (q & 1), QY0 = ty * TILE + 8 * (q >> 1);
    const int px = QX0 + (lane & 7), py = QY0 + (lane >> 3);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    // this wave's entries: [k0, k0 + m) of the quadrant's compacted list (the whole list when chunks = 1)
    const int m_all = (int)qcount[tile * 4 + q];
    const uint32_t ks = chunk > 0 ? ck_start[(size_t)(tile * 4 + q) * (size_t)chunks + chunk] : 0u;
    if (ks >= (uint32_t)m_all) return;  // (also: chunk never begun, ~0)
    const int k0 = (int)ks;
    const uint32_t ke = chunk + 1 < chunks ? ck_start[(size_t)(tile * 4 + q) * (size_t)chunks + chunk + 1] : 0xFFFFFFFFu;
    const int m = (int)min(ke, (uint32_t)m_all) - k0;
    const uint32_t qbase = 4u * range.x + (uint32_t)q * (uint32_t)n + (uint32_t)k0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float gtot0, T0 = 1.0f;  // Gtot of pixel `lane` (what of it is not composited before entry k0), its transmittance there
    {
        // (g0, g1, g2, Gtot) and (x, y, lim): the pixel at position i of its ring meets compacted entry k at
        // step s = k + i, and the pair counts only while k < (its last contributor) <=> s < lim = ncon + i
        float4 c0 = zero4, c1 = make_float4((float)px, (float)py, __uint_as_float((uint32_t)j), 0.f);
        float3 h = make_float3(0.f, 0.f, 0.f);  // (SECOND) dL/dpixel of the second image
        if (px < W && py < H) {
            const size_t HW = (size_t)H * W;
            const size_t pid = (size_t)py * W + px;
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (dL_dpix) { g0 = dL_dpix[pid]; g1 = dL_dpix[HW + pid]; g2 = dL_dpix[2 * HW + pid]; }
            const float o0 = out_color[pid], o1 = out_color[HW + pid], o2 = out_color[2 * HW + pid];
            if (l1.target) {
                // the fused L1 loss (GsFwdArgs.l1_target): d mean|out - target| / d out = sign(out - target) / (3 H W), times
                // the loss's own gradient -- formed here instead of being written to and read from a gradient image
                const float sc = (l1.grad ? l1.grad[0] : 1.0f) * l1_inv_n;
                const float d0 = o0 - l1.target[pid], d1 = o1 - l1.target[HW + pid], d2 = o2 - l1.target[2 * HW + pid];
                g0 += d0 > 0.f ? sc : (d0 < 0.f ? -sc : 0.f);
                g1 += d1 > 0.f ? sc : (d1 < 0.f ? -sc : 0.f);
                g2 += d2 > 0.f ? sc : (d2 < 0.f ? -sc : 0.f);
            }
            c0 = make_float4(g0, g1, g2, o0 * g0 + o1 * g1 + o2 * g2);
            const uint32_t nc = ncon_c[pid];
            c1.z = __uint_as_float((nc > (uint32_t)k0 ? nc - (uint32_t)k0 : 0u) + (uint32_t)j);
            // An image whose colours are all ONE -- the opacity channel o = (1 - Tf) + Tf bg, or a second render with
            // colours (1, 1, 1) -- needs nothing in the loop: its (c . g) is the same g4 for every entry, and with
            // sum_{j <= i} alpha_j T_j = 1 - T_{i+1} its share of dL/dalpha_i = T_i g4 - Rem_i / (1 - alpha_i) collapses to
            // g4 (1 - o) / (1 - alpha_i), the same constant over (1 - alpha_i) at every entry: it is SUBTRACTED FROM Gtot
            // once, here, and the recurrence carries it (also from a checkpoint: no term there).  (Until round 3: one more
            // pixel constant read and added per step, 265 against 245 us at config 3.)
            if (OPA) {
                const float Tf = final_T[pid];
                c0.w -= dL_dopa[pid] * (Tf * (1.0f - bg[0]));  // g4 (1 - o),  1 - o = Tf (1 - bg)
            }
            if (SECOND || SONES) {
                h = make_float3(second.dL_dpix[pid], second.dL_dpix[HW + pid], second.dL_dpix[2 * HW + pid]);
                if (SECOND) c0.w += second.out_color[pid] * h.x + second.out_color[HW + pid] * h.y + second.out_color[2 * HW + pid] * h.z;
                else c0.w -= (1.0f - second.out_color[pid]) * h.x + (1.0f - second.out_color[HW + pid]) * h.y + (1.0f - second.out_color[2 * HW + pid]) * h.z;
            }
            if (chunk > 0) {
                // the forward's state before entry k0: T, and the colour composited so far (the opacity channel's is 1 - T)
                const size_t ci = ((size_t)(tile * 4 + q) * (size_t)(chunks - 1) + (size_t)(chunk - 1)) * 64 + lane;
                const float4 ck = ckpt[ci];
                T0 = ck.x;
                c0.w -= ck.y * g0 + ck.z * g1 + ck.w * g2;
                if (SECOND) {
                    const float4 ck2 = second.ckpt[ci];  // (the second render's own checkpoints: same T, its colours)
                    c0.w -= ck2.y * h.x + ck2.z * h.y + ck2.w * h.z;
                }
            }
        }
        const int slot = ring * RING_STRIDE + j;
        pix[0][slot] = pix[0][slot + RING] = make_float2(c0.x, c0.y);
        pix[1][slot] = pix[1][slot + RING] = make_float2(c0.z, c1.x);
        pix[2][slot] = pix[2][slot + RING] = make_float2(c1.y, c1.z);
        if (SECOND) {
            pix[3][slot] = pix[3][slot + RING] = make_float2(h.x, h.y);
            pix[4][slot] = pix[4][slot + RING] = make_float2(h.z, 0.f);
        }
        gtot0 = c0.w;
    }

    // record (p0,p1,p2) -> the loop's entry form, staged in LDS slot `chunk & 1` by ring 0 (the four rings hold the same
    // entries), + the gradient row of (pair, quadrant)
    float3 q2 = make_float3(0.f, 0.f, 0.f);  // (SECOND) the gathered entry's colour in the second image
    auto stage = [&](const float4 p0, const float4 p1, const float4 p2, const int chunk, uint32_t& row) {
        const uint32_t off = __float_as_uint(p2.y), rmin = __float_as_uint(p2.z), rsz = __float_as_uint(p2.w);
        const uint32_t minx = rmin & 0xFFFFu, miny = rmin >> 16, w = rsz & 0xFFFFu;
        row = gradient_row(off, w, (uint32_t)tx - minx, (uint32_t)ty - miny, (uint32_t)q);
        if (ring == 0) {
            bwd_f4* e = ent[chunk & (ENT_SLOTS - 1)][j];
            e[0] = bwd_f4{p0.x, p0.y, (-0.5f * LOG2E_F) * p0.z, -LOG2E_F * p0.w};
            e[1] = bwd_f4{(-0.5f * LOG2E_F) * p1.x, p1.y, p1.z, p1.w};
            e[2] = bwd_f4{p2.x, q2.x, q2.y, q2.z};
        }
    };
    auto gather = [&](int k, float4& p0, float4& p1, float4& p2) {
        if (k < m) {
            const uint32_t id = qlist[qbase + k];
            p0 = rec[(size_t)id * 3];
            p1 = rec[(size_t)id * 3 + 1];
            p2 = rec[(size_t)id * 3 + 2];
            if (SECOND) q2 = make_float3(second.colors[(size_t)id * 3], second.colors[(size_t)id * 3 + 1], second.colors[(size_t)id * 3 + 2]);
        }
    };

    float4 p0 = zero4, p1 = zero4, p2 = zero4;
    EntryQ cur;  // the entry this lane works on (none yet: alpha = 0)
    cur.q0 = cur.q1 = cur.q2 = bwd_f4{0.f, 0.f, 0.f, 0.f};
    uint32_t row_staged = 0;  // gradient row of this lane's entry in the chunk staged last
    gather(j, p0, p1, p2);              // (the four rings hold the same entries)
    stage(p0, p1, p2, 0, row_staged);   // chunk 0, taken by position t of every ring at step t
    gather(RING + j, p0, p1, p2);       // chunk 1 in flight during round 0
    __syncthreads();
    // LDS address of this lane's entry in the two slots
    const uint32_t ent_addr0 = (uint32_t)(uintptr_t)&ent[0][j][0], ent_addr1 = (uint32_t)(uintptr_t)&ent[1][j][0];

    // Two accumulator sets, named by the PARITY of the chunk they belong to: during round R the lanes
    // that have already taken their entry of chunk R (lane <= t) add into set R & 1, the others still
    // add into the set of chunk R - 1.  A set is therefore complete for ALL lanes when round R + 1 ends
    // ... i.e. it is stored (three full-wave stores) and cleared at the start of round R + 2, instead
    // of every lane storing / clearing its own sums at its own switch step under a one-lane exec mask
    // (where every instruction costs as much as a full-wave one).
    float accA[9], accB[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) accA[c9] = accB[c9] = 0.f;
    uint32_t rowA = 0, rowB = 0;  // gradient rows of the entries the sets belong to
    // index into pix[c][] of the pixel at this lane: ring base + (s - j) mod 16, + 16 within a round
    uint32_t pidx = (uint32_t)(ring * RING_STRIDE + ((RING - j) & (RING - 1)));
    float2 pc[NP2];
#pragma unroll
    for (int c6 = 0; c6 < NP2; c6++) pc[c6] = pix[c6][pidx];
    // state of the pixel currently at this lane: transmittance and the part of Gtot not yet composited
    // (position i of a ring starts at ring lane (16 - i) mod 16; fetch its Gtot from the lane that loaded it)
    float T = __shfl(T0, ring * RING + ((RING - j) & (RING - 1)), 64);
    float Rem = __shfl(gtot0, ring * RING + ((RING - j) & (RING - 1)), 64);

    // A finished entry's nine RAW sums go to its row in HBM.  What is constant per Gaussian -- opacity,
    // the conic combination of the two first moments, the -1/2 and 1/log2(e) factors -- is applied once
    // per Gaussian by segment_reduce_kernel.
    auto write_row = [&](size_t row, const float* a) {
        // (plain stores: the rows are read back by segment_reduce_kernel while much of them is still in the L2 / MALL --
        // streaming "nt" stores made the frame 0.15 ms slower)
        qrows[row * 2] = make_float4(a[0], a[1], a[2], a[3]);
        qrows[row * 2 + 1] = make_float4(a[4], a[5], a[6], a[7]);
        q8[row] = __float_as_uint(a[8]);  // the ninth sum doubles as the "row written" mark
    };

    // An entry k lives at position k mod 16 of every ring for the steps k .. k+15; the loop runs
    // s = 0 .. m+14, so every entry has seen all its pixels when the loop ends.
    const int total = m + RING - 1;
    // the four rings hold partial sums of the same entries: fold them (every lane ends with the total)
    auto fold_rings = [&](float (&a)[9]) {
#pragma unroll
        for (int c9 = 0; c9 < 9; c9++) {
            a[c9] += __shfl_xor(a[c9], 16, 64);
            a[c9] += __shfl_xor(a[c9], 32, 64);
        }
    };
    auto run_round = [&](const int s0, float (&X)[9], uint32_t& rowX, float (&Y)[9]) {
        // Round start (round R = s0 / 16).  X holds the finished sums of chunk R - 2: store them.  The lanes switch to
        // chunk R during this round (staged one round ago): its row is the one noted then.  The chunk that was in flight
        // (R + 1) is converted and staged for the next round; the chunk after it goes in flight.
        if (s0 >= 2 * RING) {
            fold_rings(X);
            if (ring == 0) write_row(rowX, X);
        }
        rowX = row_staged;
        const uint32_t ea = (s0 & RING) ? ent_addr1 : ent_addr0;  // this round's chunk: slot R & 1
        // position 0 of every ring takes its entry now (the other positions at the end of the step before theirs)
        switch_entry<SECOND>(cur, ea, 0ull, 0x0001000100010001ull);
        stage(p0, p1, p2, s0 / RING + 1, row_staged);  // (behind the read above: slot (R + 1) & 1 held chunk R - 1)
        gather(s0 + 2 * RING + j, p0, p1, p2);
        if (s0 > 0) pidx -= (uint32_t)RING;
#pragma unroll
        for (int c9 = 0; c9 < 9; c9++) X[c9] = 0.f;
        const int tend = min(RING, total - s0);
        unsigned long long mx = 0x0001000100010001ull;  // ring positions <= t, kept up to date step by step
        auto step = [&](const int t) {
            const uint32_t s = (uint32_t)(s0 + t);
            const float3 g = make_float3(pc[0].x, pc[0].y, pc[1].x);  // dL/dpixel of the pixel at this lane
            const float pxf = pc[1].y, pyf = pc[2].x;
            const uint32_t lim = __float_as_uint(pc[2].y);
            const float h0 = SECOND ? pc[3].x : 0.f, h1 = SECOND ? pc[3].y : 0.f, h2 = SECOND ? pc[NP2 - 1].x : 0.f;
            // next step's pixel constants, fetched now
            pidx += 1u;
#pragma unroll
            for (int c6 = 0; c6 < NP2; c6++) pc[c6] = pix[c6][pidx];
            const float dx = cur.q0.x - pxf, dy = cur.q0.y - pyf;
            // A2 dx^2 + B2 dx dy + C2 dy^2 in five operations
            const float power2 = __builtin_fmaf(cur.q0.z * dx, dx, __builtin_fmaf(cur.q0.w, dx, cur.q1.x * dy) * dy);
            const float G = __builtin_amdgcn_exp2f(power2);
            const float al = fminf(0.99f, cur.q1.y * G);
            // The pair counts iff power <= 0, the entry lies before the pixel's last contributor (s < lim; a lane that
            // has no entry yet holds zeros, i.e. alpha = 0), and alpha >= 1/255 (which implies the forward's relaxed
            // power2 >= thr pre-test): three compares whose lane masks meet in two scalar ANDs, then one select each
            // for alpha and G (rejected pairs carry alpha = 0).  (Until round 2: two selects more in place of the ANDs.)
            const bool valid = (power2 <= 0.0f) && (s < lim) && (al >= (1.0f / 255.0f));
            const float alpha = valid ? al : 0.f;
            const float Gv = valid ? G : 0.f;
            const float wgt = alpha * T;
            float cg = cur.q1.z * g.x + cur.q1.w * g.y + cur.q2.x * g.z;
            if constexpr (SECOND) cg += cur.q2.y * h0 + cur.q2.z * h1 + cur.q2.w * h2;
            Rem = __builtin_fmaf(-cg, wgt, Rem);
            const float one_m = 1.f - alpha;
            const float dL_dalpha = T * cg - Rem * __builtin_amdgcn_rcpf(one_m);
            T *= one_m;
            const float Gd = Gv * dL_dalpha;  // G dL/dalpha; times opacity it is G dL/dG (applied per Gaussian later)
            const float tdx = Gd * dx, tdy = Gd * dy;
            // sums:  0: t dx   1: t dy   2: t dx^2   3: t dx dy   4: t dy^2   5: G dL/dalpha = dL/dopacity
            //        6..8: w g_c = dL/dcolor     (t = Gd)
            // X += ... on the lanes <= t (they have taken their entry of this round's chunk), Y += ... on the others
            // ... and the positions that take their entry at the next step read it meanwhile (none after the round's last step)
            const unsigned long long mx1 = (mx << 1) | 0x0001000100010001ull;
            // ... and the pixel moves on to the next entry = the next lane: T and Rem rotate by one lane inside their ring
            split_accumulate<SECOND>(X, Y, mx, Gd, dx, dy, tdx, tdy, wgt, g.x, g.y, g.z, mx1, ea, cur.q0, cur.q1, cur.q2, T, Rem);
            mx = mx1;
        };
        // two steps per trip: the one-step-ahead pixel constants alternate between two register sets
        int t = 0;
        for (; t + 1 < tend; t += 2) {
            step(t);
            step(t + 1);
        }
        if (t < tend) step(t);
    };
    for (int s0 = 0; s0 < total; s0 += 2 * RING) {
        run_round(s0, accA, rowA, accB);
        if (s0 + RING < total) run_round(s0 + RING, accB, rowB, accA);
    }
    {
        // the two chunks still in the sets: the one taken during the last round and the one before it
        const int last_round = (total - 1) / RING;
        const bool odd = last_round & 1;
        const int k_new = RING * last_round + j, k_old = k_new - RING;
        fold_rings(accA);
        fold_rings(accB);
        if (ring == 0) {
            if (k_new <= m - 1) write_row(odd ? rowB : rowA, odd ? accB : accA);
            if (last_round >= 1 && k_old <= m - 1) write_row(odd ? rowA : rowB, odd ? accA : accB);
        }
    }
}

int launch_render_backward(const float* rec, const uint32_t* ranges, const uint32_t* order, int W, int H,
                           const QuadLists& ql, const float* out_color, const float* dL_dpix, const float* dL_dopa,
                           const float* final_T, const float* bg, float* qrows, uint32_t* q8, const SecondImage* second,
                           L1Grad l1, hipStream_t s) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int xmap = gs_tune_get(GS_TUNE_XCD_MAP);
    const int chunks = ql.ckpt ? ql.chunks : 1;
    const int bpc = render_grid_blocks(gx * gy, xmap);
    const dim3 grid((unsigned)bpc * (unsigned)chunks);
    const SecondImage none{nullptr, nullptr, nullptr, nullptr, nullptr};
    const float l1_inv_n = 1.0f / (3.0f * (float)W * (float)H);
#define GS_BWD_ARGS                                                                                                       \
    reinterpret_cast<const float4*>(rec), reinterpret_cast<const uint2*>(ranges), order, W, H, gx, gx * gy, xmap, ql.qlist, \
        ql.ncon_c, ql.qcount, out_color, dL_dpix, dL_dopa, final_T, bg, reinterpret_cast<float4*>(qrows), q8, ql.ckpt,    \
        ql.ck_start, chunks, bpc
    if (second) {
        hipLaunchKernelGGL(render_bwd_kernel<2>, grid, dim3(64), 0, s, GS_BWD_ARGS, *second, l1, l1_inv_n);  // (one of the two leaves at once)
        hipLaunchKernelGGL(render_bwd_kernel<3>, grid, dim3(64), 0, s, GS_BWD_ARGS, *second, l1, l1_inv_n);
    } else if (dL_dopa)
        hipLaunchKernelGGL(render_bwd_kernel<1>, grid, dim3(64), 0, s, GS_BWD_ARGS, none, l1, l1_inv_n);
    else
        hipLaunchKernelGGL(render_bwd_kernel<0>, grid, dim3(64), 0, s, GS_BWD_ARGS, none, l1, l1_inv_n);
#undef GS_BWD_ARGS
    GS_LAUNCH_CHECK("render_backward", 0, s);
    return GS_OK;
}
