// s360_bwd_em.h — entry-major backward composite (k_render_bwd_em).  gfx950 / wave64 only.
//
// The round-1 kernel kept the forward's layout (64 PIXELS of an 8x8 quadrant in the lanes, one list entry at a time):
// per surviving entry it paid ~50 cross-lane instructions to reduce nine gradients over the pixels, with 15.5 of 64
// lanes contributing on average.  This kernel transposes the roles:
//   * the wave walks its tile's depth-sorted list 64 entries at a time exactly like the forward, culls each chunk
//     against its quadrant and COMPACTS the survivors into a 128-slot queue in LDS (48-byte records: centre, conic,
//     opacity, colour, [depth value], list position, pair — the instance slot is gathered when the entry is popped);
//   * whenever 64 survivors are queued they are popped into the lanes — lane j holds ONE ENTRY — and the wave loops
//     over the quadrant's 64 pixels, four at a time.  Per pixel the transmittance around every entry is a scanned
//     PRODUCT of (1 - alpha) over the lanes and the colour behind it a scanned SUM: two
//     six-step DPP scans (row_shr 1/2/4/8, row_bcast 15/31) replace the nine 64-lane reductions, and every lane
//     accumulates its own entry's nine sums in registers.  The per-pixel constants (dL/dpixel, background term,
//     n_contrib) and the running (T_run, R_run) live in LDS, one pixel-contiguous array per quantity, and are read as
//     16-byte broadcasts: a four-pixel run arrives as two register pairs and the per-pixel arithmetic runs on PAIRS
//     (v_pk_fma / v_pk_mul / v_pk_add_f32): 99 M instead of 138 M wave instructions;
//   * launch order (k_order_units): tiles heavy first, the four quadrant waves of a tile on the same XCD (shared L2);
//   * back to front like upstream's backward: the walk starts at the quadrant's last contributor, the queue hands
//     out entries in DEscending list position (lane 0 = backmost), so an inclusive scan over the lanes is a SUFFIX
//     in list order: T in front of entry j = T_behind_group / prod_{i at or behind j}(1 - alpha_i), the colour
//     behind it = R_behind_group + sum_{i behind j} alpha_i T_i (c_i . dL/dpixel).  Both start from exact values
//     (final_T, 0) and every quantity keeps RELATIVE accuracy (a front-to-back variant that took R from
//     "rendered pixel - prefix" measured 1.6x faster than round 1 too, but its absolute cancellation error of
//     ~1e-7, amplified by 1 / (1 - alpha) <= 100, showed up as 2.4x the float32 oracle's own distance from the
//     float64 oracle on ill-conditioned splats).
// Accept / reject decisions per (pixel, entry) use the forward's own power2() / alpha expressions and its n_contrib,
// so they are identical to the forward's.  Results are deterministic (fixed group composition, no atomics).
// The optional depth channel (WITH_DEPTH) makes the fused depth map of s360_forward_depth differentiable: it is one more
// "colour" channel for dL/dalpha plus a per-entry gradient with respect to the entry's depth value.
#pragma once
#include "s360_device.h"

namespace s360 {

constexpr int EM_QCAP = 128;  // survivor queue slots per wave (<= 63 left over + 64 appended)


#define S360_SCAN4_STEP(op, ctrl)                                                                      \
    op " %0, %0, %0 " ctrl "\n" op " %1, %1, %1 " ctrl "\n" op " %2, %2, %2 " ctrl "\n" op " %3, %3, %3 " ctrl "\n"
// Four independent inclusive wave64 scans (lane j <- op over lanes 0..j), interleaved so that the three other chains
// sit between two dependent DPP operations on one register (VALU write -> DPP read needs two wait states).
// Kogge-Stone inside the 16-lane rows (a lane whose source falls outside its row is left unchanged: bound_ctrl off,
// destination tied to the second operand), then lane 15 -> row 1 / 3 and lane 31 -> rows 2, 3.
#define S360_SCAN4(op, a, b, c, d)                                                        \
    asm volatile("s_nop 1\n"                                                              \
                 S360_SCAN4_STEP(op, "row_shr:1 row_mask:0xf bank_mask:0xf")              \
                 S360_SCAN4_STEP(op, "row_shr:2 row_mask:0xf bank_mask:0xf")              \
                 S360_SCAN4_STEP(op, "row_shr:4 row_mask:0xf bank_mask:0xf")              \
                 S360_SCAN4_STEP(op, "row_shr:8 row_mask:0xf bank_mask:0xf")              \
                 S360_SCAN4_STEP(op, "row_bcast:15 row_mask:0xa bank_mask:0xf")           \
                 S360_SCAN4_STEP(op, "row_bcast:31 row_mask:0xc bank_mask:0xf")           \
                 "s_nop 1\n"                                                              \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// Four wave64 shifts by one lane (lane j <- lane j-1, lane 0 <- 0: bound_ctrl zero-fill): turns an inclusive scan into
// an exclusive one.
__device__ __forceinline__ void wave_shr1x4_zero(float& a, float& b, float& c, float& d) {
    float oa, ob, oc, od;
    asm volatile("s_nop 1\n"
                 "v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 "v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 "v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 : "=&v"(oa), "=&v"(ob), "=&v"(oc), "=&v"(od)
                 : "v"(a), "v"(b), "v"(c), "v"(d));
    a = oa; b = ob; c = oc; d = od;
}
// out_j = 1 - in_{j-1} (lane 0: 1 - 0): the shifted (1 - alpha) factors in one instruction each.
__device__ __forceinline__ void wave_one_minus_shr1x4(float a, float b, float c, float d, float& oa, float& ob, float& oc,
                                                      float& od) {
    const float one = 1.0f;
    asm volatile("s_nop 1\n"
                 "v_subrev_f32_dpp %0, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 "v_subrev_f32_dpp %1, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 "v_subrev_f32_dpp %2, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 "v_subrev_f32_dpp %3, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                 : "=&v"(oa), "=&v"(ob), "=&v"(oc), "=&v"(od)
                 : "v"(a), "v"(b), "v"(c), "v"(d), "v"(one));
}

__device__ __forceinline__ void wave_scan4_mul(float& a, float& b, float& c, float& d) { S360_SCAN4("v_mul_f32_dpp", a, b, c, d); }
__device__ __forceinline__ void wave_scan4_add(float& a, float& b, float& c, float& d) { S360_SCAN4("v_add_f32_dpp", a, b, c, d); }

// d(depth_value)/dz for the fused depth channel (s360_device.h depth_value; the "log" mode keeps the reference's
// swapped clamp, whose result does not depend on z whenever near < far).
__device__ __forceinline__ float depth_value_grad(float z, float nearp, float farp, int mode) {
    if (mode == 1) return -1.0f / (z * z);
    if (mode == 2) {
        const float eps = 1e-10f;
        const float disp_near = 1.0f / (nearp + eps), disp_far = 1.0f / (farp + eps), disp = 1.0f / (z + eps);
        return (disp * disp) / (disp_near - disp_far + eps);
    }
    if (mode == 3) return (z < nearp && z > farp) ? 1.0f / z : 0.0f;
    return 1.0f;
}

// S360_FLAG_SPLIT_LISTS (include/s360.h): the forward composited the rest of a long list in segments and left, per (segment slot,
// quadrant, pixel), the transmittance behind the segment (seg_t) and the colour accumulated behind it (seg_c).  The backward replays
// the head and every segment as INDEPENDENT units from those two: same entry-major arithmetic per unit, the per-pixel running state
// (T behind, colour behind) starts from the forward's values instead of (final_T, 0).
typedef SegBwdPod SegBwd;

#ifdef S360_EM_KERNEL_TU
template <bool WITH_DEPTH, bool SEG>   // SEG: the call was rendered with S360_FLAG_SPLIT_LISTS (segment units exist); false: none of that code
__global__ __launch_bounds__(64) void k_render_bwd_em(
    KParams kp, const S360View* __restrict__ views, const uint32_t* __restrict__ tile_start,
    const float4* __restrict__ surv, const uint32_t* __restrict__ surv_count, const uint2* __restrict__ slot_info,
    const float* __restrict__ depths, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dimages, const float* __restrict__ dL_dimages_scale, const float* __restrict__ dL_ddepth,
    float4* __restrict__ part,
    uint8_t* __restrict__ valid, const uint32_t* __restrict__ order, int depth_mode, float* __restrict__ pairgrad_atomic,
    uint32_t* __restrict__ dbg, SegBwd sb, uint32_t n_seg_blocks) {
    static_assert(SUB_W == 8, "entry-major backward assumes 8x8 quadrants");
    // per-pixel tables, one array per quantity (pixel-contiguous: a 16-byte read hands a four-pixel run to the lanes as two
    // register PAIRS — the operands of the packed v_pk_* arithmetic below)
    __shared__ __attribute__((aligned(16))) float s_gr[64], s_gg[64], s_gb[64], s_gd[64];  // dL/dpixel (r, g, b, depth)
    __shared__ __attribute__((aligned(16))) float s_T[64], s_R[64];  // T_run, R_run behind the entries processed so far
    __shared__ __attribute__((aligned(16))) float s_B[64];           // T_final * (bg . dL/dpixel)
    __shared__ __attribute__((aligned(16))) uint32_t s_last[64];     // n_contrib

    // (SEG = false — calls rendered without S360_FLAG_SPLIT_LISTS — compiles none of the segment code: its seven pointers and the
    // unit loop cost this kernel 12 VGPRs (SGPRs spilled into VGPR lanes), and the depth variant a wave per SIMD)
    const bool seg_blk = SEG && blockIdx.x < n_seg_blocks;   // segment units of split quadrants first: they are the heavy ones
    const int lane = threadIdx.x;
  for (uint32_t sj = blockIdx.x;; sj += n_seg_blocks) {     // one unit per workgroup, except the segment workgroups (grid-stride)
#ifdef S360_DBG_TIMING
    const long long t_begin = wall_clock64();
#endif
    uint32_t unit, n_surv, kseg = 0;
    bool split_unit = false;
    size_t sli = 0;     // split units: index of this (slot, quadrant)'s pixel 0 in seg_c / seg_t
    if (seg_blk) {
        if (sj >= sb.seg_list[0]) return;
        uint2 info = sb.seg_info[sb.seg_list[1 + sj]];   // the forward's work item: (tile, segment << 2 | quadrant)
        info.y &= ~S360_SEG_CLAIM;
        unit = 4u * info.x + (info.y & 3u);
        kseg = info.y >> 2;
        const size_t su = ((size_t)SEG_PER_CHUNK * sb.chunk_start[info.x] + kseg) * 4 + (info.y & 3u);   // slot * 4 + quadrant
        n_surv = sb.seg_cnt[su];
        split_unit = true;
        sli = su * 64;
    } else {
        unit = order ? order[blockIdx.x - n_seg_blocks] : blockIdx.x - n_seg_blocks;  // tile*4 + quadrant
        n_surv = surv_count[unit];  // survivor records in front of the quadrant's last contributor
        if (n_surv == 0) return;    // nothing reaches any pixel of this quadrant
        if (SEG && sb.seg_flag[unit] == 1u) {   // the head of a split quadrant
            split_unit = true;
            sli = ((size_t)SEG_PER_CHUNK * sb.chunk_start[unit >> 2] * 4 + (unit & 3u)) * 64;
        }
    }
    const int t = (int)(unit >> 2), wave = (int)(unit & 3u);
    const int v = t / kp.T, rem = t - v * kp.T;
    const int ty = rem / kp.gx, tx = rem - ty * kp.gx;
    const int qx = tx * 16 + sub_ox(wave), qy = ty * 16 + sub_oy(wave);
    const uint32_t start = min(tile_start[t], kp.cap), end = min(tile_start[t + 1], kp.cap);
    // this unit's records (a segment's start at its first list position: at most one record per entry in front of it)
    const float4* const sv = surv + 3 * ((size_t)4 * start + (size_t)wave * (end - start) + (size_t)kseg * SEG_LEN);
    const S360View& vw = views[view_of_image(kp, v)];  // v = image index

    // the first group's records: in flight while the pixel tables are set up
    int64_t top = (int64_t)n_surv - 1;  // lane l of a group holds record top - l (descending list position)
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, nc = na;
    if (top - lane >= 0) {
        const float4* r = sv + 3 * (size_t)(top - lane);
        na = r[0]; nb = r[1]; nc = r[2];
    }

    // ---- per-pixel constants (lane = pixel of the quadrant) ----
    {
        const int px = qx + (lane & 7), py = qy + (lane >> 3);
        const size_t hw = (size_t)kp.H * kp.W;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = make_float4(1.0f, 0.f, 0.f, 0.f);
        uint32_t last = 0;
        if (px < kp.W && py < kp.H) {
            const size_t pix = (size_t)py * kp.W + px;
            const float T_final = final_T[(size_t)v * hw + pix];
            last = n_contrib[(size_t)v * hw + pix];
            const float* dimg = dL_dimages + (size_t)v * 3 * hw;
            const float gs = dL_dimages_scale ? *dL_dimages_scale : 1.0f;  // scalar of a pre-stored loss seed (s360.h)
            pa.x = dimg[pix] * gs;
            pa.y = dimg[hw + pix] * gs;
            pa.z = dimg[2 * hw + pix] * gs;
            if (WITH_DEPTH) pa.w = dL_ddepth[(size_t)v * hw + pix];  // depth background is 0: no background term
            pb.x = T_final;
            pb.z = T_final * (vw.bg[0] * pa.x + vw.bg[1] * pa.y + vw.bg[2] * pa.z);
            if (SEG && split_unit) {   // wave-uniform: start behind this unit's backmost entry, not behind the whole list
                const float4 cb = sb.seg_c[sli + lane];
                pb.x = sb.seg_t[sli + lane];
                float r0 = cb.z * pa.z + (cb.y * pa.y + cb.x * pa.x);
                if (WITH_DEPTH) r0 = cb.w * pa.w + r0;
                pb.y = r0;      // colour (. dL/dpixel) behind the unit
            }
        }
        s_gr[lane] = pa.x; s_gg[lane] = pa.y; s_gb[lane] = pa.z;
        if (WITH_DEPTH) s_gd[lane] = pa.w;
        s_T[lane] = pb.x; s_R[lane] = pb.y; s_B[lane] = pb.z;
        s_last[lane] = last;
    }
    f2 pxc[4];  // pixel-centre x of the row's four pixel pairs
#pragma unroll
    for (int c = 0; c < 4; ++c) pxc[c] = f2{(float)(qx + 2 * c), (float)(qx + 2 * c + 1)};
    const float inv_scale = WITH_DEPTH ? 1.0f / vw.scale : 0.f;
    const float v_near = WITH_DEPTH ? vw.near_plane : 0.f, v_far = WITH_DEPTH ? vw.far_plane : 0.f;

#ifdef S360_DBG_TIMING
    uint32_t dbg_halves = 0;
#endif
    // ---- groups of up to 64 survivors in the lanes (descending list position); per group a loop over the 64 pixels ----
    for (; top >= 0; top -= 64) {
        const uint32_t n = (uint32_t)(top + 1 < 64 ? top + 1 : 64);
        const float4 qa = na, qb = nb, qc = nc;
        if (top - 64 - lane >= 0) {  // the next (nearer) group's records fly during this group's pixel loop
            const float4* r = sv + 3 * (size_t)(top - 64 - lane);
            na = r[0]; nb = r[1]; nc = r[2];
        }
        const bool lane_ok = (uint32_t)lane < n;
        const float ex = qa.x, ey = qa.y, cA = qa.z, cB = qa.w, cC = qb.x, op = qb.y, c0 = qb.z, c1 = qb.w, c2 = qc.x;
        const int erad = __float_as_int(qc.y);
        const uint32_t pos = lane_ok ? __float_as_uint(qc.z) : 0xFFFFFFFFu;  // list position; idle lanes never contribute
        // the entry's instance slot (where its partial record goes) is only needed at the very end: the gather of the pair's
        // first slot flies during the pixel loop, and only SURVIVORS pay for it
        const uint32_t pair = __float_as_uint(qc.w);
        uint2 sinfo = make_uint2(0u, 0u);   // (the pair's first slot, lean lists: which tiles of its rectangle hold instances)
        float zv = 0.f;
        if (lane_ok) {
            if (!pairgrad_atomic) sinfo = slot_info[pair];
            if (WITH_DEPTH) zv = depth_value(depths[pair] * inv_scale, v_near, v_far, depth_mode);
        }
        // positions descend with the lane: the group's frontmost entry sits in lane n - 1
        const uint32_t pos_min = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)n - 1);
        // every per-entry sum is kept as a PAIR (even / odd pixels of the runs), added up once at the end
        f2 g_op = f2{0.f, 0.f}, X = g_op, Y = g_op, XX = g_op, XY = g_op, YY = g_op, g_r = g_op, g_g = g_op, g_b = g_op, g_z = g_op;
        bool anyc = false;
        // which of the 16 four-pixel runs can still receive anything from this group (some pixel's last contributor lies at
        // or in front of the group's frontmost entry): one LDS pass up front, so that the loop below branches on a scalar bit
        // instead of waiting for the pixel table before it can decide
        uint32_t qmask;
        {
            const uint4 l4 = *reinterpret_cast<const uint4*>(&s_last[(lane & 15) * 4]);
            qmask = (uint32_t)__ballot(max(max(l4.x, l4.y), max(l4.z, l4.w)) > pos_min) & 0xFFFFu;
        }
        const f2 ex2 = f2{ex, ex}, cA2 = f2{cA, cA}, cB2 = f2{cB, cB}, op2 = f2{op, op}, c02 = f2{c0, c0}, c12 = f2{c1, c1},
                 c22 = f2{c2, c2}, zv2 = f2{zv, zv}, one2 = f2{1.0f, 1.0f};
#pragma unroll 1
        for (int row = 0; row < 8; ++row) {
            const float pyf = (float)(qy + row);
            const float dy = ey - pyf;
            const float cdy = cC * dy;
            const float cdy2 = cdy * dy;  // shared by the row's pixels: power2() = fma(fma(b,dy,a*dx), dx, (c*dy)*dy)
            const float cdy_2 = 2.0f * cdy;
            const f2 dy2 = f2{dy, dy}, cdy2_2 = f2{cdy2, cdy2}, cdy_22 = f2{cdy_2, cdy_2};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int p0 = row * 8 + half * 4;
                if (!((qmask >> (row * 2 + half)) & 1u)) continue;  // none of the four pixels reaches back to this group
#ifdef S360_DBG_TIMING
                ++dbg_halves;
#endif
                const float4 vr = *reinterpret_cast<const float4*>(&s_gr[p0]), vg = *reinterpret_cast<const float4*>(&s_gg[p0]),
                             vb = *reinterpret_cast<const float4*>(&s_gb[p0]), vT = *reinterpret_cast<const float4*>(&s_T[p0]),
                             vR = *reinterpret_cast<const float4*>(&s_R[p0]), vB = *reinterpret_cast<const float4*>(&s_B[p0]);
                const uint4 vL = *reinterpret_cast<const uint4*>(&s_last[p0]);
                float4 vd = make_float4(0.f, 0.f, 0.f, 0.f);
                if (WITH_DEPTH) vd = *reinterpret_cast<const float4*>(&s_gd[p0]);
                const f2 gr[2] = {f2{vr.x, vr.y}, f2{vr.z, vr.w}}, gg[2] = {f2{vg.x, vg.y}, f2{vg.z, vg.w}},
                         gb[2] = {f2{vb.x, vb.y}, f2{vb.z, vb.w}}, gd[2] = {f2{vd.x, vd.y}, f2{vd.z, vd.w}},
                         Tr[2] = {f2{vT.x, vT.y}, f2{vT.z, vT.w}}, Rr[2] = {f2{vR.x, vR.y}, f2{vR.z, vR.w}},
                         Bg[2] = {f2{vB.x, vB.y}, f2{vB.z, vB.w}};
                const uint32_t lastk[4] = {vL.x, vL.y, vL.z, vL.w};
                f2 dx[2], pxd[2], Gm[2], a2[2];
                float a[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dx[j] = ex2 - pxc[half * 2 + j];
                    const f2 adx = cA2 * dx[j];
                    const f2 tq = pk_fma(cB2, dy2, adx);
                    pxd[j] = tq + adx;  // d(power)/d(dx) = 2 a' dx + b' dy
                    const f2 pw = pk_fma(tq, dx[j], cdy2_2);
                    const f2 G = f2{__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
                    const f2 og = op2 * G;
                    const float al0 = fminf(0.99f, og.x), al1 = fminf(0.99f, og.y);
                    const bool act0 = pos < lastk[2 * j] && !(pw.x > 0.0f) && !(al0 < 1.0f / 255.0f);
                    const bool act1 = pos < lastk[2 * j + 1] && !(pw.y > 0.0f) && !(al1 < 1.0f / 255.0f);
                    anyc = anyc || act0 || act1;
                    a[2 * j] = act0 ? al0 : 0.0f;
                    a[2 * j + 1] = act1 ? al1 : 0.0f;
                    Gm[j] = f2{act0 ? G.x : 0.0f, act1 ? G.y : 0.0f};
                    a2[j] = f2{a[2 * j], a[2 * j + 1]};
                }
                // Qx_j = prod of (1 - alpha) over the entries strictly BEHIND j (lanes below j), Q_j includes j
                float Qx[4];
                wave_one_minus_shr1x4(a[0], a[1], a[2], a[3], Qx[0], Qx[1], Qx[2], Qx[3]);
                wave_scan4_mul(Qx[0], Qx[1], Qx[2], Qx[3]);
                f2 Tj[2], rc[2], cdp[2], w[2];
                float S[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f2 Qx2 = f2{Qx[2 * j], Qx[2 * j + 1]};
                    const f2 Q = Qx2 * (one2 - a2[j]);
                    f2 r = f2{__builtin_amdgcn_rcpf(Q.x), __builtin_amdgcn_rcpf(Q.y)};
                    r = pk_fma(pk_fma(-Q, r, one2), r, r);  // Newton step: dL/dalpha below is a difference of two nearly equal
                                                            // terms whenever an entry's colour is close to the colour behind
                                                            // it; without it the fuzz suite measures 2.7x the float32 oracle's
                                                            // distance from the float64 one
                    Tj[j] = Tr[j] * r;   // transmittance in front of entry j = T behind the group / Q_j
                    rc[j] = Qx2 * r;     // 1 / (1 - alpha_j)
                    f2 d = pk_fma(c22, gb[j], pk_fma(c12, gg[j], c02 * gr[j]));
                    if (WITH_DEPTH) d = pk_fma(zv2, gd[j], d);
                    cdp[j] = d;
                    w[j] = a2[j] * Tj[j];
                    const f2 s2 = w[j] * d;
                    S[2 * j] = s2.x;
                    S[2 * j + 1] = s2.y;
                }
                wave_scan4_add(S[0], S[1], S[2], S[3]);  // sum over the entries at or behind j of alpha_i T_i (c_i . dL/dpixel)
                float Sx[4] = {S[0], S[1], S[2], S[3]};
                wave_shr1x4_zero(Sx[0], Sx[1], Sx[2], Sx[3]);  // ... strictly behind j
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f2 Rj = Rr[j] + f2{Sx[2 * j], Sx[2 * j + 1]};  // colour (. dL/dpixel) behind entry j
                    const f2 dLda = pk_fma(Tj[j], cdp[j], -((Rj + Bg[j]) * rc[j]));
                    g_op = pk_fma(Gm[j], dLda, g_op);
                    const f2 h = (op2 * dLda) * Gm[j];  // dL/dG * G
                    const f2 hx = h * dx[j], hy = h * dy2;
                    // centre gradient per pixel (NOT 2 a' sum(h dx) + b' sum(h dy) afterwards: for an elongated splat the two
                    // sums cancel along the ridge and their rounding errors do not — 8x the float32 oracle's error on
                    // the fuzz suite's most anisotropic splat)
                    X = pk_fma(h, pxd[j], X);
                    Y = pk_fma(h, pk_fma(cB2, dx[j], cdy_22), Y);
                    XX = pk_fma(hx, dx[j], XX);
                    XY = pk_fma(hx, dy2, XY);
                    YY = pk_fma(hy, dy2, YY);
                    g_r = pk_fma(w[j], gr[j], g_r);
                    g_g = pk_fma(w[j], gg[j], g_g);
                    g_b = pk_fma(w[j], gb[j], g_b);
                    if (WITH_DEPTH) g_z = pk_fma(w[j], gd[j], g_z);
                }
                if (lane == 63) {  // lane 63 sees the whole group: the pixels' running state for the next (nearer) group
                    *reinterpret_cast<float4*>(&s_T[p0]) = make_float4(Tj[0].x, Tj[0].y, Tj[1].x, Tj[1].y);
                    *reinterpret_cast<float4*>(&s_R[p0]) = make_float4(vR.x + S[0], vR.y + S[1], vR.z + S[2], vR.w + S[3]);
                }
            }
        }
        if (pairgrad_atomic) {   // S360_FLAG_ATOMIC_GRADS (wave-uniform): ten return-less float32 atomics into the pair's record
            if (lane_ok && anyc) {
                const float ln2 = 0.6931471805599453f;
                float* pg = pairgrad_atomic + 12 * (size_t)pair;
                unsafeAtomicAdd(pg + 0, ln2 * (X.x + X.y));
                unsafeAtomicAdd(pg + 1, ln2 * (Y.x + Y.y));
                unsafeAtomicAdd(pg + 2, -0.5f * (XX.x + XX.y));
                unsafeAtomicAdd(pg + 3, -(XY.x + XY.y));
                unsafeAtomicAdd(pg + 4, -0.5f * (YY.x + YY.y));
                unsafeAtomicAdd(pg + 5, g_op.x + g_op.y);
                unsafeAtomicAdd(pg + 6, g_r.x + g_r.y);
                unsafeAtomicAdd(pg + 7, g_g.x + g_g.y);
                unsafeAtomicAdd(pg + 8, g_b.x + g_b.y);
                if (WITH_DEPTH) unsafeAtomicAdd(pg + 9, g_z.x + g_z.y);
            }
            continue;
        }
        uint32_t inst = 0xFFFFFFFFu;
        if (lane_ok && anyc) {  // position of tile (tx,ty) inside the splat's tile rectangle, in emission order
            int minx, miny, maxx, maxy;
            tile_rect(ex, ey, erad, kp.gx, kp.gy, minx, miny, maxx, maxy);
            const uint32_t idx = (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx));
            // lean lists (rectangles of up to 32 tiles): the slots follow the set bits of the hit mask
            const bool masked = (kp.flags & S360_FLAG_LEAN_LISTS) && (maxx - minx) * (maxy - miny) <= 32;
            inst = sinfo.x + (masked ? (uint32_t)__builtin_popcount(sinfo.y & ((1u << idx) - 1u)) : idx);
        }
        if (lane_ok && anyc && inst < kp.cap) {
            const float ln2 = 0.6931471805599453f;
            // dG/d(centre) = ln2 G (2 a' dx + b' dy) with the pre-scaled conic; dG/d(conic a) = -G dx^2 / 2 ...
            float4* o = part + ((size_t)inst * 4 + wave) * PREC_F4;
            o[0] = make_float4(ln2 * (X.x + X.y), ln2 * (Y.x + Y.y), -0.5f * (XX.x + XX.y), -(XY.x + XY.y));
            o[1] = make_float4(-0.5f * (YY.x + YY.y), g_op.x + g_op.y, g_r.x + g_r.y, g_g.x + g_g.y);
            o[2] = make_float4(g_b.x + g_b.y, g_z.x + g_z.y, 0.f, 0.f);
            valid[(size_t)inst * 4 + wave] = 1;
        }
    }
#ifdef S360_DBG_TIMING
    if (lane == 0 && dbg) {  // per-unit (start, duration) in 100-MHz ticks + replay length (scripts/bwdtiming.py)
        const size_t di = seg_blk ? (size_t)sb.dbg_base + sb.seg_list[1 + sj] : (size_t)unit;
        dbg[4 * di] = (uint32_t)t_begin;
        dbg[4 * di + 1] = (uint32_t)(wall_clock64() - t_begin);
        dbg[4 * di + 2] = n_surv;
        dbg[4 * di + 3] = (dbg_halves << 16) | min(n_surv, 65535u);
    }
#endif
    if (!SEG || !seg_blk) return;
  }
}
#endif  // S360_EM_KERNEL_TU

// Launcher (this kernel lives in its own translation unit, s360_backward_em.hip, which is compiled with
// -fno-slp-vectorize: the pixel pairs are packed by hand above — pixel-contiguous LDS tables, pair accumulators — which
// costs no register moves; the SLP vectoriser's own pairing of the scalar formulation paid 23 moves per half row).
void launch_render_bwd_em(bool with_depth, int n_units, hipStream_t st, const KParams& kp, const S360View* views,
                          const uint32_t* tile_start, const float4* surv, const uint32_t* surv_count, const uint2* slot_info,
                          const float* depths, const float* final_T, const uint32_t* n_contrib, const float* dL_dimages,
                          const float* dL_dimages_scale, const float* dL_ddepth, float4* part, uint8_t* valid, const uint32_t* order, int depth_mode,
                          float* pairgrad_atomic, uint32_t* dbg, const SegBwd* sbp_host, uint32_t n_seg_blocks);

}  // namespace s360
