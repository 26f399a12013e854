// s360_bwd_math.h — the per-(pixel, list entry) arithmetic of the backward composite, as a plain inline function
// (no GPU builtins) so that the SAME source is compiled into k_render_bwd and, on the host, into the bit-equivalence
// test of its two formulations (tests/test_bwd_math.py):
//   bwd_entry_scalar   the reference formulation, one float operation per line, in the oracle's operation order
//   bwd_entry_packed   the same IEEE operations in the same order per value, written on 2-vectors so that hipcc emits
//                      v_pk_mul_f32 / v_pk_add_f32 (two lanes' worth of work per VALU issue slot on gfx950)
// Both are compiled with -ffp-contract=off; the only fused operation is the explicit Newton step of the reciprocal,
// which happens before this function (rcp is an input).
#pragma once

#if defined(__HIPCC__) || defined(__CUDACC__)
#define S360_HD __host__ __device__ __forceinline__
#else
#define S360_HD inline
#endif

namespace s360 {

struct BwdPixel {     // per-pixel state carried along the list (back to front)
    float T, acc0, acc1, acc2, lc0, lc1, lc2, last_alpha;
};
struct BwdConst {     // per-pixel constants
    float dp0, dp1, dp2, T_final, bg_dot;
};
struct BwdOut {       // the nine per-pixel raster-gradient terms of one entry (summed over the pixels afterwards)
    float g_x, g_y, g_A, g_B, g_C, g_op, g_r, g_g, g_b;
};

// a_eff / G_eff: alpha and G of the entry at this pixel, or 0 where the entry does not contribute;
// rcp = 1 / (1 - a_eff); (cA, cB, cC) pre-scaled conic, op opacity, (c0, c1, c2) colour, (dx, dy) centre - pixel.
S360_HD void bwd_entry_scalar(BwdPixel& s, const BwdConst& k, float a_eff, float G_eff, float rcp, float cA, float cB,
                              float cC, float op, float c0, float c1, float c2, float dx, float dy, BwdOut& o) {
    s.T = s.T * rcp;
    const float dchannel_dcolor = a_eff * s.T;
    s.acc0 = s.last_alpha * s.lc0 + (1.f - s.last_alpha) * s.acc0;
    s.acc1 = s.last_alpha * s.lc1 + (1.f - s.last_alpha) * s.acc1;
    s.acc2 = s.last_alpha * s.lc2 + (1.f - s.last_alpha) * s.acc2;
    s.lc0 = c0; s.lc1 = c1; s.lc2 = c2;
    float dL_dalpha = (c0 - s.acc0) * k.dp0 + (c1 - s.acc1) * k.dp1 + (c2 - s.acc2) * k.dp2;
    o.g_r = dchannel_dcolor * k.dp0;
    o.g_g = dchannel_dcolor * k.dp1;
    o.g_b = dchannel_dcolor * k.dp2;
    dL_dalpha *= s.T;
    s.last_alpha = a_eff;
    dL_dalpha += (-k.T_final * rcp) * k.bg_dot;
    const float dL_dG = op * dL_dalpha;
    const float gdx = G_eff * dx, gdy = G_eff * dy;
    // dG/d(delta) = -G (a dx + b dy) = ln2 * G (2 a' dx + b' dy)   (a' = -log2e/2 a, b' = -log2e b)
    const float dG_ddelx = 0.6931471805599453f * (2.0f * gdx * cA + gdy * cB);
    const float dG_ddely = 0.6931471805599453f * (2.0f * gdy * cC + gdx * cB);
    o.g_x = dL_dG * dG_ddelx;
    o.g_y = dL_dG * dG_ddely;
    o.g_A = -0.5f * gdx * dx * dL_dG;
    o.g_B = -gdx * dy * dL_dG;
    o.g_C = -0.5f * gdy * dy * dL_dG;
    o.g_op = G_eff * dL_dalpha;
}

typedef float f2v __attribute__((ext_vector_type(2)));

// Same operations, same order per value, on 2-vectors.  Pairs: (acc0, acc1), (g_r, g_g), (x, y) for the centre
// offsets / G-weighted offsets / dG/d(delta) / (g_x, g_y), (g_A, g_C).  The third channel and g_B / g_op stay scalar.
S360_HD void bwd_entry_packed(BwdPixel& s, const BwdConst& k, float a_eff, float G_eff, float rcp, float cA, float cB,
                              float cC, float op, float c0, float c1, float c2, float dx, float dy, BwdOut& o) {
    s.T = s.T * rcp;
    const float dchannel_dcolor = a_eff * s.T;
    const float la = s.last_alpha, om = 1.f - s.last_alpha;
    const f2v lc01 = {s.lc0, s.lc1}, acc01_old = {s.acc0, s.acc1}, c01 = {c0, c1}, dp01 = {k.dp0, k.dp1};
    const f2v acc01 = la * lc01 + om * acc01_old;          // la*lc + (1-la)*acc, per channel
    const float acc2 = la * s.lc2 + om * s.acc2;
    s.acc0 = acc01.x; s.acc1 = acc01.y; s.acc2 = acc2;
    s.lc0 = c0; s.lc1 = c1; s.lc2 = c2;
    const f2v p01 = (c01 - acc01) * dp01;
    float dL_dalpha = p01.x + p01.y + (c2 - acc2) * k.dp2;  // ((p0 + p1) + p2)
    const f2v g_rg = dchannel_dcolor * dp01;
    o.g_r = g_rg.x;
    o.g_g = g_rg.y;
    o.g_b = dchannel_dcolor * k.dp2;
    dL_dalpha *= s.T;
    s.last_alpha = a_eff;
    dL_dalpha += (-k.T_final * rcp) * k.bg_dot;
    const float dL_dG = op * dL_dalpha;
    const f2v d = {dx, dy}, d_swapped = {dy, dx};
    const f2v gd = G_eff * d;                               // (gdx, gdy)
    const f2v gd_swapped = {gd.y, gd.x};
    const f2v cAC = {cA, cC};
    // x: ln2 * ((2*gdx)*cA + gdy*cB)   y: ln2 * ((2*gdy)*cC + gdx*cB)
    const f2v dG_ddel = 0.6931471805599453f * (2.0f * gd * cAC + gd_swapped * cB);
    const f2v g_xy = dL_dG * dG_ddel;
    o.g_x = g_xy.x;
    o.g_y = g_xy.y;
    const f2v g_AC = -0.5f * gd * d * dL_dG;                // ((-0.5*gdx)*dx)*dL_dG , ((-0.5*gdy)*dy)*dL_dG
    o.g_A = g_AC.x;
    o.g_C = g_AC.y;
    o.g_B = -gd.x * d_swapped.x * dL_dG;                    // ((-gdx)*dy)*dL_dG
    o.g_op = G_eff * dL_dalpha;
}

}  // namespace s360
