// s360_backward.hip — backward kernels.  gfx950 / wave64 only.
//
//   k_order_units      launch order of the work units (tile, quadrant): tiles in descending order of their replay length,
//                      sibling quadrants on the same XCD; the same launch clears the validity flags of the partial-record slots
//   k_render_bwd_em    (s360_bwd_em.h / s360_backward_em.hip) one autonomous wave per (tile, 8x8 quadrant): entry-major
//                      back-to-front replay from final_T / n_contrib — 64 culled list entries in the lanes, the
//                      pixels looped, two DPP scans per pixel; ONE partial record per (instance, quadrant) is written —
//                      the instance slot is the splat's position in emission order (grouped by (view, Gaussian) pair)
//   k_gather_slots     1 thread per instance slot: sums the slot's quadrant partials, the pair's first slot then adds its
//                      consecutive slots in a fixed order
//   k_preprocess_bwd   1 thread per Gaussian: chains conic -> cov2D -> cov3D / mean, projection -> mean, sums the
//                      V views in registers; with per-view camera centres also SH -> dL/dSH (slab through LDS)
//   k_sh_bwd           streaming SH backward for views sharing one camera centre (and for the N gathered
//                      factors of the multi-GPU exchange): dL/dSH = Y (x) dRGB (the view-direction term of dL/dmean comes
//                      from the forward's sh_jac inside k_preprocess_bwd)
// No float atomics anywhere: gradients are bit-reproducible run to run.
#include "s360_device.h"
#include "s360_prof.h"
#include "s360_bwd_em.h"
#include "s360_adapter_math.h"

#include <cstdio>
#include <cstdlib>

namespace s360 {

constexpr int GREC = 4 * PREC_F4;  // floats per partial record slot: gx gy gA gB | gC gop gr gg | gb gz - - (| pad)

// Sum of the per-(instance, quadrant) partial gradients of every (view, Gaussian) pair, SLOT-parallel: thread i owns instance
// slot i — it adds that slot's (up to four) quadrant partials in quadrant order, parks the 10-float sum in LDS, and the thread
// that holds the FIRST slot of a pair (slot_pair[i-1] != slot_pair[i]) then adds the pair's consecutive slots in slot order
// and writes the pair's 48-byte raster-gradient record.  Every global load is independent of every other (the round-1 form —
// one thread per pair walking slot bases -> validity flags -> partial records — was a chain of dependent round trips at 6 %
// VALU utilisation: 87 us for 180 MB).  Fixed summation order: deterministic.  Slots of a pair that spill into the next
// workgroup (1 pair in ~100) are re-read from global memory by the owner.
__device__ __forceinline__ void slot_sum(const float4* __restrict__ part, const uint32_t* __restrict__ valid_words, uint32_t i,
                                         float* s) {
    const uint32_t vw4 = valid_words[i];  // byte q != 0: quadrant q wrote a partial
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if ((vw4 >> (8 * q)) & 0xFFu) {
            const float4* r = part + ((size_t)i * 4 + q) * PREC_F4;
            const float4 r0 = r[0], r1 = r[1], r2 = r[2];
            s[0] += r0.x; s[1] += r0.y; s[2] += r0.z; s[3] += r0.w;
            s[4] += r1.x; s[5] += r1.y; s[6] += r1.z; s[7] += r1.w;
            s[8] += r2.x; s[9] += r2.y;
        }
    }
}

// Pairs with more than 32 instance slots (footprints of more than 32 tiles: splats close to the camera; k_emit tags their
// slot_pair entries with bit 31, so phase 1 does not even read their partials) are NOT summed by their
// owner thread — a serial chain of up to a whole image's tiles, 375 us of this kernel on the uniform stress cloud whose nearest
// splats cover 256 tiles — but by a whole wave in a second phase of the same launch: lane l adds slots l, l + 64, ... of the
// pair (rectangle order), then a fixed shuffle tree.  k_emit lists those pairs (header[4] of them, any order: each pair's sum is
// self-contained, so the result is deterministic).  Rectangles beyond 32 tiles are binned whole in either list mode, so these
// sums are the same with lean and with upstream-compatible lists.
constexpr uint32_t LONG_PAIR_SLOTS = 32;
__global__ __launch_bounds__(S360_BLOCK) void k_gather_slots(uint32_t cap, const uint32_t* __restrict__ header,
                                                            const uint32_t* __restrict__ slot_pair, const float4* __restrict__ part,
                                                            const uint32_t* __restrict__ valid_words, float4* __restrict__ pairgrad,
                                                            const uint32_t* __restrict__ long_pairs, const uint2* __restrict__ slot_info,
                                                            const uint32_t* __restrict__ tiles_touched) {
    __shared__ float s_val[S360_BLOCK][11];  // 11: odd stride, conflict-free column access
    __shared__ uint32_t s_pair[S360_BLOCK];
    const uint32_t L = min(header[0], cap);
    const uint32_t i0 = blockIdx.x * S360_BLOCK;
    const int tid = threadIdx.x;
    if (i0 < L) {   // block-uniform
        const uint32_t i = i0 + tid;
        float s[10];
        uint32_t p = 0xFFFFFFFFu;
        if (i < L) {
            p = slot_pair[i];
            if (!(p >> 31)) {    // (bit 31: a slot of a pair with more than 32 of them — phase 2 reads it, not this thread)
                slot_sum(part, valid_words, i, s);
#pragma unroll
                for (int k = 0; k < 10; ++k) s_val[tid][k] = s[k];
            }
        }
        s_pair[tid] = p;
        __syncthreads();
        bool owner = i < L && !(p >> 31) && (tid == 0 ? (i == 0 || slot_pair[i - 1] != p) : s_pair[tid - 1] != p);
        if (owner) {
            for (uint32_t j = i + 1; j < L; ++j) {  // the pair's remaining slots, in slot order
                const uint32_t tj = j - i0;
                if ((tj < S360_BLOCK ? s_pair[tj] : slot_pair[j]) != p) break;
                if (j - i >= LONG_PAIR_SLOTS) {     // a 33rd slot: this pair is left to the wave-parallel phase below
                    owner = false;
                    break;
                }
                float t[10];
                if (tj < S360_BLOCK) {
#pragma unroll
                    for (int k = 0; k < 10; ++k) t[k] = s_val[tj][k];
                } else {
                    slot_sum(part, valid_words, j, t);
                }
#pragma unroll
                for (int k = 0; k < 10; ++k) s[k] += t[k];
            }
        }
        if (owner) {
            pairgrad[(size_t)p * 3] = make_float4(s[0], s[1], s[2], s[3]);
            pairgrad[(size_t)p * 3 + 1] = make_float4(s[4], s[5], s[6], s[7]);
            pairgrad[(size_t)p * 3 + 2] = make_float4(s[8], s[9], 0.f, 0.f);
        }
    }
    // ---- phase 2: the long pairs, one wave each
    const uint32_t n_long = min(header[4], cap / LONG_PAIR_SLOTS + 1u);
    const int lane = tid & 63, wave = tid >> 6;
    for (uint32_t k = blockIdx.x * (S360_BLOCK / 64) + wave; k < n_long; k += gridDim.x * (S360_BLOCK / 64)) {
        const uint32_t p = long_pairs[k];
        const uint32_t base = slot_info[p].x, n = tiles_touched[p];
        // 64 slots at a time: the loads (the latency) in parallel, one slot per lane; the ADDS stay the owner thread's serial
        // left fold in slot order, fed by v_readlane — bit-identical to the single-thread sum (a float32 shuffle tree, and even a
        // float64 one, moved one fuzz scene's means2D error from below 5e-4 to 5.9e-4 of the largest gradient: the centre sums of
        // a large elongated splat cancel across its tiles, and the tests' bar was set with the serial order)
        float s[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) s[q] = 0.f;
        for (uint32_t j0 = 0; j0 < n; j0 += 64) {
            float t[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) t[q] = 0.f;
            const uint32_t j = j0 + (uint32_t)lane;
            if (j < n && base + j < L) slot_sum(part, valid_words, base + j, t);
            const int m = (int)min(64u, n - j0);
            for (int l = 0; l < m; ++l) {
#pragma unroll
                for (int q = 0; q < 10; ++q) s[q] += rl(t[q], l);
            }
        }
        if (lane == 0) {
            pairgrad[(size_t)p * 3] = make_float4(s[0], s[1], s[2], s[3]);
            pairgrad[(size_t)p * 3 + 1] = make_float4(s[4], s[5], s[6], s[7]);
            pairgrad[(size_t)p * 3 + 2] = make_float4(s[8], s[9], 0.f, 0.f);
        }
    }
}

// SH_PASS = true : SH backward inside this kernel (slab through LDS; required when the views have different
//                   camera centres).  SH_PASS = false (shared camera centre): the kernel only exports the
//                   clamp-masked sum of dL/dRGB per Gaussian and k_sh_bwd streams the SH slabs afterwards —
//                   two lean kernels instead of one register- and LDS-bound one.
template <bool USE_SH, bool SH_PASS>
__global__ __launch_bounds__(S360_BLOCK) void k_preprocess_bwd(
    KParams kp, const S360View* __restrict__ views, const float* __restrict__ means, const float* __restrict__ cov6,
    const float* __restrict__ shs, const uint8_t* __restrict__ vis_mask,
    const uint8_t* __restrict__ clamped, const float4* __restrict__ pairgrad, float* __restrict__ d_means3D,
    float* __restrict__ d_means2D, float* __restrict__ d_cov6, float* __restrict__ d_opac, float* __restrict__ d_shs,
    float* __restrict__ d_colors, float4* __restrict__ drgb_out, int depth_mode, const float* __restrict__ sh_jac,
    int g_begin, int g_end, float* __restrict__ d_packed, int view_stamp) {
    // [g_begin, g_end): the Gaussians of this launch (the whole cloud, or one range of the chunked multi-GPU exchange — outputs
    // land at their absolute positions either way).  d_packed != NULL: means / covariance / opacity gradients go to ONE
    // [P,10] buffer (3 + the 6 unique covariance entries + 1), the unit the all-reduce moves, instead of three arrays.
    // view_stamp >= 0: the visibility word of drgb_out is that stamp (the owning rank of the exchange) instead of the first view.
    extern __shared__ __attribute__((aligned(16))) float lds_sh[];  // [256*M*3] SH slab, then [256*V*3] dRGB
    const int tid = threadIdx.x;
    const int g0 = g_begin + blockIdx.x * S360_BLOCK;
    const int g = g0 + tid;
    const int P = g_end;
    const int nb = min(S360_BLOCK, P - g0);
    const int nfl = nb * kp.M * 3;
    float* lds_drgb = lds_sh + S360_BLOCK * kp.M * 3;  // per-thread, per-view dRGB (non-shared campos)
    const bool want_sh = USE_SH && SH_PASS;  // runs even when d_shs == NULL: dL/dmean needs the view-direction term

    if (want_sh) {
        const float* src = shs + (size_t)g0 * kp.M * 3;
        if ((((uintptr_t)src) & 15) == 0) {
            const int n4 = nfl >> 2;
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(lds_sh);
            for (int i = tid; i < n4; i += S360_BLOCK) d4[i] = s4[i];
            for (int i = (n4 << 2) + tid; i < nfl; i += S360_BLOCK) lds_sh[i] = src[i];
        } else {
            for (int i = tid; i < nfl; i += S360_BLOCK) lds_sh[i] = src[i];
        }
        __syncthreads();
    }

    const bool shared_cam = (kp.flags & S360_FLAG_SHARED_CAMPOS) != 0;
    const int n_sh = (kp.deg + 1) * (kp.deg + 1);
    if (g < P) {
        const float mx0 = means[3 * g], my0 = means[3 * g + 1], mz0 = means[3 * g + 2];
        float c60[6];
        const bool cov9 = (kp.flags & S360_FLAG_COV9) != 0;
        // SH element (k, c) inside this Gaussian's slab for either layout
        const int sk = (kp.flags & S360_FLAG_SH_CHANNEL_MAJOR) ? 1 : 3;
        const int sc_ = (kp.flags & S360_FLAG_SH_CHANNEL_MAJOR) ? kp.M : 1;
        load_cov6(cov6, g, cov9, c60);
        float dm0 = 0.f, dm1 = 0.f, dm2 = 0.f, dop = 0.f;
        float dc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float drgb_sum[3] = {0.f, 0.f, 0.f};  // clamp-masked, summed over views
        float dcol_sum[3] = {0.f, 0.f, 0.f};  // colors_precomp gradient
        bool any_visible = false;
        int first_visible = -1;
        const uint32_t vis = vis_mask[g];  // bit v: visible in view v (one byte instead of V tiles_touched words)

        for (int v = 0; v < kp.V; ++v) {
            const size_t p = (size_t)v * kp.P + g;
            float gx_ = 0.f, gy_ = 0.f;
            float drgb_v[3] = {0.f, 0.f, 0.f};
            // gradients w.r.t. the scaled cloud of this view; folded back with scale / scale^2 below
            const float sc = views[v].scale, sc2 = sc * sc;
            const float mx = mx0 * sc, my = my0 * sc, mz = mz0 * sc;
            float c6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = c60[k] * sc2;
            float dmv0 = 0.f, dmv1 = 0.f, dmv2 = 0.f;
            float dcv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if ((vis >> v) & 1u) {
                any_visible = true;
                if (first_visible < 0) first_visible = v;
                const float4 r0 = pairgrad[p * 3], r1 = pairgrad[p * 3 + 1], r2 = pairgrad[p * 3 + 2];
                gx_ = r0.x; gy_ = r0.y;
                const float gA = r0.z, gB = r0.w, gC = r1.x, gop = r1.y, gr = r1.z, gg = r1.w, gb = r2.x;
                dop += gop;
                const S360View& vw = views[v];
                if (depth_mode >= 0) {
                    // fused depth channel: value = depth_value(z_u), z_u = camera z in UNSCALED units = R_row2 . mean + t_z / scale
                    // (render_depth_cuda, cuda_splatting.py:239-251), so dz_u / dmean (unscaled) = third row of the rotation
                    const float* Vm = vw.viewmatrix;
                    const float tzs = Vm[2] * mx + Vm[6] * my + Vm[10] * mz + Vm[14];
                    const float dzu = r2.y * depth_value_grad(tzs * (1.0f / sc), vw.near_plane, vw.far_plane, depth_mode);
                    dm0 += Vm[2] * dzu;
                    dm1 += Vm[6] * dzu;
                    dm2 += Vm[10] * dzu;
                }
                const float* V = vw.viewmatrix;
                if (kp.flags & S360_FLAG_SPHERICAL) {
                    // native equirectangular splat: chain through geo_sph (oracle backward_one_sph)
                    GeoS gs;
                    geo_sph(V, kp.W, kp.H, mx, my, mz, c6, gs);
                    const float a = gs.a, b = gs.b, c = gs.c;
                    const float det = a * c - b * b;
                    const float d2inv = 1.0f / (det * det + 0.0000001f);
                    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
                    if (d2inv != 0.f) {
                        dL_da = d2inv * (-c * c * gA + b * c * gB + (det - a * c) * gC);
                        dL_dc = d2inv * (-a * a * gC + a * b * gB + (det - a * c) * gA);
                        dL_db = d2inv * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
                        const float *M0 = gs.M0, *M1 = gs.M1;
                        dcv[0] += M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
                        dcv[3] += M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
                        dcv[5] += M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
                        dcv[1] += 2.f * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + 2.f * M1[0] * M1[1] * dL_dc;
                        dcv[2] += 2.f * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + 2.f * M1[0] * M1[2] * dL_dc;
                        dcv[4] += 2.f * M0[1] * M0[2] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + 2.f * M1[1] * M1[2] * dL_dc;
                    }
                    float dJ00 = 0.f, dJ02 = 0.f, dJ10 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float dM0 = 2.f * dL_da * gs.v0[j] + dL_db * gs.v1[j];
                        const float dM1 = 2.f * dL_dc * gs.v1[j] + dL_db * gs.v0[j];
                        dJ00 += dM0 * V[j * 4 + 0];
                        dJ02 += dM0 * V[j * 4 + 2];
                        dJ10 += dM1 * V[j * 4 + 0];
                        dJ11 += dM1 * V[j * 4 + 1];
                        dJ12 += dM1 * V[j * 4 + 2];
                    }
                    const float t0 = gs.t0, t1 = gs.t1, t2 = gs.t2, rc = gs.rc, r2 = gs.r2;
                    const float c0 = -(float)kp.W / 6.283185307179586f, c1 = -(float)kp.H / 3.141592653589793f;
                    const float A = 1.0f / (rc * rc), Bq = 1.0f / (r2 * rc);
                    float dt0 = -(c0 * A) * dJ02 - (c1 * t1 * Bq) * dJ10;
                    float dt1 = -(c1 * Bq) * (t0 * dJ10 + t2 * dJ12);
                    float dt2 = (c0 * A) * dJ00 - (c1 * t1 * Bq) * dJ12;
                    const float dA = c0 * (dJ00 * t2 - dJ02 * t0);
                    const float dB = -(c1 * t1) * (dJ10 * t0 + dJ12 * t2);
                    const float dC = c1 * dJ11;
                    const float drc = dA * (-2.0f / (rc * rc * rc)) + dB * (-1.0f / (r2 * rc * rc)) + dC / r2;
                    const float dr2 = dB * (-1.0f / (r2 * r2 * rc)) + dC * (-rc / (r2 * r2));
                    dt0 += 2.0f * t0 * dr2;
                    dt1 += 2.0f * t1 * dr2;
                    dt2 += 2.0f * t2 * dr2;
                    if (gs.clamped) {
                        dt0 += drc * (0.05f * t0 / gs.r);
                        dt1 += drc * (0.05f * t1 / gs.r);
                        dt2 += drc * (0.05f * t2 / gs.r);
                    } else {
                        dt0 += drc * (t0 / gs.rho);
                        dt2 += drc * (t2 / gs.rho);
                    }
                    const float iu = c0 / gs.rho2, iv = c1 / (r2 * gs.rho);   // centre: (u, v) with the TRUE rho; pixel units
                    dt0 += gx_ * (iu * t2) + gy_ * (-(iv * t0 * t1));
                    dt1 += gy_ * (c1 * gs.rho / r2);
                    dt2 += gx_ * (-(iu * t0)) + gy_ * (-(iv * t2 * t1));
                    dmv0 += V[0] * dt0 + V[1] * dt1 + V[2] * dt2;
                    dmv1 += V[4] * dt0 + V[5] * dt1 + V[6] * dt2;
                    dmv2 += V[8] * dt0 + V[9] * dt1 + V[10] * dt2;
                } else {
                    Geo ge;
                    geo_compute(V, vw.tanfovx, vw.tanfovy, kp.W, kp.H, mx, my, mz, c6, ge);
                    const float a = ge.a, b = ge.b, c = ge.c;
                    const float det = a * c - b * b;
                    const float d2inv = 1.0f / (det * det + 0.0000001f);
                    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
                    if (d2inv != 0.f) {
                        dL_da = d2inv * (-c * c * gA + b * c * gB + (det - a * c) * gC);
                        dL_dc = d2inv * (-a * a * gC + a * b * gB + (det - a * c) * gA);
                        dL_db = d2inv * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
                        const float *M0 = ge.M0, *M1 = ge.M1;
                        dcv[0] += M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
                        dcv[3] += M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
                        dcv[5] += M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
                        dcv[1] += 2.f * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + 2.f * M1[0] * M1[1] * dL_dc;
                        dcv[2] += 2.f * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + 2.f * M1[0] * M1[2] * dL_dc;
                        dcv[4] += 2.f * M0[1] * M0[2] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + 2.f * M1[1] * M1[2] * dL_dc;
                    }
                    float dM0[3], dM1[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        dM0[j] = 2.f * dL_da * ge.v0[j] + dL_db * ge.v1[j];
                        dM1[j] = 2.f * dL_dc * ge.v1[j] + dL_db * ge.v0[j];
                    }
                    float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        dJ00 += dM0[j] * V[j * 4 + 0];
                        dJ02 += dM0[j] * V[j * 4 + 2];
                        dJ11 += dM1[j] * V[j * 4 + 1];
                        dJ12 += dM1[j] * V[j * 4 + 2];
                    }
                    const float tz = 1.f / ge.tz, tz2 = tz * tz, tz3 = tz2 * tz;
                    const float dt0 = (ge.xin ? 1.f : 0.f) * (-ge.fx * tz2 * dJ02);
                    const float dt1 = (ge.yin ? 1.f : 0.f) * (-ge.fy * tz2 * dJ12);
                    const float dt2 = -ge.fx * tz2 * dJ00 - ge.fy * tz2 * dJ11 + (2.f * ge.fx * ge.txc) * tz3 * dJ02 +
                                      (2.f * ge.fy * ge.tyc) * tz3 * dJ12;
                    dmv0 += V[0] * dt0 + V[1] * dt1 + V[2] * dt2;
                    dmv1 += V[4] * dt0 + V[5] * dt1 + V[6] * dt2;
                    dmv2 += V[8] * dt0 + V[9] * dt1 + V[10] * dt2;
                    // projection chain (NDC-scaled screen-space gradient)
                    const float m2x = gx_ * (0.5f * (float)kp.W), m2y = gy_ * (0.5f * (float)kp.H);
                    gx_ = m2x;
                    gy_ = m2y;
                    const float* Pm = vw.projmatrix;
                    const float mhx = Pm[0] * mx + Pm[4] * my + Pm[8] * mz + Pm[12];
                    const float mhy = Pm[1] * mx + Pm[5] * my + Pm[9] * mz + Pm[13];
                    const float mhw = Pm[3] * mx + Pm[7] * my + Pm[11] * mz + Pm[15];
                    const float mw = 1.f / (mhw + 0.0000001f);
                    const float mul1 = mhx * mw * mw, mul2 = mhy * mw * mw;
                    dmv0 += (Pm[0] * mw - Pm[3] * mul1) * m2x + (Pm[1] * mw - Pm[3] * mul2) * m2y;
                    dmv1 += (Pm[4] * mw - Pm[7] * mul1) * m2x + (Pm[5] * mw - Pm[7] * mul2) * m2y;
                    dmv2 += (Pm[8] * mw - Pm[11] * mul1) * m2x + (Pm[9] * mw - Pm[11] * mul2) * m2y;
                }
                if (USE_SH) {
                    const uint32_t cb = clamped[p];
                    drgb_v[0] = (cb & 1u) ? 0.f : gr;
                    drgb_v[1] = (cb & 2u) ? 0.f : gg;
                    drgb_v[2] = (cb & 4u) ? 0.f : gb;
                    drgb_sum[0] += drgb_v[0];
                    drgb_sum[1] += drgb_v[1];
                    drgb_sum[2] += drgb_v[2];
                } else {
                    dcol_sum[0] += gr;
                    dcol_sum[1] += gg;
                    dcol_sum[2] += gb;
                }
            }
            if (d_means2D) {
                d_means2D[3 * p] = gx_;
                d_means2D[3 * p + 1] = gy_;
                d_means2D[3 * p + 2] = 0.f;
            }
            if (want_sh && !shared_cam) {
                // view-direction term per view (campos differs); dRGB kept for the dSH pass
                float* dr = lds_drgb + (tid * kp.V + v) * 3;
                dr[0] = drgb_v[0]; dr[1] = drgb_v[1]; dr[2] = drgb_v[2];
                if ((vis >> v) & 1u) {
                    const S360View& vw = views[v];
                    const float ddx = mx - vw.campos[0], ddy = my - vw.campos[1], ddz = mz - vw.campos[2];
                    const float inv = 1.f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                    const float x = ddx * inv, y = ddy * inv, z = ddz * inv;
                    float bx[25], by[25], bz[25];
                    sh_basis_grad(kp.deg, x, y, z, bx, by, bz);
                    const float* sh = lds_sh + tid * kp.M * 3;
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                    for (int k = 0; k < n_sh; ++k) {
                        const float s = sh[k * sk] * drgb_v[0] + sh[k * sk + sc_] * drgb_v[1] + sh[k * sk + 2 * sc_] * drgb_v[2];
                        q0 += bx[k] * s; q1 += by[k] * s; q2 += bz[k] * s;
                    }
                    const float dot = x * q0 + y * q1 + z * q2;
                    dmv0 += (q0 - x * dot) * inv;
                    dmv1 += (q1 - y * dot) * inv;
                    dmv2 += (q2 - z * dot) * inv;
                }
            }
            dm0 += sc * dmv0;
            dm1 += sc * dmv1;
            dm2 += sc * dmv2;
#pragma unroll
            for (int k = 0; k < 6; ++k) dc[k] += sc2 * dcv[k];
        }

        if (want_sh) {
            float* sh = lds_sh + tid * kp.M * 3;
            if (shared_cam) {
                if (any_visible) {
                    const S360View& vw = views[first_visible];
                    const float sc = vw.scale;
                    const float ddx = mx0 * sc - vw.campos[0], ddy = my0 * sc - vw.campos[1], ddz = mz0 * sc - vw.campos[2];
                    const float inv = 1.f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                    const float x = ddx * inv, y = ddy * inv, z = ddz * inv;
                    float Y[25], bx[25], by[25], bz[25];
                    sh_basis(kp.deg, x, y, z, Y);
                    sh_basis_grad(kp.deg, x, y, z, bx, by, bz);
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                    for (int k = 0; k < n_sh; ++k) {
                        const float s = sh[k * sk] * drgb_sum[0] + sh[k * sk + sc_] * drgb_sum[1] + sh[k * sk + 2 * sc_] * drgb_sum[2];
                        q0 += bx[k] * s; q1 += by[k] * s; q2 += bz[k] * s;
                        sh[k * sk] = Y[k] * drgb_sum[0];
                        sh[k * sk + sc_] = Y[k] * drgb_sum[1];
                        sh[k * sk + 2 * sc_] = Y[k] * drgb_sum[2];
                    }
                    for (int k = n_sh; k < kp.M; ++k) sh[k * sk] = sh[k * sk + sc_] = sh[k * sk + 2 * sc_] = 0.f;
                    const float dot = x * q0 + y * q1 + z * q2;
                    dm0 += sc * ((q0 - x * dot) * inv);
                    dm1 += sc * ((q1 - y * dot) * inv);
                    dm2 += sc * ((q2 - z * dot) * inv);
                } else {
                    for (int k = 0; k < kp.M * 3; ++k) sh[k] = 0.f;
                }
            } else {
                for (int k = 0; k < kp.M * 3; ++k) sh[k] = 0.f;
                for (int v = 0; v < kp.V; ++v) {
                    if (!((vis >> v) & 1u)) continue;
                    const S360View& vw = views[v];
                    const float sc = vw.scale;
                    const float ddx = mx0 * sc - vw.campos[0], ddy = my0 * sc - vw.campos[1], ddz = mz0 * sc - vw.campos[2];
                    const float inv = 1.f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                    float Y[25];
                    sh_basis(kp.deg, ddx * inv, ddy * inv, ddz * inv, Y);
                    const float* dr = lds_drgb + (tid * kp.V + v) * 3;
                    for (int k = 0; k < n_sh; ++k) {
                        sh[k * sk] += Y[k] * dr[0];
                        sh[k * sk + sc_] += Y[k] * dr[1];
                        sh[k * sk + 2 * sc_] += Y[k] * dr[2];
                    }
                }
            }
        }
        if (USE_SH && !SH_PASS && sh_jac && any_visible) {
            // view-direction term of dL/dmean from the forward's d(rgb)/d(mean) (k_sh_eval): no SH slab re-read
            const float* J = sh_jac + 9 * (size_t)g;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                dm0 += drgb_sum[ch] * J[3 * ch];
                dm1 += drgb_sum[ch] * J[3 * ch + 1];
                dm2 += drgb_sum[ch] * J[3 * ch + 2];
            }
        }
        if (USE_SH && !SH_PASS && drgb_out)
            drgb_out[g] = make_float4(drgb_sum[0], drgb_sum[1], drgb_sum[2],
                                      __int_as_float(view_stamp >= 0 && first_visible >= 0 ? view_stamp : first_visible));
        if (d_packed) {
            float* o = d_packed + 10 * (size_t)g;
            o[0] = dm0; o[1] = dm1; o[2] = dm2;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[3 + k] = dc[k];
            o[9] = dop;
        } else {
            d_means3D[3 * g] = dm0;
            d_means3D[3 * g + 1] = dm1;
            d_means3D[3 * g + 2] = dm2;
            if (cov9) {
                // adjoint of the upper-triangle gather: lower triangle receives no gradient
                float* o = d_cov6 + 9 * (size_t)g;
                o[0] = dc[0]; o[1] = dc[1]; o[2] = dc[2];
                o[3] = 0.f;   o[4] = dc[3]; o[5] = dc[4];
                o[6] = 0.f;   o[7] = 0.f;   o[8] = dc[5];
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) d_cov6[6 * (size_t)g + k] = dc[k];
            }
            d_opac[g] = dop;
        }
        if (!USE_SH && d_colors) {
            d_colors[3 * g] = dcol_sum[0];
            d_colors[3 * g + 1] = dcol_sum[1];
            d_colors[3 * g + 2] = dcol_sum[2];
        }
    }
    if (want_sh && d_shs) {
        __syncthreads();
        float* dst = d_shs + (size_t)g0 * kp.M * 3;
        if ((((uintptr_t)dst) & 15) == 0) {
            const int n4 = nfl >> 2;
            float4* o4 = reinterpret_cast<float4*>(dst);
            const float4* l4 = reinterpret_cast<const float4*>(lds_sh);
            for (int i = tid; i < n4; i += S360_BLOCK) o4[i] = l4[i];
            for (int i = (n4 << 2) + tid; i < nfl; i += S360_BLOCK) dst[i] = lds_sh[i];
        } else {
            for (int i = tid; i < nfl; i += S360_BLOCK) dst[i] = lds_sh[i];
        }
    }
}

// dL/dSH for views sharing one camera centre: a pure streaming WRITE kernel, ONE WAVE per workgroup so that the
// compute / store phases of the (up to 8) waves on a CU overlap freely.  Per Gaussian a lane reads the summed dL/dRGB
// (16 B) and its mean (12 B), evaluates the SH basis at the view direction and forms dL/dSH = Y_k * dRGB_c; the 64 output
// slabs (19.2 KB, contiguous in memory) go through LDS so the global stores are fully coalesced 16-byte writes
// (lane-strided stores of partial lines cost ~2x here).  The SH coefficients themselves are NOT read: the view-direction
// term of dL/dmean they used to be needed for comes from the forward's sh_jac (k_preprocess_bwd).
// n_groups (view, summed dL/dRGB) pairs per Gaussian: 1 for a local backward; N when the factors of the rank-1 products
// Y (x) dRGB of N ranks were all-gathered instead of all-reducing N full SH gradients (slab accumulated in LDS).
template <bool CH_MAJOR>
__global__ __launch_bounds__(64) void k_sh_bwd(KParams kp, const S360View* __restrict__ views, const float* __restrict__ means,
                                              const float4* __restrict__ drgb_in, int n_groups, float* __restrict__ d_shs) {
    extern __shared__ __attribute__((aligned(16))) float lds_o[];  // [64][M*3]
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const int g = g0 + lane;
    const int slab = kp.M * 3;
    float* mine = lds_o + lane * slab;
    const int n_sh = (kp.deg + 1) * (kp.deg + 1);
    const int sk = CH_MAJOR ? 1 : 3, sc_ = CH_MAJOR ? kp.M : 1;
    if (g < kp.P) {
        for (int k = 0; k < slab; ++k) mine[k] = 0.f;
        const float m0 = means[3 * g], m1 = means[3 * g + 1], m2 = means[3 * g + 2];
#pragma unroll 1
        for (int j = 0; j < n_groups; ++j) {
            const float4 dr = drgb_in[(size_t)j * kp.P + g];
            const int fv = __float_as_int(dr.w);
            if (fv < 0) continue;  // invisible in that group's views: no contribution
            const S360View& vw = views[fv];
            const float sc = vw.scale;
            const float ddx = m0 * sc - vw.campos[0], ddy = m1 * sc - vw.campos[1], ddz = m2 * sc - vw.campos[2];
            const float inv = 1.f / sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
            float Y[25];
            sh_basis(kp.deg, ddx * inv, ddy * inv, ddz * inv, Y);
#pragma unroll
            for (int k = 0; k < 25; ++k) {
                if (k < n_sh) {
                    mine[k * sk] += Y[k] * dr.x;
                    mine[k * sk + sc_] += Y[k] * dr.y;
                    mine[k * sk + 2 * sc_] += Y[k] * dr.z;
                }
            }
        }
    }
    __syncthreads();  // single wave: orders the LDS writes above before the cooperative read below
    const int nb = min(64, kp.P - g0);
    const int nfl = nb * slab;
    float* dst = d_shs + (size_t)g0 * slab;
    if ((((uintptr_t)dst) & 15) == 0) {
        const int n4 = nfl >> 2;
        // non-temporal stores: the 315 MB of dL/dSH are consumed by the optimiser / the exchange, never by this library — kept
        // out of L2 / Infinity Cache they no longer sit, dirty, in front of the next step's SH read (k_sh_eval3_jac measured
        // 98 us right behind this kernel's plain stores, 72 us on its own)
        typedef float f4v __attribute__((ext_vector_type(4)));
        for (int i = lane; i < n4; i += 64) __builtin_nontemporal_store(reinterpret_cast<const f4v*>(lds_o)[i], reinterpret_cast<f4v*>(dst) + i);
        for (int i = (n4 << 2) + lane; i < nfl; i += 64) dst[i] = lds_o[i];
    } else {
        for (int i = lane; i < nfl; i += 64) dst[i] = lds_o[i];
    }
}


// ------------------------------------------------------------------------------ backward down to the encoder's raw outputs
// Last kernel of s360_backward_raw: per Gaussian, from the rasteriser's own per-Gaussian gradients (dL/dcov6, the clamp-masked
// sum of dL/dRGB over the call's views, optionally dL/dmean) straight to dL/d(raw record) and dL/ddepth — the adapter tail's
// backward (k_adapter_bwd's expressions: scale map, quaternion normalisation, Sigma = (C R) diag(s^2) (C R)^T) with its dL/dharmonics
// input replaced by the rank-1 form it always has here,
//     dL/d raw_sh[c][k] = (mask . D_v^T Y(dir))[k] * dL/dRGB[c],
// so the [P,3,25] dL/dSH buffer (written by k_sh_bwd, re-read by the adapter's backward: 600 B/Gaussian) does not exist.
// Reads 28 B (raw geometry words kept by the forward) + 4 (depth) + 24 (dL/dcov6) + 16 (dL/dRGB) + 12 (mean) [+ 12 dL/dmean],
// writes the 328-byte gradient record through LDS with coalesced non-temporal stores + 4 (dL/ddepth).
struct RawBwd {
    const float* extrinsics;
    const float* depths;
    const float* geo7;
    const float* sh_rot;
    const float* means;        // [P,3] (forward output: the view direction)
    const float* d_means;      // [P,3] or null (means detached, the reference's behaviour)
    const float* d_cov6;       // [P,6]
    const float4* d_rgb;       // [n_groups][P] (.w = index of the group's camera record, int32 bits, or -1)
    float* d_depths;           // [P]
    float* d_raw;              // [P,82]
    int Gv, H, W, per_ray, conv, n_groups;
    float smin, smax, eps;
};
constexpr int RAWB_C = 82;

// Round 6: workgroups dealt per context view (the view's Wigner-D matrix is wave-uniform: scalar loads, SGPR operands — round 5 read it
// from LDS, one ds_read per multiply-add, on all three waves redundantly); the rotated masked basis is computed ONCE per Gaussian, its
// 165 multiply-adds shared by waves 1 and 2 (degree 4 | degrees 0..3) through LDS, while wave 0 runs the geometry chain.
template <bool ROT>
__global__ __launch_bounds__(192) void k_raw_bwd(KParams kp, const S360View* __restrict__ views, RawBwd rb) {
    __shared__ __attribute__((aligned(16))) float s_out[64 * RAWB_C];
    __shared__ float s_yp[25 * 64];      // (mask . D^T Y)[k] of lane l at [k * 64 + l]
    __shared__ float4 s_w[64];           // the group's dL/dRGB (0 where the Gaussian is invisible in the group's views)
    const int tid = threadIdx.x, d = tid >> 6, l = tid & 63;
    const int v = blockIdx.y, gi0 = blockIdx.x * 64;
    const int nb = min(64, rb.Gv - gi0), g0 = v * rb.Gv + gi0, g = g0 + l;
    const bool live = l < nb;
    const float* D = ROT ? rb.sh_rot + (size_t)v * 625 : nullptr;   // wave-uniform
    if (d == 0 && live) {
        // ---- geometry: k_adapter_bwd's chain (scale map, quaternion, covariance), 6-entry covariance gradient — wave 0, while waves 1
        // and 2 rotate the basis
        const int gi = gi0 + l;
        const float* E = rb.extrinsics + 16 * v;
        const float* rw = rb.geo7 + 7 * (size_t)g;
        const float depth = rb.depths[g];
        const float px = 1.0f / (float)max(rb.W, rb.H);
        float sig[3], base[3], sc3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sig[k] = sigmoidf(rw[k]);
            base[k] = rb.smin + (rb.smax - rb.smin) * sig[k];
            sc3[k] = (base[k] * depth) * px;
        }
        QuatGeom qg;
        const float qr[4] = {rw[3], rw[4], rw[5], rw[6]};
        quat_geom(qr, rb.eps, qg);
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a][b] = E[4 * a] * qg.R[0][b] + E[4 * a + 1] * qg.R[1][b] + E[4 * a + 2] * qg.R[2][b];
        float G[3][3];
        {
            const float* gcv = rb.d_cov6 + 6 * (size_t)g;
            G[0][0] = gcv[0]; G[0][1] = gcv[1]; G[0][2] = gcv[2]; G[1][1] = gcv[3]; G[1][2] = gcv[4]; G[2][2] = gcv[5];
            G[1][0] = G[2][0] = G[2][1] = 0.f;
        }
        float Gs[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Gs[a][b] = G[a][b] + G[b][a];
        float dM[3][3], ds[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) t[a] = Gs[a][0] * M[0][k] + Gs[a][1] * M[1][k] + Gs[a][2] * M[2][k];
            const float s2 = sc3[k] * sc3[k];
#pragma unroll
            for (int a = 0; a < 3; ++a) dM[a][k] = t[a] * s2;
            ds[k] = 2.0f * sc3[k] * (0.5f * (M[0][k] * t[0] + M[1][k] * t[1] + M[2][k] * t[2]));
        }
        float dR[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) dR[a][b] = E[a] * dM[0][b] + E[4 + a] * dM[1][b] + E[8 + a] * dM[2][b];
        const float qi = qg.q[0], qj = qg.q[1], qk = qg.q[2], qr_ = qg.q[3], a = qg.a;
        const float B[3][3] = {{-(qj * qj + qk * qk), qi * qj - qk * qr_, qi * qk + qj * qr_},
                               {qi * qj + qk * qr_, -(qi * qi + qk * qk), qj * qk - qi * qr_},
                               {qi * qk - qj * qr_, qj * qk + qi * qr_, -(qi * qi + qj * qj)}};
        float dLda = 0.f;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) dLda += dR[x][y] * B[x][y];
        float dq[4];
        dq[0] = a * (dR[0][1] * qj + dR[0][2] * qk + dR[1][0] * qj - 2.0f * qi * dR[1][1] - dR[1][2] * qr_ + dR[2][0] * qk + dR[2][1] * qr_ - 2.0f * qi * dR[2][2]);
        dq[1] = a * (-2.0f * qj * dR[0][0] + dR[0][1] * qi + dR[0][2] * qr_ + dR[1][0] * qi + dR[1][2] * qk - dR[2][0] * qr_ + dR[2][1] * qk - 2.0f * qj * dR[2][2]);
        dq[2] = a * (-2.0f * qk * dR[0][0] - dR[0][1] * qr_ + dR[0][2] * qi + dR[1][0] * qr_ - 2.0f * qk * dR[1][1] + dR[1][2] * qj + dR[2][0] * qi + dR[2][1] * qj);
        dq[3] = a * (-dR[0][1] * qk + dR[0][2] * qj + dR[1][0] * qk - dR[1][2] * qi - dR[2][0] * qj + dR[2][1] * qi);
        const float da = -a * a * dLda;
#pragma unroll
        for (int k = 0; k < 4; ++k) dq[k] += da * qg.q[k];
        const float dot = qr[0] * dq[0] + qr[1] * dq[1] + qr[2] * dq[2] + qr[3] * dq[3];
        const float f = qg.n > 0.f ? dot / (qg.n * qg.m * qg.m) : 0.f;
        float* orec = s_out + l * RAWB_C;
#pragma unroll
        for (int k = 0; k < 4; ++k) orec[3 + k] = dq[k] / qg.m - qr[k] * f;
        float dd = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            orec[k] = ds[k] * depth * px * (rb.smax - rb.smin) * sig[k] * (1.0f - sig[k]);
            dd += ds[k] * base[k] * px;
        }
        if (rb.d_means) {   // opt-in: the reference's means are detached (sphere_projection.py:14-86)
            float dir[3];
            erp_dir(gi / rb.per_ray, rb.H, rb.W, rb.conv, dir);
            const float* gm = rb.d_means + 3 * (size_t)g;
#pragma unroll
            for (int x = 0; x < 3; ++x) dd += (E[x] * gm[0] + E[4 + x] * gm[1] + E[8 + x] * gm[2]) * dir[x];
        }
        rb.d_depths[g] = dd;
    }
    {
        // ---- colour channel d: sum over the camera groups of (mask . D^T Y(dir_j)) * dL/dRGB_j[d]
        float m0 = 0.f, m1 = 0.f, m2 = 1.f;
        if (live && d >= 1) { m0 = rb.means[3 * (size_t)g]; m1 = rb.means[3 * (size_t)g + 1]; m2 = rb.means[3 * (size_t)g + 2]; }
        float acc[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) acc[k] = 0.f;
#pragma unroll 1
        for (int j = 0; j < rb.n_groups; ++j) {
            if (j > 0) __syncthreads();   // the previous group's s_yp / s_w were read
            if (d >= 1) {
                float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
                int fv = 0;
                if (live) {
                    dr = rb.d_rgb[(size_t)j * kp.P + g];
                    fv = __float_as_int(dr.w);
                    if (fv < 0) { fv = 0; dr = make_float4(0.f, 0.f, 0.f, 0.f); }   // invisible in that group's views: contributes 0
                }
                const S360View& vw = views[fv];
                const float sc = vw.scale;
                const float dx = m0 * sc - vw.campos[0], dy = m1 * sc - vw.campos[1], dz = m2 * sc - vw.campos[2];
                const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
                float Y[25], Yp[25];
                sh_basis(4, dx * inv, dy * inv, dz * inv, Y);
                if (d == 1) {
                    sh_rotate_basis25<ROT, 1>(D, Y, Yp);
#pragma unroll
                    for (int k = 16; k < 25; ++k) s_yp[k * 64 + l] = Yp[k];
                    s_w[l] = dr;
                } else {
                    sh_rotate_basis25<ROT, 2>(D, Y, Yp);
#pragma unroll
                    for (int k = 0; k < 16; ++k) s_yp[k * 64 + l] = Yp[k];
                }
            }
            __syncthreads();
            const float4 dr = s_w[l];
            const float w = d == 0 ? dr.x : (d == 1 ? dr.y : dr.z);
#pragma unroll
            for (int k = 0; k < 25; ++k) acc[k] = __builtin_fmaf(s_yp[k * 64 + l], w, acc[k]);
        }
        if (live) {
            float* orec = s_out + l * RAWB_C + 7 + 25 * d;
#pragma unroll
            for (int k = 0; k < 25; ++k) orec[k] = acc[k];
        }
    }
    __syncthreads();
    const int nfl = nb * RAWB_C;
    float* dst = rb.d_raw + (size_t)g0 * RAWB_C;
    if ((((uintptr_t)dst) & 15) == 0) {
        const int n4 = nfl >> 2;
        typedef float f4v __attribute__((ext_vector_type(4)));
        for (int i = tid; i < n4; i += 192) __builtin_nontemporal_store(reinterpret_cast<const f4v*>(s_out)[i], reinterpret_cast<f4v*>(dst) + i);
        for (int i = (n4 << 2) + tid; i < nfl; i += 192) dst[i] = s_out[i];
    } else {
        for (int i = tid; i < nfl; i += 192) dst[i] = s_out[i];
    }
}

}  // namespace s360

using namespace s360;

#ifdef S360_DEBUG_LAUNCH  /* compile-time diagnostic (-DS360_DEBUG_LAUNCH): no environment reads in the host path */
#define S360_LAUNCH_DIAG(e) fprintf(stderr, "s360: %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__)
#else
#define S360_LAUNCH_DIAG(e) ((void)0)
#endif
#define S360_CHECK_LAUNCH()                                                                          \
    do {                                                                                             \
        hipError_t e_ = hipGetLastError();                                                           \
        if (e_ != hipSuccess) {                                                                      \
            S360_LAUNCH_DIAG(e_);                                                                    \
            return S360_E_LAUNCH;                                                                    \
        }                                                                                            \
    } while (0)

static int launch_sh_bwd(const KParams& kp, const S360View* views, const float* means3D, const float4* drgb, int n_groups,
                         float* d_shs, hipStream_t st) {
    const int wblk = (kp.P + 63) / 64;
    const size_t wlds = (size_t)64 * kp.M * 3 * 4;
    if (wlds > 64 * 1024) return S360_E_UNSUPPORTED;
    ProfScope ps(PS_SH_BWD, st);
    if (kp.flags & S360_FLAG_SH_CHANNEL_MAJOR)
        hipLaunchKernelGGL((k_sh_bwd<true>), dim3(wblk), dim3(64), wlds, st, kp, views, means3D, drgb, n_groups, d_shs);
    else
        hipLaunchKernelGGL((k_sh_bwd<false>), dim3(wblk), dim3(64), wlds, st, kp, views, means3D, drgb, n_groups, d_shs);
    return S360_OK;
}

struct BwdCtx {
    KParams kp;
    S360Layout L;
    int nt;
    float4* part;
    uint32_t* valid_words;
    uint32_t* order;
    float4* pairgrad;
    uint32_t* seg_list;   // S360_FLAG_SPLIT_LISTS: [0] = count, then the segment units that hold survivor records
    size_t n_slots;       // segment slots of the forward workspace
};

static int bwd_ctx(const S360Params* prm, const void* workspace, size_t workspace_bytes, void* bwd_workspace,
                   size_t bwd_workspace_bytes, BwdCtx& c) {
    if (!prm || !workspace || !bwd_workspace) return S360_E_BADARG;
    if (prm->flags & S360_FLAG_FORWARD_ONLY) return S360_E_BADARG;
    int rc = s360_layout(prm, &c.L);
    if (rc) return rc;
    if (workspace_bytes < c.L.total_bytes || bwd_workspace_bytes < c.L.backward_bytes) return S360_E_WORKSPACE;
    KParams& kp = c.kp;
    kp.P = prm->P; kp.V = prm->V; kp.H = prm->H; kp.W = prm->W; kp.deg = s360_effective_degree(prm); kp.M = prm->M;
    kp.gx = (prm->W + 15) / 16; kp.gy = (prm->H + 15) / 16; kp.T = kp.gx * kp.gy;
    kp.flags = prm->flags; kp.cap = prm->max_instances;
    const bool sph = (kp.flags & S360_FLAG_SPHERICAL) != 0;
    if (sph && (kp.V & 1)) return S360_E_BADARG;
    c.nt = (sph ? kp.V / 2 : kp.V) * kp.T;
    // backward scratch: [cap] x 4 quadrant partial records of 48 B, [cap] x 4 validity bytes, the launch order, one gathered
    // 48-byte record per pair, [P] summed dL/dRGB
    if (kp.flags & S360_FLAG_ATOMIC_GRADS) {   // no partial slots, no validity flags: order table, then the pair records
        c.part = nullptr;
        c.valid_words = nullptr;
        c.order = (uint32_t*)bwd_workspace;
    } else {
        c.part = (float4*)bwd_workspace;
        c.valid_words = (uint32_t*)((char*)bwd_workspace + (size_t)kp.cap * 4 * GREC * 4);
        c.order = c.valid_words + kp.cap;
    }
    c.pairgrad = (float4*)((char*)(c.order + c.nt * 4) + 256 - ((uintptr_t)(c.order + c.nt * 4) & 255));
    c.seg_list = nullptr;
    c.n_slots = seg_slots_of(prm);
    if (kp.flags & S360_FLAG_SPLIT_LISTS) {   // behind the pair records and the [P] summed dL/dRGB
        char* e = (char*)(c.pairgrad + (size_t)kp.V * (kp.P > 0 ? kp.P : 1) * 3 + (size_t)(kp.P > 0 ? kp.P : 1));
        c.seg_list = (uint32_t*)(e + 256 - ((uintptr_t)e & 255)) + 32;   // 32 words in front of it hold the SegBwd record
    }
    return S360_OK;
}

// Composite part of the backward: every (tile, quadrant) replay + the per-pair sum of the partial records.  Leaves one
// 48-byte raster-gradient record per (view, Gaussian) pair in the backward workspace.
static int backward_composite(const BwdCtx& c, const S360View* views, const void* workspace, const float* dL_dimages,
                              const float* dL_dimages_scale, const float* dL_ddepth, int depth_mode, hipStream_t st) {
    const KParams& kp = c.kp;
    const S360Layout& L = c.L;
    const char* ws = (const char*)workspace;
    const uint32_t* header = (const uint32_t*)(ws + L.header);
    const bool with_depth = dL_ddepth != nullptr;
    const uint32_t* surv_count = (const uint32_t*)(ws + L.surv_count);  // per-unit replay length: also the work estimate
    SegBwd sb{};
    if (c.seg_list) {
        sb.seg_flag = (const uint32_t*)(ws + L.seg_flag);
        sb.chunk_start = (const uint32_t*)(ws + L.chunk_start);
        sb.seg_c = (const float4*)(ws + L.seg_c);
        sb.seg_t = (const float*)(ws + L.seg_t);
        sb.seg_cnt = (const uint32_t*)(ws + L.seg_cnt);
        sb.seg_info = (const uint2*)(ws + L.seg_info);
        sb.seg_list = c.seg_list;
        sb.n_seg_blocks = (uint32_t)min((size_t)S360_SEG_BWD_BLOCKS, c.n_slots * 4);
        sb.dbg_base = (uint32_t)c.nt * 4u;
    }
    {
        ProfScope ps(PS_ORDER, st);
        hipLaunchKernelGGL(k_order_units, dim3(1 + 512), dim3(1024), 0, st, surv_count, c.order, c.nt * 4, c.valid_words, header, kp.cap,
                           (kp.flags & S360_FLAG_ATOMIC_GRADS) ? c.pairgrad : (float4*)nullptr, (const uint8_t*)(ws + L.vis_mask), kp.P, kp.V,
                           (const uint32_t*)(ws + L.seg_cnt), (const uint32_t*)(ws + L.chunk_start), c.seg_list, (const uint2*)(ws + L.seg_info), sb);
    }
    {
        ProfScope ps(PS_RENDER_BWD, st);
        launch_render_bwd_em(with_depth, c.nt * 4, st, kp, views, (const uint32_t*)(ws + L.tile_start), (const float4*)(ws + L.surv),
                             surv_count, (const uint2*)(ws + L.slot_base), (const float*)(ws + L.depths),
                             (const float*)(ws + L.final_T), (const uint32_t*)(ws + L.n_contrib), dL_dimages, dL_dimages_scale,
                             dL_ddepth, c.part, (uint8_t*)c.valid_words, c.order, depth_mode,
                             (kp.flags & S360_FLAG_ATOMIC_GRADS) ? (float*)c.pairgrad : (float*)nullptr,
#ifdef S360_DBG_TIMING
                             (uint32_t*)(ws + L.keys_alt), c.seg_list ? &sb : (const SegBwd*)nullptr, sb.n_seg_blocks);  // the forward's merge buffer is free by now
#else
                             (uint32_t*)nullptr, c.seg_list ? &sb : (const SegBwd*)nullptr, sb.n_seg_blocks);
#endif
    }
    S360_CHECK_LAUNCH();
    if (kp.flags & S360_FLAG_ATOMIC_GRADS) return S360_OK;   // the pair records are complete: nothing to gather
    ProfScope ps(PS_GATHER, st);
    hipLaunchKernelGGL(k_gather_slots, dim3((unsigned)(((size_t)kp.cap + S360_BLOCK - 1) / S360_BLOCK)), dim3(S360_BLOCK), 0, st,
                       kp.cap, header, (const uint32_t*)(ws + L.slot_pair), c.part, c.valid_words, c.pairgrad,
                       (const uint32_t*)(ws + L.long_pairs), (const uint2*)(ws + L.slot_base), (const uint32_t*)(ws + L.tiles_touched));
    S360_CHECK_LAUNCH();
    return S360_OK;
}

// Per-Gaussian part for Gaussians [g_begin, g_end): geometry chains of the V views (+ the SH pass unless the views share a
// camera centre, where k_sh_bwd follows or the caller defers it).
static int backward_gaussians(const BwdCtx& c, const S360View* views, const float* means3D, const float* cov6, const float* shs,
                              const void* workspace, bool with_depth, int depth_mode, int g_begin, int g_end, float* d_means3D,
                              float* d_means2D, float* d_cov6, float* d_opacities, float* d_shs, float* d_colors, float* d_rgb_sum,
                              float* d_packed, int view_stamp, hipStream_t st) {
    const KParams& kp = c.kp;
    const S360Layout& L = c.L;
    const char* ws = (const char*)workspace;
    const uint8_t* vis_mask = (const uint8_t*)(ws + L.vis_mask);
    const uint8_t* clamped = (const uint8_t*)(ws + L.clamped);
    const int dmode = with_depth ? depth_mode : -1;
    const int n = g_end - g_begin;
    if (n <= 0) return S360_OK;
    const int nblk = (n + S360_BLOCK - 1) / S360_BLOCK;
    float4* drgb = d_rgb_sum ? (float4*)d_rgb_sum : c.pairgrad + (size_t)kp.V * kp.P * 3;  // [P] summed dL/dRGB (+ first visible view)
    if (shs) {
        const bool shared = (kp.flags & S360_FLAG_SHARED_CAMPOS) != 0;
        if ((d_rgb_sum || d_packed) && !shared) return S360_E_UNSUPPORTED;  // the split form needs one camera centre per call
        if (shared) {
            // dRGB/d(view direction) reaches dL/dmean here (from the forward's sh_jac), whether or not dL/dSH is wanted
            // (harmonics frozen: d_shs == NULL), as upstream does
            {
                ProfScope ps(PS_PREPROCESS_BWD, st);
                hipLaunchKernelGGL((k_preprocess_bwd<true, false>), dim3(nblk), dim3(S360_BLOCK), 0, st, kp, views, means3D, cov6, shs,
                                   vis_mask, clamped, c.pairgrad, d_means3D, d_means2D, d_cov6, d_opacities, d_shs,
                                   d_colors, drgb, dmode, (const float*)(ws + L.sh_jac), g_begin, g_end, d_packed, view_stamp);
            }
            if (!d_rgb_sum && d_shs) {
                if (g_begin != 0 || g_end != kp.P) return S360_E_UNSUPPORTED;
                const int rc2 = launch_sh_bwd(kp, views, means3D, drgb, 1, d_shs, st);
                if (rc2) return rc2;
            }
        } else {
            if (g_begin != 0 || g_end != kp.P) return S360_E_UNSUPPORTED;
            size_t lds = (size_t)S360_BLOCK * kp.M * 3 * 4 + (size_t)S360_BLOCK * kp.V * 3 * 4;
            if (lds > 160 * 1024) return S360_E_UNSUPPORTED;
            {
                static bool attr_done[64] = {};
                int dev = 0;
                if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_done[dev]) {
                    (void)hipFuncSetAttribute((const void*)k_preprocess_bwd<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipGetLastError();
                    attr_done[dev] = true;
                }
            }
            ProfScope ps(PS_PREPROCESS_BWD, st);
            hipLaunchKernelGGL((k_preprocess_bwd<true, true>), dim3(nblk), dim3(S360_BLOCK), lds, st, kp, views, means3D, cov6, shs,
                               vis_mask, clamped, c.pairgrad, d_means3D, d_means2D, d_cov6, d_opacities, d_shs,
                               d_colors, (float4*)nullptr, dmode, (const float*)nullptr, 0, kp.P, (float*)nullptr, -1);
        }
    } else {
        if (d_packed) return S360_E_UNSUPPORTED;
        ProfScope ps(PS_PREPROCESS_BWD, st);
        hipLaunchKernelGGL((k_preprocess_bwd<false, false>), dim3(nblk), dim3(S360_BLOCK), 0, st, kp, views, means3D, cov6, shs,
                           vis_mask, clamped, c.pairgrad, d_means3D, d_means2D, d_cov6, d_opacities, d_shs,
                           d_colors, (float4*)nullptr, dmode, (const float*)nullptr, g_begin, g_end, (float*)nullptr, -1);
    }
    S360_CHECK_LAUNCH();
    return S360_OK;
}

static int backward_impl(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                         const float* opacities, const float* shs, const float* colors_precomp,
                         const void* workspace, size_t workspace_bytes, const float* dL_dimages,
                         const float* dL_dimages_scale, const float* dL_ddepth, int depth_mode, float* d_means3D,
                         float* d_means2D, float* d_cov6, float* d_opacities, float* d_shs, float* d_colors,
                         float* d_rgb_sum, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream_) {
    (void)opacities;
    if (!prm || !views || !workspace || !dL_dimages || !bwd_workspace) return S360_E_BADARG;
    if (prm->flags & S360_FLAG_FORWARD_ONLY) return S360_E_BADARG;
    if (prm->P > 0 && (shs == nullptr) == (colors_precomp == nullptr)) return S360_E_BADARG;
    if (prm->P > 0 && (!means3D || !cov6 || !d_means3D || !d_cov6 || !d_opacities)) return S360_E_BADARG;
    const bool with_depth = dL_ddepth != nullptr;
    if (with_depth && (depth_mode < 0 || depth_mode > 3)) return S360_E_BADARG;
    BwdCtx c;
    int rc = bwd_ctx(prm, workspace, workspace_bytes, bwd_workspace, bwd_workspace_bytes, c);
    if (rc) return rc;
    if (prm->P == 0) return S360_OK;
    hipStream_t st = (hipStream_t)stream_;
    rc = backward_composite(c, views, workspace, dL_dimages, dL_dimages_scale, dL_ddepth, depth_mode, st);
    if (rc) return rc;
    return backward_gaussians(c, views, means3D, cov6, shs, workspace, with_depth, depth_mode, 0, prm->P, d_means3D, d_means2D,
                              d_cov6, d_opacities, d_shs, d_colors, d_rgb_sum, nullptr, -1, st);
}

extern "C" int s360_backward(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                             const float* opacities, const float* shs, const float* colors_precomp,
                             const void* workspace, size_t workspace_bytes, const float* dL_dimages,
                             const float* dL_dimages_scale, const float* dL_ddepth, int32_t depth_mode, float* d_means3D,
                             float* d_means2D, float* d_cov6, float* d_opacities, float* d_shs, float* d_colors,
                             void* bwd_workspace, size_t bwd_workspace_bytes, void* stream_) {
    return backward_impl(prm, views, means3D, cov6, opacities, shs, colors_precomp, workspace, workspace_bytes,
                         dL_dimages, dL_dimages_scale, dL_ddepth, depth_mode, d_means3D, d_means2D, d_cov6, d_opacities, d_shs,
                         d_colors, nullptr, bwd_workspace, bwd_workspace_bytes, stream_);
}

extern "C" int s360_backward_split(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                                   const float* opacities, const float* shs, const void* workspace, size_t workspace_bytes,
                                   const float* dL_dimages, const float* dL_dimages_scale,
                                   const float* dL_ddepth, int32_t depth_mode, float* d_means3D, float* d_means2D,
                                   float* d_cov6, float* d_opacities, float* d_rgb_sum, void* bwd_workspace,
                                   size_t bwd_workspace_bytes, void* stream_) {
    if (!shs || !d_rgb_sum) return S360_E_BADARG;
    return backward_impl(prm, views, means3D, cov6, opacities, shs, nullptr, workspace, workspace_bytes, dL_dimages,
                         dL_dimages_scale, dL_ddepth, depth_mode, d_means3D, d_means2D, d_cov6, d_opacities, nullptr, nullptr,
                         d_rgb_sum, bwd_workspace, bwd_workspace_bytes, stream_);
}

extern "C" int s360_sh_backward(const S360Params* prm, int32_t n_groups, const S360View* views, const float* means3D,
                                const float* d_rgb_sums, float* d_shs, void* stream_) {
    if (!prm || !views || !means3D || !d_rgb_sums || !d_shs || n_groups < 1) return S360_E_BADARG;
    if (prm->M < 1 || prm->sh_degree < 0 || prm->sh_degree > 4 || (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M)
        return S360_E_BADARG;
    if (prm->P == 0) return S360_OK;
    KParams kp;
    kp.P = prm->P; kp.V = prm->V; kp.H = prm->H; kp.W = prm->W; kp.deg = s360_effective_degree(prm); kp.M = prm->M;
    kp.gx = kp.gy = kp.T = 0;
    kp.flags = prm->flags; kp.cap = prm->max_instances;
    const int rc = launch_sh_bwd(kp, views, means3D, (const float4*)d_rgb_sums, n_groups, d_shs, (hipStream_t)stream_);
    if (rc) return rc;
    S360_CHECK_LAUNCH();
    return S360_OK;
}

extern "C" int s360_backward_composite(const S360Params* prm, const S360View* views, const void* workspace, size_t workspace_bytes,
                                       const float* dL_dimages, const float* dL_dimages_scale, const float* dL_ddepth,
                                       int32_t depth_mode, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream_) {
    if (!views || !dL_dimages) return S360_E_BADARG;
    if (dL_ddepth && (depth_mode < 0 || depth_mode > 3)) return S360_E_BADARG;
    BwdCtx c;
    const int rc = bwd_ctx(prm, workspace, workspace_bytes, bwd_workspace, bwd_workspace_bytes, c);
    if (rc) return rc;
    if (prm->P == 0) return S360_OK;
    return backward_composite(c, views, workspace, dL_dimages, dL_dimages_scale, dL_ddepth, depth_mode, (hipStream_t)stream_);
}

extern "C" int s360_backward_gaussians(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                                       const float* shs, const void* workspace, size_t workspace_bytes, int32_t with_depth,
                                       int32_t depth_mode, int32_t g_begin, int32_t g_count, int32_t rank_stamp,
                                       float* d_packed, float* d_means2D, float* d_rgb_sum, void* bwd_workspace,
                                       size_t bwd_workspace_bytes, void* stream_) {
    if (!views || !means3D || !cov6 || !shs || !d_packed || !d_rgb_sum) return S360_E_BADARG;
    if (with_depth && (depth_mode < 0 || depth_mode > 3)) return S360_E_BADARG;
    BwdCtx c;
    const int rc = bwd_ctx(prm, workspace, workspace_bytes, bwd_workspace, bwd_workspace_bytes, c);
    if (rc) return rc;
    if (g_begin < 0 || g_count < 0 || (long long)g_begin + g_count > prm->P) return S360_E_BADARG;
    if (g_count == 0) return S360_OK;
    return backward_gaussians(c, views, means3D, cov6, shs, workspace, with_depth != 0, depth_mode, g_begin, g_begin + g_count,
                              nullptr, d_means2D, nullptr, nullptr, nullptr, nullptr, d_rgb_sum, d_packed, rank_stamp,
                              (hipStream_t)stream_);
}

namespace s360 {
// Measurement aid (no reference counterpart): how many (pixel, list entry) pairs actually CONTRIBUTE to the rendered images of a
// training workspace — alpha >= 1/255, in front of the pixel's last contributor: what the composites' arithmetic is for.  One
// wave per (tile, quadrant) replays the forward's survivor records with the forward's own accept test.  out[0] += contributing
// pairs, out[1] += pairs the backward composite evaluates (survivor records in front of the quadrant's last contributor x 64
// pixels), out[2] += survivor records (all).  bench.py turns these into the work-based VALU figure next to the issue-rate one.
__global__ __launch_bounds__(64) void k_count_pairs(KParams kp, const uint32_t* __restrict__ tile_start, const float4* __restrict__ surv,
                                                   const uint32_t* __restrict__ surv_count, const uint32_t* __restrict__ n_contrib,
                                                   unsigned long long* __restrict__ out, const uint32_t* __restrict__ seg_flag,
                                                   const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ seg_cnt) {
    __shared__ uint32_t s_last[64];
    const uint32_t unit = blockIdx.x;
    const int t = (int)(unit >> 2), wave = (int)(unit & 3u), lane = threadIdx.x;
    const int v = t / kp.T, rem = t - v * kp.T;
    const int ty = rem / kp.gx, tx = rem - ty * kp.gx;
    const int qx = tx * 16 + sub_ox(wave), qy = ty * 16 + sub_oy(wave);
    const uint32_t start = min(tile_start[t], kp.cap), end = min(tile_start[t + 1], kp.cap);
    {
        const int px = qx + (lane & 7), py = qy + (lane >> 3);
        s_last[lane] = (px < kp.W && py < kp.H) ? n_contrib[((size_t)v * kp.H + py) * kp.W + px] : 0u;
    }
    __syncthreads();
    uint32_t cnt = 0;
    unsigned long long evaluated = 0ull;
    // a split quadrant (S360_FLAG_SPLIT_LISTS): its head, then every segment's own records
    const bool split = seg_flag && seg_flag[unit] == 1u;
    const uint32_t nseg = split ? (end - start + SEG_LEN - 1) / SEG_LEN : 1u;
  for (uint32_t ks = 0; ks < nseg; ks = ks ? ks + 1 : (split ? SEG_K0 : 1u)) {
    const uint32_t n_surv = ks == 0 ? surv_count[unit] : seg_cnt[((size_t)SEG_PER_CHUNK * chunk_start[t] + ks) * 4 + wave];
    const float4* const sv = surv + 3 * ((size_t)4 * start + (size_t)wave * (end - start) + (size_t)ks * SEG_LEN);
    evaluated += (unsigned long long)n_surv * 64ull;
    for (uint32_t g0 = 0; g0 < n_surv; g0 += 64) {
        const bool ok = g0 + lane < n_surv;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
        if (ok) {
            const float4* r = sv + 3 * (size_t)(g0 + lane);
            a = r[0]; b = r[1]; c = r[2];
        }
        const uint32_t pos = ok ? __float_as_uint(c.z) : 0xFFFFFFFFu;
        for (int p = 0; p < 64; ++p) {
            const float dx = a.x - (float)(qx + (p & 7)), dy = a.y - (float)(qy + (p >> 3));
            const float power = power2(a.z, a.w, b.x, dx, dy);
            const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power));
            cnt += (pos < s_last[p] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f)) ? 1u : 0u;
        }
    }
  }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
    if (lane == 0) {
        atomicAdd(&out[0], (unsigned long long)cnt);
        atomicAdd(&out[1], evaluated);
    }
}

// Measurement aid, second form: WHERE the backward composite's evaluated (entry, pixel) slots go.  One wave per
// (tile, quadrant) replays the survivor records in k_render_bwd_em's own structure — groups of 64 records in the lanes, back to front,
// sixteen four-pixel runs per group, a run skipped when none of its pixels reaches back to the group — and classifies every lane x
// pixel slot of every EXECUTED run:
//   out[0] slots executed (64 lanes x 4 pixels per run)      out[1] padding: lanes beyond the group's records
//   out[2] "stopped": the entry lies at or behind the pixel's last contributor (the pixel saturated in front of it, or nothing of
//          the list reaches it any more)                      out[3] "miss": power > 0 or alpha < 1/255 (the splat does not reach)
//   out[4] contributing                                       out[5] slots of runs the qmask test skipped (not executed)
//   out[6] units with records                                 out[7] groups executed
//   out[8 + b], b = 0..9: executed slots of the units whose contributing fraction falls in [b/10, (b+1)/10)
//   out[18 + r], r = 0..7: executed runs by the number of records among the group's 64 that contribute to at least one of the
//          run's four pixels: r = 0: none, 1: 1-4, 2: 5-8, 3: 9-16, 4: 17-24, 5: 25-32, 6: 33-48, 7: 49-64
__global__ __launch_bounds__(64) void k_count_bwd_slots(KParams kp, const uint32_t* __restrict__ tile_start, const float4* __restrict__ surv,
                                                       const uint32_t* __restrict__ surv_count, const uint32_t* __restrict__ n_contrib,
                                                       unsigned long long* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t s_last[64];
    const uint32_t unit = blockIdx.x;
    const int t = (int)(unit >> 2), wave = (int)(unit & 3u), lane = threadIdx.x;
    const int v = t / kp.T, rem = t - v * kp.T;
    const int ty = rem / kp.gx, tx = rem - ty * kp.gx;
    const int qx = tx * 16 + sub_ox(wave), qy = ty * 16 + sub_oy(wave);
    const uint32_t start = min(tile_start[t], kp.cap), end = min(tile_start[t + 1], kp.cap);
    {
        const int px = qx + (lane & 7), py = qy + (lane >> 3);
        s_last[lane] = (px < kp.W && py < kp.H) ? n_contrib[((size_t)v * kp.H + py) * kp.W + px] : 0u;
    }
    __syncthreads();
    const uint32_t n_surv = surv_count[unit];
    if (n_surv == 0) return;
    const float4* const sv = surv + 3 * ((size_t)4 * start + (size_t)wave * (end - start));
    uint32_t c_pad = 0, c_stop = 0, c_miss = 0, c_hit = 0;
    uint32_t h_top = 0, h_bot = 0, h_both = 0;    // records whose splat reaches the quadrant's upper / lower 8x4 half (the composites' own box test)
    unsigned long long runs_exec = 0, runs_skip = 0, groups = 0;
    unsigned long long rh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t top = (int64_t)n_surv - 1; top >= 0; top -= 64) {
        const uint32_t n = (uint32_t)(top + 1 < 64 ? top + 1 : 64);
        const bool lane_ok = (uint32_t)lane < n;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
        if (lane_ok) {
            const float4* r = sv + 3 * (size_t)(top - lane);
            a = r[0]; b = r[1]; c = r[2];
        }
        const uint32_t pos = lane_ok ? __float_as_uint(c.z) : 0xFFFFFFFFu;
        if (lane_ok) {
            const float ka = -a.w / (2.0f * a.z), kb = -a.w / (2.0f * b.x);
            const bool ht = box_hit_rt(a.x, a.y, a.z, a.w, b.x, b.y, ka, kb, (float)qx, (float)qy, 7.0f, 3.0f);
            const bool hb = box_hit_rt(a.x, a.y, a.z, a.w, b.x, b.y, ka, kb, (float)qx, (float)(qy + 4), 7.0f, 3.0f);
            h_top += ht; h_bot += hb; h_both += ht && hb;
        }
        const uint32_t pos_min = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)n - 1);
        const uint4 l4 = *reinterpret_cast<const uint4*>(&s_last[(lane & 15) * 4]);
        const uint32_t qmask = (uint32_t)__ballot(max(max(l4.x, l4.y), max(l4.z, l4.w)) > pos_min) & 0xFFFFu;
        ++groups;
        for (int run = 0; run < 16; ++run) {
            if (!((qmask >> run) & 1u)) { ++runs_skip; continue; }
            ++runs_exec;
            bool any = false;
            for (int k = 0; k < 4; ++k) {
                const int p = run * 4 + k;
                if (!lane_ok) { ++c_pad; continue; }
                const float dx = a.x - (float)(qx + (p & 7)), dy = a.y - (float)(qy + (p >> 3));
                const float power = power2(a.z, a.w, b.x, dx, dy);
                const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power));
                if (!(pos < s_last[p])) ++c_stop;
                else if (power > 0.0f || alpha < 1.0f / 255.0f) ++c_miss;
                else { ++c_hit; any = true; }
            }
            const int na = __popcll(__ballot(any));
            ++rh[na == 0 ? 0 : na <= 4 ? 1 : na <= 8 ? 2 : na <= 16 ? 3 : na <= 24 ? 4 : na <= 32 ? 5 : na <= 48 ? 6 : 7];
        }
    }
    uint32_t tot[7] = {c_pad, c_stop, c_miss, c_hit, h_top, h_bot, h_both};
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot[k] += (uint32_t)__shfl_xor((int)tot[k], o);
    if (lane == 0) {
        const unsigned long long exec = runs_exec * 256ull;
        atomicAdd(&out[0], exec);
        atomicAdd(&out[1], (unsigned long long)tot[0]);
        atomicAdd(&out[2], (unsigned long long)tot[1]);
        atomicAdd(&out[3], (unsigned long long)tot[2]);
        atomicAdd(&out[4], (unsigned long long)tot[3]);
        atomicAdd(&out[5], runs_skip * 256ull);
        atomicAdd(&out[6], 1ull);
        atomicAdd(&out[7], groups);
        const int bin = exec ? min(9, (int)(10ull * tot[3] / exec)) : 0;
        atomicAdd(&out[8 + bin], exec);
        for (int r = 0; r < 8; ++r) atomicAdd(&out[18 + r], rh[r]);
        // out[26..28]: records reaching the upper half / the lower half / both; out[29]: records; out[30]: 8-run iterations of a
        // composite that walks the two halves' records side by side, 32 + 32 lanes (max over the halves of ceil(records / 32))
        atomicAdd(&out[26], (unsigned long long)tot[4]);
        atomicAdd(&out[27], (unsigned long long)tot[5]);
        atomicAdd(&out[28], (unsigned long long)tot[6]);
        atomicAdd(&out[29], (unsigned long long)n_surv);
        atomicAdd(&out[30], (unsigned long long)max((tot[4] + 31u) / 32u, (tot[5] + 31u) / 32u));
    }
}

// [P,10] packed gradients (3 mean + 6 unique covariance entries + 1 opacity) -> the three tensors of the reference's layouts
__global__ __launch_bounds__(S360_BLOCK) void k_unpack_gradients(const float* __restrict__ packed, int P, int cov9,
                                                                 float* __restrict__ d_means, float* __restrict__ d_cov,
                                                                 float* __restrict__ d_opac) {
    const int g = blockIdx.x * S360_BLOCK + threadIdx.x;
    if (g >= P) return;
    const float* s = packed + 10 * (size_t)g;
    d_means[3 * g] = s[0]; d_means[3 * g + 1] = s[1]; d_means[3 * g + 2] = s[2];
    if (cov9) {
        float* o = d_cov + 9 * (size_t)g;
        o[0] = s[3]; o[1] = s[4]; o[2] = s[5];
        o[3] = 0.f;  o[4] = s[6]; o[5] = s[7];
        o[6] = 0.f;  o[7] = 0.f;  o[8] = s[8];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) d_cov[6 * (size_t)g + k] = s[3 + k];
    }
    d_opac[g] = s[9];
}

// The local reduction of the exchange's "gather" form fused with the unpack: n_blocks gathered [count,10] row blocks (one per rank,
// block_stride floats apart) are summed in block order and written straight into rows [g_begin, g_begin + count) of the three tensors.
__global__ __launch_bounds__(S360_BLOCK) void k_reduce_unpack(const float* __restrict__ blocks, int n_blocks, size_t block_stride, int g_begin,
                                                              int count, int cov9, float* __restrict__ d_means, float* __restrict__ d_cov,
                                                              float* __restrict__ d_opac) {
    const int i = blockIdx.x * S360_BLOCK + threadIdx.x;
    if (i >= count) return;
    float acc[10];
    {
        const float* s = blocks + 10 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = s[k];
    }
    for (int b = 1; b < n_blocks; ++b) {
        const float* s = blocks + (size_t)b * block_stride + 10 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] += s[k];
    }
    const size_t g = (size_t)g_begin + i;
    d_means[3 * g] = acc[0]; d_means[3 * g + 1] = acc[1]; d_means[3 * g + 2] = acc[2];
    if (cov9) {
        float* o = d_cov + 9 * g;
        o[0] = acc[3]; o[1] = acc[4]; o[2] = acc[5];
        o[3] = 0.f;    o[4] = acc[6]; o[5] = acc[7];
        o[6] = 0.f;    o[7] = 0.f;    o[8] = acc[8];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) d_cov[6 * g + k] = acc[3 + k];
    }
    d_opac[g] = acc[9];
}
}  // namespace s360

extern "C" int s360_reduce_unpack_gradients(const float* packed_blocks, int32_t n_blocks, int32_t g_begin, int32_t g_count, int32_t cov9,
                                            float* d_means3D, float* d_cov, float* d_opacities, void* stream_) {
    if (n_blocks < 1 || g_begin < 0 || g_count < 0 || (g_count > 0 && (!packed_blocks || !d_means3D || !d_cov || !d_opacities))) return S360_E_BADARG;
    if (g_count == 0) return S360_OK;
    hipLaunchKernelGGL(s360::k_reduce_unpack, dim3((g_count + S360_BLOCK - 1) / S360_BLOCK), dim3(S360_BLOCK), 0, (hipStream_t)stream_,
                       packed_blocks, n_blocks, (size_t)g_count * 10, g_begin, g_count, cov9, d_means3D, d_cov, d_opacities);
    S360_CHECK_LAUNCH();
    return S360_OK;
}

extern "C" int s360_unpack_gradients(const float* packed, int32_t P, int32_t cov9, float* d_means3D, float* d_cov, float* d_opacities,
                                     void* stream_) {
    if (P < 0 || (P > 0 && (!packed || !d_means3D || !d_cov || !d_opacities))) return S360_E_BADARG;
    if (P == 0) return S360_OK;
    hipLaunchKernelGGL(s360::k_unpack_gradients, dim3((P + S360_BLOCK - 1) / S360_BLOCK), dim3(S360_BLOCK), 0, (hipStream_t)stream_,
                       packed, P, cov9, d_means3D, d_cov, d_opacities);
    S360_CHECK_LAUNCH();
    return S360_OK;
}

extern "C" int s360_count_contributions(const S360Params* prm, const void* workspace, size_t workspace_bytes, uint64_t* counts, void* stream_) {
    if (!prm || !workspace || !counts) return S360_E_BADARG;
    if (prm->flags & S360_FLAG_FORWARD_ONLY) return S360_E_BADARG;   // survivor records exist in training workspaces only
    S360Layout L;
    const int rc = s360_layout(prm, &L);
    if (rc) return rc;
    if (workspace_bytes < L.total_bytes) return S360_E_WORKSPACE;
    KParams kp;
    kp.P = prm->P; kp.V = prm->V; kp.H = prm->H; kp.W = prm->W; kp.deg = s360_effective_degree(prm); kp.M = prm->M;
    kp.gx = (prm->W + 15) / 16; kp.gy = (prm->H + 15) / 16; kp.T = kp.gx * kp.gy;
    kp.flags = prm->flags; kp.cap = prm->max_instances;
    const int nt = ((kp.flags & S360_FLAG_SPHERICAL) ? kp.V / 2 : kp.V) * kp.T;
    const char* ws = (const char*)workspace;
    hipStream_t st = (hipStream_t)stream_;
    if (hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), st) != hipSuccess) return S360_E_LAUNCH;
    if (prm->P == 0) return S360_OK;
    const bool split = (kp.flags & S360_FLAG_SPLIT_LISTS) != 0;
    hipLaunchKernelGGL(s360::k_count_pairs, dim3(nt * 4), dim3(64), 0, st, kp, (const uint32_t*)(ws + L.tile_start), (const float4*)(ws + L.surv),
                       (const uint32_t*)(ws + L.surv_count), (const uint32_t*)(ws + L.n_contrib), (unsigned long long*)counts,
                       split ? (const uint32_t*)(ws + L.seg_flag) : (const uint32_t*)nullptr, (const uint32_t*)(ws + L.chunk_start),
                       (const uint32_t*)(ws + L.seg_cnt));
    S360_CHECK_LAUNCH();
    return S360_OK;
}

// the last kernel of the raw backward: per-Gaussian dL/dcov6 [, dL/dmean] and n_groups clamp-masked dL/dRGB sums (one per camera
// centre: one for a single-GPU call, one per rank after the multi-GPU exchange) -> dL/d(raw record), dL/ddepth
static int raw_tail(const S360Params* prm, const S360View* views, int n_groups, const S360RawInputs* raw, const float* means, const void* workspace,
                    const float* d_means_or_null, const float* d_cov6, const float* d_rgb_sums, float* d_depths, float* d_raw_gaussians,
                    void* stream_) {
    S360Layout L;
    const int rc = s360_layout(prm, &L);
    if (rc) return rc;
    KParams kp;
    kp.P = prm->P; kp.V = prm->V; kp.H = prm->H; kp.W = prm->W; kp.deg = 4; kp.M = 25;
    kp.gx = (prm->W + 15) / 16; kp.gy = (prm->H + 15) / 16; kp.T = kp.gx * kp.gy;
    kp.flags = prm->flags; kp.cap = prm->max_instances;
    RawBwd rb{raw->extrinsics, raw->depths, (const float*)((const char*)workspace + L.geo7), raw->sh_rotation, means,
              d_means_or_null, d_cov6, (const float4*)d_rgb_sums, d_depths, d_raw_gaussians,
              raw->per_view > 0 ? raw->per_view : 1, raw->H, raw->W, raw->per_ray, raw->erp_convention, n_groups, raw->scale_min, raw->scale_max, raw->eps};
    {
        ProfScope ps(PS_SH_BWD, (hipStream_t)stream_);
        const dim3 bgrid((rb.Gv + 63) / 64, (unsigned)(prm->P / rb.Gv));   // per context view: its rotation matrix is wave-uniform
        if (rb.sh_rot) hipLaunchKernelGGL(k_raw_bwd<true>, bgrid, dim3(192), 0, (hipStream_t)stream_, kp, views, rb);
        else hipLaunchKernelGGL(k_raw_bwd<false>, bgrid, dim3(192), 0, (hipStream_t)stream_, kp, views, rb);
    }
    S360_CHECK_LAUNCH();
    return S360_OK;
}

extern "C" int s360_backward_raw(const S360Params* prm, const S360View* views, const S360RawInputs* raw, const float* means, const float* cov6,
                                 const float* opacities, const void* workspace, size_t workspace_bytes, const float* dL_dimages,
                                 const float* dL_dimages_scale, const float* dL_ddepth, int32_t depth_mode, int32_t differentiable_means,
                                 float* d_means3D, float* d_cov6, float* d_opacities, float* d_rgb_sum, float* d_depths,
                                 float* d_raw_gaussians, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream_) {
    if (!prm || !raw || !means || !cov6 || !d_means3D || !d_cov6 || !d_opacities || !d_rgb_sum || !d_depths || !d_raw_gaussians)
        return S360_E_BADARG;
    if (!(prm->flags & S360_FLAG_RAW_INPUTS) || !(prm->flags & S360_FLAG_SHARED_CAMPOS) || (prm->flags & (S360_FLAG_COV9 | S360_FLAG_SPHERICAL)))
        return S360_E_BADARG;
    if (prm->M != 25 || prm->sh_degree != 4) return S360_E_UNSUPPORTED;
    if ((long long)raw->n_views * raw->per_view != (long long)prm->P) return S360_E_BADARG;
    // the rasteriser's own backward without its dL/dSH pass (the s360_backward_split form: per-Gaussian dL/dmean, dL/dcov6, dL/dopacity
    // and the clamp-masked sum of dL/dRGB) ...
    const float* dummy_sh = raw->raw_gaussians;   // never read: views sharing a camera centre take the colour terms from sh_jac
    int rc = backward_impl(prm, views, means, cov6, opacities, dummy_sh, nullptr, workspace, workspace_bytes, dL_dimages, dL_dimages_scale,
                           dL_ddepth, depth_mode, d_means3D, nullptr, d_cov6, d_opacities, nullptr, nullptr, d_rgb_sum, bwd_workspace,
                           bwd_workspace_bytes, stream_);
    if (rc || prm->P == 0) return rc;
    // ... then ONE kernel down to the encoder's outputs
    return raw_tail(prm, views, 1, raw, means, workspace, differentiable_means ? d_means3D : (const float*)nullptr, d_cov6, d_rgb_sum, d_depths,
                    d_raw_gaussians, stream_);
}

extern "C" int s360_count_backward_slots(const S360Params* prm, const void* workspace, size_t workspace_bytes, uint64_t* counts, void* stream_) {
    if (!prm || !workspace || !counts) return S360_E_BADARG;
    if (prm->flags & (S360_FLAG_FORWARD_ONLY | S360_FLAG_SPLIT_LISTS | S360_FLAG_SPHERICAL)) return S360_E_BADARG;   // the unsplit training composite's structure
    S360Layout L;
    const int rc = s360_layout(prm, &L);
    if (rc) return rc;
    if (workspace_bytes < L.total_bytes) return S360_E_WORKSPACE;
    KParams kp;
    kp.P = prm->P; kp.V = prm->V; kp.H = prm->H; kp.W = prm->W; kp.deg = s360_effective_degree(prm); kp.M = prm->M;
    kp.gx = (prm->W + 15) / 16; kp.gy = (prm->H + 15) / 16; kp.T = kp.gx * kp.gy;
    kp.flags = prm->flags; kp.cap = prm->max_instances;
    const int nt = kp.V * kp.T;
    const char* ws = (const char*)workspace;
    hipStream_t st = (hipStream_t)stream_;
    if (hipMemsetAsync(counts, 0, 32 * sizeof(uint64_t), st) != hipSuccess) return S360_E_LAUNCH;
    if (prm->P == 0) return S360_OK;
    hipLaunchKernelGGL(s360::k_count_bwd_slots, dim3(nt * 4), dim3(64), 0, st, kp, (const uint32_t*)(ws + L.tile_start), (const float4*)(ws + L.surv),
                       (const uint32_t*)(ws + L.surv_count), (const uint32_t*)(ws + L.n_contrib), (unsigned long long*)counts);
    S360_CHECK_LAUNCH();
    return S360_OK;
}

extern "C" int s360_backward_raw_tail(const S360Params* prm, const S360View* group_views, int32_t n_groups, const S360RawInputs* raw,
                                      const float* means, const void* workspace, size_t workspace_bytes, const float* d_means3D,
                                      const float* d_cov6, const float* d_rgb_sums, float* d_depths, float* d_raw_gaussians, void* stream_) {
    if (!prm || !group_views || !raw || !means || !workspace || !d_cov6 || !d_rgb_sums || !d_depths || !d_raw_gaussians || n_groups < 1)
        return S360_E_BADARG;
    if (!(prm->flags & S360_FLAG_RAW_INPUTS) || (prm->flags & (S360_FLAG_COV9 | S360_FLAG_SPHERICAL | S360_FLAG_FORWARD_ONLY))) return S360_E_BADARG;
    if (prm->M != 25 || prm->sh_degree != 4) return S360_E_UNSUPPORTED;
    if ((long long)raw->n_views * raw->per_view != (long long)prm->P) return S360_E_BADARG;
    S360Layout L;
    const int rc = s360_layout(prm, &L);
    if (rc) return rc;
    if (workspace_bytes < L.total_bytes) return S360_E_WORKSPACE;
    if (prm->P == 0) return S360_OK;
    return raw_tail(prm, group_views, n_groups, raw, means, workspace, d_means3D, d_cov6, d_rgb_sums, d_depths, d_raw_gaussians, stream_);
}
