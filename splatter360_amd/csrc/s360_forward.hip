// s360_forward.hip — forward kernels: fused multi-view preprocess, scans, tile binning,
// per-tile depth sort, front-to-back composite.  gfx950 / wave64 only.
//
// Pipeline (all asynchronous on the caller's stream, no host sync, no library-global state):
//   k_preprocess     1 thread / Gaussian, loops the V views in registers: cull, EWA cov2D, conic,
//                    radius, tile rect; SH->RGB once per Gaussian when the views share campos
//                    (each lane streams its own 300-byte slab with 16-byte loads).
//   k_tile_scan      exclusive scan of the per-tile instance counts -> tile ranges (+ sort-chunk table)
//   k_emit           scatter (depth bits << 32 | pair) keys into their tile's bucket; training calls: reserve the
//                    pairs' instance slots (one atomic per block) and write the slot owner table
//   k_sort_stage1 / k_merge_all
//                    per-tile ascending sort of the unique 64-bit keys (== stable radix sort by
//                    (tile, depth) with ascending-index emission): LDS merge sort per list or per
//                    4096-key chunk, global merge-path passes for multi-chunk lists
//   k_render         1 workgroup per 16x16 tile = 4 autonomous waves of 8x8 pixels (no LDS, no barriers)
#include "s360_device.h"
#include "s360_prof.h"
#include "s360_adapter_math.h"

#include <cstdio>
#include <cstdlib>

namespace s360 {

typedef float f4v __attribute__((ext_vector_type(4)));   // a 128-bit register tuple (vector loads, inline-asm operands)

// ------------------------------------------------------------------------------ SH -> RGB (shared camera centre)
// When all views of a call share one camera centre (the six faces of a panorama) the colour of a Gaussian does not
// depend on the view: a streaming kernel evaluates it ONCE per Gaussian ahead of the geometry pass, and the geometry
// kernel (k_preprocess<.., EAGER = true>) carries no slab code.  Training calls (JAC) also store
// J_c = d(rgb_c)/d(mean) through the view direction — (G_c - dir (dir . G_c)) * scale / |d| with
// G_c = sum_k grad Y_k(dir) sh_kc — 36 bytes with which the backward adds that term to dL/dmean and never re-reads the slab.
// k_sh_eval is the general form (either layout, any degree): one lane per Gaussian streams its own slab with 16-byte loads,
// ALL of them (19 for 75 coefficients) in flight before the first use — every byte of every cache line is consumed by the
// same lane within a few instructions, so HBM traffic stays 1x whatever the occupancy (loading one colour channel at a
// time, the round-1 form, re-fetched every line three times: 365 us for the forward's SH + geometry pair instead of 140).
// The reference's own layout (channel-major, degree 4) takes k_sh_eval3 / k_sh_eval3_jac below.
template <bool CH_MAJOR, bool JAC>
__global__ __launch_bounds__(S360_BLOCK) void k_sh_eval(KParams kp, const S360View* __restrict__ views,
                                                       const float* __restrict__ means, const float* __restrict__ shs,
                                                       float4* __restrict__ rgbc, float* __restrict__ sh_jac,
                                                       uint32_t* __restrict__ zero_ptr, int zero_words) {
    const int g = blockIdx.x * S360_BLOCK + threadIdx.x;
    if (g < zero_words) zero_ptr[g] = 0u;  // the call's tile histogram + slot tickets (first kernel of the call: no memset node)
    if (g >= kp.P) return;
    const bool fast = kp.M == 25 && kp.deg == 4;
    const float* sh = shs + (size_t)g * kp.M * 3;
    float c[75];
    if (fast) load75(sh, c);
    const S360View& vw = views[0];
    const float sc = vw.scale;
    const float dx = means[3 * g] * sc - vw.campos[0], dy = means[3 * g + 1] * sc - vw.campos[1], dz = means[3 * g + 2] * sc - vw.campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx * inv, y = dy * inv, z = dz * inv;
    const int n = (kp.deg + 1) * (kp.deg + 1);
    const int sk = CH_MAJOR ? 1 : 3, sc_ = CH_MAJOR ? kp.M : 1;
    float acc[3], G[3][3];
    // (Measured alternatives for the training variant, whose four 25-entry basis tables + 75 coefficients need 200 VGPRs =
    // 2 waves per SIMD: one table at a time in four passes compiles to 176 VGPRs — still 2 waves; forcing 3 waves spills
    // and runs 121 us instead of 106 us.)
    float Y[25], bx[25], by[25], bz[25];
    sh_basis(kp.deg, x, y, z, Y);
    if (JAC) sh_basis_grad(kp.deg, x, y, z, bx, by, bz);
    // coefficient-major loop: the three colour sums (sequential, unfused: the oracle's rounding) and the nine jacobian sums
    // (fused multiply-adds) are twelve independent dependency chains — at 2 waves per SIMD the issue stalls of a
    // channel-major loop (four 25-deep chains at a time) were 56 % of this kernel's wave cycles
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        acc[ch] = 0.f;
        G[ch][0] = G[ch][1] = G[ch][2] = 0.f;
    }
    if (fast) {
#pragma unroll
        for (int k = 0; k < 25; ++k) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float ck = c[CH_MAJOR ? 25 * ch + k : 3 * k + ch];
                acc[ch] += Y[k] * ck;
                if (JAC) {
                    G[ch][0] = __builtin_fmaf(bx[k], ck, G[ch][0]);
                    G[ch][1] = __builtin_fmaf(by[k], ck, G[ch][1]);
                    G[ch][2] = __builtin_fmaf(bz[k], ck, G[ch][2]);
                }
            }
        }
    } else {
        for (int k = 0; k < n; ++k) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float ck = sh[k * sk + ch * sc_];
                acc[ch] += Y[k] * ck;
                if (JAC) {
                    G[ch][0] = __builtin_fmaf(bx[k], ck, G[ch][0]);
                    G[ch][1] = __builtin_fmaf(by[k], ck, G[ch][1]);
                    G[ch][2] = __builtin_fmaf(bz[k], ck, G[ch][2]);
                }
            }
        }
    }
    const float a0 = acc[0] + 0.5f, a1 = acc[1] + 0.5f, a2 = acc[2] + 0.5f;
    const uint32_t clampbits = (a0 < 0.f ? 1u : 0u) | (a1 < 0.f ? 2u : 0u) | (a2 < 0.f ? 4u : 0u);
    rgbc[g] = make_float4(fmaxf(a0, 0.f), fmaxf(a1, 0.f), fmaxf(a2, 0.f), __uint_as_float(clampbits));
    if (JAC) {
        float* o = sh_jac + 9 * (size_t)g;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float dot = x * G[ch][0] + y * G[ch][1] + z * G[ch][2];
            o[3 * ch] = sc * ((G[ch][0] - x * dot) * inv);
            o[3 * ch + 1] = sc * ((G[ch][1] - y * dot) * inv);
            o[3 * ch + 2] = sc * ((G[ch][2] - z * dot) * inv);
        }
    }
}

// Channel-major slabs ([P,3,25], the reference's Gaussians.harmonics layout) at degree 4, inference form: ONE LANE PER
// (Gaussian, colour channel).  The workgroup's 64 slabs are 19 200 contiguous bytes: coalesced 16-byte loads into LDS, then
// thread t owns floats [25 t, 25 t + 25) (lane stride 25 words: conflict-free) — a lane keeps 25 coefficients instead of 75
// (49 VGPRs against 126), the channel's sum is the same sequential sum (bit-identical colours), and the three channels meet
// in LDS for the packed colour record.  63 us for 1 M Gaussians (343 MB: 5.4 TB/s) against 70 us one lane per Gaussian.
// (Per-lane unaligned 16-byte loads of the 100-byte runs instead of the LDS stage: 112 us — at 8 waves per SIMD a lane's
// lines are evicted from the 32-KB L1 between its successive loads.)
constexpr int SHE3_G = 64;  // Gaussians per workgroup (192 threads)
__global__ __launch_bounds__(SHE3_G * 3) void k_sh_eval3(KParams kp, const S360View* __restrict__ views, const float* __restrict__ means,
                                                        const float* __restrict__ shs, float4* __restrict__ rgbc,
                                                        uint32_t* __restrict__ zero_ptr, int zero_words) {
    __shared__ float s_rgb[SHE3_G * 3];
    if ((int)(blockIdx.x * (SHE3_G * 3) + threadIdx.x) < zero_words) zero_ptr[blockIdx.x * (SHE3_G * 3) + threadIdx.x] = 0u;
    __shared__ __attribute__((aligned(16))) float s_sh[7 * SHE3_G * 3 * 4];  // the workgroup's 64 slabs (19 200 contiguous bytes) + pad
    const int tid = threadIdx.x;
    const int gl = tid / 3;
    const int g0 = blockIdx.x * SHE3_G;
    const int g = g0 + gl;
    {
        const float* src = shs + (size_t)g0 * 75;
        const int nfl = min(SHE3_G, kp.P - g0) * 75;
        if ((((uintptr_t)src) & 15) == 0 && nfl == SHE3_G * 75) {
            // full workgroup: 1 200 float4 = 6.25 per thread.  Loads AND stores unconditional (the 7th round's surplus lanes
            // re-read the last vector into the pad): a guarded load compiles to a branch with a full wait per round
            float4* d4 = reinterpret_cast<float4*>(s_sh);
            float4 q[7];
            // (f4v loads non-temporal: every slab byte is read once per call, 63 -> 54 us)
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src) + min(tid + r * (SHE3_G * 3), SHE3_G * 75 / 4 - 1));
                q[r] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) d4[tid + r * (SHE3_G * 3)] = q[r];
        } else {
            for (int i = tid; i < nfl; i += SHE3_G * 3) s_sh[i] = src[i];
        }
    }
    __syncthreads();
    if (g < kp.P) {
        const S360View& vw = views[0];
        const float sc = vw.scale;
        const float dx = means[3 * g] * sc - vw.campos[0], dy = means[3 * g + 1] * sc - vw.campos[1], dz = means[3 * g + 2] * sc - vw.campos[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        float Y[25];
        sh_basis(4, dx * inv, dy * inv, dz * inv, Y);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 25; ++k) acc += Y[k] * s_sh[tid * 25 + k];
        s_rgb[tid] = acc + 0.5f;
    }
    __syncthreads();
    if (tid < SHE3_G && g0 + tid < kp.P) {
        const float a0 = s_rgb[3 * tid], a1 = s_rgb[3 * tid + 1], a2 = s_rgb[3 * tid + 2];
        const uint32_t clampbits = (a0 < 0.f ? 1u : 0u) | (a1 < 0.f ? 2u : 0u) | (a2 < 0.f ? 4u : 0u);
        rgbc[g0 + tid] = make_float4(fmaxf(a0, 0.f), fmaxf(a1, 0.f), fmaxf(a2, 0.f), __uint_as_float(clampbits));
    }
}

// Training form of the kernel above (colours + J = d rgb / d mean).  With one lane per (Gaussian, channel) every lane would
// evaluate all 75 basis derivatives (700 VALU instructions per lane).  Here WAVE d of the workgroup owns derivative component d (x, y, z) and colour channel
// d of the 64 Gaussians: a lane evaluates the basis, ONE component of its gradient (the zero entries skipped), its
// channel's colour sum (same sequential sum: bit-identical colours) and G[ch][d] = sum_k dY_k/dd sh[ch][k] for the three
// channels, reading the 75 coefficients from the staged slab (lane stride 75 words: conflict-free).  The nine sums meet in
// LDS; thread t then writes floats [3 t, 3 t + 3) of sh_jac (row t % 3 of Gaussian t / 3).
// In a training loop this kernel measures ~100 us although it runs 72 us on its own (first step of a run): it follows the
// previous step's k_sh_bwd, whose 315 MB of dL/dSH are still being written back from the L2 / Infinity Cache while it reads
// (ablations: without the jacobian stores 96 us, without the gradient evaluation 99 us — neither is the cost).
constexpr uint32_t SHG_NZ_X = 0x1E7FFD8u, SHG_NZ_Y = 0x1CFFF72u, SHG_NZ_Z = 0x0FE7CE4u;  // non-zero entries of sh_basis_grad's dx / dy / dz at degree 4
template <int D>
__device__ __forceinline__ void sh_jac_component(float x, float y, float z, const float* __restrict__ slab, float* G) {
    float b0[25], b1[25], b2[25];
    sh_basis_grad(4, x, y, z, b0, b1, b2);
    const float* b = D == 0 ? b0 : (D == 1 ? b1 : b2);
    constexpr uint32_t NZ = D == 0 ? SHG_NZ_X : (D == 1 ? SHG_NZ_Y : SHG_NZ_Z);
    G[0] = G[1] = G[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        if (!((NZ >> k) & 1u)) continue;  // fma(0, c, G) == G
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) G[ch] = __builtin_fmaf(b[k], slab[25 * ch + k], G[ch]);
    }
}

__global__ __launch_bounds__(SHE3_G * 3) void k_sh_eval3_jac(KParams kp, const S360View* __restrict__ views, const float* __restrict__ means,
                                                            const float* __restrict__ shs, float4* __restrict__ rgbc,
                                                            float* __restrict__ sh_jac, uint32_t* __restrict__ zero_ptr, int zero_words) {
    __shared__ float s_rgb[SHE3_G * 3];
    __shared__ float s_G[9 * SHE3_G];   // [ch][d][Gaussian]
    if ((int)(blockIdx.x * (SHE3_G * 3) + threadIdx.x) < zero_words) zero_ptr[blockIdx.x * (SHE3_G * 3) + threadIdx.x] = 0u;
    __shared__ float4 s_dir[SHE3_G];    // (x, y, z, scale / |d|)
    __shared__ __attribute__((aligned(16))) float s_sh[7 * SHE3_G * 3 * 4];
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * SHE3_G;
    {
        const float* src = shs + (size_t)g0 * 75;
        const int nfl = min(SHE3_G, kp.P - g0) * 75;
        if ((((uintptr_t)src) & 15) == 0 && nfl == SHE3_G * 75) {
            float4* d4 = reinterpret_cast<float4*>(s_sh);
            float4 q[7];
            // non-temporal loads: every slab byte is read exactly once per call
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src) + min(tid + r * (SHE3_G * 3), SHE3_G * 75 / 4 - 1));
                q[r] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) d4[tid + r * (SHE3_G * 3)] = q[r];
        } else {
            for (int i = tid; i < nfl; i += SHE3_G * 3) s_sh[i] = src[i];
        }
    }
    const int d = tid >> 6, l = tid & 63;  // wave d: derivative component / colour channel d
    const int g = g0 + l;
    const S360View& vw = views[0];
    const float sc = vw.scale;
    float x = 0.f, y = 0.f, z = 1.f, inv = 0.f;
    if (g < kp.P) {
        const float dx = means[3 * g] * sc - vw.campos[0], dy = means[3 * g + 1] * sc - vw.campos[1], dz = means[3 * g + 2] * sc - vw.campos[2];
        inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        x = dx * inv; y = dy * inv; z = dz * inv;
    }
    __syncthreads();
    if (g < kp.P) {
        const float* slab = s_sh + l * 75;
        float Y[25];
        sh_basis(4, x, y, z, Y);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 25; ++k) acc += Y[k] * slab[25 * d + k];
        s_rgb[3 * l + d] = acc + 0.5f;
        float G[3];
        if (d == 0) sh_jac_component<0>(x, y, z, slab, G);
        else if (d == 1) sh_jac_component<1>(x, y, z, slab, G);
        else sh_jac_component<2>(x, y, z, slab, G);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) s_G[(3 * ch + d) * SHE3_G + l] = G[ch];
        if (d == 0) s_dir[l] = make_float4(x, y, z, sc * inv);
    }
    __syncthreads();
    {
        const int gl = tid / 3, ch = tid - 3 * gl;
        if (g0 + gl < kp.P) {
            const float G0 = s_G[(3 * ch) * SHE3_G + gl], G1 = s_G[(3 * ch + 1) * SHE3_G + gl], G2 = s_G[(3 * ch + 2) * SHE3_G + gl];
            const float4 dr = s_dir[gl];
            const float dot = dr.x * G0 + dr.y * G1 + dr.z * G2;
            float* o = sh_jac + 3 * ((size_t)g0 * 3 + tid);
            o[0] = (G0 - dr.x * dot) * dr.w;
            o[1] = (G1 - dr.y * dot) * dr.w;
            o[2] = (G2 - dr.z * dot) * dr.w;
        }
    }
    if (tid < SHE3_G && g0 + tid < kp.P) {
        const float a0 = s_rgb[3 * tid], a1 = s_rgb[3 * tid + 1], a2 = s_rgb[3 * tid + 2];
        const uint32_t clampbits = (a0 < 0.f ? 1u : 0u) | (a1 < 0.f ? 2u : 0u) | (a2 < 0.f ? 4u : 0u);
        rgbc[g0 + tid] = make_float4(fmaxf(a0, 0.f), fmaxf(a1, 0.f), fmaxf(a2, 0.f), __uint_as_float(clampbits));
    }
}

// ------------------------------------------------------------------------------ raw encoder outputs -> geometry + colours
// s360_forward_raw (the purpose of the adapter row): the first kernel of the call reads what the ENCODER emits — per
// context pixel a depth and a raw record of 3 scale logits, a quaternion and 3 x 25 SH coefficients
// (/root/reference/src/model/encoder/common/gaussian_adapter_erp.py:50-119) — instead of the [G,3,25] harmonics / [G,3,3] covariances /
// means the stand-alone adapter would materialise for it to read back (340 B/Gaussian written + 340 read).  One launch:
//   * the workgroup's 64 raw records (64 x 328 contiguous bytes) are staged into LDS with coalesced non-temporal 16-byte loads;
//   * wave 0: scales, normalised quaternion, Sigma = (C R) diag(s^2) (C R)^T, mean = C (ray * depth) + t with the adapter's own
//     expressions (s360_adapter_math.h: golden-pinned against the reference module) -> means[G,3], cov6[G,6] (36 B: what the
//     geometry pass reads) and the 7 raw geometry words compacted for the backward (28 B);
//   * all three waves (wave d = colour channel d and derivative component d, as in k_sh_eval3_jac): the SH basis at the view
//     direction is carried through the adapter's coefficient transform instead of the coefficients — harmonics = D_v (mask . raw)
//     per degree (gaussian_adapter_erp.py:86,113; rotate_sh, src/misc/sh_rotation.py:10-30), so
//         rgb = Y . harmonics = (mask . D_v^T Y) . raw,
//     165 multiply-adds per basis vector against 75 x 25 per Gaussian for rotating the coefficients — and the same for the basis
//     derivative behind sh_jac.  D_v (the context view's 25 x 25 block-diagonal matrix) sits in LDS.
// Colours differ from the two-step path (adapter kernel, then k_sh_eval3_jac) by float association only (<= 2e-6).
struct RawIn {
    const float* extrinsics;   // [n_views,4,4] context-panorama camera-to-world
    const float* depths;       // [P]
    const float* raw;          // [P, 82]
    const float* sh_rot;       // [n_views,25,25] or null (identity)
    float* means_out;          // [P,3]
    float* cov6_out;           // [P,6]
    float* geo7;               // [P,7] (workspace): the raw geometry words, for the backward
    int Gv, H, W, per_ray, conv;
    float smin, smax, eps;
};
constexpr int RAW_C = 82;   // 7 + 3 * 25 floats per raw record

// Round 6: the transform is applied to the COEFFICIENTS, once per (Gaussian, channel), with the view's matrix in SGPRs
// (sh_rotate_coefs25: 165 multiply-adds on wave ch, scalar loads of D — the kernel's workgroups are dealt per context view, so D is
// wave-uniform), written back into the staged record in place; colours and jacobian are then k_sh_eval3_jac's own code on that slab.
// Round 5 carried the BASIS through the transform instead (mask . D^T Y and the same for the basis derivative: 330 LDS-broadcast
// multiply-adds per lane on every one of the three waves, 145 us); rotating three coefficient vectors is less work than rotating
// four basis vectors, and the harmonics — hence colours, sh_jac, images — are now bit-identical to the two-step path's
// (adapter kernel, then k_sh_eval3_jac).
template <bool JAC, bool ROT, bool MFMA = false>
__global__ __launch_bounds__(SHE3_G * 3) void k_raw_eval(KParams kp, const S360View* __restrict__ views, RawIn rin, float4* __restrict__ rgbc,
                                                        float* __restrict__ sh_jac, uint32_t* __restrict__ zero_ptr, int zero_words) {
    __shared__ __attribute__((aligned(16))) float s_raw[7 * SHE3_G * 3 * 4];   // 64 records x 82 floats = 5 248 floats (+ pad to 7 rounds of 192 float4)
    __shared__ float s_rgb[SHE3_G * 3];
    __shared__ float s_G[9 * SHE3_G];
    __shared__ float4 s_dir[SHE3_G];
    const int tid = threadIdx.x;
    {
        const int lin = (int)((blockIdx.y * gridDim.x + blockIdx.x) * (SHE3_G * 3)) + tid;
        if (lin < zero_words) zero_ptr[lin] = 0u;
    }
    const int v = blockIdx.y;                                   // context view: wave-uniform (pose, SH rotation matrix in SGPRs)
    const int gi0 = blockIdx.x * SHE3_G;                        // first Gaussian of the block inside its view
    const int nb = min(SHE3_G, rin.Gv - gi0);
    const int g0 = v * rin.Gv + gi0;
    {
        const float* src = rin.raw + (size_t)g0 * RAW_C;
        const int nfl = nb * RAW_C;
        if ((((uintptr_t)src) & 15) == 0 && nb == SHE3_G) {
            float4* d4 = reinterpret_cast<float4*>(s_raw);
            float4 q[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src) + min(tid + r * (SHE3_G * 3), SHE3_G * RAW_C / 4 - 1));
                q[r] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) d4[tid + r * (SHE3_G * 3)] = q[r];
        } else {
            for (int i = tid; i < nfl; i += SHE3_G * 3) s_raw[i] = src[i];
        }
    }
    // Wave 0 un-projects the means (erp_dir: four full-precision sin / cos, ~280 instructions) while the staging loads fly, and hands
    // the view direction to the other two through LDS (round 5 let every wave do it: 9 M of the launch's 39 M VALU instructions,
    // profiles/r06_raw_pmc.txt); the covariance chain rides on wave 1 and the raw-geometry copy on wave 2, both behind their colour work.
    // (Measured and dropped: PERSISTENT workgroups walking the blocks grid-stride with the next block's records prefetched into
    // registers — 145 VGPRs, 3 waves per SIMD, 168 us instead of 92: a block's three barrier-separated phases are a serial chain,
    // and what hides it is the number of OTHER workgroups resident on the CU, not the depth of one workgroup's prefetch.)
    const int d = tid >> 6, l = tid & 63;
    const int g = g0 + l;
    const bool live = l < nb;
    const S360View& vw = views[0];
    const float sc = vw.scale;
    const float* E = rin.extrinsics + 16 * v;
    if (d == 0 && live) {
        const float depth = rin.depths[g];
        float dr[3];
        erp_dir((gi0 + l) / rin.per_ray, rin.H, rin.W, rin.conv, dr);
        const float p[3] = {dr[0] * depth, dr[1] * depth, dr[2] * depth};
        float mn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) mn[a] = (E[4 * a] * p[0] + E[4 * a + 1] * p[1] + E[4 * a + 2] * p[2]) + E[4 * a + 3];
        rin.means_out[3 * (size_t)g] = mn[0]; rin.means_out[3 * (size_t)g + 1] = mn[1]; rin.means_out[3 * (size_t)g + 2] = mn[2];
        const float dx = mn[0] * sc - vw.campos[0], dy = mn[1] * sc - vw.campos[1], dz = mn[2] * sc - vw.campos[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        s_dir[l] = make_float4(dx * inv, dy * inv, dz * inv, sc * inv);
    }
    // (Measured and dropped: wave 1's covariance chain in front of the barrier too, its seven geometry words loaded straight from
    // global memory — the non-temporal staging lines are not retained, FETCH_SIZE +113 MB, 98 us instead of 92.)
    float depth = 0.f;
    if (d == 1 && live) depth = rin.depths[g];
    __syncthreads();      // the staged records, the view directions
    float* slab = s_raw + l * RAW_C + 7;     // [3][25] channel-major coefficients of this lane's Gaussian
    float x = 0.f, y = 0.f, z = 1.f;
    if (MFMA) {
        // MFMA form (ROT only): the wave rotates its channel of all 64 records on the matrix cores, in place (every lane takes part:
        // rows of a partial block compute on stale LDS words and are never read); the colour sum then reads the harmonics back
        sh_rotate_coefs25_mfma(rin.sh_rot + (size_t)v * 625, s_raw, d, l);
        __syncthreads();  // the three channels' harmonics (also orders this wave's own LDS writes before its reads below)
        if (live) {
            const float4 dr = s_dir[l];
            x = dr.x; y = dr.y; z = dr.z;
            float Y[25];
            sh_basis(4, x, y, z, Y);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 25; ++k) acc += Y[k] * slab[25 * d + k];
            s_rgb[3 * l + d] = acc + 0.5f;
        }
    } else
    if (live) {
        const float4 dr = s_dir[l];
        x = dr.x; y = dr.y; z = dr.z;
        // channel d: harmonics = D (mask . raw), the adapter kernel's own expression
        const float* D = ROT ? rin.sh_rot + (size_t)v * 625 : nullptr;
        float c[25], h[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) c[k] = slab[25 * d + k];
        sh_rotate_coefs25<ROT>(D, c, h);
        float Y[25];
        sh_basis(4, x, y, z, Y);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 25; ++k) acc += Y[k] * h[k];     // k_sh_eval3_jac's sum, term for term
        s_rgb[3 * l + d] = acc + 0.5f;
        if (JAC) {
#pragma unroll
            for (int k = 0; k < 25; ++k) slab[25 * d + k] = h[k];   // in place: only wave d ever touched these 25 words
        }
    }
    if (JAC) {
        if (!MFMA) __syncthreads();  // the three channels' harmonics
        if (live) {
            float G[3];
            if (d == 0) sh_jac_component<0>(x, y, z, slab, G);     // wave-uniform: each wave compiles ONE derivative component
            else if (d == 1) sh_jac_component<1>(x, y, z, slab, G);
            else sh_jac_component<2>(x, y, z, slab, G);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) s_G[(3 * ch + d) * SHE3_G + l] = G[ch];
        }
    }
    if (d == 1 && live) {
        // ---- covariance: the adapter tail's own expressions (k_adapter_fwd)
        const float* rw = s_raw + l * RAW_C;
        const float px = 1.0f / (float)max(rin.W, rin.H);
        float sc3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) sc3[k] = ((rin.smin + (rin.smax - rin.smin) * sigmoidf(rw[k])) * depth) * px;
        QuatGeom qg;
        const float qr[4] = {rw[3], rw[4], rw[5], rw[6]};
        quat_geom(qr, rin.eps, qg);
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a][b] = E[4 * a] * qg.R[0][b] + E[4 * a + 1] * qg.R[1][b] + E[4 * a + 2] * qg.R[2][b];
        const float s2[3] = {sc3[0] * sc3[0], sc3[1] * sc3[1], sc3[2] * sc3[2]};
        float S[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) S[a][b] = S[b][a] = M[a][0] * s2[0] * M[b][0] + M[a][1] * s2[1] * M[b][1] + M[a][2] * s2[2] * M[b][2];
        float* oc = rin.cov6_out + 6 * (size_t)g;
        oc[0] = S[0][0]; oc[1] = S[0][1]; oc[2] = S[0][2]; oc[3] = S[1][1]; oc[4] = S[1][2]; oc[5] = S[2][2];
    }
    if (d == 2 && live && rin.geo7) {
        const float* rw = s_raw + l * RAW_C;
        float* o7 = rin.geo7 + 7 * (size_t)g;
#pragma unroll
        for (int k = 0; k < 7; ++k) o7[k] = rw[k];
    }
    __syncthreads();
    if (JAC) {
        const int gl = tid / 3, ch = tid - 3 * gl;
        if (gl < nb) {
            const float G0 = s_G[(3 * ch) * SHE3_G + gl], G1 = s_G[(3 * ch + 1) * SHE3_G + gl], G2 = s_G[(3 * ch + 2) * SHE3_G + gl];
            const float4 dr = s_dir[gl];
            const float dot = dr.x * G0 + dr.y * G1 + dr.z * G2;
            float* o = sh_jac + 3 * ((size_t)g0 * 3 + tid);
            o[0] = (G0 - dr.x * dot) * dr.w;
            o[1] = (G1 - dr.y * dot) * dr.w;
            o[2] = (G2 - dr.z * dot) * dr.w;
        }
    }
    if (tid < nb) {
        const float a0 = s_rgb[3 * tid], a1 = s_rgb[3 * tid + 1], a2 = s_rgb[3 * tid + 2];
        const uint32_t clampbits = (a0 < 0.f ? 1u : 0u) | (a1 < 0.f ? 2u : 0u) | (a2 < 0.f ? 4u : 0u);
        rgbc[g0 + tid] = make_float4(fmaxf(a0, 0.f), fmaxf(a1, 0.f), fmaxf(a2, 0.f), __uint_as_float(clampbits));
    }
}

// ------------------------------------------------------------------------------ preprocess
// COOP (S360_FLAG_COOP_WALK): rectangles of more than 32 tiles are counted by the whole wave, one tile per lane
template <bool USE_SH, bool CH_MAJOR, bool EAGER = false, bool COOP = false>  // EAGER: colours come from k_sh_eval (rgbc), no slab code here
__global__ __launch_bounds__(S360_BLOCK) void k_preprocess(
    KParams kp, const S360View* __restrict__ views, const float* __restrict__ means,
    const float* __restrict__ cov6, const float* __restrict__ opac, const float* __restrict__ shs,
    const float* __restrict__ colors, int32_t* __restrict__ radii, uint32_t* __restrict__ tiles_touched,
    float4* __restrict__ recA, float4* __restrict__ recB, float4* __restrict__ recC,
    uint8_t* __restrict__ clamped, float* __restrict__ depths, uint32_t* __restrict__ tile_count, int lds_hist,
    const float4* __restrict__ rgbc, uint8_t* __restrict__ vis_mask, uint2* __restrict__ slot_info) {
    // dynamic LDS: tile histogram V*T uint32 (lds_hist).  SH coefficients are NOT staged: every lane
    // streams its own Gaussian's 300-byte slab with 16-byte loads (all bytes of every cache line are
    // consumed by the same lane within a few instructions, so HBM traffic stays 1x) — this keeps the
    // kernel register-limited (12 waves/CU) instead of LDS-limited (4 waves/CU with a 77 KB slab).
    extern __shared__ __attribute__((aligned(16))) float lds_sh[];
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * S360_BLOCK;
    const int P = kp.P;
    // every thread runs the view loop (the per-view histogram mode below has barriers in it): a thread past the end works on the
    // last Gaussian again and writes nothing
    const bool act = g0 + tid < P;
    const int g = act ? g0 + tid : P - 1;
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds_sh);
    const bool lean = (kp.flags & S360_FLAG_LEAN_LISTS) != 0;
    // lds_hist: 1 = the block's tile histogram of ALL images in LDS (V*T words <= 48 KB); 2 = one image at a time (T words <= 48 KB:
    // faces beyond 512^2 with six views), flushed after every view; 0 = global atomics per instance (a 5-ms kernel at 16 M Gaussians
    // on 1024^2 faces before mode 2 existed)
    const int nhist = lds_hist == 2 ? kp.T : (image_of_view(kp, kp.V - 1) + 1) * kp.T;
    if (lds_hist)
        for (int i = tid; i < nhist; i += S360_BLOCK) hist[i] = 0u;

    __syncthreads();
    {
    const float mx0 = means[3 * g], my0 = means[3 * g + 1], mz0 = means[3 * g + 2];
    float c60[6];
    load_cov6(cov6, g, (kp.flags & S360_FLAG_COV9) != 0, c60);
    const float op = opac[g];
    constexpr bool ch_major = CH_MAJOR;

    const bool shared_cam = (kp.flags & S360_FLAG_SHARED_CAMPOS) != 0;
    float rgb[3] = {0.f, 0.f, 0.f};
    uint32_t clampbits = 0;
    bool have_rgb = false;
    if (!USE_SH) {
        rgb[0] = colors[3 * g];
        rgb[1] = colors[3 * g + 1];
        rgb[2] = colors[3 * g + 2];
        have_rgb = true;
    }

    uint32_t vis = 0;
    for (int v = 0; v < kp.V; ++v) {
        const S360View& vw = views[v];
        const size_t p = (size_t)v * P + g;
        int radius = 0;
        uint32_t touched = 0;
        uint32_t coop_lo = 0, coop_hi = 0;   // COOP: this lane's rectangle, if the wave is to count it (min | max corners, x | y << 16)
        // scale-invariant rescale fused here (cuda_splatting.py:68-69): same f32 products as torch
        const float sc = vw.scale, sc2 = sc * sc;
        const float mx = mx0 * sc, my = my0 * sc, mz = mz0 * sc;
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = c60[k] * sc2;
        // mode-specific projection: (front, cov2D a b c incl. the dilation, pixel centre, sort key)
        bool front;
        float ga = 0.f, gb = 0.f, gc = 0.f, px = 0.f, py = 0.f, zkey = 0.f;
        if (!(kp.flags & S360_FLAG_SPHERICAL)) {
            float pvx, pvy, pvz;
            xform43(vw.viewmatrix, mx, my, mz, pvx, pvy, pvz);
            front = pvz > 0.2f;
            if (front) {
                const float* Pm = vw.projmatrix;
                const float phx = Pm[0] * mx + Pm[4] * my + Pm[8] * mz + Pm[12];
                const float phy = Pm[1] * mx + Pm[5] * my + Pm[9] * mz + Pm[13];
                const float phw = Pm[3] * mx + Pm[7] * my + Pm[11] * mz + Pm[15];
                const float pw = 1.0f / (phw + 0.0000001f);
                const float prx = phx * pw, pry = phy * pw;
                Geo ge;
                geo_compute(vw.viewmatrix, vw.tanfovx, vw.tanfovy, kp.W, kp.H, mx, my, mz, c6, ge);
                ga = ge.a; gb = ge.b; gc = ge.c;
                px = ((prx + 1.0f) * (float)kp.W - 1.0f) * 0.5f;
                py = ((pry + 1.0f) * (float)kp.H - 1.0f) * 0.5f;
                zkey = pvz;
            }
        } else {  // native equirectangular splat (oracle geo_sph); radial distance is the cull and sort quantity
            GeoS gs;
            geo_sph(vw.viewmatrix, kp.W, kp.H, mx, my, mz, c6, gs);
            front = gs.r > 0.2f;
            ga = gs.a; gb = gs.b; gc = gs.c;
            px = gs.u; py = gs.v;
            zkey = gs.r;
        }
        if (front) {
            const float det = ga * gc - gb * gb;
            if (det != 0.0f) {
                const float det_inv = 1.0f / det;
                const float conA = gc * det_inv, conB = -gb * det_inv, conC = ga * det_inv;
                const float mid = 0.5f * (ga + gc);
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lam1 = mid + sq, lam2 = mid - sq;
                const int rad = (int)ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));
                bool keep = true;
                if ((kp.flags & S360_FLAG_SPHERICAL) && (v & 1)) {  // seam ghost: the same splat one panorama width away
                    keep = rad < kp.W / 2;
                    px = px < 0.5f * (float)kp.W ? px + (float)kp.W : px - (float)kp.W;
                }
                int minx, miny, maxx, maxy;
                tile_rect(px, py, rad, kp.gx, kp.gy, minx, miny, maxx, maxy);
                const int area = (keep && act) ? (maxx - minx) * (maxy - miny) : 0;
                if (area != 0) {
                    radius = rad;
                    // conic stored pre-scaled for the composite: exponent in base 2, -1/2 folded in; ka / kb: slopes of the two
                    // parabola-vertex lines of the exact cull (dx* = ka dy maximises the exponent on a row, dy* = kb dx on a column)
                    const float ra = conA * kConicDiag, rb = conB * kConicOff, rc_ = conC * kConicDiag;
                    const float ka = -rb / (2.0f * ra), kb = -rb / (2.0f * rc_);
                    // block-local histogram in LDS; one global atomic per (block, touched tile) below
                    uint32_t* tc = lds_hist == 2 ? hist : (lds_hist ? hist : tile_count) + (size_t)image_of_view(kp, v) * kp.T;
                    uint32_t hmask = 0xFFFFFFFFu;
                    // lean_safe: the bit-identity of the lean lists rests on a 0.02-log2 margin between the tile test
                    // and the quadrants' tests over the float32 rounding of power2().  The cancelling terms a dx^2, b dx dy,
                    // c dy^2 of a thin diagonal splat that spans many tiles reach 1e4 ... 1e5, where that rounding approaches
                    // 1e-2 (at the bound below: <= 1e4 x 4 roundings x 6e-8 = 2.4e-3): such a splat (an axis ratio beyond ~30:1 at a
                    // 5 x 6-tile footprint — none in any workload here) is binned
                    // whole, like the rectangles beyond 32 tiles.  Bound: every |dx|, |dy| inside the rectangle is < rad + 16.
                    const float rext = (float)rad + 16.0f;
                    const bool lean_safe = (fabsf(ra) + fabsf(rb) + fabsf(rc_)) * (rext * rext) < 1.0e4f;
                    if (lean && area <= 32 && lean_safe) {
                        // lean lists: count only the tiles the splat can reach (tile_hit: the composites' cull at tile size) and
                        // remember them as one bit per tile of the rectangle, scan order — k_emit places exactly these, the
                        // backward finds an instance's slot as the rank of its tile's bit.  Larger rectangles are binned whole.
                        const float lop = __builtin_amdgcn_logf(op);
                        uint32_t bit = 1u;
                        hmask = 0u;
                        for (int y = miny; y < maxy; ++y)
                            for (int x = minx; x < maxx; ++x) {
                                if (tile_hit(px, py, ra, rb, rc_, lop, ka, kb, x, y)) {
                                    atomicAdd(&tc[y * kp.gx + x], 1u);
                                    hmask |= bit;
                                }
                                bit <<= 1;
                            }
                        touched = (uint32_t)__builtin_popcount(hmask);
                    } else if (COOP && area > 32) {
                        touched = (uint32_t)area;
                        coop_lo = (uint32_t)minx | ((uint32_t)miny << 16);
                        coop_hi = (uint32_t)maxx | ((uint32_t)maxy << 16);
                    } else {
                        touched = (uint32_t)area;
                        for (int y = miny; y < maxy; ++y)
                            for (int x = minx; x < maxx; ++x) atomicAdd(&tc[y * kp.gx + x], 1u);
                    }
                    if (touched) {   // (lean: a splat too faint to reach any pixel is on no list; its radius is still reported)
                        if (USE_SH && EAGER) {
                            if (!have_rgb) {
                                const float4 cc = rgbc[g];
                                rgb[0] = cc.x; rgb[1] = cc.y; rgb[2] = cc.z;
                                clampbits = __float_as_uint(cc.w);
                                have_rgb = true;
                            }
                        } else if (USE_SH && (!have_rgb || !shared_cam)) {
                            const float dx = mx - vw.campos[0], dy = my - vw.campos[1], dz = mz - vw.campos[2];
                            const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                            const float x = dx * inv, y = dy * inv, z = dz * inv;
                            float Y[25];
                            sh_basis(kp.deg, x, y, z, Y);
                            const int n = (kp.deg + 1) * (kp.deg + 1);
                            const float* sh = shs + (size_t)g * kp.M * 3;
                            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                            // sequential (unfused) accumulation: same rounding as the CPU oracle
                            if (kp.M == 25 && kp.deg == 4 && ch_major) {
                                // one colour channel (25 contiguous floats) at a time: 3x fewer live registers
                                float acc[3];
#pragma unroll 1
                                for (int ch = 0; ch < 3; ++ch) {
                                    float c[25];
                                    load25(sh + 25 * ch, c);
                                    float a = 0.f;
#pragma unroll
                                    for (int k = 0; k < 25; ++k) a += Y[k] * c[k];
                                    acc[ch] = a;
                                }
                                a0 = acc[0]; a1 = acc[1]; a2 = acc[2];
                            } else if (kp.M == 25 && kp.deg == 4) {
                                // interleaved [k][rgb]: five chunks of 5 coefficients x 3 channels (15 floats)
#pragma unroll  // fully unrolled: Y[] indexed by constants stays in registers (a rolled loop put it in scratch)
                                for (int q = 0; q < 5; ++q) {
                                    float c[15];
                                    load15(sh + 15 * q, c);
#pragma unroll
                                    for (int k = 0; k < 5; ++k) {
                                        a0 += Y[5 * q + k] * c[k * 3 + 0];
                                        a1 += Y[5 * q + k] * c[k * 3 + 1];
                                        a2 += Y[5 * q + k] * c[k * 3 + 2];
                                    }
                                }
                            } else if (ch_major) {
                                for (int k = 0; k < n; ++k) {
                                    a0 += Y[k] * sh[k];
                                    a1 += Y[k] * sh[kp.M + k];
                                    a2 += Y[k] * sh[2 * kp.M + k];
                                }
                            } else {
                                for (int k = 0; k < n; ++k) {
                                    a0 += Y[k] * sh[k * 3 + 0];
                                    a1 += Y[k] * sh[k * 3 + 1];
                                    a2 += Y[k] * sh[k * 3 + 2];
                                }
                            }
                            a0 += 0.5f;
                            a1 += 0.5f;
                            a2 += 0.5f;
                            clampbits = (a0 < 0.f ? 1u : 0u) | (a1 < 0.f ? 2u : 0u) | (a2 < 0.f ? 4u : 0u);
                            rgb[0] = fmaxf(a0, 0.f);
                            rgb[1] = fmaxf(a1, 0.f);
                            rgb[2] = fmaxf(a2, 0.f);
                            have_rgb = true;
                        }
                        recA[3 * (size_t)(p)] = make_float4(px, py, ra, rb);
                        recA[3 * (size_t)(p) + 1] = make_float4(rc_, op, rgb[0], rgb[1]);
                        recA[3 * (size_t)(p) + 2] = make_float4(rgb[2], __int_as_float(rad), ka, kb);
                        depths[p] = zkey;
                        clamped[p] = (uint8_t)clampbits;
                        if (lean) slot_info[p].y = hmask;
                    }
                }
            }
        }
        if (COOP) {   // (wave-uniform here: every thread runs the view loop)
            uint32_t* tc = lds_hist == 2 ? hist : (lds_hist ? hist : tile_count) + (size_t)image_of_view(kp, v) * kp.T;
            const int lane = tid & 63;
            for (unsigned long long big = __ballot(coop_hi != 0u); big; big &= big - 1ull) {
                const int src = (int)__builtin_ctzll(big);
                const uint32_t lo = (uint32_t)__shfl((int)coop_lo, src), hi = (uint32_t)__shfl((int)coop_hi, src);
                const int minx = lo & 0xFFFFu, miny = lo >> 16, w = (int)(hi & 0xFFFFu) - minx, n = w * ((int)(hi >> 16) - miny);
                for (int i = lane; i < n; i += 64) {
                    const int y = i / w;
                    atomicAdd(&tc[(miny + y) * kp.gx + minx + (i - y * w)], 1u);
                }
            }
        }
        if (act && radii) radii[p] = radius;
        // visibility of the V (<= 8) views in ONE byte per Gaussian: k_emit and the backward test that instead of V words
        // (24 MB written here and read twice for six views of 1 M Gaussians); tiles_touched only exists for visible pairs
        if (touched) {
            tiles_touched[p] = touched;
            vis |= 1u << v;
        }
        if (lds_hist == 2) {   // block-uniform: this view's counts go out, the histogram is reused by the next view
            __syncthreads();
            uint32_t* tcg = tile_count + (size_t)image_of_view(kp, v) * kp.T;
            for (int i = tid; i < nhist; i += S360_BLOCK) {
                const uint32_t c = hist[i];
                if (c) {
                    atomicAdd(&tcg[i], c);
                    hist[i] = 0u;
                }
            }
            __syncthreads();
        }
    }
    if (act) vis_mask[g] = (uint8_t)vis;
    }
    if (lds_hist == 1) {
        __syncthreads();
        for (int i = tid; i < nhist; i += S360_BLOCK) {
            const uint32_t c = hist[i];
            if (c) atomicAdd(&tile_count[i], c);
        }
    }
}

// ------------------------------------------------------------------------------ scans

template <int NW = S360_BLOCK / 64>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
    // wave-level inclusive scan by shuffles, then 4 wave totals through LDS
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// single block: tile_start[0..nt] = exclusive scan of tile_count; header bookkeeping.
// Lists longer than SORT_SHORT keys are sorted as chunks of SORT_CHUNK keys (k_sort_stage1: one 512-thread
// workgroup per chunk) followed, when there is more than one chunk, by global merge passes (k_merge_all);
// chunk_start[t] = number of such chunks before tile t (0 chunks for the short tiles, which have their own class).
#ifndef S360_SORT_THREADS
#define S360_SORT_THREADS 512   // workgroup of the sort kernels; a chunk is 8 keys per thread
#endif
constexpr int SORT_THREADS = S360_SORT_THREADS;
constexpr uint32_t SORT_SHORT = 2048;
constexpr uint32_t SORT_CHUNK = 8 * SORT_THREADS;
static_assert(SORT_SHORT % SORT_THREADS == 0 && SORT_SHORT <= SORT_CHUNK, "short lists: SORT_SHORT / SORT_THREADS keys per thread");
constexpr uint32_t MAX_PASSES = 4;
__device__ __forceinline__ uint32_t ceil_log2_u32(uint32_t x) { return x <= 1 ? 0u : 32u - (uint32_t)__builtin_clz(x - 1); }
// Lists of 2 .. LDSM_CHUNKS chunks (up to 16 384 keys: every long list of the encoder-like cloud, most of the surface-like one's) are
// finished by ONE workgroup in LDS (merge_tile_lds: 128 KB of keys + padding, two in-LDS merge passes) — ONE write step after the chunk
// sorts, so they count as one pass in the ping-pong parity; longer lists take ceil(log2 chunks) global merge passes.
constexpr uint32_t LDSM_CHUNKS = 4;
constexpr int LDSM_E = 32;   // outputs per thread and pass: SORT_THREADS * LDSM_E == LDSM_CHUNKS * SORT_CHUNK
__device__ __forceinline__ uint32_t merge_passes_of(uint32_t nch) { return nch <= 1 ? 0u : nch <= LDSM_CHUNKS ? 1u : ceil_log2_u32(nch); }

// S360_FLAG_SPLIT_LISTS state (S360Layout part_* / seg_*), by value into the composites
struct SegBufs {
    const uint32_t* chunk_start;  // null: splitting off
    uint32_t* seg_flag;
    uint32_t* seg_arrive;         // phase-1 deliveries per (tile, quadrant)
    uint32_t* seg_arrive2;        // phase-2 deliveries per (tile, quadrant)
    uint32_t n_slots;             // segment slots the workspace holds (quadrants whose segments do not fit are not split)
    float4* part_c;
    float* part_t;
    float* part_e;
    uint32_t* part_l;
    uint32_t* part_n;
    float4* seg_c;
    float* seg_t;
    uint32_t* seg_cnt;
    uint2* seg_info;
    uint32_t* header;
};
static_assert(sizeof(SegBufs) <= (64 - S360_HDR_SEGBUFS) * 4, "SegBufs is parked in the workspace header");

constexpr int TS_BLOCK = 1024;  // one workgroup; 16 waves: 1 536 tiles in two sweeps (a 256-thread block needed six: 10 us of barriers)
__global__ __launch_bounds__(TS_BLOCK) void k_tile_scan(const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_start,
                                                         uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ tile_max_contrib,
                                                         int nt, uint32_t cap, uint32_t* __restrict__ header,
                                                         uint32_t* __restrict__ chunk_start, unsigned long long* __restrict__ header_mirror,
                                                         SegBufs sg_in) {
    __shared__ uint32_t lds[TS_BLOCK / 64];
    __shared__ uint32_t lds_max, lds_nlong;
    if (threadIdx.x == 0) { lds_max = 0; lds_nlong = 0; }
    __syncthreads();
    uint32_t carry = 0, mx = 0, ccarry = 0;
    // two adjacent tiles per thread: the headline's 1 536 tiles are ONE sweep (two block scans, one round of global loads) instead of two
    for (int b = 0; b < nt; b += 2 * TS_BLOCK) {
        const int i0 = b + 2 * (int)threadIdx.x, i1 = i0 + 1;
        uint32_t v0 = 0u, v1 = 0u;
        if (i1 < nt) {
            const uint2 vv = *reinterpret_cast<const uint2*>(tile_count + i0);   // (i0 is even: 8-byte aligned)
            v0 = vv.x; v1 = vv.y;
        } else if (i0 < nt) {
            v0 = tile_count[i0];
        }
        mx = max(mx, max(v0, v1));
        uint32_t tot;
        const uint32_t ex0 = block_exclusive_scan<TS_BLOCK / 64>(v0 + v1, lds, tot), ex1 = ex0 + v0;
        // the sort kernels see list lengths clamped to the binning capacity
        const uint32_t nc0 = min(carry + ex0 + v0, cap) - min(carry + ex0, cap), nc1 = min(carry + ex1 + v1, cap) - min(carry + ex1, cap);
        const uint32_t nch0 = nc0 > SORT_SHORT ? (nc0 + SORT_CHUNK - 1) / SORT_CHUNK : 0u;
        const uint32_t nch1 = nc1 > SORT_SHORT ? (nc1 + SORT_CHUNK - 1) / SORT_CHUNK : 0u;
        if (nch0 || nch1) atomicAdd(&lds_nlong, (nch0 ? 1u : 0u) + (nch1 ? 1u : 0u));
        uint32_t ctot;
        const uint32_t cex0 = block_exclusive_scan<TS_BLOCK / 64>(nch0 + nch1, lds, ctot), cex1 = cex0 + nch0;
        if (i0 < nt) {
            tile_start[i0] = carry + ex0;
            tile_cursor[i0] = 0;
            tile_max_contrib[i0] = 0;
            chunk_start[i0] = ccarry + cex0;
        }
        if (i1 < nt) {
            tile_start[i1] = carry + ex1;
            tile_cursor[i1] = 0;
            tile_max_contrib[i1] = 0;
            chunk_start[i1] = ccarry + cex1;
        }
        carry += tot;
        ccarry += ctot;
    }
    atomicMax(&lds_max, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        chunk_start[nt] = ccarry;
        tile_start[nt] = carry;
        header[0] = carry;                    // num_instances ("num_rendered")
        header[1] = carry > cap ? 1u : 0u;    // overflow flag
        header[2] = lds_max;                  // longest tile list
        {   // merge passes the longest (capacity-clamped) list needs: k_merge_all enters no pass beyond it
            const uint32_t nmax = min(lds_max, cap);
            const uint32_t nch = nmax > SORT_SHORT ? (nmax + SORT_CHUNK - 1) / SORT_CHUNK : 0u;
            header[3] = merge_passes_of(nch);
        }
        header[4] = 0;                        // pairs with more than 32 instance slots (k_emit counts and lists them)
        // S360_FLAG_SPLIT_LISTS: the segment-state pointers, parked in the header for k_render — which takes ONE pointer to them and
        // reads them only on its rare hand-over path (as thirteen by-value kernel arguments they cost k_render, at its 80-VGPR cap,
        // ten more SGPRs spilled into VGPR lanes and a scratch dword: +15 us on the headline)
        *reinterpret_cast<SegBufs*>(header + S360_HDR_SEGBUFS) = sg_in;
        header[S360_HDR_SPLIT] = 0;           // split (tile, quadrant) units (k_render counts them)
        header[6] = header[7] = 0;
        // S360Params.header_mirror: the count and the overflow flag as ONE 64-bit store into host-visible memory — the caller's
        // next call sizes its buffers from it without ever synchronising with the device
        if (header_mirror) __hip_atomic_store(header_mirror, (unsigned long long)carry | ((unsigned long long)(carry > cap ? 1u : 0u) << 32) |
                                                             ((unsigned long long)ccarry << 33),   // + sort chunks of the long lists: sizes max_segments
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        header[S360_HDR_NLONG] = lds_nlong;  // tiles whose lists are sorted as chunks (> SORT_SHORT keys): the first workgroups of k_render<.., SPLIT>
        for (int i = 8; i < 25; ++i) header[i] = 0;  // debug counters; [16, 24): deferred-loss hand-over (S360_HDR_LOSS), set by k_render; [24]: S360_HDR_TILES_DONE
    }
}

// ------------------------------------------------------------------------------ emit
// grid (ceil(P/256), V).  LDS_BIN: block-local two-pass binning — count the block's instances per
// tile in LDS, reserve one contiguous range per (block, tile) with a single returning global
// atomic, then place.  (Order inside a tile's bucket is irrelevant: the bucket is sorted next.)
#ifndef S360_EMIT_PPT
#define S360_EMIT_PPT 4
#endif
constexpr int EMIT_PPT = S360_EMIT_PPT;  // pairs per thread: more instances per (block, tile) => fewer global atomics

// COOP (S360_FLAG_COOP_WALK): pairs of more than 32 tiles are counted and placed by the whole wave, one tile per lane
template <bool LDS_BIN, bool COOP = false>
__global__ __launch_bounds__(S360_BLOCK) void k_emit(KParams kp, const uint32_t* __restrict__ tiles_touched,
                                                    const uint8_t* __restrict__ vis_mask, const float4* __restrict__ recA, const float4* __restrict__ recC,
                                                    const float* __restrict__ depths, const uint32_t* __restrict__ tile_start,
                                                    uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ keys,
                                                    uint2* __restrict__ slot_info, uint32_t* __restrict__ slot_pair,
                                                    uint32_t* __restrict__ slot_ticket, uint32_t* __restrict__ header,
                                                    uint32_t* __restrict__ long_pairs) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_bin[];  // [T] counts/cursors, [T] bases
    __shared__ uint32_t s_scan[S360_BLOCK / 64];
    __shared__ uint32_t s_slot0;
    const int v = blockIdx.y;
    const int g0 = blockIdx.x * (S360_BLOCK * EMIT_PPT) + threadIdx.x;
    const size_t tb = (size_t)image_of_view(kp, v) * kp.T;
    uint32_t* cnt = lds_bin;
    uint32_t* base = lds_bin + kp.T;
    const bool lean = (kp.flags & S360_FLAG_LEAN_LISTS) != 0;
    if (LDS_BIN) {
        for (int i = threadIdx.x; i < kp.T; i += S360_BLOCK) cnt[i] = 0u;
        __syncthreads();
    }
    // The block's visible pairs — one in six of its 1 024 (Gaussian, view) candidates — are first COMPACTED (ballot-free: a
    // block scan of the per-thread counts, pair indices into LDS) and then dealt out again one per thread: every visible pair's
    // record, depth and tile count are then fetched in ONE round of independent gathers and kept in registers for both binning
    // phases, instead of up to four sequential rounds per phase behind the thread that happened to own several visible pairs.
    __shared__ uint32_t s_p[S360_BLOCK * EMIT_PPT];
    uint32_t vis = 0;
#pragma unroll
    for (int j = 0; j < EMIT_PPT; ++j) {
        const int g = g0 + j * S360_BLOCK;
        if (g < kp.P && ((vis_mask[g] >> v) & 1u)) vis |= 1u << j;   // one visibility byte per Gaussian
    }
    uint32_t nvis;
    {
        uint32_t off = block_exclusive_scan((uint32_t)__builtin_popcount(vis), s_scan, nvis);
        for (uint32_t m = vis; m; m &= m - 1) s_p[off++] = (uint32_t)v * (uint32_t)kp.P + (uint32_t)(g0 + __builtin_ctz(m) * S360_BLOCK);
    }
    __syncthreads();
    // this thread's pairs: compact entries tid, tid + 256, ... (usually one)
    // hm: S360_FLAG_LEAN_LISTS — the tiles of the rectangle (scan order, rectangles of up to 32 tiles) that hold an instance,
    // as k_preprocess counted them; all ones otherwise (every tile of the rectangle)
    uint32_t pr[EMIT_PPT], tt[EMIT_PPT], rlo[EMIT_PPT], rhi[EMIT_PPT], dbits[EMIT_PPT], hm[EMIT_PPT];
#pragma unroll
    for (int j = 0; j < EMIT_PPT; ++j) {
        const uint32_t e = threadIdx.x + j * S360_BLOCK;
        tt[j] = 0u; pr[j] = 0u; rlo[j] = rhi[j] = dbits[j] = 0u; hm[j] = 0xFFFFFFFFu;
        if (e < nvis) {
            const uint32_t p = s_p[e];
            const float4 rc = recA[3 * (size_t)p + 2];
            const float4 ra = recA[3 * (size_t)p];
            pr[j] = p;
            tt[j] = tiles_touched[p];
            dbits[j] = __float_as_uint(depths[p]);
            int minx, miny, maxx, maxy;
            tile_rect(ra.x, ra.y, __float_as_int(rc.y), kp.gx, kp.gy, minx, miny, maxx, maxy);
            rlo[j] = (uint32_t)minx | ((uint32_t)miny << 16);
            rhi[j] = (uint32_t)maxx | ((uint32_t)maxy << 16);
            if (lean && (maxx - minx) * (maxy - miny) <= 32) hm[j] = slot_info[p].y;
        }
    }
    // Training calls: the pair's instance slots — where the backward composite leaves its partial gradients, one record
    // per (pair, tile) — are `touched` consecutive slots reserved here: block total -> ONE returning atomic on the
    // image's ticket (its slots start at tile_start[first tile of the image]; one ticket per image, 256 bytes apart:
    // 14 000 returning atomics on a single address cost 20 us), exclusive scan inside the block.  Which block gets which
    // range varies from run to run; nothing depends on it (k_gather_slots sums a pair's slots in slot order = the fixed
    // tile order of its rectangle), and the upstream point_offsets scan over all V*P pairs (a 25 us kernel) is not needed.
    uint32_t sl[EMIT_PPT], ticket = 0;   // first slot of each of this thread's pairs: slots follow the compact (= Gaussian) order
#pragma unroll
    for (int j = 0; j < EMIT_PPT; ++j) sl[j] = 0u;
    if (slot_pair) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < EMIT_PPT; ++j) {
            if ((uint32_t)j * S360_BLOCK >= nvis) break;   // block-uniform: usually one round (a block holds ~175 visible pairs)
            uint32_t tot;
            sl[j] = carry + block_exclusive_scan(tt[j], s_scan, tot);
            carry += tot;
        }
        // issued now, consumed after the counting phase
        if (threadIdx.x == 0 && carry) ticket = tile_start[tb] + atomicAdd(&slot_ticket[image_of_view(kp, v) * 64], carry);
    }
    const int lane = threadIdx.x & 63;
    if (LDS_BIN) {
#pragma unroll
        for (int j = 0; j < EMIT_PPT; ++j) {
            if (COOP && (uint32_t)j * S360_BLOCK >= nvis) break;   // block-uniform
            const bool wide = COOP && tt[j] > 32u;   // (more than 32 instances = a rectangle of more than 32 tiles, binned whole)
            if (tt[j] && !wide) {
                const int minx = rlo[j] & 0xFFFFu, miny = rlo[j] >> 16, maxx = rhi[j] & 0xFFFFu, maxy = rhi[j] >> 16;
                uint32_t m = hm[j];
                for (int y = miny; y < maxy; ++y)
                    for (int x = minx; x < maxx; ++x) {
                        if (m & 1u) atomicAdd(&cnt[y * kp.gx + x], 1u);
                        m = (m >> 1) | 0x80000000u;   // rectangles beyond 32 tiles: all ones
                    }
            }
            if (COOP) {
                for (unsigned long long big = __ballot(wide); big; big &= big - 1ull) {
                    const int src = (int)__builtin_ctzll(big);
                    const uint32_t lo = (uint32_t)__shfl((int)rlo[j], src), hi = (uint32_t)__shfl((int)rhi[j], src);
                    const int minx = lo & 0xFFFFu, miny = lo >> 16, w = (int)(hi & 0xFFFFu) - minx, n = w * ((int)(hi >> 16) - miny);
                    for (int i = lane; i < n; i += 64) {
                        const int y = i / w;
                        atomicAdd(&cnt[(miny + y) * kp.gx + minx + (i - y * w)], 1u);
                    }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kp.T; i += S360_BLOCK) {
            const uint32_t c = cnt[i];
            if (c) {
                base[i] = tile_start[tb + i] + atomicAdd(&tile_cursor[tb + i], c);
                cnt[i] = 0u;
            }
        }
    }
    if (slot_pair || LDS_BIN) {
        if (slot_pair && threadIdx.x == 0) s_slot0 = ticket;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < EMIT_PPT; ++j) {
        if (COOP && (uint32_t)j * S360_BLOCK >= nvis) break;   // block-uniform
        const bool wide = COOP && tt[j] > 32u;
        const uint32_t p = pr[j];
        // owner table of the pair's instance slots (slot = slot_base[p] + rank of the tile among the rectangle's instance-holding
        // tiles in this emission order: its position inside the rectangle when every tile holds one)
        uint32_t slot = 0;
        if (tt[j]) {
            if (slot_pair) {
                slot = sl[j] + s_slot0;
                slot_info[p].x = slot;
            }
            if (tt[j] > 32u) {   // more than 32 slots: summed by a whole wave in the backward (k_gather_slots, second phase);
                                 // the count also tells the caller when S360_FLAG_COOP_WALK pays (header_mirror word 2)
                const uint32_t k = atomicAdd(&header[4], 1u);
                if (slot_pair && k < kp.cap / 32u + 1u) long_pairs[k] = p;
            }
        }
        // (slot_pair: bit 31 marks the slots of a pair with more than 32 of them — k_gather_slots' slot-parallel pass skips those)
        const uint32_t ptag = p | (tt[j] > 32u ? 0x80000000u : 0u);
        if (tt[j] && !wide) {
            const int minx = rlo[j] & 0xFFFFu, miny = rlo[j] >> 16, maxx = rhi[j] & 0xFFFFu, maxy = rhi[j] >> 16;
            const uint64_t key = ((uint64_t)dbits[j] << 32) | (uint64_t)p;
            uint32_t m = hm[j];
            for (int y = miny; y < maxy; ++y)
                for (int x = minx; x < maxx; ++x) {
                    if (m & 1u) {
                        const int t = y * kp.gx + x;
                        const uint32_t pos = LDS_BIN ? base[t] + atomicAdd(&cnt[t], 1u)
                                                     : tile_start[tb + t] + atomicAdd(&tile_cursor[tb + t], 1u);
                        if (pos < kp.cap) keys[pos] = key;
                        if (slot_pair) {
                            if (slot < kp.cap) slot_pair[slot] = ptag;
                            ++slot;
                        }
                    }
                    m = (m >> 1) | 0x80000000u;
                }
        }
        if (COOP) {
            for (unsigned long long big = __ballot(wide); big; big &= big - 1ull) {
                const int src = (int)__builtin_ctzll(big);
                const uint32_t lo = (uint32_t)__shfl((int)rlo[j], src), hi = (uint32_t)__shfl((int)rhi[j], src);
                const uint32_t ps = (uint32_t)__shfl((int)p, src), ds = (uint32_t)__shfl((int)dbits[j], src), s0 = (uint32_t)__shfl((int)slot, src);
                const uint64_t key = ((uint64_t)ds << 32) | (uint64_t)ps;
                const int minx = lo & 0xFFFFu, miny = lo >> 16, w = (int)(hi & 0xFFFFu) - minx, n = w * ((int)(hi >> 16) - miny);
                for (int i = lane; i < n; i += 64) {   // tile i of the rectangle in scan order = instance slot s0 + i
                    const int y = i / w;
                    const int t = (miny + y) * kp.gx + minx + (i - y * w);
                    const uint32_t pos = LDS_BIN ? base[t] + atomicAdd(&cnt[t], 1u)
                                                 : tile_start[tb + t] + atomicAdd(&tile_cursor[tb + t], 1u);
                    if (pos < kp.cap) keys[pos] = key;
                    if (slot_pair && s0 + (uint32_t)i < kp.cap) slot_pair[s0 + (uint32_t)i] = ps | 0x80000000u;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ per-tile sort
// Merge sort of one tile's bucket in LDS: every thread sorts E keys in registers (odd-even transposition
// network), then log2(THREADS) merge passes; in each pass a thread finds its slice of the two runs being
// merged with a merge-path binary search and merges E outputs serially.  ~2 barriers and ~(2E + log n)
// LDS accesses per thread per pass — against log^2(n)/2 barriers and 4 accesses per compare-exchange of
// the bitonic network.  Keys are unique, so the result is the one ascending order (== stable radix sort
// by (tile, depth), index-ordered emission).  Measured alternatives that were slower on this workload:
// 4-ary merge-path search + in-register bitonic merge of 2E candidates (more instructions; the kernel is
// issue-bound on 64-bit compares/selects, not LDS-latency-bound), and an LDS radix sort.
template <int THREADS, int E>
__device__ __forceinline__ void block_merge_sort(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                 uint32_t* __restrict__ list, uint32_t n, uint64_t* lds_m) {
    // [CAP + CAP/E]: one pad slot per E keys.  Threads walk the runs with a stride of ~E (or ~E/2) keys;
    // without the skew those 64 / 128-byte strides land on 4 / 2 bank groups (16- / 32-way conflicts).
#define S360_PHYS(i) ((i) + (i) / E)
    const int tid = threadIdx.x;
    uint64_t k[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const uint32_t i = (uint32_t)(tid * E + q);
        k[q] = i < n ? in[i] : ~0ull;
    }
    // in-register sort of the thread's E keys
#pragma unroll
    for (int r = 0; r < E; ++r) {
#pragma unroll
        for (int q = (r & 1); q + 1 < E; q += 2) {
            const uint64_t a = k[q], b = k[q + 1];
            k[q] = a < b ? a : b;
            k[q + 1] = a < b ? b : a;
        }
    }
#pragma unroll
    for (int q = 0; q < E; ++q) lds_m[S360_PHYS((uint32_t)(tid * E + q))] = k[q];
    // While a pair of runs (2 * width keys) lies inside the 64 * E keys of ONE wave, only that wave reads and writes them: the
    // workgroup barrier of those passes (6 of the 9 for a full list, each waiting for the slowest of 8 waves) is replaced by
    // wave-level ordering — DS operations of a wave execute in order; the fence keeps the compiler from moving them.
#define S360_SORT_SYNC(w)                                                      \
    do {                                                                       \
        if (2u * (w) <= 64u * E) {                                             \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             \
            __builtin_amdgcn_wave_barrier();                                   \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");             \
        } else {                                                               \
            __syncthreads();                                                   \
        }                                                                      \
    } while (0)
    S360_SORT_SYNC(E);
    // runs of length `width` are sorted; merge neighbouring pairs until one run remains.  Runs that lie
    // entirely in the padding (start >= n) never need merging, so stop once width covers n.
    uint32_t npad = E;
    while (npad < n) npad <<= 1;
    // A thread whose E outputs all lie at or beyond n would merge padding into padding: every position >= n holds +inf from the
    // load above and keeps it (the real keys of a pair sort in front of its padding), so such a thread only takes part in the
    // barriers — whole waves of a 1 456-key list (the mean) in a 2 048-key capacity skip the search and the merge: a quarter of the
    // stage's instructions.
    const uint32_t out0 = (uint32_t)tid * E;                      // first output element of this thread
    const bool live = out0 < n;
    for (uint32_t width = E; width < npad; width <<= 1) {
        const uint32_t pair = out0 / (2 * width) * (2 * width);   // start of the run pair
        const uint32_t pa = pair, pb = pair + width;              // run starts (logical indices)
#define S360_A(x) lds_m[S360_PHYS(pa + (x))]
#define S360_B(x) lds_m[S360_PHYS(pb + (x))]
        if (live) {
            const uint32_t diag = out0 - pair;
            uint32_t lo_a = diag > width ? diag - width : 0u, hi_a = diag < width ? diag : width;
            while (lo_a < hi_a) {  // merge path: first a with A[a] > B[diag-1-a]
                const uint32_t mid = (lo_a + hi_a) >> 1;
                if (S360_A(mid) <= S360_B(diag - 1 - mid)) lo_a = mid + 1; else hi_a = mid;
            }
            uint32_t a = lo_a, b = diag - lo_a;
            uint64_t ka = a < width ? S360_A(a) : ~0ull, kb = b < width ? S360_B(b) : ~0ull;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const bool take_a = ka <= kb;
                k[q] = take_a ? ka : kb;
                if (take_a) {
                    ++a;
                    ka = a < width ? S360_A(a) : ~0ull;
                } else {
                    ++b;
                    kb = b < width ? S360_B(b) : ~0ull;
                }
            }
        }
        S360_SORT_SYNC(width);          // every read of this pass precedes its writes
        if (live) {
#pragma unroll
            for (int q = 0; q < E; ++q) lds_m[S360_PHYS(out0 + q)] = k[q];
        }
        S360_SORT_SYNC(2u * width);     // ... which the NEXT pass (runs of 2 * width, pairs of 4 * width) reads
#undef S360_A
#undef S360_B
    }
    __syncthreads();
#undef S360_SORT_SYNC
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const uint32_t i = (uint32_t)(tid * E + q);
        if (i < n) {
            const uint64_t kq = lds_m[S360_PHYS(i)];
            out[i] = kq;
            if (list) list[i] = (uint32_t)kq;
        }
    }
#undef S360_PHYS
}

// ---- long lists (n > SORT_CHUNK): chunk sort + global merge passes ------------------------------------------
// Work unit = one SORT_CHUNK-sized output chunk (t, k) of a long tile, found from the block index by a binary
// search over chunk_start[].  A tile with c chunks needs P = ceil(log2 c) merge passes; it ping-pongs between
// `keys` and `alt` such that the LAST pass lands in `keys`: the buffer holding the runs before pass i is
// `alt` when (P - i) is odd.  Tiles needing more than max_passes passes are left to the global-memory network (sort_tiles_global_body).


struct ChunkUnit {
    uint32_t s, n, k, passes, t;  // tile start (clamped), tile length, chunk index inside the tile, merge passes of the tile, tile
    bool valid;
};
__device__ __forceinline__ ChunkUnit chunk_unit(const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ chunk_start,
                                                 int nt, uint32_t cap, uint32_t b) {
    ChunkUnit u;
    u.valid = b < chunk_start[nt];
    u.s = u.n = u.k = u.passes = u.t = 0;
    if (!u.valid) return u;
    // last t with chunk_start[t] <= b (chunk_start is non-decreasing, chunk_start[0] = 0): 64-ary search by the whole
    // wave — 2 dependent global round trips for up to 4 096 tiles instead of the 12 of a per-thread binary search
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = nt;
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;
        const int t = lo + lane * step;
        const bool p = t < hi && chunk_start[t] <= b;
        const int cnt = __popcll(__ballot(p));  // >= 1: the predicate holds at lo
        const int nhi = lo + cnt * step;
        lo = lo + (cnt - 1) * step;
        hi = nhi < hi ? nhi : hi;
    }
    u.s = min(tile_start[lo], cap);
    u.n = min(tile_start[lo + 1], cap) - u.s;
    u.k = b - chunk_start[lo];
    u.t = (uint32_t)lo;
    u.passes = merge_passes_of((u.n + SORT_CHUNK - 1) / SORT_CHUNK);
    return u;
}

// ONE launch for the first sorting stage: chunk blocks first in the grid (their results feed the merge passes, the critical
// path), then one block per tile for the short lists, and a last block that deals the composite's tile order — round 1
// ran the short-list sort and the ordering on a process-wide side stream (fork / join events, a host mutex): ~25 us of
// cross-stream latency inside a 100-us stage, and library-global state.
__global__ __launch_bounds__(SORT_THREADS) void k_sort_stage1(const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ chunk_start,
                                                    int nt, uint64_t* __restrict__ keys, uint64_t* __restrict__ alt,
                                                    uint32_t* __restrict__ list, uint32_t cap, uint32_t max_passes, uint32_t cgrid,
                                                    const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_order,
                                                    const uint32_t* __restrict__ header, unsigned long long* __restrict__ wide_mirror) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_m[];
    const uint32_t bid = blockIdx.x;
    if (bid < cgrid) {  // 4 096-key chunks of the long lists, grid-stride over the chunk table
        const uint32_t nchunks = chunk_start[nt];
        for (uint32_t b = bid; b < nchunks; b += cgrid) {
            const ChunkUnit u = chunk_unit(tile_start, chunk_start, nt, cap, b);
            if (u.valid && u.passes <= max_passes) {
                const uint32_t c0 = u.k * SORT_CHUNK, len = min(SORT_CHUNK, u.n - c0);
                uint64_t* dst = ((u.passes & 1u) ? alt : keys) + u.s + c0;
                block_merge_sort<SORT_THREADS, 8>(keys + u.s + c0, dst, u.passes == 0 ? list + u.s : nullptr, len, lds_m);
            }
            __syncthreads();
        }
        return;
    }
    if (bid < cgrid + (uint32_t)nt) {  // lists of up to 2 048 keys: one workgroup each
        const uint32_t tile = bid - cgrid;
        const uint32_t s = min(tile_start[tile], cap), e = min(tile_start[tile + 1], cap);
        const uint32_t n = e - s;
        if (n == 0 || n > SORT_SHORT) return;
        block_merge_sort<SORT_THREADS, (int)(SORT_SHORT / SORT_THREADS)>(keys + s, keys + s, list + s, n, lds_m);
        return;
    }
    // S360Params.header_mirror word 2: (Gaussian, view) pairs binned over more than 32 tiles (k_emit, the previous launch, counted them)
    if (wide_mirror && threadIdx.x == 0) __hip_atomic_store(wide_mirror, (unsigned long long)header[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tile_order) order_units_body<SORT_THREADS>(tile_count, tile_order, nt);  // longest list first (dispatch order of k_render)
}

// Merge path over two sorted runs in global memory: number of A elements among the first d outputs.  Executed by
// one whole wave as a 64-ary search (64 probes per round, the ballot of the monotone predicate locates the
// boundary): 2-3 dependent global-memory round trips instead of the ~13 of a binary search.
// Device-coherent 64-bit accesses (sc1: served by / written through to memory, not a per-XCD L2 line): what the persistent merge
// kernel uses for the runs one workgroup writes and another — possibly behind a different XCD's L2 — reads in the same launch.
// (Agent-scope release / acquire FENCES instead write back / invalidate the whole 4-MB L2: 1 024 workgroups doing that made the
// merge stage 75 us slower than separate launches.)
__device__ __forceinline__ uint64_t ld_dev(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t merge_path_global_wave(const uint64_t* __restrict__ A, uint32_t la,
                                                           const uint64_t* __restrict__ B, uint32_t lb, uint32_t d, int lane) {
    uint32_t lo = d > lb ? d - lb : 0u, hi = d < la ? d : la;  // answer in [lo, hi]; pred(a) := A[a] <= B[d-1-a]  (answer > a)
    while (lo < hi) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t a = lo + (uint32_t)lane * step;
        const bool p = a < hi && ld_dev(A + a) <= ld_dev(B + (d - 1 - a));
        const uint32_t cnt = (uint32_t)__popcll(__ballot(p));  // true for a prefix of the probes
        if (cnt == 0) {
            hi = lo;
        } else {
            const uint32_t nhi = lo + cnt * step;
            lo = lo + (cnt - 1) * step + 1;
            hi = nhi < hi ? nhi : hi;
        }
    }
    return lo;
}

// ONE launch for everything after the chunk sorts (round 3: four k_merge_pass launches + k_sort_tiles_global, 40 us of which
// two and a half launches found nothing to do).  Work units = (pass, chunk) in pass-major order, handed out by a ticket counter
// to whichever workgroup is free; a unit of pass p >= 1 waits until ALL chunks of its tile have finished pass p - 1
// (merge_done[tile][p-1], one device-scope counter per (tile, pass); the runs themselves are written and read with
// device-coherent accesses, so no cache-wide fence is needed: stores complete, workgroup barrier, one relaxed atomic).  A unit only ever waits for units with LOWER tickets, and a ticket is only taken by
// a workgroup that is running, so the schedule cannot deadlock whatever share of the grid is resident.  Passes beyond header[3]
// (what the longest list of THIS call needs) get no tickets; lists beyond SORT_CHUNK << max_passes keys fall through to the
// global-memory network at the end of the same launch.
#ifndef S360_MERGE_GRID
#define S360_MERGE_GRID 256   // one workgroup per CU: the kernel's 135 KB of LDS (merge_tile_lds) admit no more, and workgroups beyond
                              // the resident ones would each be dispatched, take one void ticket and exit — three more rounds (1 024: +9 us)
#endif
__device__ __forceinline__ void merge_unit(const ChunkUnit& u, uint32_t pass, const uint64_t* __restrict__ keys_c, uint64_t* __restrict__ keys,
                                           uint64_t* __restrict__ alt, uint32_t* __restrict__ list, uint64_t* lds_m, uint32_t* s_part) {
    constexpr int THREADS = SORT_THREADS, E = 8;
#define S360_PHYS(i) ((i) + (i) / E)
    (void)keys_c;
    const uint32_t R = SORT_CHUNK << pass;
    const uint32_t o_tile = u.k * SORT_CHUNK, len = min(SORT_CHUNK, u.n - o_tile);
    const uint32_t pair0 = o_tile / (2 * R) * (2 * R);
    const uint32_t la = min(R, u.n - pair0), lb = min(R, u.n - pair0 - la);
    const uint64_t* src = (((u.passes - pass) & 1u) ? alt : keys) + u.s + pair0;
    uint64_t* dst = (((u.passes - pass - 1) & 1u) ? alt : keys) + u.s + o_tile;
    const uint64_t* A = src;
    const uint64_t* B = src + la;
    const uint32_t o = o_tile - pair0;
    if (threadIdx.x < 128) {  // wave 0: start of this block's output range, wave 1: its end
        const int w = threadIdx.x >> 6;
        const uint32_t r = merge_path_global_wave(A, la, B, lb, o + (uint32_t)w * len, threadIdx.x & 63);
        if ((threadIdx.x & 63) == 0) s_part[w] = r;
    }
    __syncthreads();
    const uint32_t a0 = s_part[0], a1 = s_part[1], b0 = o - a0, b1 = o + len - a1;
    const uint32_t na = a1 - a0, nb = b1 - b0;  // na + nb == len
    {   // the thread's E device-coherent loads in flight together (as a load-then-store loop: E dependent round trips)
        uint64_t kin[E];
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const uint32_t i = (uint32_t)threadIdx.x + (uint32_t)q * THREADS;
            kin[q] = i < len ? ld_dev(i < na ? A + (a0 + i) : B + (b0 + (i - na))) : 0ull;
        }
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const uint32_t i = (uint32_t)threadIdx.x + (uint32_t)q * THREADS;
            if (i < len) lds_m[S360_PHYS(i)] = kin[q];
        }
    }
    __syncthreads();
    // in-LDS merge of the two pieces: logical run A = [0, na), run B = [na, na + nb)
    const uint32_t out0 = (uint32_t)threadIdx.x * E;
    if (out0 < len) {
        uint32_t lo_a = out0 > nb ? out0 - nb : 0u, hi_a = out0 < na ? out0 : na;
        while (lo_a < hi_a) {
            const uint32_t mid = (lo_a + hi_a) >> 1;
            if (lds_m[S360_PHYS(mid)] <= lds_m[S360_PHYS(na + out0 - 1 - mid)]) lo_a = mid + 1; else hi_a = mid;
        }
        uint32_t a = lo_a, b = out0 - lo_a;
        uint64_t ka = a < na ? lds_m[S360_PHYS(a)] : ~0ull, kb = b < nb ? lds_m[S360_PHYS(na + b)] : ~0ull;
        const bool last_pass = pass + 1 == u.passes;
        uint32_t* lst = list + u.s + o_tile;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const bool take_a = ka <= kb;
            const uint64_t kq = take_a ? ka : kb;
            if (take_a) {
                ++a;
                ka = a < na ? lds_m[S360_PHYS(a)] : ~0ull;
            } else {
                ++b;
                kb = b < nb ? lds_m[S360_PHYS(na + b)] : ~0ull;
            }
            if (out0 + q < len) {
                st_dev(dst + (out0 + q), kq);
                if (last_pass) lst[out0 + q] = (uint32_t)kq;
            }
        }
    }
#undef S360_PHYS
}

// Whole-list merge in LDS for a tile of 2 .. LDSM_CHUNKS sorted chunks (n <= 16 384 keys): ONE workgroup loads the chunk-sorted runs
// (written by k_sort_stage1, the previous launch: plain loads), merges neighbouring runs in LDS until one remains — every thread finds
// its LDSM_E consecutive outputs with a merge-path binary search and merges them serially into registers, barrier, write back — and
// stores the sorted keys and the list.  Replaces two global merge passes (each: two 64-ary searches with device-coherent loads, a
// 4 096-key LDS merge, write-through stores, a drain and a counter another workgroup polls: ~15 us a pass, 3 % VALU-busy) by ~10 us of
// one workgroup's time with no cross-workgroup traffic at all.  Runs may be ragged (the last one short or missing).
static_assert(SORT_THREADS * LDSM_E == (int)(LDSM_CHUNKS * SORT_CHUNK), "a thread merges LDSM_E outputs per pass");
constexpr size_t LDSM_LDS_BYTES = (size_t)(LDSM_CHUNKS * SORT_CHUNK + LDSM_CHUNKS * SORT_CHUNK / LDSM_E) * 8;   // one pad slot per LDSM_E keys
__device__ __forceinline__ void merge_tile_lds(uint32_t n, const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t* __restrict__ lst,
                                               uint64_t* lds_m) {
    constexpr int E = LDSM_E;
#define S360_PHYS(i) ((i) + (i) / E)
    {   // all of a thread's (up to E) loads in flight at once: a load-then-store loop pays the memory latency per iteration (27 x ~0.7 us)
        uint64_t k[E];
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const uint32_t i = (uint32_t)threadIdx.x + (uint32_t)q * SORT_THREADS;
            k[q] = i < n ? src[i] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const uint32_t i = (uint32_t)threadIdx.x + (uint32_t)q * SORT_THREADS;
            if (i < n) lds_m[S360_PHYS(i)] = k[q];
        }
    }
    __syncthreads();
    const uint32_t out0 = (uint32_t)threadIdx.x * E;
    for (uint32_t width = SORT_CHUNK; width < n; width <<= 1) {
        uint64_t k[E];
        if (out0 < n) {   // (threads past the end only take part in the barriers)
            const uint32_t pa = out0 / (2 * width) * (2 * width), pb = pa + width;
            const uint32_t la = min(width, n - pa), lb = pb < n ? min(width, n - pb) : 0u;
            const uint32_t diag = out0 - pa;
            uint32_t lo = diag > lb ? diag - lb : 0u, hi = diag < la ? diag : la;
            while (lo < hi) {   // merge path: first a with A[a] > B[diag - 1 - a]
                const uint32_t mid = (lo + hi) >> 1;
                if (lds_m[S360_PHYS(pa + mid)] <= lds_m[S360_PHYS(pb + (diag - 1 - mid))]) lo = mid + 1; else hi = mid;
            }
            uint32_t a = lo, b = diag - lo;
            uint64_t ka = a < la ? lds_m[S360_PHYS(pa + a)] : ~0ull, kb = b < lb ? lds_m[S360_PHYS(pb + b)] : ~0ull;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const bool take_a = ka <= kb;
                k[q] = take_a ? ka : kb;
                if (take_a) {
                    ++a;
                    ka = a < la ? lds_m[S360_PHYS(pa + a)] : ~0ull;
                } else {
                    ++b;
                    kb = b < lb ? lds_m[S360_PHYS(pb + b)] : ~0ull;
                }
            }
        }
        __syncthreads();          // every read of this pass precedes its writes
        if (out0 < n) {
#pragma unroll
            for (int q = 0; q < E; ++q)
                if (out0 + q < n) lds_m[S360_PHYS(out0 + q)] = k[q];
        }
        __syncthreads();
    }
    {   // (LDS reads batched like the loads above: as a read-then-store loop every iteration waits for its own LDS read)
        uint64_t k[E];
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const uint32_t i = (uint32_t)threadIdx.x + (uint32_t)q * SORT_THREADS;
            k[q] = i < n ? lds_m[S360_PHYS(i)] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const uint32_t i = (uint32_t)threadIdx.x + (uint32_t)q * SORT_THREADS;
            if (i < n) {
                dst[i] = k[q];
                lst[i] = (uint32_t)k[q];
            }
        }
    }
#undef S360_PHYS
}

// Fallback for tile lists beyond the merge-pass budget (> SORT_CHUNK << MAX_PASSES keys): one workgroup sorts directly in global memory with
// the ascending-only form of the bitonic network (first sub-step of every stage compares mirrored
// positions), which tolerates VIRTUAL +inf padding at indices >= n: an ascending compare-exchange never
// moves a padding key inwards, so nothing outside [0, n) is ever read or written.  Rare and slow.
template <int THREADS>
__device__ __forceinline__ void sort_tiles_global_body(const uint32_t* __restrict__ tile_start, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ list, uint32_t lo, uint32_t cap, int nt) {
  for (int tile = blockIdx.x; tile < nt; tile += gridDim.x) {   // grid-stride over the tiles: almost always nothing to do
    const uint32_t s = min(tile_start[tile], cap), e = min(tile_start[tile + 1], cap);
    const uint32_t n = e - s;
    if (n <= lo) continue;   // block-uniform
    uint32_t npad = 1;
    while (npad < n) npad <<= 1;
    uint64_t* kk = keys + s;
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        const uint32_t half = k >> 1;
        for (uint32_t t = threadIdx.x; t < (npad >> 1); t += THREADS) {
            const uint32_t blk = t / half, off = t - blk * half;
            const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < n) {
                const uint64_t a = kk[i], b = kk[l];
                if (a > b) {
                    kk[i] = b;
                    kk[l] = a;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (npad >> 1); t += THREADS) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                if (l < n) {
                    const uint64_t a = kk[i], b = kk[l];
                    if (a > b) {
                        kk[i] = b;
                        kk[l] = a;
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) list[s + i] = (uint32_t)kk[i];
    __syncthreads();
  }
}

__global__ __launch_bounds__(SORT_THREADS) void k_merge_all(const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ chunk_start,
                                                  int nt, uint64_t* __restrict__ keys, uint64_t* __restrict__ alt,
                                                  uint32_t* __restrict__ list, uint32_t cap, uint32_t max_passes,
                                                  const uint32_t* __restrict__ header, uint32_t* __restrict__ merge_done,
                                                  uint32_t global_lo) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_m[];  // LDSM_LDS_BYTES: [SORT_CHUNK + SORT_CHUNK/E] skewed for the pair merges, the whole list for merge_tile_lds
    __shared__ uint32_t s_part[2];
    __shared__ uint32_t s_ticket;
    const uint32_t npass = min(header[3], max_passes);   // merge passes the longest (capacity-clamped) list of this call needs
    const uint32_t nchunks = npass ? chunk_start[nt] : 0u;
    const uint32_t nunits = npass * nchunks;
    uint32_t* const queue = merge_done + (size_t)nt * MAX_PASSES;   // the work queue's ticket counter (cleared with the counters)
    for (;;) {
        if (threadIdx.x == 0) s_ticket = nunits ? __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __syncthreads();
        const uint32_t tk = s_ticket;
        __syncthreads();
        if (tk >= nunits) break;
        const uint32_t pass = tk / nchunks, blk = tk - pass * nchunks;
        const ChunkUnit u = chunk_unit(tile_start, chunk_start, nt, cap, blk);
        if (!u.valid || u.passes > max_passes || pass >= u.passes) continue;  // block-uniform
        const uint32_t nch = (u.n + SORT_CHUNK - 1) / SORT_CHUNK;
        if (nch <= LDSM_CHUNKS) {   // (one pass by merge_passes_of: only pass 0 gets here) the whole list, by the unit of its first chunk
            if (u.k == 0) merge_tile_lds(u.n, alt + u.s, keys + u.s, list + u.s, lds_m);
            continue;               // (the barrier behind the next ticket separates this unit's LDS reads from the next one's writes)
        }
        if (pass > 0) {
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(&merge_done[(size_t)u.t * MAX_PASSES + pass - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nch)
                    __builtin_amdgcn_s_sleep(4);
            }
            __syncthreads();
        }
        merge_unit(u, pass, keys, keys, alt, list, lds_m, s_part);   // runs read and written with device-coherent accesses
        // Every thread DRAINS its own write-through stores (s_waitcnt vmcnt(0): a store's counter is released when the memory side
        // has acknowledged it) before the workgroup barrier, and only then does thread 0 publish the pass.  A workgroup-scope
        // release fence alone compiles to `s_waitcnt lgkmcnt(0)` on gfx950 — the sc1 global stores were not waited on, and a
        // workgroup behind another XCD could read a run before it had landed.  No cache-wide write-back is
        // needed: the stores are device-coherent themselves.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();                                              // (also: LDS and s_part are reused next)
        if (threadIdx.x == 0 && pass + 1 < u.passes)
            __hip_atomic_fetch_add(&merge_done[(size_t)u.t * MAX_PASSES + pass], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (header[2] > global_lo) sort_tiles_global_body<SORT_THREADS>(tile_start, keys, list, global_lo, cap, nt);
}

// ------------------------------------------------------------------------------ composite
// Wave-autonomous quadrants.  A workgroup is the four 8x8 pixel quadrants of one 16x16 tile, but the
// four waves never synchronise: each wave walks the tile's depth-sorted list on its own, 64 entries
// at a time, with lane l fetching entry l straight into registers (list index prefetched two
// chunks ahead, splat records one chunk ahead).  The wave's register file is the broadcast source:
//   * cull   : every lane tests ITS entry against the quadrant with the exact cull half-extents;
//              the ballot is the work list (culled entries fail alpha >= 1/255 on every pixel of
//              the quadrant, so skipping them changes nothing);
//   * dense  : surviving entries are broadcast one at a time with v_readlane (SGPR operands, no
//              LDS, no exec-mask juggling) while the 64 pixels sit in the lanes;
//   * sparse : once <= SPARSE_PIXELS pixels of the quadrant are still unsaturated the roles flip —
//              64 ENTRIES in the lanes, one live pixel per iteration, and only entries passing the
//              tests are replayed in list order.  Same arithmetic per (pixel, entry) pair.
// A wave retires as soon as its own 64 pixels are saturated (no tile-wide barrier to wait for).
#ifndef S360_SPARSE_PIXELS
#define S360_SPARSE_PIXELS 6
#endif
constexpr int SPARSE_PIXELS = S360_SPARSE_PIXELS;

// Optional loss epilogue of the composite store (the reference computes it as separate torch ops right after
// the decoder: LossMse, src/loss/loss_mse.py:30-31; compute_psnr, src/evaluation/metrics.py:11-21).
struct MseEp {
    const float* target;  // [V,3,H,W] or null (epilogue off)
    float* d_images;      // [V,3,H,W]  grad_scale * (image - target)
    float* partials;      // [V*T*4, 2] per 8x8 quadrant: sum (image-target)^2, sum (clip01(image)-clip01(target))^2
    float grad_scale;
    float* loss_out;      // [1 + V] or null: (grad_scale / 2) * sum of all plain partials, per-view clipped MSE
};

// Final reduction of the loss epilogue in one launch (torch needs four: two reductions and two scalings).  One workgroup;
// fixed assignment of partials to threads, fixed shuffle tree, fixed wave order: deterministic.  Its ~9 us are a chain of
// ~2-us memory round trips after a launch (the partials were just written behind other XCDs' L2s); running the same
// reduction in the LAST workgroup of k_render instead (write-through partials, one device-scope count per workgroup) measured
// exactly the same 9 us at the end of k_render — built, parity-green, not kept.
__global__ __launch_bounds__(MSE_BLOCK) void k_mse_finish(const float* __restrict__ partials, int n_per_view, int V, float loss_scale,
                                                         float inv_elems, float* __restrict__ out) {
    mse_finish_body(partials, n_per_view, V, loss_scale, inv_elems, out);
}


// ---- shared by k_render's phase-1 workers and k_render_tail (S360_FLAG_SPLIT_LISTS; see k_render_tail below)
__device__ __forceinline__ void st_dev32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_dev32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_devf(float* p, float v) { st_dev32(reinterpret_cast<uint32_t*>(p), __float_as_uint(v)); }
__device__ __forceinline__ float ld_devf(const float* p) { return __uint_as_float(ld_dev32(reinterpret_cast<const uint32_t*>(p))); }

__device__ __forceinline__ void st_dev64(uint2* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld_dev64(const uint2* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ float4 ld_dev_f4(const float4* p) {   // one device-coherent 16-byte load
    f4v r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\ns_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void st_dev_f4(float4* p, float4 v) {
    float* q = reinterpret_cast<float*>(p);
    st_devf(q, v.x); st_devf(q + 1, v.y); st_devf(q + 2, v.z); st_devf(q + 3, v.w);
}

struct WaveLds {   // one wave's slices of the compaction arrays (see k_render)
    float *x, *y, *a, *b, *c, *o;
    float2 *rg, *bz;
    uint32_t* pos;
};

// k_render's chunk loop over list positions [b0, b1) of the tile that starts at `start` (b0 - start a multiple of 64), on the
// running per-pixel state (T, C01, C2D, last, done); sv != null: survivor records appended at sv[3 (scount + rank)].
// T_ONLY: transmittance and stop logic alone (phase 1).
template <bool WITH_DEPTH, bool T_ONLY>
__device__ __forceinline__ void composite_segment(const uint32_t* __restrict__ list, const float4* __restrict__ recA, const float* __restrict__ depths,
                                                  uint32_t start, uint32_t b0, uint32_t b1, float pxf, float pyf, float x0, float ys0, int lane,
                                                  const WaveLds& L, float inv_scale, float v_near, float v_far, int depth_mode, float& T, f2& C01,
                                                  f2& C2D, uint32_t& last, bool& done, float4* sv, uint32_t& scount) {
    uint32_t p_n1 = 0, p_n2 = 0;
    if (b0 + lane < b1) p_n1 = list[b0 + lane];
    if (b0 + 64 + lane < b1) p_n2 = list[b0 + 64 + lane];
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, nc = na;
    float nz = 0.f;
    if (b0 + lane < b1) {
        na = recA[3 * (size_t)(p_n1)];
        nb = recA[3 * (size_t)(p_n1) + 1];
        nc = recA[3 * (size_t)(p_n1) + 2];
        if (WITH_DEPTH && !T_ONLY) nz = depths[p_n1];
    }
    for (uint32_t b = b0; b < b1; b += 64) {
        const unsigned long long act = __ballot(!done);
        if (act == 0ull) break;
        const float4 ea = na, eb = nb;
        float ez = 0.f;
        if (WITH_DEPTH && !T_ONLY) ez = depth_value(nz * inv_scale, v_near, v_far, depth_mode);
        const float ec = nc.x, erad = nc.y, eka = nc.z, ekb = nc.w;
        const bool ev = b + lane < b1;
        const uint32_t epair = p_n1;
        p_n1 = p_n2;
        if (b + 64 + lane < b1) {
            na = recA[3 * (size_t)(p_n1)];
            nb = recA[3 * (size_t)(p_n1) + 1];
            nc = recA[3 * (size_t)(p_n1) + 2];
            if (WITH_DEPTH && !T_ONLY) nz = depths[p_n1];
        }
        if (b + 128 + lane < b1) p_n2 = list[b + 128 + lane];
        const bool hit = ev && quadrant_hit(ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eka, ekb, x0, ys0);
        const unsigned long long m = __ballot(hit);
        if (m == 0ull) continue;
        const uint32_t rel = b - start;  // list position of this chunk's lane 0
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (!T_ONLY && sv && hit) {
            float4* o = sv + 3 * (size_t)(scount + rank);
            o[0] = ea;
            o[1] = eb;
            o[2] = make_float4(ec, erad, __uint_as_float(rel + (uint32_t)lane), __uint_as_float(epair));
        }
        scount += (uint32_t)__popcll(m);
        if (__popcll(act) > SPARSE_PIXELS) {
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (hit) {
                L.x[rank] = ea.x; L.y[rank] = ea.y; L.a[rank] = ea.z; L.b[rank] = ea.w;
                L.c[rank] = eb.x; L.o[rank] = eb.y;
                if (!T_ONLY) {
                    L.rg[rank] = make_float2(eb.z, eb.w);
                    L.bz[rank] = make_float2(ec, ez);
                    L.pos[rank] = rel + (uint32_t)lane + 1u;
                }
            }
            if (lane < 3) {  // null records: opacity 0
                L.x[cnt + lane] = 0.f; L.y[cnt + lane] = 0.f; L.a[cnt + lane] = 0.f; L.b[cnt + lane] = 0.f;
                L.c[cnt + lane] = 0.f; L.o[cnt + lane] = 0.f;
                if (!T_ONLY) {
                    L.rg[cnt + lane] = make_float2(0.f, 0.f);
                    L.bz[cnt + lane] = make_float2(0.f, 0.f);
                    L.pos[cnt + lane] = 0u;
                }
            }
            const f2 pxf2 = f2{pxf, pxf}, pyf2 = f2{pyf, pyf};
            for (uint32_t i = 0; i < cnt; i += 4) {
                const float4 vx = *reinterpret_cast<const float4*>(&L.x[i]), vy = *reinterpret_cast<const float4*>(&L.y[i]),
                             va = *reinterpret_cast<const float4*>(&L.a[i]), vb = *reinterpret_cast<const float4*>(&L.b[i]),
                             vc = *reinterpret_cast<const float4*>(&L.c[i]), vo = *reinterpret_cast<const float4*>(&L.o[i]);
                float al[4], om[4];
                bool ok[4];
                bool any_ok = false;
#pragma unroll
                for (int j = 0; j < 2; ++j) {  // entries 2j, 2j+1 as one register pair: k_render's operations in k_render's order
                    const f2 X = j ? f2{vx.z, vx.w} : f2{vx.x, vx.y}, Y = j ? f2{vy.z, vy.w} : f2{vy.x, vy.y};
                    const f2 A = j ? f2{va.z, va.w} : f2{va.x, va.y}, B = j ? f2{vb.z, vb.w} : f2{vb.x, vb.y};
                    const f2 Cc = j ? f2{vc.z, vc.w} : f2{vc.x, vc.y}, O = j ? f2{vo.z, vo.w} : f2{vo.x, vo.y};
                    const f2 dx = X - pxf2, dy = Y - pyf2;
                    const f2 t = pk_fma(B, dy, A * dx);
                    const f2 pw = pk_fma(t, dx, (Cc * dy) * dy);
                    const f2 og = O * f2{__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
                    al[2 * j] = fminf(0.99f, og.x);
                    al[2 * j + 1] = fminf(0.99f, og.y);
                    const f2 o2 = f2{1.0f, 1.0f} - f2{al[2 * j], al[2 * j + 1]};
                    om[2 * j] = o2.x;
                    om[2 * j + 1] = o2.y;
                    ok[2 * j] = !done && !(pw.x > 0.0f) && !(al[2 * j] < 1.0f / 255.0f);
                    ok[2 * j + 1] = !done && !(pw.y > 0.0f) && !(al[2 * j + 1] < 1.0f / 255.0f);
                    any_ok = any_ok || ok[2 * j] || ok[2 * j + 1];
                }
                if (__ballot(any_ok) == 0ull) continue;  // wave-uniform
                if (T_ONLY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool v = ok[e] && !done;
                        const float test_T = T * om[e];
                        const bool stop = v && test_T < 0.0001f;
                        done = done || stop;
                        T = (v && !stop) ? test_T : T;
                    }
                } else {
                    const float4 rg01 = *reinterpret_cast<const float4*>(&L.rg[i]), rg23 = *reinterpret_cast<const float4*>(&L.rg[i + 2]);
                    const float4 bz01 = *reinterpret_cast<const float4*>(&L.bz[i]), bz23 = *reinterpret_cast<const float4*>(&L.bz[i + 2]);
                    const uint4 vp = *reinterpret_cast<const uint4*>(&L.pos[i]);
                    const f2 rg[4] = {f2{rg01.x, rg01.y}, f2{rg01.z, rg01.w}, f2{rg23.x, rg23.y}, f2{rg23.z, rg23.w}};
                    const f2 bz[4] = {f2{bz01.x, bz01.y}, f2{bz01.z, bz01.w}, f2{bz23.x, bz23.y}, f2{bz23.z, bz23.w}};
                    const uint32_t posk[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool v = ok[e] && !done;
                        const float test_T = T * om[e];
                        const bool stop = v && test_T < 0.0001f;
                        const bool contrib = v && !stop;
                        done = done || stop;
                        const float w = contrib ? al[e] * T : 0.0f;
                        const f2 w2 = f2{w, w};
                        C01 = C01 + rg[e] * w2;
                        if (WITH_DEPTH) C2D = C2D + bz[e] * w2;
                        else C2D.x = C2D.x + bz[e].x * w;
                        T = contrib ? test_T : T;
                        last = contrib ? posk[e] : last;
                    }
                }
            }
        } else {
            unsigned long long am = act;
            while (am) {
                const int pl = __builtin_ctzll(am);
                am &= am - 1;
                const float ppx = rl(pxf, pl), ppy = rl(pyf, pl);
                const float dx = ea.x - ppx, dy = ea.y - ppy;
                const float power = power2(ea.z, ea.w, eb.x, dx, dy);
                const float alpha = fminf(0.99f, eb.y * __builtin_amdgcn_exp2f(power));
                unsigned long long vm = __ballot(hit && !(power > 0.0f) && !(alpha < 1.0f / 255.0f));
                const bool mine = lane == pl;
                while (vm) {
                    const int eb_ = __builtin_ctzll(vm);
                    vm &= vm - 1;
                    const float a_s = rl(alpha, eb_);
                    const float test_T = T * (1.0f - a_s);
                    const bool stop = mine && test_T < 0.0001f;
                    const bool contrib = mine && !stop;
                    if (!T_ONLY) {
                        const float w = contrib ? a_s * T : 0.0f;
                        C01.x += rl(eb.z, eb_) * w;
                        C01.y += rl(eb.w, eb_) * w;
                        C2D.x += rl(ec, eb_) * w;
                        if (WITH_DEPTH) C2D.y += rl(ez, eb_) * w;
                        last = contrib ? rel + (uint32_t)eb_ + 1u : last;
                    }
                    T = contrib ? test_T : T;
                    done = done || stop;
                    if (__ballot(stop) != 0ull) break;
                }
            }
        }
    }
}

// SPLIT (S360_FLAG_SPLIT_LISTS): the hand-over to segment waves is compiled in; false: the kernel only REPORTS quadrants that would
// have handed over (one store into the caller's host-visible mirror), so that a caller which splits adaptively — the Python layer:
// the flag is set for the calls that follow a report — pays nothing for the feature on clouds that never split (the hand-over code
// costs this kernel, at its 80-VGPR cap, a scratch dword and ~10 us on the headline; the second launch another 3 us).
template <bool WITH_DEPTH, bool COH>
__device__ __forceinline__ void seg_item(const KParams& kp, const S360View* __restrict__ views, const uint32_t* __restrict__ tile_start,
                                         const uint32_t* __restrict__ list, const float4* __restrict__ recA, float* __restrict__ images,
                                         float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib,
                                         uint32_t* __restrict__ strip_last, const float* __restrict__ depths, float* __restrict__ depth_maps,
                                         int depth_mode, const MseEp& ep, float4* __restrict__ surv, uint32_t* __restrict__ surv_count,
                                         const SegBufs& sg, int nt, uint32_t* __restrict__ dbg, const WaveLds& L, int lane, uint2 item, uint32_t wi,
                                         uint32_t nwork);

#ifndef S360_SPLIT_WAVES
#define S360_SPLIT_WAVES 4      // waves per SIMD of the SPLIT instance of k_render (128 VGPRs: its segment workers need them)
#endif
#ifndef S360_P1_GRID
#define S360_P1_GRID 1024
#endif
#ifndef S360_P1_FIRST
#define S360_P1_FIRST 0         // 0: the phase-1 workers behind ALL tiles (in front of the phase-2 workers); 1: right behind the long tiles
                                // — measured on the 1 M surface-like cloud: forward composite 220 us (0) against 277 (1): the 1 024
                                // phase-1 workgroups then sit in front of 1 100 ordinary tiles for two residency rounds
#endif
#ifndef S360_P2_GRID
#define S360_P2_GRID 1024
#endif
#ifndef S360_SEGS_SPINS
#define S360_SEGS_SPINS 600     // polls (~1.7 us apart) without finding anything before a segment wave gives up: ~1 ms
#endif
template <bool WITH_DEPTH, bool SPLIT>
__global__ __launch_bounds__(S360_BLOCK) __attribute__((amdgpu_waves_per_eu(SPLIT ? S360_SPLIT_WAVES : 6, SPLIT ? S360_SPLIT_WAVES : 6))) void k_render(KParams kp, const S360View* __restrict__ views,
                                                      const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ list,
                                                      const float4* __restrict__ recA, const float4* __restrict__ recB,
                                                      const float4* __restrict__ recC, float* __restrict__ images,
                                                      float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                      uint32_t* __restrict__ tile_max_contrib, uint32_t* __restrict__ strip_last,
                                                      uint32_t* __restrict__ dbg, const float* __restrict__ depths,
                                                      float* __restrict__ depth_maps, int depth_mode, MseEp ep,
                                                      const uint32_t* __restrict__ tile_order, float4* __restrict__ surv,
                                                      uint32_t* __restrict__ surv_count, uint32_t* __restrict__ hdr_loss,
                                                      const SegBufs* __restrict__ sgp, const uint32_t* __restrict__ chunk_start,
                                                      unsigned long long* __restrict__ cand_mirror, int nt, const uint32_t* __restrict__ list_all,
                                                      const float* __restrict__ depths_all) {
#ifdef S360_DBG_TIMING
    const long long t_begin = wall_clock64();
#endif
    // per wave: the current chunk's culled records (+ padding), one array per quantity (entry-contiguous: a uniform-address
    // 16-byte read hands FOUR entries to every lane as two register pairs — the operands of the packed v_pk_* arithmetic);
    // the colours are kept as (r, g) and (b, depth value) pairs per entry for the packed accumulation
    __shared__ __attribute__((aligned(16))) float s_x[S360_BLOCK / 64][68], s_y[S360_BLOCK / 64][68], s_a[S360_BLOCK / 64][68],
        s_b[S360_BLOCK / 64][68], s_c[S360_BLOCK / 64][68], s_o[S360_BLOCK / 64][68];
    __shared__ __attribute__((aligned(16))) float2 s_rg[S360_BLOCK / 64][68], s_bz[S360_BLOCK / 64][68];
    __shared__ __attribute__((aligned(16))) uint32_t s_pos[S360_BLOCK / 64][68];
    // tiles are dealt longest list first (LPT: the sequential per-pixel chains of the long polar lists would
    // otherwise form the tail of the kernel): 249 -> 224 us.  (Single-wave workgroups per (tile, quadrant), as in
    // the backward, bring nothing more here: 229 us.)
    if (hdr_loss && blockIdx.x == 0 && threadIdx.x == 0) {
        // S360_FLAG_DEFER_LOSS: where the loss reduction goes, for the backward's first launch (k_order_units)
        const uint64_t pp = (uint64_t)(uintptr_t)ep.partials, po = (uint64_t)(uintptr_t)ep.loss_out;
        hdr_loss[0] = (uint32_t)pp; hdr_loss[1] = (uint32_t)(pp >> 32);
        hdr_loss[2] = (uint32_t)po; hdr_loss[3] = (uint32_t)(po >> 32);
        hdr_loss[4] = (uint32_t)(kp.T * 4); hdr_loss[5] = (uint32_t)kp.V;
        hdr_loss[6] = __float_as_uint(0.5f * ep.grad_scale);
        hdr_loss[7] = __float_as_uint(1.0f / (3.0f * (float)kp.H * (float)kp.W));
        // until the backward has reduced them, loss / clipped MSE read as NaN — a premature read (a NaN guard, a logger, Lightning's
        // returned loss) is then visibly wrong instead of uninitialised memory
        for (int i = 0; i <= kp.V; ++i) ep.loss_out[i] = __uint_as_float(0x7FC00000u);
    }
    // SPLIT (round 6): ONE launch does everything.  Workgroup roles by launch position — the hardware dispatches in blockIdx order:
    //   [0, n_long)                   the tiles whose lists may split (tile_order deals the longest lists first): their heads
    //   [n_long, n_long + P1)         phase-1 workers: the own transmittance T_k of every segment of every such list (no dependency)
    //   [n_long + P1, nt + P1)        all other tiles
    //   [nt + P1, nt + P1 + P2)       phase-2 workers: the (tile, quadrant, segment) items the heads publish, from the pixels' true
    //                                 incoming transmittance, + the combine by whoever delivers last
    // A worker only ever waits (bounded) for workgroups IN FRONT of it in that order, which are running or done by the time it is
    // dispatched; a phase-2 wave that gives up leaves its items unclaimed and k_render_tail (the next launch) takes them.
    // (Until round 6 phase 2 was that second launch: strictly behind the last tile workgroup.)
    __shared__ uint32_t s_retired;
    uint32_t tile_slot = blockIdx.x;
    if (SPLIT) {
        const SegBufs& sgr = *sgp;
        const uint32_t n_long = S360_P1_FIRST ? sgr.header[S360_HDR_NLONG] : (uint32_t)nt;
        const uint32_t b = blockIdx.x;
        if (b >= n_long && b < n_long + (uint32_t)S360_P1_GRID) {
            // ---- phase-1 worker
            const SegBufs sg = sgr;
            const int pwave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const WaveLds L{s_x[pwave], s_y[pwave], s_a[pwave], s_b[pwave], s_c[pwave], s_o[pwave], s_rg[pwave], s_bz[pwave], s_pos[pwave]};
            const uint32_t nunits = min((uint32_t)(SEG_PER_CHUNK * chunk_start[nt]), sg.n_slots);
            for (uint32_t u = b - n_long; u < nunits; u += (uint32_t)S360_P1_GRID) {
#ifdef S360_DBG_TIMING
                const long long t_p1 = wall_clock64();
#endif
                const ChunkUnit cu = chunk_unit(tile_start, chunk_start, nt, kp.cap, u / SEG_PER_CHUNK);
                const uint32_t k = cu.k * SEG_PER_CHUNK + (u % SEG_PER_CHUNK);
                if (!cu.valid || k < SEG_K0 || k * SEG_LEN >= cu.n || cu.n < SEG_HEAD + SEG_MIN_REST) continue;   // block-uniform
                const int t1 = (int)cu.t, v1 = t1 / kp.T, rem1 = t1 - v1 * kp.T;
                const int ty1 = rem1 / kp.gx, tx1 = rem1 - ty1 * kp.gx;
                const int px1 = tx1 * 16 + sub_ox(pwave) + lane % SUB_W, py1 = ty1 * 16 + sub_oy(pwave) + lane / SUB_W;
                const bool inside1 = px1 < kp.W && py1 < kp.H;
                float T1 = 1.0f;
                f2 c01 = f2{0.f, 0.f}, c2d = f2{0.f, 0.f};
                uint32_t last1 = 0, cnt1 = 0;
                bool done1 = !inside1;
                const uint32_t b0 = cu.s + k * SEG_LEN, b1 = min(b0 + SEG_LEN, cu.s + cu.n);
                composite_segment<WITH_DEPTH, true>(list, recA, depths, cu.s, b0, b1, (float)px1, (float)py1, (float)(tx1 * 16 + sub_ox(pwave)),
                                                    (float)(ty1 * 16 + sub_oy(pwave)), lane, L, 0.f, 0.f, 0.f, depth_mode, T1, c01, c2d, last1, done1,
                                                    nullptr, cnt1);
                st_devf(sg.part_t + ((size_t)u * 4 + pwave) * 64 + lane, (done1 && inside1) ? 0.0f : T1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&sg.seg_arrive[4 * t1 + pwave], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef S360_DBG_TIMING
                if (lane == 0) {   // phase-1 records behind the item records: (workgroup start, unit start, unit duration, tile << 8 | k)
                    const size_t di = 4 * ((size_t)4 * nt + (size_t)4 * sg.n_slots + (size_t)u * 4 + pwave);
                    dbg[di] = (uint32_t)t_begin; dbg[di + 1] = (uint32_t)t_p1; dbg[di + 2] = (uint32_t)(wall_clock64() - t_p1);
                    dbg[di + 3] = ((uint32_t)t1 << 8) | k;
                }
#endif
            }
            return;
        }
        if (b >= (uint32_t)nt + (uint32_t)S360_P1_GRID) {
            // ---- phase-2 worker: item i belongs to wave slot i mod (4 P2)
            const SegBufs sg = sgr;
            const int pwave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const WaveLds L{s_x[pwave], s_y[pwave], s_a[pwave], s_b[pwave], s_c[pwave], s_o[pwave], s_rg[pwave], s_bz[pwave], s_pos[pwave]};
            const uint32_t n_items_cap = sg.n_slots * 4u;
            const uint32_t stride = (S360_BLOCK / 64) * (uint32_t)S360_P2_GRID;
            uint32_t wi = (uint32_t)pwave * (uint32_t)S360_P2_GRID + (b - (uint32_t)nt - (uint32_t)S360_P1_GRID);
#ifdef S360_DBG_TIMING
            if (lane == 0) dbg[4 * ((size_t)4 * nt + (size_t)8 * sg.n_slots) + wi] = (uint32_t)t_begin;    // when this worker wave started
#endif
            uint32_t spins = 0;
            while (wi < n_items_cap) {
                uint2 item = ld_dev64(sg.seg_info + wi);         // every lane loads the same address: one request, one value
                item.x = (uint32_t)__shfl((int)item.x, 0);
                item.y = (uint32_t)__shfl((int)item.y, 0);
                if (item.y == 0u) {     // not published (yet)
                    bool finished = false;
                    if ((spins & 3u) == 0u) {
                        uint32_t td = ld_dev32(sg.header + S360_HDR_TILES_DONE);
                        td = (uint32_t)__shfl((int)td, 0);
                        if (td >= (uint32_t)nt) {      // every tile workgroup has retired: the item count is final
                            uint32_t nw = ld_dev32(sg.header + S360_HDR_SEGWORK);
                            nw = (uint32_t)__shfl((int)nw, 0);
                            finished = wi >= nw;
                        }
                    }
                    if (finished || ++spins > (uint32_t)S360_SEGS_SPINS) break;
                    __builtin_amdgcn_s_sleep(64);
                    continue;
                }
                if (item.y & S360_SEG_CLAIM) { wi += stride; continue; }
                const int ti = (int)item.x, q = (int)(item.y & 3u);
                bool ready = (uint32_t)ti < (uint32_t)nt;
                if (ready) {   // all phase-1 products of the quadrant delivered?
                    const uint32_t n = min(tile_start[ti + 1], kp.cap) - min(tile_start[ti], kp.cap);
                    const uint32_t need = (n + SEG_LEN - 1) / SEG_LEN - SEG_K0;
                    uint32_t w2 = 0;
                    for (;;) {
                        uint32_t got = ld_dev32(sg.seg_arrive + 4 * ti + q);
                        got = (uint32_t)__shfl((int)got, 0);
                        if (got >= need) break;
                        if (++w2 > (uint32_t)S360_SEGS_SPINS) { ready = false; break; }
                        __builtin_amdgcn_s_sleep(32);
                    }
                }
                if (!ready) break;      // gives up: this and the wave's remaining items stay unclaimed (k_render_tail takes them)
                uint32_t old = 0;
                if (lane == 0) old = __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(sg.seg_info + wi) + 1, S360_SEG_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = (uint32_t)__shfl((int)old, 0);
                if (!(old & S360_SEG_CLAIM))
                    seg_item<WITH_DEPTH, true>(kp, views, tile_start, list, recA, images, final_T, n_contrib, tile_max_contrib, strip_last, depths, depth_maps,
                                               depth_mode, ep, surv, surv_count, sg, nt, dbg, L, lane, item, wi, 0xFFFFFFFFu);
                wi += stride;
                spins = 0;
            }
            return;
        }
        tile_slot = b < n_long ? b : b - (uint32_t)S360_P1_GRID;
        if (threadIdx.x == 0) s_retired = 0u;
        __syncthreads();
    }
    const int t = tile_order ? (int)tile_order[tile_slot] : (int)tile_slot;
    // (s_setprio by launch-order quartile — the longest lists take the SIMD's issue slots first, all 6 144 waves being resident at
    // once — measured no change: 141.5 vs 141.8 us; the slowest waves are ordinary tiles whose pixels never saturate.)
    const int v = t / kp.T, rem = t - v * kp.T;
    const int ty = rem / kp.gx, tx = rem - ty * kp.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = sub_ox(wave) + lane % SUB_W, ly = sub_oy(wave) + lane / SUB_W;
    const int px = tx * 16 + lx, py = ty * 16 + ly;
    const bool inside = px < kp.W && py < kp.H;
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)(tx * 16 + sub_ox(wave)), ys0 = (float)(ty * 16 + sub_oy(wave));  // footprint origin

    const uint32_t start = min(tile_start[t], kp.cap), end = min(tile_start[t + 1], kp.cap);
    float T = 1.0f;
    f2 C01 = f2{0.f, 0.f}, C2D = f2{0.f, 0.f};  // (r, g) and (b, depth) accumulators
    uint32_t last = 0;
    bool done = !inside;
    // Training calls: every entry that survives this quadrant's exact cull is appended (48-byte record, list order) to the
    // unit's slice of `surv`; the backward composite streams those records back to front instead of walking and culling the
    // tile list again (its walk was a third of its time: 48-byte gathers through L2 for three entries in four that it then
    // dropped).  sv_* remember the last chunk in which any pixel of the quadrant took a contribution: its cull ballot turns the
    // quadrant's final replay length (a list position) into a count of survivor records.
    float4* const sv = surv ? surv + 3 * ((size_t)4 * start + (size_t)wave * (end - start)) : nullptr;
    uint32_t scount = 0, sv_cnt = 0, sv_rel = 0;
    unsigned long long sv_m = 0ull;
    const int vcam = view_of_image(kp, v);  // v = image index
    const float inv_scale = WITH_DEPTH ? 1.0f / views[vcam].scale : 0.f;
    const float v_near = WITH_DEPTH ? views[vcam].near_plane : 0.f, v_far = WITH_DEPTH ? views[vcam].far_plane : 0.f;

    // software pipeline: list indices two chunks ahead, records one chunk ahead
    uint32_t p_n1 = 0, p_n2 = 0;
    if (start + lane < end) p_n1 = list[start + lane];
    if (start + 64 + lane < end) p_n2 = list[start + 64 + lane];
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, nc = na;
    float nz = 0.f;
    if (start + lane < end) {
        na = recA[3 * (size_t)(p_n1)];
        nb = recA[3 * (size_t)(p_n1) + 1];
        nc = recA[3 * (size_t)(p_n1) + 2];
        if (WITH_DEPTH) nz = depths[p_n1];
    }
    // S360_FLAG_SPLIT_LISTS: where this quadrant may hand the rest of its list over to segment waves (k_render_tail)
    // (lists beyond SORT_SHORT keys only: those own segment slots through the sort's chunk table)
    const uint32_t split_at = ((SPLIT || cand_mirror) && end - start >= SEG_HEAD + SEG_MIN_REST && end - start > SORT_SHORT) ? start + SEG_HEAD : 0xFFFFFFFFu;
    bool went = false;
    // (s_setprio 3 on the waves that may hand over — the 1 024-entry heads are this variant's critical path, ~135 us of its ~158 —
    // measured no change: 137 us.  They are not short of issue slots.)
    for (uint32_t b = start; b < end; b += 64) {
        const unsigned long long act = __ballot(!done);
        if (act == 0ull) break;
        if (b == split_at) {   // wave-uniform.  Hand over when some pixel is still FAR from saturating (a pixel about to stop would make
                               // the segment waves speculate for nothing: the headline cloud's polar lists) and the slots exist
            const bool far_px = !done && T >= SEG_T_FAR;
            if (__ballot(far_px) != 0ull) {
                // tell the host (next call): this cloud has quadrants worth splitting
                if (cand_mirror && lane == 0) __hip_atomic_store(cand_mirror, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (word 1 of the mirror)
                if (SPLIT && (SEG_PER_CHUNK * chunk_start[t] + (end - start + SEG_LEN - 1) / SEG_LEN) <= sgp->n_slots) {
                    went = true;
                    break;
                }
            }
        }
        const float4 ea = na, eb = nb;
        // fused depth "colour" of this lane's entry: camera z in unscaled units, then the reference's mode
        float ez = 0.f;
        if (WITH_DEPTH) ez = depth_value(nz * inv_scale, v_near, v_far, depth_mode);
        const float ec = nc.x, erad = nc.y, eka = nc.z, ekb = nc.w;
        const bool ev = b + lane < end;
        const uint32_t epair = p_n1;
        // issue the next chunk's loads before touching this one
        p_n1 = p_n2;
        if (b + 64 + lane < end) {
            na = recA[3 * (size_t)(p_n1)];
            nb = recA[3 * (size_t)(p_n1) + 1];
            nc = recA[3 * (size_t)(p_n1) + 2];
            if (WITH_DEPTH) nz = depths[p_n1];
        }
        if (b + 128 + lane < end) p_n2 = list[b + 128 + lane];

        // (Round 4 culled each chunk against the bounding box of the quadrant's still-unsaturated pixels instead — mask_bbox8 /
        // box_hit_rt in s360_device.h — so that a clump of splats on already saturated pixels is dropped: on the surface-like
        // cloud the backward went 445 -> 323 us (fewer survivor records), on the headline cloud forward +4 / backward -8 us.  Not
        // kept: the unsaturated set at a chunk's start depends on where the chunk boundaries fall, i.e. on the list mode, so the
        // survivor records — and with them the rounding of the backward's scans — were no longer identical between the lean
        // and the upstream-compatible lists.)
        const bool hit = ev && quadrant_hit(ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eka, ekb, x0, ys0);
        unsigned long long m = __ballot(hit);
#ifdef S360_DBG_COUNT
        {
            const unsigned long long qe = __ballot(ev);
            if (lane == 0) atomicAdd(&dbg[3], (uint32_t)__popcll(qe));  // entries walked
        }
#endif
        if (m == 0ull) continue;
        const uint32_t rel = b - start;  // list position of this chunk's lane 0
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (sv && hit) {
            float4* o = sv + 3 * (size_t)(scount + rank);
            o[0] = ea;
            o[1] = eb;
            o[2] = make_float4(ec, erad, __uint_as_float(rel + (uint32_t)lane), __uint_as_float(epair));
        }
        if (__popcll(act) > SPARSE_PIXELS) {
            // Survivors are COMPACTED into the wave's LDS slice (rank = prefix count of the cull ballot) and padded with
            // null records (opacity 0 => alpha 0 => rejected like any other miss) to a multiple of four, so the loop below is
            // a plain counted loop over consecutive records: no bit scanning on the scalar unit, uniform-address
            // ds_read_b128 broadcasts instead of 9 v_readlane per entry, and the alpha evaluations of FOUR entries per
            // iteration are two packed (v_pk_*) chains over entry pairs; the T / colour updates stay strictly sequential
            // (colour accumulators as (r, g) / (b, depth) pairs).  Same operations in the same order as the scalar form:
            // bit-identical images (148 -> 142 us).
            const uint32_t cnt = (uint32_t)__popcll(m);
            {
                if (hit) {
                    s_x[wave][rank] = ea.x; s_y[wave][rank] = ea.y; s_a[wave][rank] = ea.z; s_b[wave][rank] = ea.w;
                    s_c[wave][rank] = eb.x; s_o[wave][rank] = eb.y;
                    s_rg[wave][rank] = make_float2(eb.z, eb.w);
                    s_bz[wave][rank] = make_float2(ec, ez);
                    s_pos[wave][rank] = rel + (uint32_t)lane + 1u;
                }
                if (lane < 3) {  // null records: opacity 0
                    s_x[wave][cnt + lane] = 0.f; s_y[wave][cnt + lane] = 0.f; s_a[wave][cnt + lane] = 0.f; s_b[wave][cnt + lane] = 0.f;
                    s_c[wave][cnt + lane] = 0.f; s_o[wave][cnt + lane] = 0.f;
                    s_rg[wave][cnt + lane] = make_float2(0.f, 0.f);
                    s_bz[wave][cnt + lane] = make_float2(0.f, 0.f);
                    s_pos[wave][cnt + lane] = 0u;
                }
            }
            const f2 pxf2 = f2{pxf, pxf}, pyf2 = f2{pyf, pyf};
            for (uint32_t i = 0; i < cnt; i += 4) {
                const float4 vx = *reinterpret_cast<const float4*>(&s_x[wave][i]), vy = *reinterpret_cast<const float4*>(&s_y[wave][i]),
                             va = *reinterpret_cast<const float4*>(&s_a[wave][i]), vb = *reinterpret_cast<const float4*>(&s_b[wave][i]),
                             vc = *reinterpret_cast<const float4*>(&s_c[wave][i]), vo = *reinterpret_cast<const float4*>(&s_o[wave][i]);
                const float4 rg01 = *reinterpret_cast<const float4*>(&s_rg[wave][i]), rg23 = *reinterpret_cast<const float4*>(&s_rg[wave][i + 2]);
                const float4 bz01 = *reinterpret_cast<const float4*>(&s_bz[wave][i]), bz23 = *reinterpret_cast<const float4*>(&s_bz[wave][i + 2]);
                const uint4 vp = *reinterpret_cast<const uint4*>(&s_pos[wave][i]);
                float al[4], om[4];
                bool ok[4];
                bool any_ok = false;
#pragma unroll
                for (int j = 0; j < 2; ++j) {  // entries 2j, 2j+1 as one register pair
                    const f2 X = j ? f2{vx.z, vx.w} : f2{vx.x, vx.y}, Y = j ? f2{vy.z, vy.w} : f2{vy.x, vy.y};
                    const f2 A = j ? f2{va.z, va.w} : f2{va.x, va.y}, B = j ? f2{vb.z, vb.w} : f2{vb.x, vb.y};
                    const f2 Cc = j ? f2{vc.z, vc.w} : f2{vc.x, vc.y}, O = j ? f2{vo.z, vo.w} : f2{vo.x, vo.y};
                    const f2 dx = X - pxf2, dy = Y - pyf2;
                    const f2 t = pk_fma(B, dy, A * dx);            // power2(): fma(fma(b, dy, a dx), dx, (c dy) dy)
                    const f2 pw = pk_fma(t, dx, (Cc * dy) * dy);
                    const f2 og = O * f2{__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
                    al[2 * j] = fminf(0.99f, og.x);
                    al[2 * j + 1] = fminf(0.99f, og.y);
                    const f2 o2 = f2{1.0f, 1.0f} - f2{al[2 * j], al[2 * j + 1]};
                    om[2 * j] = o2.x;
                    om[2 * j + 1] = o2.y;
                    ok[2 * j] = !done && !(pw.x > 0.0f) && !(al[2 * j] < 1.0f / 255.0f);
                    ok[2 * j + 1] = !done && !(pw.y > 0.0f) && !(al[2 * j + 1] < 1.0f / 255.0f);
                    any_ok = any_ok || ok[2 * j] || ok[2 * j + 1];
                }
#ifdef S360_DBG_COUNT
                if (lane == 0) atomicAdd(&dbg[0], min(4u, cnt - i));
#endif
                if (__ballot(any_ok) == 0ull) continue;  // wave-uniform
                const f2 rg[4] = {f2{rg01.x, rg01.y}, f2{rg01.z, rg01.w}, f2{rg23.x, rg23.y}, f2{rg23.z, rg23.w}};
                const f2 bz[4] = {f2{bz01.x, bz01.y}, f2{bz01.z, bz01.w}, f2{bz23.x, bz23.y}, f2{bz23.z, bz23.w}};
                const uint32_t posk[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool v = ok[e] && !done;
                    const float test_T = T * om[e];
                    const bool stop = v && test_T < 0.0001f;
                    const bool contrib = v && !stop;
                    done = done || stop;
                    const float w = contrib ? al[e] * T : 0.0f;
                    const f2 w2 = f2{w, w};
                    C01 = C01 + rg[e] * w2;   // unfused multiply-add, like the scalar form (the oracle's rounding)
                    if (WITH_DEPTH) C2D = C2D + bz[e] * w2;
                    else C2D.x = C2D.x + bz[e].x * w;
                    T = contrib ? test_T : T;
                    last = contrib ? posk[e] : last;
                }
            }
        } else {
            unsigned long long am = act;
            while (am) {
                const int pl = __builtin_ctzll(am);
                am &= am - 1;
                const float ppx = rl(pxf, pl), ppy = rl(pyf, pl);
                const float dx = ea.x - ppx, dy = ea.y - ppy;
                const float power = power2(ea.z, ea.w, eb.x, dx, dy);
                const float alpha = fminf(0.99f, eb.y * __builtin_amdgcn_exp2f(power));
                unsigned long long vm = __ballot(hit && !(power > 0.0f) && !(alpha < 1.0f / 255.0f));
                const bool mine = lane == pl;
                while (vm) {
                    const int eb_ = __builtin_ctzll(vm);
                    vm &= vm - 1;
                    const float a_s = rl(alpha, eb_);
                    const float test_T = T * (1.0f - a_s);
                    const bool stop = mine && test_T < 0.0001f;
                    const bool contrib = mine && !stop;
                    const float w = contrib ? a_s * T : 0.0f;
                    C01.x += rl(eb.z, eb_) * w;
                    C01.y += rl(eb.w, eb_) * w;
                    C2D.x += rl(ec, eb_) * w;
                    if (WITH_DEPTH) C2D.y += rl(ez, eb_) * w;
                    T = contrib ? test_T : T;
                    last = contrib ? rel + (uint32_t)eb_ + 1u : last;
                    done = done || stop;
                    if (__ballot(stop) != 0ull) break;
                }
            }
        }
        if (sv) {  // wave-uniform
            if (__ballot(last > rel) != 0ull) {  // some pixel's last contributor (so far) lies in this chunk
                sv_m = m;
                sv_cnt = scount;
                sv_rel = rel;
            }
            scount += (uint32_t)__popcll(m);
        }
    }
    if (SPLIT && went) {
        // The exact sequential state of this quadrant after SEG_HEAD entries, per pixel, in slot k = 0 of the tile; the segment waves
        // (k_render_segs, running beside this kernel) composite [SEG_HEAD, end) in parallel and the last of them to finish combines and
        // writes the pixels.  Everything they read is stored device-coherently and drained BEFORE the work items appear.
        const SegBufs sg = *sgp;     // (loaded here only)
        const size_t slot = (size_t)SEG_PER_CHUNK * chunk_start[t];
        const size_t li = (slot * 4 + wave) * 64 + lane;
        st_dev_f4(sg.part_c + li, make_float4(C01.x, C01.y, C2D.x, C2D.y));
        st_devf(sg.part_t + li, T);
        st_dev32(sg.part_l + li, last | (done ? 0x80000000u : 0u));
        const uint32_t wmh = wave_max_u32(inside ? last : 0u);
        // the quadrant's segments join the work list: (tile, segment << 2 | quadrant), in ascending segment order
        const uint32_t nseg = (end - start + SEG_LEN - 1) / SEG_LEN - SEG_K0;
        uint32_t wbase = 0;
        if (lane == 0) {
            st_dev32(sg.part_n + slot * 4 + wave, scount);        // survivor records of the head
            // ... and how many of them lie in front of the head's last contributor (what the backward replays if no segment adds one)
            st_dev32(sg.seg_cnt + slot * 4 + wave, wmh ? sv_cnt + (uint32_t)__popcll(sv_m & ((2ull << (wmh - 1u - sv_rel)) - 1ull)) : 0u);
            st_dev32(sg.seg_flag + 4 * t + wave, 1u);
            atomicAdd(&sg.header[S360_HDR_SPLIT], 1u);
            wbase = atomicAdd(&sg.header[S360_HDR_SEGWORK], nseg);
        }
        wbase = (uint32_t)__shfl((int)wbase, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the state above (and this wave's survivor records) before the items
        for (uint32_t i = (uint32_t)lane; i < nseg; i += 64u) st_dev64(sg.seg_info + wbase + i, make_uint2((uint32_t)t, ((SEG_K0 + i) << 2) | (uint32_t)wave));
#ifdef S360_DBG_TIMING
        if (lane == 0) {
            dbg[4 * (4 * t + wave)] = (uint32_t)t_begin;
            dbg[4 * (4 * t + wave) + 1] = (uint32_t)(wall_clock64() - t_begin);
            dbg[4 * (4 * t + wave) + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            dbg[4 * (4 * t + wave) + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && atomicAdd(&s_retired, 1u) == 3u) __hip_atomic_fetch_add(&sg.header[S360_HDR_TILES_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    float sq = 0.f, sqc = 0.f;
    if (inside) {
        const S360View& vw = views[vcam];
        const size_t hw = (size_t)kp.H * kp.W;
        const size_t pix = (size_t)py * kp.W + px;
        float* img = images + (size_t)v * 3 * hw;
        const float o0 = C01.x + T * vw.bg[0], o1 = C01.y + T * vw.bg[1], o2 = C2D.x + T * vw.bg[2];
        img[pix] = o0;
        img[hw + pix] = o1;
        img[2 * hw + pix] = o2;
        if (ep.target) {
            const float* gt = ep.target + (size_t)v * 3 * hw;
            float* dg = ep.d_images + (size_t)v * 3 * hw;
            const float g0 = gt[pix], g1 = gt[hw + pix], g2 = gt[2 * hw + pix];
            const float d0 = o0 - g0, d1 = o1 - g1, d2 = o2 - g2;
            dg[pix] = ep.grad_scale * d0;
            dg[hw + pix] = ep.grad_scale * d1;
            dg[2 * hw + pix] = ep.grad_scale * d2;
            sq = d0 * d0 + d1 * d1 + d2 * d2;
            const float c0 = fminf(fmaxf(g0, 0.f), 1.f) - fminf(fmaxf(o0, 0.f), 1.f);
            const float c1 = fminf(fmaxf(g1, 0.f), 1.f) - fminf(fmaxf(o1, 0.f), 1.f);
            const float c2 = fminf(fmaxf(g2, 0.f), 1.f) - fminf(fmaxf(o2, 0.f), 1.f);
            sqc = c0 * c0 + c1 * c1 + c2 * c2;
        }
        final_T[(size_t)v * hw + pix] = T;
        n_contrib[(size_t)v * hw + pix] = last;
        if (WITH_DEPTH) depth_maps[(size_t)v * hw + pix] = C2D.y;  // background depth is 0 (cuda_splatting.py:258)
    }
    if (ep.target) {  // wave-uniform
        const float s0 = wave_sum1_lane63(sq), s1 = wave_sum1_lane63(sqc);
        if (lane == 63) {
            ep.partials[2 * (4 * (size_t)t + wave)] = s0;
            ep.partials[2 * (4 * (size_t)t + wave) + 1] = s1;
        }
    }
    const uint32_t wm = wave_max_u32(inside ? last : 0u);
    if (lane == 0) {
        strip_last[4 * t + wave] = wm;  // per-quadrant replay length (list positions)
        if (wm) atomicMax(&tile_max_contrib[t], wm);  // zeroed by k_tile_scan
        // survivor records in front of the quadrant's last contributor (list position wm - 1, itself a survivor of chunk
        // sv_rel): what the backward replays, and its work estimate
        if (surv_count) surv_count[4 * t + wave] = wm ? sv_cnt + (uint32_t)__popcll(sv_m & ((2ull << (wm - 1u - sv_rel)) - 1ull)) : 0u;
    }
#ifdef S360_DBG_TIMING
    if (lane == 0) {
        dbg[4 * (4 * t + wave)] = (uint32_t)t_begin;
        dbg[4 * (4 * t + wave) + 1] = (uint32_t)(wall_clock64() - t_begin);
        dbg[4 * (4 * t + wave) + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
        dbg[4 * (4 * t + wave) + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    }
#endif
    if (SPLIT) {   // the segment waves stop looking for work once every tile workgroup has retired
        if (lane == 0 && atomicAdd(&s_retired, 1u) == 3u)
            __hip_atomic_fetch_add(&sgp->header[S360_HDR_TILES_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ------------------------------------------------------------------------------ split lists: segment waves + combine
// S360_FLAG_SPLIT_LISTS, second launch of the composite.  k_render left, for every split (tile, quadrant), the exact per-pixel state
// after SEG_HEAD entries and, from its phase-1 workers, the own transmittance T_k of every SEG_LEN-entry segment k of the rest of the
// list (per pixel, from T = 1; 0 where the pixel stops inside the segment).  Here every wave takes (tile, quadrant, segment) work
// items from the dense list k_render queued:
//   phase 2  the wave composites its segment with k_render's sequential rule from the pixel's true incoming transmittance
//            T_in(k) = T_head * T_K0 * ... * T_(k-1)  (a fixed left-to-right product); a pixel with T_in(k) < 1e-4 has stopped in an
//            earlier segment and is skipped.  It appends the segment's survivor records (training calls) and delivers its colour
//            contribution, the transmittance behind it and its last contributor;
//   combine  the wave that delivers last for a quadrant adds the contributions in list order, writes the pixels exactly like
//            k_render's epilogue, and leaves per segment what the backward starts from: transmittance behind it, colour accumulated
//            behind it.
// Every (pixel, entry) pair is evaluated with k_render's arithmetic and k_render's stop rule, in list order, from a transmittance
// that differs from the sequential product only by floating-point association (T_in is a product of segment products): images
// within 1e-6 of the unsplit composite, stop decisions identical except where a product lands within rounding of 1e-4.
// No wave ever waits for another: the phase-1 results come from the previous launch, and the combine is done by whoever arrives last
// (device-coherent stores, drained with s_waitcnt vmcnt(0), then ONE agent-scope counter — the pattern of k_merge_all).  The
// arrival count is broadcast BY VALUE (__shfl), never v_readfirstlane: a build that handed out work through a ticket loop the
// compiler treated as divergent left the lanes that are not lane 0 with ticket 0 for ever and hung the GPU.
#ifndef S360_TAIL_GRID
#define S360_TAIL_GRID 1024
#endif
#ifndef S360_TAIL_WAVES
#define S360_TAIL_WAVES 4   // waves per SIMD of k_render_tail (128 VGPRs; 3 = unconstrained: 141 VGPRs)
#endif
// One (tile, quadrant, segment) work item: phase 2 + delivery + (for the wave that delivers last) the combine.  COH: the head's state and
// the phase-1 products were written inside the CURRENT launch window by other CUs (k_render / k_render_segs running side by side) — read
// them device-coherently; false: they come from completed launches (k_render_tail, the mop-up).
template <bool WITH_DEPTH, bool COH>
__device__ __forceinline__ void seg_item(const KParams& kp, const S360View* __restrict__ views, const uint32_t* __restrict__ tile_start,
                                         const uint32_t* __restrict__ list, const float4* __restrict__ recA, float* __restrict__ images,
                                         float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max_contrib,
                                         uint32_t* __restrict__ strip_last, const float* __restrict__ depths, float* __restrict__ depth_maps,
                                         int depth_mode, const MseEp& ep, float4* __restrict__ surv, uint32_t* __restrict__ surv_count,
                                         const SegBufs& sg, int nt, uint32_t* __restrict__ dbg, const WaveLds& L, int lane, uint2 item, uint32_t wi,
                                         uint32_t nwork) {
#ifdef S360_DBG_TIMING
        const long long t_begin = wall_clock64();
#endif
        const int t = (int)item.x, wave = (int)(item.y & 3u);
        const uint32_t k = item.y >> 2;
        if ((uint32_t)t >= (uint32_t)nt) {   // corrupt work item: an error word (RasterState.split_errors), never a wild access
            if (lane == 0 && atomicAdd(&sg.header[7], 0x10000u) == 0u) {
                sg.header[8] = item.x; sg.header[9] = item.y; sg.header[12] = wi; sg.header[13] = nwork;
            }
            return;
        }
        const uint32_t start = min(tile_start[t], kp.cap), end = min(tile_start[t + 1], kp.cap), n = end - start;
        if (k < SEG_K0 || k * SEG_LEN >= n || (COH ? ld_dev32(sg.seg_flag + 4 * t + wave) : sg.seg_flag[4 * t + wave]) != 1u) {
            if (lane == 0 && atomicAdd(&sg.header[7], 0x100u) == 0u) {
                sg.header[8] = item.x; sg.header[9] = item.y; sg.header[10] = n; sg.header[12] = wi; sg.header[13] = nwork;
            }
            return;
        }
        const size_t u = (size_t)SEG_PER_CHUNK * sg.chunk_start[t] + k;     // the segment's slot
        const int v = t / kp.T, rem = t - v * kp.T;
        const int ty = rem / kp.gx, tx = rem - ty * kp.gx;
        const int lx = sub_ox(wave) + lane % SUB_W, ly = sub_oy(wave) + lane / SUB_W;
        const int px = tx * 16 + lx, py = ty * 16 + ly;
        const bool inside = px < kp.W && py < kp.H;
        const float pxf = (float)px, pyf = (float)py;
        const float x0 = (float)(tx * 16 + sub_ox(wave)), ys0 = (float)(ty * 16 + sub_oy(wave));
        const int vcam = view_of_image(kp, v);
        const float inv_scale = WITH_DEPTH ? 1.0f / views[vcam].scale : 0.f;
        const float v_near = WITH_DEPTH ? views[vcam].near_plane : 0.f, v_far = WITH_DEPTH ? views[vcam].far_plane : 0.f;
        const uint32_t K = (n + SEG_LEN - 1) / SEG_LEN;   // segments of the list, the head's SEG_K0 included
        const uint32_t b0 = start + k * SEG_LEN, b1 = min(b0 + SEG_LEN, end);
        const size_t slot0 = u - k;                       // slot of segment 0 of this tile (= SEG_PER_CHUNK * chunk_start[t])
        const size_t li0 = (slot0 * 4 + wave) * 64 + lane, li = ((size_t)u * 4 + wave) * 64 + lane;
        // ---- the segment itself, from the pixel's true incoming transmittance (phase-1 results: k_render's workers, the previous launch)
        float T;
        bool head_done;
        {   // the head's state (written by k_render: the previous launch) and the fixed left-to-right product of the segments in front
            const uint32_t l = COH ? ld_dev32(sg.part_l + li0) : sg.part_l[li0];
            head_done = (l >> 31) != 0u || !inside;
            T = COH ? ld_devf(sg.part_t + li0) : sg.part_t[li0];
#pragma unroll 8
            for (uint32_t kk = SEG_K0; kk < k; ++kk) {
                const float* q = sg.part_t + ((slot0 + kk) * 4 + wave) * 64 + lane;
                T = T * (COH ? ld_devf(q) : *q);
            }
        }
        {
            f2 C01 = f2{0.f, 0.f}, C2D = f2{0.f, 0.f};
            uint32_t last = 0, scount = 0;
            bool done = head_done || T < 0.0001f;    // stopped in an earlier segment (its product took T below the stop threshold)
            const bool live_in = !done;
            float4* const sv_unit = surv ? surv + 3 * ((size_t)4 * start + (size_t)wave * n) : nullptr;
            composite_segment<WITH_DEPTH, false>(list, recA, depths, start, b0, b1, pxf, pyf, x0, ys0, lane, L, inv_scale, v_near, v_far, depth_mode, T,
                                                 C01, C2D, last, done, sv_unit ? sv_unit + 3 * (size_t)(k * SEG_LEN) : nullptr, scount);
            float* pc = reinterpret_cast<float*>(sg.part_c + li);
            st_devf(pc, C01.x); st_devf(pc + 1, C01.y); st_devf(pc + 2, C2D.x); st_devf(pc + 3, C2D.y);
            st_devf(sg.part_e + li, T);         // transmittance behind the segment (unchanged where the pixel took nothing)
            st_dev32(sg.part_l + li, last | ((live_in && done) ? 0x80000000u : 0u));   // bit 31: the stop test fired INSIDE this segment
            if (lane == 0) st_dev32(sg.part_n + (size_t)u * 4 + wave, scount);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(&sg.seg_arrive2[4 * t + wave], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived = (uint32_t)__shfl((int)arrived, 0);
#ifdef S360_DBG_TIMING
        if (lane == 0) {
            const size_t di = 4 * ((size_t)4 * nt + wi);
            dbg[di] = (uint32_t)t_begin;
            dbg[di + 1] = dbg[di + 2] = (uint32_t)(wall_clock64() - t_begin);
            dbg[di + 3] = ((uint32_t)t << 12) | (k << 2) | (uint32_t)wave;
        }
#endif
        if (arrived + 1u != K - SEG_K0) return;   // wave-uniform: another segment of this quadrant is still out
        // ---- combine: this wave delivered last
        f2 C01, C2D;
        uint32_t last;
        {
            const float4 c = COH ? ld_dev_f4(sg.part_c + li0) : sg.part_c[li0];
            C01 = f2{c.x, c.y}; C2D = f2{c.z, c.w};
            last = (COH ? ld_dev32(sg.part_l + li0) : sg.part_l[li0]) & 0x7FFFFFFFu;
            T = COH ? ld_devf(sg.part_t + li0) : sg.part_t[li0];
        }
        const float head_T = T;
        uint32_t kstop = SEG_K0;   // the pixel was live in segments [SEG_K0, kstop) (the products only shrink: one change-over)
        {
            float Tin = head_T;     // the same left-to-right product the segment waves formed: which segments this pixel was live in
            bool alive = !head_done;
            // The other waves' results are read with device-coherent loads, FOUR segments (12 loads) in flight per wait: one load per
            // round trip — what a loop of atomic loads compiles to — cost the combining wave of a 34-segment list ~100 us.
            for (uint32_t kb = SEG_K0; kb < K; kb += 4) {
                size_t ix[4];
                float tq[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    ix[j] = ((slot0 + min(kb + j, K - 1)) * 4 + wave) * 64 + lane;
                    tq[j] = COH ? ld_devf(sg.part_t + ix[j]) : sg.part_t[ix[j]];      // (phase 1)
                }
                f4v cq[4];
                float teq[4];
                uint32_t plq[4];
                asm volatile(
                    "global_load_dwordx4 %0, %12, off sc1\n"
                    "global_load_dwordx4 %1, %13, off sc1\n"
                    "global_load_dwordx4 %2, %14, off sc1\n"
                    "global_load_dwordx4 %3, %15, off sc1\n"
                    "global_load_dword %4, %16, off sc1\n"
                    "global_load_dword %5, %17, off sc1\n"
                    "global_load_dword %6, %18, off sc1\n"
                    "global_load_dword %7, %19, off sc1\n"
                    "global_load_dword %8, %20, off sc1\n"
                    "global_load_dword %9, %21, off sc1\n"
                    "global_load_dword %10, %22, off sc1\n"
                    "global_load_dword %11, %23, off sc1\n"
                    "s_waitcnt vmcnt(0)\n"
                    : "=&v"(cq[0]), "=&v"(cq[1]), "=&v"(cq[2]), "=&v"(cq[3]), "=&v"(teq[0]), "=&v"(teq[1]), "=&v"(teq[2]), "=&v"(teq[3]),
                      "=&v"(plq[0]), "=&v"(plq[1]), "=&v"(plq[2]), "=&v"(plq[3])
                    : "v"(sg.part_c + ix[0]), "v"(sg.part_c + ix[1]), "v"(sg.part_c + ix[2]), "v"(sg.part_c + ix[3]),
                      "v"(sg.part_e + ix[0]), "v"(sg.part_e + ix[1]), "v"(sg.part_e + ix[2]), "v"(sg.part_e + ix[3]),
                      "v"(sg.part_l + ix[0]), "v"(sg.part_l + ix[1]), "v"(sg.part_l + ix[2]), "v"(sg.part_l + ix[3])
                    : "memory");
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    if (kb + j < K) {
                        alive = alive && !(Tin < 0.0001f);
                        Tin = Tin * tq[j];
                        if (alive) {
                            C01 = C01 + f2{cq[j].x, cq[j].y}; C2D = C2D + f2{cq[j].z, cq[j].w};
                            T = teq[j];
                            last = (plq[j] & 0x7FFFFFFFu) ? (plq[j] & 0x7FFFFFFFu) : last;
                            kstop = kb + j + 1;
                            // the segment's own sequential stop decides, not the rounding of the product of products: a
                            // pixel that stopped inside segment k is dead in every later one, for the forward's sums and the backward's units alike
                            alive = (plq[j] >> 31) == 0u;
                        }
                        sg.seg_t[ix[j]] = T;                                   // transmittance behind the segment
                    }
                }
            }
        }
        // ---- the pixels, exactly as k_render's epilogue writes them
        float sq = 0.f, sqc = 0.f;
        if (inside) {
            const S360View& vw = views[vcam];
            const size_t hw = (size_t)kp.H * kp.W;
            const size_t pix = (size_t)py * kp.W + px;
            float* img = images + (size_t)v * 3 * hw;
            const float o0 = C01.x + T * vw.bg[0], o1 = C01.y + T * vw.bg[1], o2 = C2D.x + T * vw.bg[2];
            img[pix] = o0;
            img[hw + pix] = o1;
            img[2 * hw + pix] = o2;
            if (ep.target) {
                const float* gt = ep.target + (size_t)v * 3 * hw;
                float* dg = ep.d_images + (size_t)v * 3 * hw;
                const float g0 = gt[pix], g1 = gt[hw + pix], g2 = gt[2 * hw + pix];
                const float d0 = o0 - g0, d1 = o1 - g1, d2 = o2 - g2;
                dg[pix] = ep.grad_scale * d0;
                dg[hw + pix] = ep.grad_scale * d1;
                dg[2 * hw + pix] = ep.grad_scale * d2;
                sq = d0 * d0 + d1 * d1 + d2 * d2;
                const float q0 = fminf(fmaxf(g0, 0.f), 1.f) - fminf(fmaxf(o0, 0.f), 1.f);
                const float q1 = fminf(fmaxf(g1, 0.f), 1.f) - fminf(fmaxf(o1, 0.f), 1.f);
                const float q2 = fminf(fmaxf(g2, 0.f), 1.f) - fminf(fmaxf(o2, 0.f), 1.f);
                sqc = q0 * q0 + q1 * q1 + q2 * q2;
            }
            final_T[(size_t)v * hw + pix] = T;
            n_contrib[(size_t)v * hw + pix] = last;
            if (WITH_DEPTH) depth_maps[(size_t)v * hw + pix] = C2D.y;
        }
        if (ep.target) {  // wave-uniform
            const float s0 = wave_sum1_lane63(sq), s1 = wave_sum1_lane63(sqc);
            if (lane == 63) {
                ep.partials[2 * (4 * (size_t)t + wave)] = s0;
                ep.partials[2 * (4 * (size_t)t + wave) + 1] = s1;
            }
        }
        const uint32_t wm = wave_max_u32(inside ? last : 0u);
        if (lane == 0) {
            strip_last[4 * t + wave] = wm;
            if (wm) atomicMax(&tile_max_contrib[t], wm);
            // the backward's units of this quadrant: the head (all of its survivor records once a later segment contributes, else
            // those in front of its own last contributor) and every segment that starts in front of the last contributor
            if (surv_count) surv_count[4 * t + wave] = wm > SEG_HEAD ? ld_dev32(sg.part_n + slot0 * 4 + wave) : ld_dev32(sg.seg_cnt + slot0 * 4 + wave);
        }
        for (uint32_t kk = SEG_K0 + (uint32_t)lane; kk < K; kk += 64)      // (one segment per lane: the loads overlap)
            sg.seg_cnt[(slot0 + kk) * 4 + wave] = (surv_count && kk * SEG_LEN < wm) ? ld_dev32(sg.part_n + (slot0 + kk) * 4 + wave) : 0u;
        // ---- colour accumulated BEHIND every segment (back to front: the small terms first) and behind the head: a second pass
        // over the segments' contributions, four coherent loads in flight per wait
        {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
            for (uint32_t kt = K; kt > SEG_K0;) {
                // segments kt-1, kt-2, kt-3, kt-4 (those below SEG_K0 clamp to a valid slot and are skipped)
                size_t ix[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) ix[j] = ((slot0 + (kt - SEG_K0 > j ? kt - 1 - j : (uint32_t)SEG_K0)) * 4 + wave) * 64 + lane;
                f4v dq[4];
                asm volatile(
                    "global_load_dwordx4 %0, %4, off sc1\n"
                    "global_load_dwordx4 %1, %5, off sc1\n"
                    "global_load_dwordx4 %2, %6, off sc1\n"
                    "global_load_dwordx4 %3, %7, off sc1\n"
                    "s_waitcnt vmcnt(0)\n"
                    : "=&v"(dq[0]), "=&v"(dq[1]), "=&v"(dq[2]), "=&v"(dq[3])
                    : "v"(sg.part_c + ix[0]), "v"(sg.part_c + ix[1]), "v"(sg.part_c + ix[2]), "v"(sg.part_c + ix[3])
                    : "memory");
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    if (kt - SEG_K0 > j) {
                        const uint32_t kk = kt - 1 - j;
                        sg.seg_c[ix[j]] = make_float4(acc.x, acc.y, acc.z, acc.w);
                        if (kk < kstop) acc = acc + dq[j];      // (a segment the pixel had stopped in front of contributes nothing)
                    }
                }
                kt = kt - SEG_K0 > 4 ? kt - 4 : (uint32_t)SEG_K0;
            }
            sg.seg_c[li0] = make_float4(acc.x, acc.y, acc.z, acc.w);
            sg.seg_t[li0] = head_T;
        }
#ifdef S360_DBG_TIMING
        if (lane == 0) {   // the combining wave: phase-2 segment + combine, marked
            const size_t di = 4 * ((size_t)4 * nt + wi);
            dbg[di + 2] = (uint32_t)(wall_clock64() - t_begin);
            dbg[di + 3] |= 0x80000000u;
        }
#endif
}

// The mop-up behind k_render: every work item no phase-2 worker of k_render claimed —
// none, normally: then this launch reads the item list once and returns.  (Until round 6 this launch did ALL the segment work,
// strictly after the last tile workgroup: 62 us on the 1 M surface-like cloud behind a 159-us k_render, and no gain at 4 M / 512^2.)
template <bool WITH_DEPTH>
__global__ __launch_bounds__(S360_BLOCK) __attribute__((amdgpu_waves_per_eu(S360_TAIL_WAVES, S360_TAIL_WAVES))) void k_render_tail(KParams kp, const S360View* __restrict__ views, const uint32_t* __restrict__ tile_start,
                                                            const uint32_t* __restrict__ list, const float4* __restrict__ recA,
                                                            float* __restrict__ images, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                            uint32_t* __restrict__ tile_max_contrib, uint32_t* __restrict__ strip_last,
                                                            const float* __restrict__ depths, float* __restrict__ depth_maps, int depth_mode, MseEp ep,
                                                            float4* __restrict__ surv, uint32_t* __restrict__ surv_count, SegBufs sg, int nt,
                                                            uint32_t* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) float s_x[S360_BLOCK / 64][68], s_y[S360_BLOCK / 64][68], s_a[S360_BLOCK / 64][68],
        s_b[S360_BLOCK / 64][68], s_c[S360_BLOCK / 64][68], s_o[S360_BLOCK / 64][68];
    __shared__ __attribute__((aligned(16))) float2 s_rg[S360_BLOCK / 64][68], s_bz[S360_BLOCK / 64][68];
    __shared__ __attribute__((aligned(16))) uint32_t s_pos[S360_BLOCK / 64][68];
    if (sg.header[S360_HDR_SPLIT] == 0u) return;   // no quadrant of this call split
    const int pwave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // pwave: this wave's LDS slices; its quadrant comes with the work item
    const WaveLds L{s_x[pwave], s_y[pwave], s_a[pwave], s_b[pwave], s_c[pwave], s_o[pwave], s_rg[pwave], s_bz[pwave], s_pos[pwave]};
    const uint32_t nwork = sg.header[S360_HDR_SEGWORK];       // (tile, quadrant, segment) items k_render queued
    // items are dealt STATICALLY, item i to wave slot i mod (4 gridDim) (a ticket counter — one agent-scope atomic per item on ONE
    // address — let the last wave start 27 us into the kernel)
    for (uint32_t wi = (uint32_t)pwave * gridDim.x + blockIdx.x; wi < nwork; wi += (S360_BLOCK / 64) * gridDim.x) {
        uint2 item = sg.seg_info[wi];
        if (item.y & S360_SEG_CLAIM) continue;       // a phase-2 worker of k_render took it
        seg_item<WITH_DEPTH, false>(kp, views, tile_start, list, recA, images, final_T, n_contrib, tile_max_contrib, strip_last, depths, depth_maps,
                                    depth_mode, ep, surv, surv_count, sg, nt, dbg, L, lane, item, wi, nwork);
    }
}

}  // namespace s360

// ------------------------------------------------------------------------------ host side
using namespace s360;

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

extern "C" int s360_layout(const S360Params* prm, S360Layout* out) {
    if (!prm || !out) return S360_E_BADARG;
    if (prm->P < 0 || prm->V < 1 || prm->V > S360_MAX_VIEWS || prm->H < 1 || prm->W < 1) return S360_E_BADARG;
    if ((uint64_t)prm->V * (uint64_t)prm->P >= (1ull << 31)) return S360_E_BADARG;   // pair indices carry a tag in bit 31 (slot_pair)
    const size_t np = (size_t)prm->V * (size_t)(prm->P > 0 ? prm->P : 1);
    const size_t gx = (prm->W + 15) / 16, gy = (prm->H + 15) / 16;
    const size_t nt = (size_t)prm->V * gx * gy;
    const size_t cap = prm->max_instances ? prm->max_instances : 1;
    const size_t npix = (size_t)prm->V * prm->H * prm->W;
    const bool fwd_only = (prm->flags & S360_FLAG_FORWARD_ONLY) != 0;  // inference calls keep no backward state
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes);
        return r;
    };
    out->header = take(64 * 4);
    out->tiles_touched = take(np * 4);
    out->vis_mask = take((size_t)(prm->P > 0 ? prm->P : 1));
    out->slot_base = take(np * 8);   // (first slot, hit mask) per pair
    out->rec_a = take(np * 48);  // one 48-byte record per pair: rec_b / rec_c are the 2nd / 3rd float4 of it
    out->rec_b = out->rec_a + 16;
    out->rec_c = out->rec_a + 32;
    out->clamped = take(np);
    out->depths = take(np * 4);
    out->tile_count = take(nt * 4);
    out->slot_ticket = take((size_t)prm->V * 256);  // per-image instance-slot tickets, 256 B apart; cleared with tile_count
    out->merge_done = take((nt * MAX_PASSES + 1) * 4);   // completion counters of the merge passes, one per (tile, pass), + the ticket counter of their work queue; cleared with tile_count
    out->seg_flag = take(nt * 4 * 4);                    // S360_FLAG_SPLIT_LISTS: split mark / arrival counters per (tile, quadrant); cleared with tile_count
    out->seg_arrive = take(nt * 4 * 4);
    out->seg_arrive2 = take((nt * 4 + 1) * 4);           // ... + the ticket counter of k_render_tail's work queue
    out->tile_start = take((nt + 1) * 4);
    out->tile_cursor = take(nt * 4);
    out->chunk_start = take((nt + 1) * 4);
    out->tile_order = take(nt * 4);
    out->keys = take(cap * 8);
    out->keys_alt = take(cap * 8);
    out->list = take(cap * 4);
    out->final_T = take(npix * 4);
    out->n_contrib = take(npix * 4);
    out->tile_max_contrib = take(nt * 4);
    out->strip_last = take(nt * 4 * 4);
    out->slot_pair = take(cap * 4);
    out->long_pairs = take(fwd_only ? 16 : (cap / 32 + 1) * 4);
    out->rgbc = take((size_t)(prm->P > 0 ? prm->P : 1) * 16);
    out->sh_jac = take((size_t)(prm->P > 0 ? prm->P : 1) * 36);
    out->surv = take(fwd_only ? 16 : cap * 4 * 48);
    out->surv_count = take(nt * 4 * 4);
    {   // segment state of S360_FLAG_SPLIT_LISTS (44 B per pixel of a (segment slot, quadrant): ~11 B per instance of capacity)
        const bool split = (prm->flags & S360_FLAG_SPLIT_LISTS) != 0;
        const size_t ns = split ? seg_slots_of(prm) : 1, nl = ns * 4 * 64;
        out->part_c = take(nl * 16);
        out->part_t = take(nl * 4);
        out->part_e = take(nl * 4);
        out->part_l = take(nl * 4);
        out->part_n = take(ns * 4 * 4);
        out->seg_c = take(nl * 16);
        out->seg_t = take(nl * 4);
        out->seg_cnt = take(ns * 4 * 4);
        out->seg_info = take(ns * 4 * 8);
    }
    // s360_forward_raw: the 7 raw geometry words per Gaussian (scale logits, quaternion), compacted for s360_backward_raw
    out->geo7 = take((prm->flags & S360_FLAG_RAW_INPUTS) && !fwd_only ? (size_t)(prm->P > 0 ? prm->P : 1) * 28 : 16);
    out->total_bytes = o;
    // backward scratch: 4 quadrant-partial raster-gradient records (12 floats) + 4 validity bytes per instance
    // ... + tile order [V*T] + one gathered 48-byte record per (view, Gaussian) pair
    // ... + tile order [V*T] + one gathered 48-byte record per (view, Gaussian) pair.  S360_FLAG_ATOMIC_GRADS: the composite
    // adds into the pair records directly — no partial slots, no validity flags
    const bool atomic = (prm->flags & S360_FLAG_ATOMIC_GRADS) != 0;
    out->backward_bytes = (atomic ? 0 : align_up(cap * 4 * (size_t)(16 * S360_PREC_F4)) + align_up(cap * 4)) + align_up(nt * 4 * 4) + 512 +
                          align_up(np * 48) + align_up((size_t)(prm->P > 0 ? prm->P : 1) * 16) + 256 +
                          ((prm->flags & S360_FLAG_SPLIT_LISTS) ? align_up((seg_slots_of(prm) * 4 + 64 + 32) * 4) + 256 : 0);   // launch list of the split segments' units
    return S360_OK;
}

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

static int forward_impl(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                        const float* opacities, const float* shs, const float* colors_precomp, float* images,
                        float* depth_maps, int depth_mode, int32_t* radii, void* workspace, size_t workspace_bytes,
                        void* stream_, MseEp ep = MseEp{nullptr, nullptr, nullptr, 0.f, nullptr}, const RawIn* rawin = nullptr) {
    if (!prm || !views || !images || !workspace) return S360_E_BADARG;
    if (prm->P > 0 && (shs == nullptr) == (colors_precomp == nullptr)) return S360_E_BADARG;
    if (((prm->flags & S360_FLAG_RAW_INPUTS) != 0) != (rawin != nullptr)) return S360_E_BADARG;   // the flag sizes the workspace for this entry point
    if (prm->P > 0 && (!means3D || !cov6 || !opacities)) return S360_E_BADARG;
    if (shs && (prm->M < 1 || prm->sh_degree < 0 || prm->sh_degree > 4 ||
                (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M))
        return S360_E_BADARG;
    if (prm->W > 65535 * 16 || prm->H > 65535 * 16) return S360_E_UNSUPPORTED;
    S360Layout L;
    int rc = s360_layout(prm, &L);
    if (rc) return rc;
    if (workspace_bytes < L.total_bytes) return S360_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;

    KParams kp;
    kp.P = prm->P; kp.V = prm->V; kp.H = prm->H; kp.W = prm->W; kp.deg = s360_effective_degree(prm); kp.M = prm->M;
    kp.gx = (prm->W + 15) / 16; kp.gy = (prm->H + 15) / 16; kp.T = kp.gx * kp.gy;
    kp.flags = prm->flags; kp.cap = prm->max_instances;
    const bool sph = (kp.flags & S360_FLAG_SPHERICAL) != 0;
    if (sph && (kp.V & 1)) return S360_E_BADARG;  // views come in (camera, seam ghost) pairs
    const int nt = (sph ? kp.V / 2 : kp.V) * kp.T;
    const size_t np = (size_t)kp.V * kp.P;

    uint32_t* header = (uint32_t*)(ws + L.header);
    uint32_t* tiles_touched = (uint32_t*)(ws + L.tiles_touched);
    uint8_t* vis_mask = (uint8_t*)(ws + L.vis_mask);
    uint2* slot_info = (uint2*)(ws + L.slot_base);
    uint32_t* slot_ticket = (uint32_t*)(ws + L.slot_ticket);
    float4* recA = (float4*)(ws + L.rec_a);
    float4* recB = (float4*)(ws + L.rec_b);
    float4* recC = (float4*)(ws + L.rec_c);
    uint8_t* clamped = (uint8_t*)(ws + L.clamped);
    float* depths = (float*)(ws + L.depths);
    uint32_t* tile_count = (uint32_t*)(ws + L.tile_count);
    uint32_t* tile_start = (uint32_t*)(ws + L.tile_start);
    uint32_t* tile_cursor = (uint32_t*)(ws + L.tile_cursor);
    uint64_t* keys = (uint64_t*)(ws + L.keys);
    uint64_t* keys_alt = (uint64_t*)(ws + L.keys_alt);
    uint32_t* chunk_start = (uint32_t*)(ws + L.chunk_start);
    uint32_t* tile_order = (uint32_t*)(ws + L.tile_order);
    uint32_t* list = (uint32_t*)(ws + L.list);
    float* final_T = (float*)(ws + L.final_T);
    uint32_t* n_contrib = (uint32_t*)(ws + L.n_contrib);
    uint32_t* tile_max_contrib = (uint32_t*)(ws + L.tile_max_contrib);
    uint32_t* strip_last = (uint32_t*)(ws + L.strip_last);

    // one clear for the tile histogram and the (adjacent) instance-slot tickets: done by the call's first kernel when that is
    // the SH colour kernel (no memset node: 4 us), else by a memset
    const bool eager_sh = kp.P > 0 && shs && (kp.flags & S360_FLAG_SHARED_CAMPOS) && (!(kp.flags & S360_FLAG_FORWARD_ONLY) || kp.V >= 2 || rawin);
    const size_t zero_words_sz = (L.tile_start - L.tile_count) / 4;
    const bool fold_clear = eager_sh && zero_words_sz <= (size_t)kp.P;
    uint32_t* zero_ptr = fold_clear ? tile_count : (uint32_t*)nullptr;
    const int zero_words = fold_clear ? (int)zero_words_sz : 0;
    if (!fold_clear && hipMemsetAsync(tile_count, 0, L.tile_start - L.tile_count, st) != hipSuccess) return S360_E_LAUNCH;
    if (kp.P > 0) {
        {
        const int nblk = (kp.P + S360_BLOCK - 1) / S360_BLOCK;
        // the block's tile histogram: all images in LDS when they fit, else one image at a time, else global atomics
        const int lds_hist = (size_t)nt * 4 <= 48 * 1024 ? 1 : ((size_t)kp.T * 4 <= 48 * 1024 ? 2 : 0);
        const size_t hist_bytes = lds_hist == 2 ? (size_t)kp.T * 4 : (size_t)nt * 4;
        // views sharing one camera centre: SH colours once per Gaussian in their own streaming kernel.  Always in training
        // calls (the backward relies on sh_jac); in inference calls only when several views amortise the full-cloud read
        // (a single-face drop-in call sees ~17 % of the cloud and keeps the lazy in-kernel evaluation).
        const bool eager = eager_sh;
        float4* rgbc = (float4*)(ws + L.rgbc);
        if (eager) {
            ProfScope ps(PS_SH_EVAL, st);
            float* sh_jac = (float*)(ws + L.sh_jac);
            const bool jac = !(kp.flags & S360_FLAG_FORWARD_ONLY);
            const bool chm = (kp.flags & S360_FLAG_SH_CHANNEL_MAJOR) != 0;
#define S360_SHE(A, B) hipLaunchKernelGGL((k_sh_eval<A, B>), dim3(nblk), dim3(S360_BLOCK), 0, st, kp, views, means3D, shs, rgbc, sh_jac, zero_ptr, zero_words)
            if (rawin) {   // s360_forward_raw: geometry + colours straight from the encoder's raw records
                RawIn rin = *rawin;
                rin.geo7 = jac ? (float*)(ws + L.geo7) : nullptr;
                // workgroups are dealt per context view (grid.y): pose and SH rotation matrix are wave-uniform
                const dim3 rgrid((rin.Gv + SHE3_G - 1) / SHE3_G, (unsigned)(kp.P / rin.Gv));
                const bool rot = rin.sh_rot != nullptr;
#define S360_RAWE(J, R, M) hipLaunchKernelGGL((k_raw_eval<J, R, M>), rgrid, dim3(SHE3_G * 3), 0, st, kp, views, rin, rgbc, sh_jac, zero_ptr, zero_words)
                // S360_RAW_MFMA=1: the coefficient rotation on the matrix cores (v_mfma_f32_16x16x4_f32) instead of SGPR-operand
                // multiply-adds — same bits; measured side by side in DESIGN.md section 6
                static const bool raw_mfma = [] { const char* e = getenv("S360_RAW_MFMA"); return e && e[0] == '1'; }();
                if (rot && raw_mfma) { if (jac) S360_RAWE(true, true, true); else S360_RAWE(false, true, true); }
                else if (jac) { if (rot) S360_RAWE(true, true, false); else S360_RAWE(true, false, false); }
                else { if (rot) S360_RAWE(false, true, false); else S360_RAWE(false, false, false); }
#undef S360_RAWE
            } else if (chm && kp.M == 25 && kp.deg == 4) {  // the reference's harmonics: one lane per (Gaussian, channel)
                const int nb3 = (kp.P + SHE3_G - 1) / SHE3_G;
                if (jac) hipLaunchKernelGGL(k_sh_eval3_jac, dim3(nb3), dim3(SHE3_G * 3), 0, st, kp, views, means3D, shs, rgbc, sh_jac, zero_ptr, zero_words);
                else hipLaunchKernelGGL(k_sh_eval3, dim3(nb3), dim3(SHE3_G * 3), 0, st, kp, views, means3D, shs, rgbc, zero_ptr, zero_words);
            } else if (chm && jac) S360_SHE(true, true); else if (chm) S360_SHE(true, false); else if (jac) S360_SHE(false, true); else S360_SHE(false, false);
#undef S360_SHE
        }
        ProfScope ps(PS_PREPROCESS, st);
        if (shs) {
            const size_t lds = lds_hist ? hist_bytes : 0;
            if (eager && (kp.flags & S360_FLAG_COOP_WALK))
                hipLaunchKernelGGL((k_preprocess<true, true, true, true>), dim3(nblk), dim3(S360_BLOCK), lds, st, kp, views, means3D, cov6,
                                   opacities, shs, colors_precomp, radii, tiles_touched, recA, recB, recC, clamped, depths,
                                   tile_count, lds_hist, rgbc, vis_mask, slot_info);
            else if (eager)
                hipLaunchKernelGGL((k_preprocess<true, true, true>), dim3(nblk), dim3(S360_BLOCK), lds, st, kp, views, means3D, cov6,
                                   opacities, shs, colors_precomp, radii, tiles_touched, recA, recB, recC, clamped, depths,
                                   tile_count, lds_hist, rgbc, vis_mask, slot_info);
            else if (kp.flags & S360_FLAG_SH_CHANNEL_MAJOR)
                hipLaunchKernelGGL((k_preprocess<true, true>), dim3(nblk), dim3(S360_BLOCK), lds, st, kp, views, means3D, cov6,
                                   opacities, shs, colors_precomp, radii, tiles_touched, recA, recB, recC, clamped, depths,
                                   tile_count, lds_hist, rgbc, vis_mask, slot_info);
            else
                hipLaunchKernelGGL((k_preprocess<true, false>), dim3(nblk), dim3(S360_BLOCK), lds, st, kp, views, means3D, cov6,
                                   opacities, shs, colors_precomp, radii, tiles_touched, recA, recB, recC, clamped, depths,
                                   tile_count, lds_hist, rgbc, vis_mask, slot_info);
        } else {
            hipLaunchKernelGGL((k_preprocess<false, false>), dim3(nblk), dim3(S360_BLOCK), lds_hist ? hist_bytes : 0, st, kp, views,
                               means3D, cov6, opacities, shs, colors_precomp, radii, tiles_touched, recA, recB, recC, clamped,
                               depths, tile_count, lds_hist, rgbc, vis_mask, slot_info);
        }
        }
        S360_CHECK_LAUNCH();
    }
    {
        ProfScope ps(PS_TILE_SCAN, st);
        const bool split0 = kp.P > 0 && (kp.flags & S360_FLAG_SPLIT_LISTS);
        SegBufs sg0{split0 ? chunk_start : (const uint32_t*)nullptr, (uint32_t*)(ws + L.seg_flag), (uint32_t*)(ws + L.seg_arrive),
                    (uint32_t*)(ws + L.seg_arrive2), (uint32_t)seg_slots_of(prm),
                    (float4*)(ws + L.part_c), (float*)(ws + L.part_t), (float*)(ws + L.part_e), (uint32_t*)(ws + L.part_l), (uint32_t*)(ws + L.part_n),
                    (float4*)(ws + L.seg_c), (float*)(ws + L.seg_t), (uint32_t*)(ws + L.seg_cnt), (uint2*)(ws + L.seg_info), header};
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(TS_BLOCK), 0, st, tile_count, tile_start, tile_cursor, tile_max_contrib, nt, kp.cap, header,
                           chunk_start, (unsigned long long*)prm->header_mirror, sg0);
    }
    S360_CHECK_LAUNCH();
    if (kp.P > 0) {
        const dim3 egrid((kp.P + S360_BLOCK * EMIT_PPT - 1) / (S360_BLOCK * EMIT_PPT), kp.V);
        uint32_t* slot_pair = (kp.flags & S360_FLAG_FORWARD_ONLY) ? nullptr : (uint32_t*)(ws + L.slot_pair);
        {
        ProfScope ps(PS_EMIT, st);
        if ((size_t)kp.T * 8 <= 64 * 1024 && (kp.flags & S360_FLAG_COOP_WALK))
            hipLaunchKernelGGL((k_emit<true, true>), egrid, dim3(S360_BLOCK), (size_t)kp.T * 8, st, kp, tiles_touched, vis_mask, recA, recC,
                               depths, tile_start, tile_cursor, keys, slot_info, slot_pair, slot_ticket, header, (uint32_t*)(ws + L.long_pairs));
        else if ((size_t)kp.T * 8 <= 64 * 1024)
            hipLaunchKernelGGL(k_emit<true>, egrid, dim3(S360_BLOCK), (size_t)kp.T * 8, st, kp, tiles_touched, vis_mask, recA, recC,
                               depths, tile_start, tile_cursor, keys, slot_info, slot_pair, slot_ticket, header, (uint32_t*)(ws + L.long_pairs));
        else
            hipLaunchKernelGGL(k_emit<false>, egrid, dim3(S360_BLOCK), 0, st, kp, tiles_touched, vis_mask, recA, recC, depths, tile_start,
                               tile_cursor, keys, slot_info, slot_pair, slot_ticket, header, (uint32_t*)(ws + L.long_pairs));
        }
        S360_CHECK_LAUNCH();
        ProfScope ps(PS_SORT, st);
        {
            // Lists of up to 2 048 keys (the bulk) are sorted by one workgroup each; longer ones as 4 096-key chunks followed,
            // for multi-chunk lists, by `passes` global merge passes (k_merge_all).  The upper bound of the pass count is fixed on
            // the host (no read-back): enough for the longest possible list, capped at MAX_PASSES (4 096 << 4 = 65 536 keys);
            // passes no list of the call needs return at once (header[3]); anything longer falls through to the global network.
            const size_t cap_keys = kp.cap < (uint32_t)kp.P ? kp.cap : (size_t)kp.P;   // a tile holds a Gaussian at most once
            uint32_t passes = 0;
            while (passes < MAX_PASSES && ((size_t)SORT_CHUNK << passes) < cap_keys) ++passes;
            const uint32_t global_lo = SORT_CHUNK << passes;  // lists longer than this go to the global-memory network
            // the chunk blocks walk the chunk table grid-stride, so a moderate grid serves any count
            const unsigned cgrid = (unsigned)min((size_t)1024, (size_t)kp.cap / SORT_CHUNK + (size_t)kp.cap / SORT_SHORT + 2);
            const size_t lds512 = (size_t)(SORT_CHUNK + SORT_CHUNK / 8) * 8;
            hipLaunchKernelGGL(k_sort_stage1, dim3(cgrid + nt + 1), dim3(SORT_THREADS), lds512, st, tile_start, chunk_start, nt, keys, keys_alt, list,
                               kp.cap, passes, cgrid, tile_count, tile_order, header,
                               prm->header_mirror ? (unsigned long long*)prm->header_mirror + 2 : (unsigned long long*)nullptr);
            // everything after the chunk sorts in ONE launch: persistent workgroups, pass by pass behind per-tile completion
            // counters; passes no list of the call needs are never entered, the global-memory fallback runs in the same launch
            if (passes > 0) {
                const unsigned mgrid = cgrid < (unsigned)S360_MERGE_GRID ? cgrid : (unsigned)S360_MERGE_GRID;
                {   // 135 KB of dynamic LDS needs the opt-in, once per device (idempotent: a race sets it twice)
                    static bool attr_done[64] = {};
                    int dev = 0;
                    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_done[dev]) {
                        (void)hipFuncSetAttribute((const void*)k_merge_all, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSM_LDS_BYTES);
                        (void)hipGetLastError();
                        attr_done[dev] = true;
                    }
                }
                hipLaunchKernelGGL(k_merge_all, dim3(mgrid), dim3(SORT_THREADS), LDSM_LDS_BYTES, st, tile_start, chunk_start, nt, keys, keys_alt, list,
                                   kp.cap, passes, header, (uint32_t*)(ws + L.merge_done), (size_t)global_lo < cap_keys ? global_lo : 0xFFFFFFFFu);
            }
        }
        S360_CHECK_LAUNCH();
    }
    {
        ProfScope ps(PS_RENDER, st);
        const dim3 rgrid(nt), rblock(S360_BLOCK);
        const bool training = !(kp.flags & S360_FLAG_FORWARD_ONLY);
        float4* surv = training ? (float4*)(ws + L.surv) : nullptr;
        uint32_t* surv_count = training ? (uint32_t*)(ws + L.surv_count) : nullptr;
#ifdef S360_DBG_TIMING
        uint32_t* dbg = (uint32_t*)keys_alt;  // [4 * nt * 4] per-wave (start, duration ticks, HW_ID, XCC_ID): the merge buffer is free by now
#else
        uint32_t* dbg = header + 8;           // S360_DBG_COUNT counters
#endif
        // S360_FLAG_DEFER_LOSS (training calls with a loss epilogue): no reduction launch here — k_render leaves the pointers in the
        // header and the backward's first launch reduces (loss_out is complete once s360_backward* has run on this workspace)
        const bool defer = training && kp.P > 0 && ep.target && ep.loss_out && (kp.flags & S360_FLAG_DEFER_LOSS);
        uint32_t* hdr_loss = defer ? header + S360_HDR_LOSS : nullptr;
        // S360_FLAG_SPLIT_LISTS: quadrants still busy after SEG_HEAD entries of a long list stop there (k_render) and the rest of
        // the list is composited segment-parallel by a second launch (k_render_tail: returns at once when nothing split)
        const bool split = kp.P > 0 && (kp.flags & S360_FLAG_SPLIT_LISTS);
        SegBufs sg{split ? chunk_start : (const uint32_t*)nullptr, (uint32_t*)(ws + L.seg_flag), (uint32_t*)(ws + L.seg_arrive),
                   (uint32_t*)(ws + L.seg_arrive2), (uint32_t)seg_slots_of(prm),
                   (float4*)(ws + L.part_c), (float*)(ws + L.part_t), (float*)(ws + L.part_e), (uint32_t*)(ws + L.part_l), (uint32_t*)(ws + L.part_n),
                   (float4*)(ws + L.seg_c), (float*)(ws + L.seg_t), (uint32_t*)(ws + L.seg_cnt), (uint2*)(ws + L.seg_info), header};
        unsigned long long* cand = prm->header_mirror ? (unsigned long long*)prm->header_mirror + 1 : nullptr;   // word 1 of the mirror
        const SegBufs* sgp = split ? (const SegBufs*)(header + S360_HDR_SEGBUFS) : (const SegBufs*)nullptr;
        // S360_FLAG_SPLIT_LISTS: the segment work runs INSIDE k_render's launch (phase-1 workers behind the long tiles, phase-2 workers
        // behind all tiles); k_render_tail behind it takes whatever no phase-2 worker claimed (normally nothing)
        if (split && hipMemsetAsync(ws + L.seg_info, 0, seg_slots_of(prm) * 4 * 8, st) != hipSuccess) return S360_E_LAUNCH;   // a published work item is non-zero
#define S360_LAUNCH_RENDER(WD, SP)                                                                                                         \
    hipLaunchKernelGGL((k_render<WD, SP>), SP ? dim3(nt + S360_P1_GRID + S360_P2_GRID) : rgrid, rblock, 0, st, kp, views, tile_start, list, recA, recB, recC, images, final_T, n_contrib,      \
                       tile_max_contrib, strip_last, dbg, depths, depth_maps, depth_mode, ep, kp.P > 0 ? tile_order : (const uint32_t*)nullptr, \
                       surv, surv_count, hdr_loss, sgp, chunk_start, cand, nt, list, depths)
        if (depth_maps) { if (split) S360_LAUNCH_RENDER(true, true); else S360_LAUNCH_RENDER(true, false); }
        else { if (split) S360_LAUNCH_RENDER(false, true); else S360_LAUNCH_RENDER(false, false); }
#undef S360_LAUNCH_RENDER
        if (split) {
            const unsigned tgrid = (unsigned)min((size_t)S360_TAIL_GRID, seg_slots_of(prm));
            if (depth_maps)
                hipLaunchKernelGGL(k_render_tail<true>, dim3(tgrid), rblock, 0, st, kp, views, tile_start, list, recA, images, final_T, n_contrib,
                                   tile_max_contrib, strip_last, depths, depth_maps, depth_mode, ep, surv, surv_count, sg, nt, dbg);
            else
                hipLaunchKernelGGL(k_render_tail<false>, dim3(tgrid), rblock, 0, st, kp, views, tile_start, list, recA, images, final_T, n_contrib,
                                   tile_max_contrib, strip_last, depths, depth_maps, depth_mode, ep, surv, surv_count, sg, nt, dbg);
        }
        if (ep.target && ep.loss_out && !defer)
            hipLaunchKernelGGL(k_mse_finish, dim3(1), dim3(MSE_BLOCK), 0, st, ep.partials, kp.T * 4, kp.V, 0.5f * ep.grad_scale,
                               1.0f / (3.0f * (float)kp.H * (float)kp.W), ep.loss_out);
    }
    S360_CHECK_LAUNCH();
    return S360_OK;
}

extern "C" int s360_forward(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                            const float* opacities, const float* shs, const float* colors_precomp, float* images,
                            int32_t* radii, void* workspace, size_t workspace_bytes, void* stream_) {
    return forward_impl(prm, views, means3D, cov6, opacities, shs, colors_precomp, images, nullptr, 0, radii, workspace,
                        workspace_bytes, stream_);
}

extern "C" int s360_forward_depth(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                                  const float* opacities, const float* shs, const float* colors_precomp, float* images,
                                  float* depth_maps, int32_t depth_mode, int32_t* radii, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    if (!depth_maps || depth_mode < 0 || depth_mode > 3) return S360_E_BADARG;
    return forward_impl(prm, views, means3D, cov6, opacities, shs, colors_precomp, images, depth_maps, depth_mode, radii,
                        workspace, workspace_bytes, stream_);
}

extern "C" int s360_forward_mse(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                                const float* opacities, const float* shs, const float* colors_precomp, float* images,
                                float* depth_maps, int32_t depth_mode, int32_t* radii, const float* target,
                                float grad_scale, float* d_images, float* partials, float* loss_out, void* workspace,
                                size_t workspace_bytes, void* stream_) {
    if (!target || !d_images || !partials) return S360_E_BADARG;
    if (depth_maps && (depth_mode < 0 || depth_mode > 3)) return S360_E_BADARG;
    return forward_impl(prm, views, means3D, cov6, opacities, shs, colors_precomp, images, depth_maps, depth_mode, radii,
                        workspace, workspace_bytes, stream_, s360::MseEp{target, d_images, partials, grad_scale, loss_out});
}

extern "C" int s360_forward_raw(const S360Params* prm, const S360View* views, const S360RawInputs* raw, const float* opacities,
                                float* means_out, float* cov6_out, float* images, float* depth_maps, int32_t depth_mode, int32_t* radii,
                                const float* target, float grad_scale, float* d_images, float* partials, float* loss_out, void* workspace,
                                size_t workspace_bytes, void* stream_) {
    if (!prm || !raw || !means_out || !cov6_out) return S360_E_BADARG;
    if (!(prm->flags & S360_FLAG_RAW_INPUTS) || !(prm->flags & S360_FLAG_SHARED_CAMPOS) || (prm->flags & (S360_FLAG_COV9 | S360_FLAG_SPHERICAL)))
        return S360_E_BADARG;   // one camera centre per call; the geometry pass reads the 6-entry covariances this call writes
    if (prm->M != 25 || prm->sh_degree != 4) return S360_E_UNSUPPORTED;   // the reference's configuration (costvolume.yaml:16)
    if (prm->P > 0 && (!raw->extrinsics || !raw->depths || !raw->raw_gaussians)) return S360_E_BADARG;
    if (raw->n_views < 0 || raw->per_view < 0 || raw->H < 1 || raw->W < 1 || raw->per_ray < 1 ||
        (long long)raw->n_views * raw->per_view != (long long)prm->P || (long long)raw->H * raw->W * raw->per_ray != (long long)raw->per_view)
        return S360_E_BADARG;
    if (raw->erp_convention < 0 || raw->erp_convention > 3 || (raw->erp_convention != 0 && (raw->H < 2 || raw->W < 2))) return S360_E_BADARG;
    if (depth_maps && (depth_mode < 0 || depth_mode > 3)) return S360_E_BADARG;
    if (target && (!d_images || !partials)) return S360_E_BADARG;
    s360::RawIn rin{raw->extrinsics, raw->depths, raw->raw_gaussians, raw->sh_rotation, means_out, cov6_out, nullptr,
                    raw->per_view > 0 ? raw->per_view : 1, raw->H, raw->W, raw->per_ray, raw->erp_convention, raw->scale_min, raw->scale_max, raw->eps};
    const float* dummy_sh = reinterpret_cast<const float*>(raw->raw_gaussians ? raw->raw_gaussians : (const float*)workspace);   // never read: the colours come from k_raw_eval
    return forward_impl(prm, views, means_out, cov6_out, opacities, dummy_sh, nullptr, images, depth_maps, depth_mode, radii, workspace,
                        workspace_bytes, stream_, target ? s360::MseEp{target, d_images, partials, grad_scale, loss_out}
                                                         : s360::MseEp{nullptr, nullptr, nullptr, 0.f, nullptr}, &rin);
}
