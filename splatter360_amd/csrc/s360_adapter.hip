// s360_adapter.hip — fused Gaussian-adapter tail (forward + backward).  gfx950 only.
//
// The step of the reference's encoder that PRODUCES the per-Gaussian buffers the rasteriser reads
// (src/model/encoder/common/gaussian_adapter_erp.py:50-119), one thread per Gaussian:
//   scales   = (smin + (smax - smin) sigmoid(raw_s)) * depth / max(w, h)                      (:63-78)
//   q^       = q / (|q| + eps)                                                                (:82)
//   Sigma    = C R(q^) diag(scales^2) R(q^)^T C^T   (R: gaussians.py:8-31, xyzw; C = c2w rotation)   (:89-92)
//   mean     = C (dir(pixel) * depth) + t           (sphere_projection.py:6-86, utils360.py:93-104,148-153)
//   harmonics= D_l (sh * sh_mask) per degree l      (:38-47,86; rotate_sh, src/misc/sh_rotation.py:10-30 — the
//              Wigner-D blocks, one d_sh x d_sh matrix per view, come from k_sh_rotation_blocks below or from the caller)
// instead of ~20 torch launches with their [G,3,3] intermediates.  grid.y = view, so everything that depends on the
// view only (pose, SH rotation blocks) is wave-uniform.  Streaming: 336 B read, 352 B (cov6: 340 B) written per Gaussian.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s360.h"
#include "s360_shrot_tables.h"
#include "s360_adapter_math.h"

namespace s360 {

__global__ __launch_bounds__(256) void k_adapter_fwd(AdapterParams ap, const float* __restrict__ extrinsics,
                                                     const float* __restrict__ depths, const float* __restrict__ raw,
                                                     const float* __restrict__ sh_rot, float* __restrict__ means,
                                                     float* __restrict__ cov, float* __restrict__ harmonics,
                                                     float* __restrict__ scales_out, float* __restrict__ rot_out) {
    const int v = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= ap.Gv) return;
    const size_t i = (size_t)v * ap.Gv + g;
    const float* E = extrinsics + 16 * v;  // wave-uniform
    const int c_in = 7 + 3 * ap.d_sh;
    const float* rw = raw + i * c_in;
    const float depth = depths[i];
    // scales
    const float px = 1.0f / (float)max(ap.W, ap.H);
    float s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = ((ap.smin + (ap.smax - ap.smin) * sigmoidf(rw[k])) * depth) * px;
    // rotation
    QuatGeom qg;
    const float qr[4] = {rw[3], rw[4], rw[5], rw[6]};
    quat_geom(qr, ap.eps, qg);
    // M = C R ; Sigma = M diag(s^2) M^T
    float M[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) M[a][b] = E[4 * a] * qg.R[0][b] + E[4 * a + 1] * qg.R[1][b] + E[4 * a + 2] * qg.R[2][b];
    const float s2[3] = {s[0] * s[0], s[1] * s[1], s[2] * s[2]};
    float S[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b) S[a][b] = S[b][a] = M[a][0] * s2[0] * M[b][0] + M[a][1] * s2[1] * M[b][1] + M[a][2] * s2[2] * M[b][2];
    if (ap.cov9) {
        float* o = cov + 9 * i;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) o[3 * a + b] = S[a][b];
    } else {
        float* o = cov + 6 * i;
        o[0] = S[0][0]; o[1] = S[0][1]; o[2] = S[0][2]; o[3] = S[1][1]; o[4] = S[1][2]; o[5] = S[2][2];
    }
    // mean
    float d[3];
    erp_dir(g / ap.per_ray, ap.H, ap.W, ap.conv, d);
    const float p[3] = {d[0] * depth, d[1] * depth, d[2] * depth};
#pragma unroll
    for (int a = 0; a < 3; ++a) means[3 * i + a] = (E[4 * a] * p[0] + E[4 * a + 1] * p[1] + E[4 * a + 2] * p[2]) + E[4 * a + 3];
    if (scales_out) {
        scales_out[3 * i] = s[0]; scales_out[3 * i + 1] = s[1]; scales_out[3 * i + 2] = s[2];
    }
    if (rot_out) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rot_out[4 * i + k] = qg.q[k];
    }
    // harmonics: per colour channel, per degree: D_l (sh * mask)
    const int deg = ap.d_sh == 25 ? 4 : ap.d_sh == 16 ? 3 : ap.d_sh == 9 ? 2 : ap.d_sh == 4 ? 1 : 0;
    const float* D = sh_rot ? sh_rot + (size_t)v * ap.d_sh * ap.d_sh : nullptr;  // wave-uniform
    for (int c = 0; c < 3; ++c) {
        const float* src = rw + 7 + c * ap.d_sh;
        float* dst = harmonics + (i * 3 + c) * ap.d_sh;
        float mask = 1.0f;
        for (int l = 0; l <= deg; ++l) {
            const int o = l * l, nl = 2 * l + 1;
            mask = kShMask[l];
            for (int a = 0; a < nl; ++a) {
                float acc;
                if (D) {
                    acc = 0.f;
                    for (int b = 0; b < nl; ++b) acc += D[(o + a) * ap.d_sh + o + b] * (src[o + b] * mask);
                } else {
                    acc = src[o + a] * mask;
                }
                dst[o + a] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_adapter_bwd(AdapterParams ap, const float* __restrict__ extrinsics,
                                                     const float* __restrict__ depths, const float* __restrict__ raw,
                                                     const float* __restrict__ sh_rot, const float* __restrict__ d_means,
                                                     const float* __restrict__ d_cov, const float* __restrict__ d_harm,
                                                     float* __restrict__ d_depths, float* __restrict__ d_raw) {
    const int v = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= ap.Gv) return;
    const size_t i = (size_t)v * ap.Gv + g;
    const float* E = extrinsics + 16 * v;
    const int c_in = 7 + 3 * ap.d_sh;
    const float* rw = raw + i * c_in;
    float* dr = d_raw + i * c_in;
    const float depth = depths[i];
    const float px = 1.0f / (float)max(ap.W, ap.H);
    float sig[3], base[3], s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sig[k] = sigmoidf(rw[k]);
        base[k] = ap.smin + (ap.smax - ap.smin) * sig[k];
        s[k] = (base[k] * depth) * px;
    }
    QuatGeom qg;
    const float qr[4] = {rw[3], rw[4], rw[5], rw[6]};
    quat_geom(qr, ap.eps, qg);
    float M[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) M[a][b] = E[4 * a] * qg.R[0][b] + E[4 * a + 1] * qg.R[1][b] + E[4 * a + 2] * qg.R[2][b];
    // G = dL/dSigma as a full 3x3 (cov6: the off-diagonal entries stand for one variable each -> upper triangle only)
    float G[3][3];
    if (ap.cov9) {
        const float* gcv = d_cov + 9 * i;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) G[a][b] = gcv[3 * a + b];
    } else {
        const float* gcv = d_cov + 6 * i;
        G[0][0] = gcv[0]; G[0][1] = gcv[1]; G[0][2] = gcv[2]; G[1][1] = gcv[3]; G[1][2] = gcv[4]; G[2][2] = gcv[5];
        G[1][0] = G[2][0] = G[2][1] = 0.f;
    }
    // Sigma = M D M^T, D = diag(s^2):  dM = (G + G^T) M D,  dD_k = (M^T G M)_kk
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
        for (int a = 0; a < 3; ++a) t[a] = Gs[a][0] * M[0][k] + Gs[a][1] * M[1][k] + Gs[a][2] * M[2][k];  // (Gs M)[:,k]
        const float s2 = s[k] * s[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) dM[a][k] = t[a] * s2;
        // (M^T G M)_kk = 1/2 (M^T Gs M)_kk
        ds[k] = 2.0f * s[k] * (0.5f * (M[0][k] * t[0] + M[1][k] * t[1] + M[2][k] * t[2]));
    }
    // dR = C^T dM
    float dR[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) dR[a][b] = E[a] * dM[0][b] + E[4 + a] * dM[1][b] + E[8 + a] * dM[2][b];
    // R(q^) -> q^   (a = two_s depends on q^ as well)
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
    const float da = -a * a * dLda;  // da/dq^ = -a^2 q^
#pragma unroll
    for (int k = 0; k < 4; ++k) dq[k] += da * qg.q[k];
    // q^ = q / (|q| + eps):  dq = dq^ / m - q (q . dq^) / (n m^2)
    const float dot = qr[0] * dq[0] + qr[1] * dq[1] + qr[2] * dq[2] + qr[3] * dq[3];
    const float f = qg.n > 0.f ? dot / (qg.n * qg.m * qg.m) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) dr[3 + k] = dq[k] / qg.m - qr[k] * f;
    // scales and depth
    float dd = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dr[k] = ds[k] * depth * px * (ap.smax - ap.smin) * sig[k] * (1.0f - sig[k]);
        dd += ds[k] * base[k] * px;
    }
    float d[3];
    erp_dir(g / ap.per_ray, ap.H, ap.W, ap.conv, d);
    if (d_means) {  // opt-in: the reference un-projects under torch.no_grad() (sphere_projection.py:14-86) — its means are
                    // detached and depth receives gradient through the scales only (d_means == NULL, the default)
        const float* gm = d_means + 3 * i;
#pragma unroll
        for (int x = 0; x < 3; ++x) dd += (E[x] * gm[0] + E[4 + x] * gm[1] + E[8 + x] * gm[2]) * d[x];  // (C^T dmean) . dir
    }
    d_depths[i] = dd;
    // harmonics
    const int deg = ap.d_sh == 25 ? 4 : ap.d_sh == 16 ? 3 : ap.d_sh == 9 ? 2 : ap.d_sh == 4 ? 1 : 0;
    const float* D = sh_rot ? sh_rot + (size_t)v * ap.d_sh * ap.d_sh : nullptr;
    for (int c = 0; c < 3; ++c) {
        const float* gh = d_harm + (i * 3 + c) * ap.d_sh;
        float* dst = dr + 7 + c * ap.d_sh;
        float mask = 1.0f;
        for (int l = 0; l <= deg; ++l) {
            const int o = l * l, nl = 2 * l + 1;
            mask = kShMask[l];
            for (int b = 0; b < nl; ++b) {
                float acc;
                if (D) {
                    acc = 0.f;
                    for (int a2 = 0; a2 < nl; ++a2) acc += D[(o + a2) * ap.d_sh + o + b] * gh[o + a2];
                } else {
                    acc = gh[o + b];
                }
                dst[o + b] = acc * mask;
            }
        }
    }
}

// ---- degree-4 fast path (d_sh = 25: the reference's configuration, costvolume.yaml:16) — round 6 -----------------------------------
// The generic kernels above give a thread one Gaussian and let it walk its 328-byte record with 4-byte accesses 328 bytes apart (and
// write 300 bytes of harmonics the same way): 9.05 ms / 6.51 ms per million Gaussians, ~1 % of HBM (profiles/r05_adapter_kernel_stats.csv).
// Here a workgroup of three waves takes 64 consecutive Gaussians of ONE view (grid.y = view): the 64 records are 20 992 contiguous
// bytes, staged into LDS with coalesced non-temporal 16-byte loads; wave c rotates colour channel c's 25 coefficients IN PLACE with
// the view's Wigner-D matrix in SGPRs (sh_rotate_coefs25); the 64 x 75 harmonics leave through coalesced non-temporal 16-byte stores;
// the geometry (scale map, quaternion -> Sigma, sphere un-projection: the very expressions of k_adapter_fwd) rides on the waves behind
// their channel, writing its 48 (cov6: 36) bytes per Gaussian directly.
constexpr int AD_G = 64, AD_C = 82, AD_T = 192;
typedef float ad_f4v __attribute__((ext_vector_type(4)));

template <bool ROT>
__global__ __launch_bounds__(AD_T) void k_adapter_fwd25(AdapterParams ap, const float* __restrict__ extrinsics,
                                                       const float* __restrict__ depths, const float* __restrict__ raw,
                                                       const float* __restrict__ sh_rot, float* __restrict__ means,
                                                       float* __restrict__ cov, float* __restrict__ harmonics,
                                                       float* __restrict__ scales_out, float* __restrict__ rot_out) {
    __shared__ __attribute__((aligned(16))) float s_raw[7 * AD_T * 4];   // 64 x 82 floats (+ pad to 7 rounds of 192 float4)
    const int tid = threadIdx.x, d = tid >> 6, l = tid & 63;
    const int v = blockIdx.y, gi0 = blockIdx.x * AD_G;
    const int nb = min(AD_G, ap.Gv - gi0);
    const size_t i0 = (size_t)v * ap.Gv + gi0;
    {
        const float* src = raw + i0 * AD_C;
        if ((((uintptr_t)src) & 15) == 0 && nb == AD_G) {
            float4* d4 = reinterpret_cast<float4*>(s_raw);
            float4 q[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) {   // unconditional loads (a guarded load compiles to a branch + full wait per round)
                const ad_f4v t = __builtin_nontemporal_load(reinterpret_cast<const ad_f4v*>(src) + min(tid + r * AD_T, AD_G * AD_C / 4 - 1));
                q[r] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) d4[tid + r * AD_T] = q[r];
        } else {
            for (int i = tid; i < nb * AD_C; i += AD_T) s_raw[i] = src[i];
        }
    }
    const bool live = l < nb;
    const size_t i = i0 + l;
    const float* E = extrinsics + 16 * v;  // wave-uniform
    float depth = 0.f;
    if (live) depth = depths[i];
    if (live && d == 0) {
        // mean (in flight with the staging loads)
        float dr[3];
        erp_dir((gi0 + l) / ap.per_ray, ap.H, ap.W, ap.conv, dr);
        const float p[3] = {dr[0] * depth, dr[1] * depth, dr[2] * depth};
#pragma unroll
        for (int a = 0; a < 3; ++a) means[3 * i + a] = (E[4 * a] * p[0] + E[4 * a + 1] * p[1] + E[4 * a + 2] * p[2]) + E[4 * a + 3];
    }
    __syncthreads();
    float* rec = s_raw + l * AD_C;
    if (live) {
        float c[25], h[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) c[k] = rec[7 + 25 * d + k];
        sh_rotate_coefs25<ROT>(ROT ? sh_rot + (size_t)v * 625 : nullptr, c, h);
#pragma unroll
        for (int k = 0; k < 25; ++k) rec[7 + 25 * d + k] = h[k];
    }
    if (live && d == 1) {
        const float px = 1.0f / (float)max(ap.W, ap.H);
        float s[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] = ((ap.smin + (ap.smax - ap.smin) * sigmoidf(rec[k])) * depth) * px;
        QuatGeom qg;
        const float qr[4] = {rec[3], rec[4], rec[5], rec[6]};
        quat_geom(qr, ap.eps, qg);
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a][b] = E[4 * a] * qg.R[0][b] + E[4 * a + 1] * qg.R[1][b] + E[4 * a + 2] * qg.R[2][b];
        const float s2[3] = {s[0] * s[0], s[1] * s[1], s[2] * s[2]};
        float S[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) S[a][b] = S[b][a] = M[a][0] * s2[0] * M[b][0] + M[a][1] * s2[1] * M[b][1] + M[a][2] * s2[2] * M[b][2];
        if (ap.cov9) {
            float* o = cov + 9 * i;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) o[3 * a + b] = S[a][b];
        } else {
            float* o = cov + 6 * i;
            o[0] = S[0][0]; o[1] = S[0][1]; o[2] = S[0][2]; o[3] = S[1][1]; o[4] = S[1][2]; o[5] = S[2][2];
        }
        if (scales_out) {
            scales_out[3 * i] = s[0]; scales_out[3 * i + 1] = s[1]; scales_out[3 * i + 2] = s[2];
        }
        if (rot_out) {
#pragma unroll
            for (int k = 0; k < 4; ++k) rot_out[4 * i + k] = qg.q[k];
        }
    }
    __syncthreads();
    // the block's 64 x 75 harmonics are contiguous in global memory: 1 200 float4 per full block, gathered from the 82-word records
    float* dst = harmonics + i0 * 75;
    const int nfl = nb * 75;
    if ((((uintptr_t)dst) & 15) == 0) {
        const int n4 = nfl >> 2;
        for (int q = tid; q < n4; q += AD_T) {
            ad_f4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = 4 * q + e, gl = idx / 75, j = idx - 75 * gl;
                o[e] = s_raw[gl * AD_C + 7 + j];
            }
            __builtin_nontemporal_store(o, reinterpret_cast<ad_f4v*>(dst) + q);
        }
        for (int idx = (n4 << 2) + tid; idx < nfl; idx += AD_T) dst[idx] = s_raw[(idx / 75) * AD_C + 7 + idx % 75];
    } else {
        for (int idx = tid; idx < nfl; idx += AD_T) dst[idx] = s_raw[(idx / 75) * AD_C + 7 + idx % 75];
    }
}

// Backward of the above: dL/dharmonics (64 x 75 contiguous floats) staged with coalesced non-temporal loads straight into the layout
// of the OUTPUT records; wave c applies D^T and the mask to channel c in place; wave 0 first runs k_adapter_bwd's geometry chain on
// the seven raw geometry words (read straight from the record: 28 of its 328 bytes) into words 0..6; the 64 gradient records
// (20 992 contiguous bytes) leave through coalesced non-temporal 16-byte stores.
template <bool ROT>
__global__ __launch_bounds__(AD_T) void k_adapter_bwd25(AdapterParams ap, const float* __restrict__ extrinsics,
                                                       const float* __restrict__ depths, const float* __restrict__ raw,
                                                       const float* __restrict__ sh_rot, const float* __restrict__ d_means,
                                                       const float* __restrict__ d_cov, const float* __restrict__ d_harm,
                                                       float* __restrict__ d_depths, float* __restrict__ d_raw) {
    __shared__ __attribute__((aligned(16))) float s_out[AD_G * AD_C];
    const int tid = threadIdx.x, d = tid >> 6, l = tid & 63;
    const int v = blockIdx.y, gi0 = blockIdx.x * AD_G;
    const int nb = min(AD_G, ap.Gv - gi0);
    const size_t i0 = (size_t)v * ap.Gv + gi0;
    {
        const float* src = d_harm + i0 * 75;
        const int nfl = nb * 75;
        if ((((uintptr_t)src) & 15) == 0 && nb == AD_G) {
            ad_f4v q[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) q[r] = __builtin_nontemporal_load(reinterpret_cast<const ad_f4v*>(src) + min(tid + r * AD_T, AD_G * 75 / 4 - 1));
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int q4 = tid + r * AD_T;
                if (q4 < AD_G * 75 / 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = 4 * q4 + e, gl = idx / 75, j = idx - 75 * gl;
                        s_out[gl * AD_C + 7 + j] = q[r][e];
                    }
                }
            }
        } else {
            for (int idx = tid; idx < nfl; idx += AD_T) s_out[(idx / 75) * AD_C + 7 + idx % 75] = src[idx];
        }
    }
    const bool live = l < nb;
    const size_t i = i0 + l;
    const float* E = extrinsics + 16 * v;
    float* rec = s_out + l * AD_C;
    if (live && d == 0) {
        const float* rw = raw + i * AD_C;
        const float depth = depths[i];
        const float px = 1.0f / (float)max(ap.W, ap.H);
        float sig[3], base[3], s[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sig[k] = sigmoidf(rw[k]);
            base[k] = ap.smin + (ap.smax - ap.smin) * sig[k];
            s[k] = (base[k] * depth) * px;
        }
        QuatGeom qg;
        const float qr[4] = {rw[3], rw[4], rw[5], rw[6]};
        quat_geom(qr, ap.eps, qg);
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a][b] = E[4 * a] * qg.R[0][b] + E[4 * a + 1] * qg.R[1][b] + E[4 * a + 2] * qg.R[2][b];
        float G[3][3];
        if (ap.cov9) {
            const float* gcv = d_cov + 9 * i;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) G[a][b] = gcv[3 * a + b];
        } else {
            const float* gcv = d_cov + 6 * i;
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
            const float s2 = s[k] * s[k];
#pragma unroll
            for (int a = 0; a < 3; ++a) dM[a][k] = t[a] * s2;
            ds[k] = 2.0f * s[k] * (0.5f * (M[0][k] * t[0] + M[1][k] * t[1] + M[2][k] * t[2]));
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
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[3 + k] = dq[k] / qg.m - qr[k] * f;
        float dd = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rec[k] = ds[k] * depth * px * (ap.smax - ap.smin) * sig[k] * (1.0f - sig[k]);
            dd += ds[k] * base[k] * px;
        }
        if (d_means) {   // opt-in: the reference un-projects under torch.no_grad() (sphere_projection.py:14-86)
            float dir[3];
            erp_dir((gi0 + l) / ap.per_ray, ap.H, ap.W, ap.conv, dir);
            const float* gm = d_means + 3 * i;
#pragma unroll
            for (int x = 0; x < 3; ++x) dd += (E[x] * gm[0] + E[4 + x] * gm[1] + E[8 + x] * gm[2]) * dir[x];
        }
        d_depths[i] = dd;
    }
    __syncthreads();   // the staged dL/dharmonics
    if (live) {
        float gh[25], o[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) gh[k] = rec[7 + 25 * d + k];
        sh_rotate_basis25<ROT, -1>(ROT ? sh_rot + (size_t)v * 625 : nullptr, gh, o);
#pragma unroll
        for (int k = 0; k < 25; ++k) rec[7 + 25 * d + k] = o[k];
    }
    __syncthreads();
    const int nfl = nb * AD_C;
    float* dst = d_raw + i0 * AD_C;
    if ((((uintptr_t)dst) & 15) == 0) {
        const int n4 = nfl >> 2;
        for (int q = tid; q < n4; q += AD_T) __builtin_nontemporal_store(reinterpret_cast<const ad_f4v*>(s_out)[q], reinterpret_cast<ad_f4v*>(dst) + q);
        for (int q = (n4 << 2) + tid; q < nfl; q += AD_T) dst[q] = s_out[q];
    } else {
        for (int q = tid; q < nfl; q += AD_T) dst[q] = s_out[q];
    }
}

// ---- rotate_sh's matrices (src/misc/sh_rotation.py:19-24: wigner_D(l, *matrix_to_angles(R)) per degree) --------------------
// D^l(R) is defined by Y^l(R d) = D^l(R) Y^l(d) in e3nn's real basis (polar axis y, azimuth from z towards x, m = -l..l, no
// Condon-Shortley phase: l = 1 is (x, y, z), so D^1 = R).  One thread per (view, degree): evaluate Y^l at the 2l+1 tabulated
// directions rotated by R and multiply by the tabulated inverse of the unrotated basis matrix (s360_shrot_tables.h).  float64:
// the work is a few hundred flops per view and the blocks multiply every Gaussian's coefficients.
__device__ inline void e3nn_sh(int l, double x, double y, double z, double* out) {
    double q[5];  // d^m P_l / dy^m
    switch (l) {
        case 0: q[0] = 1.0; break;
        case 1: q[0] = y; q[1] = 1.0; break;
        case 2: q[0] = (3.0 * y * y - 1.0) * 0.5; q[1] = 3.0 * y; q[2] = 3.0; break;
        case 3: q[0] = (5.0 * y * y * y - 3.0 * y) * 0.5; q[1] = (15.0 * y * y - 3.0) * 0.5; q[2] = 15.0 * y; q[3] = 15.0; break;
        default: q[0] = (35.0 * y * y * y * y - 30.0 * y * y + 3.0) * 0.125; q[1] = (35.0 * y * y * y - 15.0 * y) * 0.5;
                 q[2] = (105.0 * y * y - 15.0) * 0.5; q[3] = 105.0 * y; q[4] = 105.0; break;
    }
    out[l] = q[0];
    double re = 1.0, im = 0.0, fact = 1.0;  // (z + i x)^m; fact = (l+m)!/(l-m)!
    for (int m = 1; m <= l; ++m) {
        const double nre = re * z - im * x, nim = re * x + im * z;
        re = nre; im = nim;
        fact *= (double)((l + m) * (l - m + 1));
        const double n = sqrt(2.0 / fact);
        out[l - m] = n * q[m] * im;
        out[l + m] = n * q[m] * re;
    }
}

__global__ void k_sh_rotation_blocks(const float* __restrict__ rotations, int stride, int n_views, int d_sh, int deg,
                                     float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = t / 5, l = t - 5 * v;
    if (v >= n_views) return;
    float* o = out + (size_t)v * d_sh * d_sh;
    if (l == 0)  // this thread also clears the off-block entries of the view's matrix
        for (int i = 0; i < d_sh; ++i)
            for (int j = 0; j < d_sh; ++j) {
                int li = 0, lj = 0;
                while ((li + 1) * (li + 1) <= i) ++li;
                while ((lj + 1) * (lj + 1) <= j) ++lj;
                if (li != lj) o[i * d_sh + j] = 0.f;
            }
    if (l > deg) return;
    const float* Rm = rotations + (size_t)v * stride;  // row-major 3x3, row stride 3 (stride 9) or 4 (a [4,4] pose, stride 16)
    const int rs = stride == 16 ? 4 : 3;
    const int n = 2 * l + 1;
    const double(*dirs)[3] = l == 0 ? kShRotDir0 : l == 1 ? kShRotDir1 : l == 2 ? kShRotDir2 : l == 3 ? kShRotDir3 : kShRotDir4;
    double B[9][9];  // B[m][i] = Y_m(R d_i)
    for (int i = 0; i < n; ++i) {
        const double dx = dirs[i][0], dy = dirs[i][1], dz = dirs[i][2];
        double u[3];
        for (int a = 0; a < 3; ++a) u[a] = (double)Rm[a * rs] * dx + (double)Rm[a * rs + 1] * dy + (double)Rm[a * rs + 2] * dz;
        const double inv = 1.0 / sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);  // R is only float32-orthonormal
        double y[9];
        e3nn_sh(l, u[0] * inv, u[1] * inv, u[2] * inv, y);
        for (int m = 0; m < n; ++m) B[m][i] = y[m];
    }
    for (int m = 0; m < n; ++m)
        for (int m2 = 0; m2 < n; ++m2) {
            double acc = 0.0;
            for (int i = 0; i < n; ++i) {
                const double a = l == 0 ? kShRotInv0[i][m2] : l == 1 ? kShRotInv1[i][m2] : l == 2 ? kShRotInv2[i][m2]
                                 : l == 3 ? kShRotInv3[i][m2] : kShRotInv4[i][m2];
                acc += B[m][i] * a;
            }
            o[(l * l + m) * d_sh + l * l + m2] = (float)acc;
        }
}

}  // namespace s360


using namespace s360;

static int adapter_args_ok(const float* extrinsics, const float* depths, const float* raw, int V, int Gv, int H, int W,
                           int per_ray, int d_sh) {
    if (!extrinsics || !depths || !raw || V < 0 || Gv < 0 || H < 1 || W < 1 || per_ray < 1) return 0;
    if (d_sh != 1 && d_sh != 4 && d_sh != 9 && d_sh != 16 && d_sh != 25) return 0;
    if ((long long)H * W * per_ray != (long long)Gv) return 0;
    return 1;
}

extern "C" int s360_adapter_forward(const float* extrinsics, const float* depths, const float* raw_gaussians,
                                    const float* sh_rotation, int32_t n_views, int32_t per_view, int32_t H, int32_t W,
                                    int32_t per_ray, int32_t d_sh, float scale_min, float scale_max, float eps,
                                    float* means, float* covariances, int32_t cov9, float* harmonics, float* scales_out,
                                    float* rotations_out, int32_t erp_convention, void* stream) {
    if (erp_convention < 0 || erp_convention > 3 || (erp_convention != 0 && (H < 2 || W < 2))) return S360_E_BADARG;
    if (!adapter_args_ok(extrinsics, depths, raw_gaussians, n_views, per_view, H, W, per_ray, d_sh) || !means || !covariances ||
        !harmonics)
        return S360_E_BADARG;
    if (n_views == 0 || per_view == 0) return S360_OK;
    AdapterParams ap = {n_views, per_view, H, W, per_ray, d_sh, cov9, erp_convention, scale_min, scale_max, eps};
    if (d_sh == 25) {   // degree 4: LDS-staged, coalesced (k_adapter_fwd25)
        const dim3 grid((per_view + AD_G - 1) / AD_G, n_views);
        if (sh_rotation)
            hipLaunchKernelGGL(k_adapter_fwd25<true>, grid, dim3(AD_T), 0, (hipStream_t)stream, ap, extrinsics, depths, raw_gaussians,
                               sh_rotation, means, covariances, harmonics, scales_out, rotations_out);
        else
            hipLaunchKernelGGL(k_adapter_fwd25<false>, grid, dim3(AD_T), 0, (hipStream_t)stream, ap, extrinsics, depths, raw_gaussians,
                               sh_rotation, means, covariances, harmonics, scales_out, rotations_out);
    } else
    hipLaunchKernelGGL(k_adapter_fwd, dim3((per_view + 255) / 256, n_views), dim3(256), 0, (hipStream_t)stream, ap, extrinsics,
                       depths, raw_gaussians, sh_rotation, means, covariances, harmonics, scales_out, rotations_out);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}

extern "C" int s360_adapter_backward(const float* extrinsics, const float* depths, const float* raw_gaussians,
                                     const float* sh_rotation, int32_t n_views, int32_t per_view, int32_t H, int32_t W,
                                     int32_t per_ray, int32_t d_sh, float scale_min, float scale_max, float eps,
                                     const float* d_means, const float* d_covariances, int32_t cov9, const float* d_harmonics,
                                     float* d_depths, float* d_raw_gaussians, int32_t erp_convention, void* stream) {
    if (erp_convention < 0 || erp_convention > 3 || (erp_convention != 0 && (H < 2 || W < 2))) return S360_E_BADARG;
    if (!adapter_args_ok(extrinsics, depths, raw_gaussians, n_views, per_view, H, W, per_ray, d_sh) ||
        !d_covariances || !d_harmonics || !d_depths || !d_raw_gaussians)
        return S360_E_BADARG;
    if (n_views == 0 || per_view == 0) return S360_OK;
    AdapterParams ap = {n_views, per_view, H, W, per_ray, d_sh, cov9, erp_convention, scale_min, scale_max, eps};
    if (d_sh == 25) {
        const dim3 grid((per_view + AD_G - 1) / AD_G, n_views);
        if (sh_rotation)
            hipLaunchKernelGGL(k_adapter_bwd25<true>, grid, dim3(AD_T), 0, (hipStream_t)stream, ap, extrinsics, depths, raw_gaussians,
                               sh_rotation, d_means, d_covariances, d_harmonics, d_depths, d_raw_gaussians);
        else
            hipLaunchKernelGGL(k_adapter_bwd25<false>, grid, dim3(AD_T), 0, (hipStream_t)stream, ap, extrinsics, depths, raw_gaussians,
                               sh_rotation, d_means, d_covariances, d_harmonics, d_depths, d_raw_gaussians);
    } else
    hipLaunchKernelGGL(k_adapter_bwd, dim3((per_view + 255) / 256, n_views), dim3(256), 0, (hipStream_t)stream, ap, extrinsics,
                       depths, raw_gaussians, sh_rotation, d_means, d_covariances, d_harmonics, d_depths, d_raw_gaussians);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}

extern "C" int s360_sh_rotation_blocks(const float* rotations, int32_t row_major_stride, int32_t n_views, int32_t d_sh,
                                       float* sh_rotation_out, void* stream) {
    if (!rotations || !sh_rotation_out || n_views < 0 || (row_major_stride != 9 && row_major_stride != 16)) return S360_E_BADARG;
    if (d_sh != 1 && d_sh != 4 && d_sh != 9 && d_sh != 16 && d_sh != 25) return S360_E_BADARG;
    if (n_views == 0) return S360_OK;
    const int deg = d_sh == 25 ? 4 : d_sh == 16 ? 3 : d_sh == 9 ? 2 : d_sh == 4 ? 1 : 0;
    const int threads = n_views * 5;
    hipLaunchKernelGGL(k_sh_rotation_blocks, dim3((threads + 63) / 64), dim3(64), 0, (hipStream_t)stream, rotations,
                       row_major_stride, n_views, d_sh, deg, sh_rotation_out);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}
