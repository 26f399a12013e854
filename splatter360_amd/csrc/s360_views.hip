// s360_views.hip — camera records of a rasteriser call in ONE kernel launch.  gfx950 only.
//
// Replaces the ~60 tiny torch / rocSOLVER launches of the reference's per-call camera glue
// (src/model/decoder/cuda_splatting.py:64-71,80-87: scale-invariant rescale, get_fov at
// src/geometry/projection.py:233-247, get_projection_matrix at cuda_splatting.py:17-44, two matrix
// inverses, one matrix product) for all N views of a call: one thread per view, everything in registers.
// The arithmetic follows the reference's formulas in the reference's order; the two inverses are Gauss-Jordan
// eliminations with partial pivoting instead of LAPACK's LU, so results agree with the torch glue to a few
// ulp (pinned by tests/test_gpu_views.py), not bit for bit — the drop-in render_cuda keeps the torch glue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s360.h"

namespace s360 {

// in-place inverse of an n x n matrix (row-major, n <= 4) by Gauss-Jordan elimination with partial pivoting
template <int N>
__device__ void invert(float (&a)[N][N], float (&inv)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) inv[i][j] = i == j ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        int piv = c;
        float best = fabsf(a[c][c]);
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            const float v = fabsf(a[r][c]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            if (r == piv) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float t = a[c][j]; a[c][j] = a[r][j]; a[r][j] = t;
                    t = inv[c][j]; inv[c][j] = inv[r][j]; inv[r][j] = t;
                }
            }
        }
        const float d = 1.0f / a[c][c];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            a[c][j] *= d;
            inv[c][j] *= d;
        }
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r == c) continue;
            const float f = a[r][c];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                a[r][j] -= f * a[c][j];
                inv[r][j] -= f * inv[c][j];
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_pack_views(const float* __restrict__ extrinsics, const float* __restrict__ intrinsics,
                                                   const float* __restrict__ near_, const float* __restrict__ far_,
                                                   const float* __restrict__ background, int bg_stride, int n,
                                                   int scale_invariant, S360View* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float nr = near_[i], fr = far_[i];
    // cuda_splatting.py:64-71
    const float scale = scale_invariant ? 1.0f / nr : 1.0f;
    float E[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) E[r][c] = extrinsics[16 * i + 4 * r + c];
    float near_s = nr, far_s = fr;
    if (scale_invariant) {
        E[0][3] *= scale;
        E[1][3] *= scale;
        E[2][3] *= scale;
        near_s = nr * scale;
        far_s = fr * scale;
    }
    const float campos[3] = {E[0][3], E[1][3], E[2][3]};
    // get_fov (projection.py:233-247): angle between the un-projected mid-edge rays
    float K[3][3], Kinv[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) K[r][c] = intrinsics[9 * i + 3 * r + c];
    invert<3>(K, Kinv);
    const float vecs[4][3] = {{0.f, 0.5f, 1.f}, {1.f, 0.5f, 1.f}, {0.5f, 0.f, 1.f}, {0.5f, 1.f, 1.f}};
    float ray[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float r[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) r[a] = Kinv[a][0] * vecs[q][0] + Kinv[a][1] * vecs[q][1] + Kinv[a][2] * vecs[q][2];
        const float nrm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
#pragma unroll
        for (int a = 0; a < 3; ++a) ray[q][a] = r[a] / nrm;
    }
    const float fov_x = acosf(ray[0][0] * ray[1][0] + ray[0][1] * ray[1][1] + ray[0][2] * ray[1][2]);
    const float fov_y = acosf(ray[2][0] * ray[3][0] + ray[2][1] * ray[3][1] + ray[2][2] * ray[3][2]);
    const float tan_x = tanf(0.5f * fov_x), tan_y = tanf(0.5f * fov_y);
    // get_projection_matrix (cuda_splatting.py:17-44) with the rescaled planes
    const float top = tan_y * near_s, bottom = -top, right = tan_x * near_s, left = -right;
    float P[4][4] = {};
    P[0][0] = 2.0f * near_s / (right - left);
    P[1][1] = 2.0f * near_s / (top - bottom);
    P[0][2] = (right + left) / (right - left);
    P[1][2] = (top + bottom) / (top - bottom);
    P[3][2] = 1.0f;
    P[2][2] = far_s / (far_s - near_s);
    P[2][3] = -(far_s * near_s) / (far_s - near_s);
    // view_matrix = inverse(c2w)^T, full_projection = view_matrix @ P^T   (cuda_splatting.py:85-87)
    float W2C[4][4];
    invert<4>(E, W2C);
    S360View o;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) o.viewmatrix[4 * r + c] = W2C[c][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += W2C[k][r] * P[c][k];  // view[r][k] * P^T[k][c]
            o.projmatrix[4 * r + c] = s;
        }
    o.campos[0] = campos[0];
    o.campos[1] = campos[1];
    o.campos[2] = campos[2];
    o.tanfovx = tan_x;
    o.tanfovy = tan_y;
    const float* bg = background + (size_t)bg_stride * i;
    o.bg[0] = bg[0];
    o.bg[1] = bg[1];
    o.bg[2] = bg[2];
    o.scale = scale;
    o.near_plane = nr;
    o.far_plane = fr;
    o._pad = 0.f;
    out[i] = o;
}

}  // namespace s360

extern "C" int s360_pack_views(const float* extrinsics, const float* intrinsics, const float* near_planes,
                               const float* far_planes, const float* background, int32_t background_per_view,
                               int32_t n_views, int32_t scale_invariant, S360View* views_out, void* stream) {
    if (!extrinsics || !intrinsics || !near_planes || !far_planes || !background || !views_out || n_views < 0) return S360_E_BADARG;
    if (n_views == 0) return S360_OK;
    hipLaunchKernelGGL(s360::k_pack_views, dim3((n_views + 63) / 64), dim3(64), 0, (hipStream_t)stream, extrinsics, intrinsics,
                       near_planes, far_planes, background, background_per_view ? 3 : 0, n_views, scale_invariant, views_out);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}
