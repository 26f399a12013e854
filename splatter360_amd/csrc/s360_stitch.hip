// s360_stitch.hip — cube -> equirectangular stitch (Cube2Equirec) as one gather kernel, plus
// library-level entry points.  gfx950 only.
//
// Semantics restated from torch's 5-D grid_sample as the reference uses it
// (/root/reference/src/geometry/layers.py:108-116): input [C, D=6, fw, fw], grid (x=u, y=v, z=face),
// mode trilinear, padding_mode "border", align_corners=True:
//   i = ((g + 1) / 2) * (size - 1), clipped to [0, size-1]; 8 corner taps with the usual
//   (1-f) / f weights, taps outside the volume contribute nothing.
// The face-z coordinate lands on an integer face +- 6e-8, so up to two faces are blended with
// a ~1e-7 weight — reproduced here rather than "fixed" (SURVEY.md §8 a10).
#include "s360_device.h"

namespace s360 {

struct FaceMap {
    int src[6];   // source face index for Cube2Equirec slot s
    int flip[6];  // 1: read the face flipped on both image axes
};

__device__ __forceinline__ float unnorm_clip(float g, int size) {
    float i = ((g + 1.0f) / 2.0f) * (float)(size - 1);
    return fminf((float)(size - 1), fmaxf(i, 0.0f));
}

__global__ __launch_bounds__(S360_BLOCK) void k_cube2erp_fwd(const float* __restrict__ faces, const float* __restrict__ grid,
                                                            float* __restrict__ erp, int C, int fw, int eh, int ew, FaceMap fm) {
    const size_t n = (size_t)eh * ew;
    const size_t i = (size_t)blockIdx.x * S360_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float ix = unnorm_clip(grid[3 * i], fw), iy = unnorm_clip(grid[3 * i + 1], fw), iz = unnorm_clip(grid[3 * i + 2], 6);
    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const float fx = ix - x0f, fy = iy - y0f, fz = iz - z0f;
    const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy}, wz[2] = {1.0f - fz, fz};
    const size_t fsz = (size_t)fw * fw;
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int z = z0 + dz;
            if (z < 0 || z > 5) continue;
            const int sf = fm.src[z];
            const float* fp = faces + ((size_t)sf * C + c) * fsz;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int y = y0 + dy;
                if (y < 0 || y >= fw) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int x = x0 + dx;
                    if (x < 0 || x >= fw) continue;
                    const int yy = fm.flip[z] ? fw - 1 - y : y, xx = fm.flip[z] ? fw - 1 - x : x;
                    acc += fp[(size_t)yy * fw + xx] * (wx[dx] * wy[dy] * wz[dz]);
                }
            }
        }
        erp[(size_t)c * n + i] = acc;
    }
}

__global__ __launch_bounds__(S360_BLOCK) void k_cube2erp_bwd(const float* __restrict__ d_erp, const float* __restrict__ grid,
                                                            float* __restrict__ d_faces, int C, int fw, int eh, int ew, FaceMap fm) {
    const size_t n = (size_t)eh * ew;
    const size_t i = (size_t)blockIdx.x * S360_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float ix = unnorm_clip(grid[3 * i], fw), iy = unnorm_clip(grid[3 * i + 1], fw), iz = unnorm_clip(grid[3 * i + 2], 6);
    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const float fx = ix - x0f, fy = iy - y0f, fz = iz - z0f;
    const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy}, wz[2] = {1.0f - fz, fz};
    const size_t fsz = (size_t)fw * fw;
    for (int c = 0; c < C; ++c) {
        const float g = d_erp[(size_t)c * n + i];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int z = z0 + dz;
            if (z < 0 || z > 5) continue;
            float* fp = d_faces + ((size_t)fm.src[z] * C + c) * fsz;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int y = y0 + dy;
                if (y < 0 || y >= fw) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int x = x0 + dx;
                    if (x < 0 || x >= fw) continue;
                    const float w = wx[dx] * wy[dy] * wz[dz];
                    if (w == 0.f) continue;
                    const int yy = fm.flip[z] ? fw - 1 - y : y, xx = fm.flip[z] ? fw - 1 - x : x;
                    atomicAdd(&fp[(size_t)yy * fw + xx], g * w);
                }
            }
        }
    }
}

static bool make_face_map(const int32_t* face_map_host, FaceMap& fm) {
    for (int s = 0; s < 6; ++s) {
        const int v = face_map_host ? face_map_host[s] : s;
        fm.src[s] = v & 7;
        fm.flip[s] = (v >> 3) & 1;
        if (fm.src[s] > 5) return false;
    }
    return true;
}

}  // namespace s360

using namespace s360;

extern "C" int s360_cube2erp_forward(const float* faces, const float* grid, float* erp, int32_t channels, int32_t face_w,
                                     int32_t equ_h, int32_t equ_w, const int32_t* face_map_host, void* stream) {
    if (!faces || !grid || !erp || channels < 1 || face_w < 1 || equ_h < 1 || equ_w < 1) return S360_E_BADARG;
    FaceMap fm;
    if (!make_face_map(face_map_host, fm)) return S360_E_BADARG;
    const size_t n = (size_t)equ_h * equ_w;
    hipLaunchKernelGGL(k_cube2erp_fwd, dim3((unsigned)((n + S360_BLOCK - 1) / S360_BLOCK)), dim3(S360_BLOCK), 0,
                       (hipStream_t)stream, faces, grid, erp, channels, face_w, equ_h, equ_w, fm);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}

extern "C" int s360_cube2erp_backward(const float* d_erp, const float* grid, float* d_faces, int32_t channels,
                                      int32_t face_w, int32_t equ_h, int32_t equ_w, const int32_t* face_map_host,
                                      void* stream) {
    if (!d_erp || !grid || !d_faces || channels < 1 || face_w < 1 || equ_h < 1 || equ_w < 1) return S360_E_BADARG;
    FaceMap fm;
    if (!make_face_map(face_map_host, fm)) return S360_E_BADARG;
    const size_t n = (size_t)equ_h * equ_w;
    if (hipMemsetAsync(d_faces, 0, (size_t)6 * channels * face_w * face_w * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return S360_E_LAUNCH;
    hipLaunchKernelGGL(k_cube2erp_bwd, dim3((unsigned)((n + S360_BLOCK - 1) / S360_BLOCK)), dim3(S360_BLOCK), 0,
                       (hipStream_t)stream, d_erp, grid, d_faces, channels, face_w, equ_h, equ_w, fm);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}

extern "C" int s360_abi_version(void) { return S360_ABI_VERSION; }

extern "C" const char* s360_error_string(int code) {
    switch (code) {
        case S360_OK: return "ok";
        case S360_E_BADARG: return "bad argument";
        case S360_E_WORKSPACE: return "workspace too small";
        case S360_E_LAUNCH: return "HIP launch / runtime error";
        case S360_E_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
