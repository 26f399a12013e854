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
#include "s360_prof.h"

#include <mutex>
#include <vector>

namespace s360 {

struct FaceMap {
    int src[6];   // source face index for Cube2Equirec slot s
    int flip[6];  // 1: read the face flipped on both image axes
    long long fs, cs, rs;  // element strides between faces / channels / rows of the face tensor
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
    // The eight taps once per pixel (offset + weight), then per channel eight INDEPENDENT loads in flight before the first use
    // (the per-channel, per-tap branches of the plain loop nest issued one dependent load at a time: 19 us for 19 MB).  A tap
    // outside the volume keeps a clamped (valid) address and weight 0: acc + v * 0 == acc for the finite pixels of a render, in the
    // same dz, dy, dx order as before — bit-identical output.
    size_t off[8];
    float wgt[8];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz) {
        const int z = z0 + dz, zc = min(max(z, 0), 5);
        const bool zin = z >= 0 && z <= 5;
        const size_t fo = (size_t)fm.src[zc] * fm.fs;
        const bool fl = fm.flip[zc] != 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = y0 + dy, yc = min(max(y, 0), fw - 1);
            const bool yin = y >= 0 && y < fw;
            const int yy = fl ? fw - 1 - yc : yc;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = x0 + dx, xc = min(max(x, 0), fw - 1);
                const bool xin = x >= 0 && x < fw;
                const int xx = fl ? fw - 1 - xc : xc;
                const int k = 4 * dz + 2 * dy + dx;
                off[k] = fo + (size_t)yy * fm.rs + xx;
                wgt[k] = (zin && yin && xin) ? wx[dx] * wy[dy] * wz[dz] : 0.0f;
            }
        }
    }
    for (int c = 0; c < C; ++c) {
        const float* fp = faces + (size_t)c * fm.cs;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fp[off[k]];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k] * wgt[k];
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
    for (int c = 0; c < C; ++c) {
        const float g = d_erp[(size_t)c * n + i];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int z = z0 + dz;
            if (z < 0 || z > 5) continue;
            float* fp = d_faces + (size_t)fm.src[z] * fm.fs + (size_t)c * fm.cs;
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
                    atomicAdd(&fp[(size_t)yy * fm.rs + xx], g * w);
                }
            }
        }
    }
}

static bool make_face_map(const int32_t* face_map_host, const int64_t* strides_host, int C, int fw, FaceMap& fm) {
    fm.fs = strides_host ? strides_host[0] : (long long)C * fw * fw;
    fm.cs = strides_host ? strides_host[1] : (long long)fw * fw;
    fm.rs = strides_host ? strides_host[2] : (long long)fw;
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
                                     int32_t equ_h, int32_t equ_w, const int32_t* face_map_host,
                                     const int64_t* strides_host, void* stream) {
    if (!faces || !grid || !erp || channels < 1 || face_w < 1 || equ_h < 1 || equ_w < 1) return S360_E_BADARG;
    FaceMap fm;
    if (!make_face_map(face_map_host, strides_host, channels, face_w, fm)) return S360_E_BADARG;
    const size_t n = (size_t)equ_h * equ_w;
    ProfScope ps(PS_STITCH, (hipStream_t)stream);
    hipLaunchKernelGGL(k_cube2erp_fwd, dim3((unsigned)((n + S360_BLOCK - 1) / S360_BLOCK)), dim3(S360_BLOCK), 0,
                       (hipStream_t)stream, faces, grid, erp, channels, face_w, equ_h, equ_w, fm);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}

extern "C" int s360_cube2erp_backward(const float* d_erp, const float* grid, float* d_faces, int32_t channels,
                                      int32_t face_w, int32_t equ_h, int32_t equ_w, const int32_t* face_map_host,
                                      const int64_t* strides_host, void* stream) {
    if (!d_erp || !grid || !d_faces || channels < 1 || face_w < 1 || equ_h < 1 || equ_w < 1) return S360_E_BADARG;
    if (strides_host) return S360_E_UNSUPPORTED;  // the adjoint writes a dense [6,C,fw,fw] tensor it zeroes itself
    FaceMap fm;
    if (!make_face_map(face_map_host, strides_host, channels, face_w, fm)) return S360_E_BADARG;
    const size_t n = (size_t)equ_h * equ_w;
    ProfScope ps(PS_STITCH_BWD, (hipStream_t)stream);
    if (hipMemsetAsync(d_faces, 0, (size_t)6 * channels * face_w * face_w * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return S360_E_LAUNCH;
    hipLaunchKernelGGL(k_cube2erp_bwd, dim3((unsigned)((n + S360_BLOCK - 1) / S360_BLOCK)), dim3(S360_BLOCK), 0,
                       (hipStream_t)stream, d_erp, grid, d_faces, channels, face_w, equ_h, equ_w, fm);
    return hipGetLastError() == hipSuccess ? S360_OK : S360_E_LAUNCH;
}

// ---------------------------------------------------------------------------- profiler
namespace s360 {
namespace {
struct Rec {
    int slot;
    hipEvent_t a, b;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;          // completed or open records of the current window
std::vector<hipEvent_t> g_pool;   // recycled events
hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
bool prof_enabled() { return g_on; }
void prof_mark(int slot, hipStream_t st, bool end) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!end) {
        Rec r{slot, get_event(), get_event()};
        (void)hipEventRecord(r.a, st);
        g_recs.push_back(r);
    } else {
        for (size_t i = g_recs.size(); i-- > 0;)
            if (g_recs[i].slot == slot) {
                (void)hipEventRecord(g_recs[i].b, st);
                break;
            }
    }
}
}  // namespace s360

static const char* kSlotNames[PS_NSLOTS] = {"preprocess", "tile_scan", "emit", "sort_tiles", "render",
                                            "render_bwd", "preprocess_bwd", "cube2erp", "cube2erp_bwd", "sh_eval", "gather_slots",
                                            "sh_bwd", "order_units"};

extern "C" int s360_profile_slots(void) { return PS_NSLOTS; }
extern "C" const char* s360_profile_slot_name(int slot) { return slot >= 0 && slot < PS_NSLOTS ? kSlotNames[slot] : ""; }
extern "C" int s360_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(s360::g_mu);
    s360::g_on = on != 0;
    return S360_OK;
}
extern "C" int s360_profile_collect(float* total_ms, int32_t* calls) {
    if (!total_ms || !calls) return S360_E_BADARG;
    std::lock_guard<std::mutex> lk(s360::g_mu);
    for (int i = 0; i < PS_NSLOTS; ++i) {
        total_ms[i] = 0.f;
        calls[i] = 0;
    }
    for (auto& r : s360::g_recs) {
        if (hipEventSynchronize(r.b) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                total_ms[r.slot] += ms;
                calls[r.slot] += 1;
            }
        }
        s360::g_pool.push_back(r.a);
        s360::g_pool.push_back(r.b);
    }
    s360::g_recs.clear();
    return S360_OK;
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
