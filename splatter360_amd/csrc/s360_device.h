// s360_device.h — device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
//
// The whole library is compiled with -ffp-contract=off: the per-Gaussian geometry below keeps
// the exact expression order of the CPU oracle so that every integer intermediate (radius, tile
// rect, tiles_touched, sort order) is bit-identical; hot loops that want FMAs say so explicitly
// with __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s360.h"

#define S360_WAVE 64
#define S360_BLOCK 256

// Pixel footprint of one wave inside a 16x16 tile: SUB_W x (64 / SUB_W) pixels, four of them per tile.
// 16 -> four 16x4 strips stacked vertically; 8 -> four 8x8 quadrants (fewer (splat, wave) overlaps for the
// roughly isotropic footprints of the encoder's Gaussians).
#ifndef S360_SUB_W
#define S360_SUB_W 8
#endif

// Active SH degree of a call: S360_FLAG_SH_DEG4_IGNORED turns a requested degree 4 into degree 3 (coefficients
// 16..24 neither read nor given gradient) — the fallback if the reference's rasteriser fork has no degree-4 table.
static inline int s360_effective_degree(const S360Params* prm) {
    return (prm->sh_degree > 3 && (prm->flags & S360_FLAG_SH_DEG4_IGNORED)) ? 3 : prm->sh_degree;
}

// float4s per (instance, quadrant) partial raster-gradient record of the backward: 3 = packed 48-byte records, 4 = 64-byte aligned
// slots (a record never straddles a 128-byte line or a 32-byte write granule)
#ifndef S360_PREC_F4
#define S360_PREC_F4 4
#endif

namespace s360 {

constexpr int PREC_F4 = S360_PREC_F4;
constexpr int SUB_W = S360_SUB_W, SUB_H = 64 / S360_SUB_W;
__device__ __forceinline__ int sub_ox(int w) { return SUB_W == 16 ? 0 : 8 * (w & 1); }
__device__ __forceinline__ int sub_oy(int w) { return SUB_W == 16 ? 4 * w : 8 * (w >> 1); }


// Problem description passed by value to every kernel (lands in SGPRs).
struct KParams {
    int P, V, H, W, deg, M;
    int gx, gy, T;  // tiles per row / column / view
    uint32_t flags, cap;
};

// S360_FLAG_SPLIT_LISTS (include/s360.h): an 8x8 quadrant that is still busy after the first SEG_HEAD entries of a tile list with
// at least SEG_MIN_REST more hands the rest over in segments of SEG_LEN entries (k_render_tail), one wave each.  Segment k covers
// list positions [SEG_LEN k, SEG_LEN (k + 1)); its slot is SEG_PER_CHUNK * chunk_start[tile] + k (chunk_start: the sort's table of
// 4 096-key chunks of the lists beyond 2 048 keys — every list that can split has chunks).
#ifndef S360_SEG_LEN
#define S360_SEG_LEN 512
#endif
#ifndef S360_SEG_HEAD
#define S360_SEG_HEAD 1024
#endif
#ifndef S360_SEG_BWD_BLOCKS
#define S360_SEG_BWD_BLOCKS 256   // workgroups (one wave each) of the backward composite that take the segment units, grid-stride
#endif
#ifndef S360_SEG_MIN_REST
#define S360_SEG_MIN_REST 512
#endif
constexpr uint32_t SEG_LEN = S360_SEG_LEN, SEG_HEAD = S360_SEG_HEAD, SEG_MIN_REST = S360_SEG_MIN_REST, SEG_PER_CHUNK = 4096 / SEG_LEN;
constexpr uint32_t SEG_K0 = SEG_HEAD / SEG_LEN;   // first segment index a segment wave takes
static_assert(4096 % SEG_LEN == 0 && SEG_HEAD % SEG_LEN == 0 && SEG_LEN % 64 == 0 && SEG_K0 >= 1, "segments tile the sort's 4 096-key chunks; slot 0 holds the head's state");
__device__ constexpr float SEG_T_FAR = 1.0f / 16.0f;   // "far from saturating": at least four more opacity-0.5 contributions to go
__host__ __device__ inline size_t seg_slots(size_t cap) { return (size_t)SEG_PER_CHUNK * (cap / 2048 + 1); }
// segment slots of a call: S360Params.max_segments, or enough for every list of the binning capacity to be long
__host__ __device__ inline size_t seg_slots_of(const S360Params* prm) {
    return prm->max_segments ? (size_t)prm->max_segments : seg_slots(prm->max_instances ? prm->max_instances : 1);
}
#define S360_HDR_SPLIT 5     /* header word: split (tile, quadrant) units of this call */
#define S360_HDR_SEGWORK 6   /* header word: (tile, quadrant, segment) work items k_render queued for k_render_tail (seg_info[]) */
#define S360_HDR_TILES_DONE 24  /* header word: tile workgroups of k_render<.., SPLIT> whose four quadrant waves have all retired (cleared by k_tile_scan) */
#define S360_HDR_NLONG 25       /* header word: tiles with more than SORT_SHORT keys (k_tile_scan) */
#define S360_SEG_CLAIM 0x80000000u  /* bit 31 of a work item's second word: a segment wave has taken the item */
#define S360_HDR_SEGBUFS 32  /* header words [32, 64): the segment-state pointers (SegBufs), written by k_tile_scan for k_render */

// compute units of the current device (256 on MI355X), cached per device: sizes the grids of the persistent kernels
inline int device_cu_count() {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        cached[dev] = n;
    }
    return cached[dev];
}

// workgroups of `fn` the whole device holds at once (occupancy API x CUs), cached per (device, slot < 16): the grid of a persistent kernel
inline int resident_blocks(const void* fn, int block_threads, int slot) {
    static int cached[64][16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || slot < 0 || slot >= 16) return 4 * 256;
    if (!cached[dev][slot]) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block_threads, 0) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); per_cu = 4; }
        cached[dev][slot] = per_cu * device_cu_count();
    }
    return cached[dev][slot];
}

// Real-SH constants (degree <= 3: public 3DGS table; degree 4: standard real-SH table).
__device__ constexpr float kC0 = 0.28209479177387814f;
__device__ constexpr float kC1 = 0.4886025119029199f;
__device__ constexpr float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                     -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                     0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                     -0.5900435899266435f};
__device__ constexpr float kC4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                     -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                     0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

// Y[0..(deg+1)^2) at unit direction (x,y,z).  DEG is a compile-time bound, deg the runtime degree.
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* Y) {
    Y[0] = kC0;
    if (deg > 0) {
        Y[1] = -kC1 * y;
        Y[2] = kC1 * z;
        Y[3] = -kC1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = kC2[0] * xy;
            Y[5] = kC2[1] * yz;
            Y[6] = kC2[2] * (2.0f * zz - xx - yy);
            Y[7] = kC2[3] * xz;
            Y[8] = kC2[4] * (xx - yy);
            if (deg > 2) {
                Y[9] = kC3[0] * y * (3.0f * xx - yy);
                Y[10] = kC3[1] * xy * z;
                Y[11] = kC3[2] * y * (4.0f * zz - xx - yy);
                Y[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                Y[13] = kC3[4] * x * (4.0f * zz - xx - yy);
                Y[14] = kC3[5] * z * (xx - yy);
                Y[15] = kC3[6] * x * (xx - 3.0f * yy);
                if (deg > 3) {
                    Y[16] = kC4[0] * xy * (xx - yy);
                    Y[17] = kC4[1] * yz * (3.0f * xx - yy);
                    Y[18] = kC4[2] * xy * (7.0f * zz - 1.0f);
                    Y[19] = kC4[3] * yz * (7.0f * zz - 3.0f);
                    Y[20] = kC4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
                    Y[21] = kC4[5] * xz * (7.0f * zz - 3.0f);
                    Y[22] = kC4[6] * (xx - yy) * (7.0f * zz - 1.0f);
                    Y[23] = kC4[7] * xz * (xx - 3.0f * yy);
                    Y[24] = kC4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
                }
            }
        }
    }
}

// dY_k/d(x,y,z) of the polynomial forms above.
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* dx, float* dy,
                                              float* dz) {
#pragma unroll
    for (int k = 0; k < 25; ++k) dx[k] = dy[k] = dz[k] = 0.0f;
    if (deg > 0) {
        dy[1] = -kC1;
        dz[2] = kC1;
        dx[3] = -kC1;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dx[4] = kC2[0] * y;  dy[4] = kC2[0] * x;
            dy[5] = kC2[1] * z;  dz[5] = kC2[1] * y;
            dx[6] = kC2[2] * (-2.0f * x); dy[6] = kC2[2] * (-2.0f * y); dz[6] = kC2[2] * (4.0f * z);
            dx[7] = kC2[3] * z;  dz[7] = kC2[3] * x;
            dx[8] = kC2[4] * (2.0f * x); dy[8] = kC2[4] * (-2.0f * y);
            if (deg > 2) {
                dx[9] = kC3[0] * (6.0f * xy); dy[9] = kC3[0] * (3.0f * xx - 3.0f * yy);
                dx[10] = kC3[1] * yz; dy[10] = kC3[1] * xz; dz[10] = kC3[1] * xy;
                dx[11] = kC3[2] * (-2.0f * xy); dy[11] = kC3[2] * (4.0f * zz - xx - 3.0f * yy); dz[11] = kC3[2] * (8.0f * yz);
                dx[12] = kC3[3] * (-6.0f * xz); dy[12] = kC3[3] * (-6.0f * yz); dz[12] = kC3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                dx[13] = kC3[4] * (4.0f * zz - 3.0f * xx - yy); dy[13] = kC3[4] * (-2.0f * xy); dz[13] = kC3[4] * (8.0f * xz);
                dx[14] = kC3[5] * (2.0f * xz); dy[14] = kC3[5] * (-2.0f * yz); dz[14] = kC3[5] * (xx - yy);
                dx[15] = kC3[6] * (3.0f * xx - 3.0f * yy); dy[15] = kC3[6] * (-6.0f * xy);
                if (deg > 3) {
                    float xyz = xy * z;
                    dx[16] = kC4[0] * y * (3.0f * xx - yy); dy[16] = kC4[0] * x * (xx - 3.0f * yy);
                    dx[17] = kC4[1] * (6.0f * xyz); dy[17] = kC4[1] * z * (3.0f * xx - 3.0f * yy); dz[17] = kC4[1] * y * (3.0f * xx - yy);
                    dx[18] = kC4[2] * y * (7.0f * zz - 1.0f); dy[18] = kC4[2] * x * (7.0f * zz - 1.0f); dz[18] = kC4[2] * (14.0f * xyz);
                    dy[19] = kC4[3] * z * (7.0f * zz - 3.0f); dz[19] = kC4[3] * y * (21.0f * zz - 3.0f);
                    dz[20] = kC4[4] * (140.0f * zz * z - 60.0f * z);
                    dx[21] = kC4[5] * z * (7.0f * zz - 3.0f); dz[21] = kC4[5] * x * (21.0f * zz - 3.0f);
                    dx[22] = kC4[6] * (2.0f * x) * (7.0f * zz - 1.0f); dy[22] = -kC4[6] * (2.0f * y) * (7.0f * zz - 1.0f); dz[22] = kC4[6] * (14.0f * z) * (xx - yy);
                    dx[23] = kC4[7] * z * (3.0f * xx - 3.0f * yy); dy[23] = -kC4[7] * (6.0f * xyz); dz[23] = kC4[7] * x * (xx - 3.0f * yy);
                    dx[24] = kC4[8] * (4.0f * xx * x - 12.0f * x * yy); dy[24] = kC4[8] * (4.0f * yy * y - 12.0f * xx * y);
                }
            }
        }
    }
}

// p' = V p with V(i,j) = m[4j+i]: the [4,4] tensors arrive transposed (cuda_splatting.py:85-87).
__device__ __forceinline__ void xform43(const float* m, float px, float py, float pz, float& ox, float& oy,
                                        float& oz) {
    ox = m[0] * px + m[4] * py + m[8] * pz + m[12];
    oy = m[1] * px + m[5] * py + m[9] * pz + m[13];
    oz = m[2] * px + m[6] * py + m[10] * pz + m[14];
}

// 75 consecutive floats (one Gaussian's degree-4 SH slab, only 4-byte aligned) as 18 x 16-byte + 3 loads.
struct __attribute__((packed, aligned(4))) F4U {
    float x, y, z, w;
};
__device__ __forceinline__ void load75(const float* __restrict__ p, float* c) {
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const F4U v = *reinterpret_cast<const F4U*>(p + 4 * i);
        c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
    }
    c[72] = p[72]; c[73] = p[73]; c[74] = p[74];
}

__device__ __forceinline__ void load15(const float* __restrict__ p, float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const F4U v = *reinterpret_cast<const F4U*>(p + 4 * i);
        c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
    }
    c[12] = p[12]; c[13] = p[13]; c[14] = p[14];
}
__device__ __forceinline__ void load25(const float* __restrict__ p, float* c) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const F4U v = *reinterpret_cast<const F4U*>(p + 4 * i);
        c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
    }
    c[24] = p[24];
}

__device__ __forceinline__ void store15(float* __restrict__ p, const float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        F4U v;
        v.x = c[4 * i]; v.y = c[4 * i + 1]; v.z = c[4 * i + 2]; v.w = c[4 * i + 3];
        *reinterpret_cast<F4U*>(p + 4 * i) = v;
    }
    p[12] = c[12]; p[13] = c[13]; p[14] = c[14];
}
__device__ __forceinline__ void store25(float* __restrict__ p, const float* c) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        F4U v;
        v.x = c[4 * i]; v.y = c[4 * i + 1]; v.z = c[4 * i + 2]; v.w = c[4 * i + 3];
        *reinterpret_cast<F4U*>(p + 4 * i) = v;
    }
    p[24] = c[24];
}

// Upper triangle (00,01,02,11,12,22) of Gaussian g's covariance from either layout.
__device__ __forceinline__ void load_cov6(const float* __restrict__ cov, int g, bool cov9, float* c6) {
    if (cov9) {
        const float* c = cov + 9 * (size_t)g;
        c6[0] = c[0]; c6[1] = c[1]; c6[2] = c[2]; c6[3] = c[4]; c6[4] = c[5]; c6[5] = c[8];
    } else {
        const float* c = cov + 6 * (size_t)g;
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = c[k];
    }
}

// Per-(Gaussian, view) geometry shared by preprocess forward and backward.
struct Geo {
    float tx, ty, tz, txc, tyc, fx, fy;
    bool xin, yin;
    float J00, J02, J11, J12;
    float M0[3], M1[3], v0[3], v1[3];
    float a, b, c;  // cov2D incl. the +0.3 low-pass dilation
};

__device__ __forceinline__ void geo_compute(const float* V, float tanfovx, float tanfovy, int W, int H,
                                            float mx, float my, float mz, const float* c6, Geo& g) {
    xform43(V, mx, my, mz, g.tx, g.ty, g.tz);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = g.tx / g.tz, tytz = g.ty / g.tz;
    g.xin = !(txtz < -limx || txtz > limx);
    g.yin = !(tytz < -limy || tytz > limy);
    g.txc = fminf(limx, fmaxf(-limx, txtz)) * g.tz;
    g.tyc = fminf(limy, fmaxf(-limy, tytz)) * g.tz;
    g.fx = (float)W / (2.0f * tanfovx);
    g.fy = (float)H / (2.0f * tanfovy);
    const float tz = g.tz;
    g.J00 = g.fx / tz;
    g.J02 = -(g.fx * g.txc) / (tz * tz);
    g.J11 = g.fy / tz;
    g.J12 = -(g.fy * g.tyc) / (tz * tz);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        g.M0[j] = g.J00 * V[j * 4 + 0] + g.J02 * V[j * 4 + 2];
        g.M1[j] = g.J11 * V[j * 4 + 1] + g.J12 * V[j * 4 + 2];
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.v0[k] = S[k][0] * g.M0[0] + S[k][1] * g.M0[1] + S[k][2] * g.M0[2];
        g.v1[k] = S[k][0] * g.M1[0] + S[k][1] * g.M1[1] + S[k][2] * g.M1[2];
    }
    g.a = g.M0[0] * g.v0[0] + g.M0[1] * g.v0[1] + g.M0[2] * g.v0[2];
    g.b = g.M1[0] * g.v0[0] + g.M1[1] * g.v0[1] + g.M1[2] * g.v0[2];
    g.c = g.M1[0] * g.v1[0] + g.M1[1] * g.v1[1] + g.M1[2] * g.v1[2];
    g.a += 0.3f;
    g.c += 0.3f;
}

// ---- native equirectangular ("spherical") projection: mirrors oracle/s360_oracle.c geo_sph operation for operation ----
__device__ __forceinline__ float s_atan_small(float z) {  // |z| <= tan(pi/16)
    const float z2 = z * z;
    return z * (1.0f + z2 * (-0.3333333333333333f + z2 * (0.2f + z2 * (-0.14285714285714285f + z2 * 0.1111111111111111f))));
}
__device__ __forceinline__ float s_atan_unit(float z) {  // 0 <= z <= 1
    const float z1 = z / (1.0f + sqrtf(1.0f + z * z));
    const float z2 = z1 / (1.0f + sqrtf(1.0f + z1 * z1));
    return 4.0f * s_atan_small(z2);
}
// atan2 from IEEE add / mul / div / sqrt only: bit-identical to the float32 oracle (|error| < 1e-7 rad)
__device__ __forceinline__ float s_atan2(float y, float x) {
    const float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    const bool swap = ay > ax;
    float a = s_atan_unit(swap ? ax / ay : ay / ax);
    if (swap) a = 1.5707963267948966f - a;
    if (x < 0.0f) a = 3.141592653589793f - a;
    return y < 0.0f ? -a : a;
}

struct GeoS {
    float t0, t1, t2, r2, r, rho2, rho, rc;
    bool clamped;
    float u, v;
    float J00, J02, J10, J11, J12;
    float M0[3], M1[3], v0[3], v1[3];
    float a, b, c;
};

__device__ __forceinline__ void geo_sph(const float* V, int W, int H, float mx, float my, float mz, const float* c6, GeoS& g) {
    xform43(V, mx, my, mz, g.t0, g.t1, g.t2);
    const float t0 = g.t0, t1 = g.t1, t2 = g.t2;
    g.rho2 = t0 * t0 + t2 * t2;
    g.r2 = g.rho2 + t1 * t1;
    g.r = sqrtf(g.r2);
    g.rho = sqrtf(g.rho2);
    g.clamped = g.rho < 0.05f * g.r;
    g.rc = g.clamped ? 0.05f * g.r : g.rho;
    const float theta = s_atan2(t0, t2), phi = s_atan2(t1, g.rho);
    g.u = (0.5f - theta / 6.283185307179586f) * (float)W - 0.5f;
    g.v = (0.5f - phi / 3.141592653589793f) * (float)H - 0.5f;
    const float c0 = -(float)W / 6.283185307179586f, c1 = -(float)H / 3.141592653589793f;
    const float A = 1.0f / (g.rc * g.rc), Bq = 1.0f / (g.r2 * g.rc), Cq = g.rc / g.r2;
    g.J00 = c0 * t2 * A;
    g.J02 = -(c0 * t0 * A);
    g.J10 = -(c1 * t0 * t1 * Bq);
    g.J11 = c1 * Cq;
    g.J12 = -(c1 * t2 * t1 * Bq);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        g.M0[j] = g.J00 * V[j * 4 + 0] + g.J02 * V[j * 4 + 2];
        g.M1[j] = g.J10 * V[j * 4 + 0] + g.J11 * V[j * 4 + 1] + g.J12 * V[j * 4 + 2];
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.v0[k] = S[k][0] * g.M0[0] + S[k][1] * g.M0[1] + S[k][2] * g.M0[2];
        g.v1[k] = S[k][0] * g.M1[0] + S[k][1] * g.M1[1] + S[k][2] * g.M1[2];
    }
    g.a = g.M0[0] * g.v0[0] + g.M0[1] * g.v0[1] + g.M0[2] * g.v0[2];
    g.b = g.M1[0] * g.v0[0] + g.M1[1] * g.v0[1] + g.M1[2] * g.v0[2];
    g.c = g.M1[0] * g.v1[0] + g.M1[1] * g.v1[1] + g.M1[2] * g.v1[2];
    g.a += 0.3f;
    g.c += 0.3f;
}

// image a view renders into (spherical mode: views 2i and 2i+1 are the camera and the seam ghost of panorama i)
__device__ __forceinline__ int image_of_view(const KParams& kp, int v) { return (kp.flags & S360_FLAG_SPHERICAL) ? (v >> 1) : v; }
__device__ __forceinline__ int view_of_image(const KParams& kp, int img) { return (kp.flags & S360_FLAG_SPHERICAL) ? 2 * img : img; }

typedef float f2 __attribute__((ext_vector_type(2)));  // a register pair: operand of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Composite exponent.  Records hold the conic pre-scaled (a' = -log2(e)/2 * a, b' = -log2(e) * b,
// c' = -log2(e)/2 * c) so that  log2 G = a' dx^2 + b' dx dy + c' dy^2  is three FMAs + two multiplies
// and G = v_exp_f32(.) with no extra multiply.  Forward and backward share this function, so both
// take identical accept / reject decisions for every (pixel, splat) pair.
__device__ constexpr float kLog2e = 1.4426950408889634f;
__device__ constexpr float kConicDiag = -0.5f * 1.4426950408889634f;
__device__ constexpr float kConicOff = -1.4426950408889634f;
__device__ __forceinline__ float power2(float a, float b, float c, float dx, float dy) {
    const float t = __builtin_fmaf(b, dy, a * dx);
    return __builtin_fmaf(t, dx, (c * dy) * dy);
}

// Exact quadrant cull shared by the forward and backward composites.  A splat contributes to a pixel only if
// alpha = opacity * 2^power >= 1/255, i.e. power >= -log2(255 opacity).  power(dx, dy) = a dx^2 + b dx dy + c dy^2 is a
// concave quadratic (maximum 0 at the centre), so its maximum over the box spanned by the quadrant's 8x8 pixel centres
// lies either at the centre (inside the box) or on one of the two box edges facing the centre; each edge maximum is a
// clamped 1-D parabola vertex (ka = -b/(2a), kb = -b/(2c) are stored in the splat record by k_preprocess).
// The test is conservative (continuous box instead of the pixel lattice, +0.02 in log2 units against float rounding of
// ~1e-5): a culled entry fails alpha >= 1/255 on every pixel of the quadrant, so skipping it changes nothing.
// Round 1 tested the ellipse's axis-aligned bounding box instead: 18 % of its survivors contributed to no pixel.
template <int BW, int BH>
__device__ __forceinline__ bool box_hit(float x, float y, float a, float b, float c, float op, float ka, float kb,
                                        float x0, float y0) {
    const float dh = x - x0, dl = dh - (float)(BW - 1);   // range of dx = x - px over the box's columns
    const float eh = y - y0, el = eh - (float)(BH - 1);   // range of dy = y - py over its rows
    const float dx1 = __builtin_amdgcn_fmed3f(0.0f, dl, dh);  // column closest to the centre
    const float dy1 = __builtin_amdgcn_fmed3f(0.0f, el, eh);
    const float dys = __builtin_amdgcn_fmed3f(kb * dx1, el, eh);  // best row on that column
    const float dxs = __builtin_amdgcn_fmed3f(ka * dy1, dl, dh);  // best column on that row
    const float pmax = fmaxf(power2(a, b, c, dx1, dys), power2(a, b, c, dxs, dy1));
    return pmax + __builtin_amdgcn_logf(op) + (7.994353436858858f + 0.02f) >= 0.0f;  // log2(255) = 7.9943...
}
__device__ __forceinline__ bool quadrant_hit(float x, float y, float a, float b, float c, float op, float ka, float kb,
                                             float x0, float y0) {
    return box_hit<SUB_W, SUB_H>(x, y, a, b, c, op, ka, kb, x0, y0);
}

// The same test against a box given at run time: origin (x0, y0) and extent (w1, h1) = (columns - 1, rows - 1) in pixels.  With
// the full quadrant, (x0, y0, 7, 7), it is quadrant_hit() operation for operation.  k_render culls every chunk against the
// bounding box of the pixels of its quadrant that are still UNSATURATED: an entry that fails it is below 1/255 on every pixel that
// can still take a contribution, so neither the forward nor the backward (whose survivor records these are) loses anything.
__device__ __forceinline__ bool box_hit_rt(float x, float y, float a, float b, float c, float op, float ka, float kb,
                                           float x0, float y0, float w1, float h1) {
    const float dh = x - x0, dl = dh - w1;
    const float eh = y - y0, el = eh - h1;
    const float dx1 = __builtin_amdgcn_fmed3f(0.0f, dl, dh);
    const float dy1 = __builtin_amdgcn_fmed3f(0.0f, el, eh);
    const float dys = __builtin_amdgcn_fmed3f(kb * dx1, el, eh);
    const float dxs = __builtin_amdgcn_fmed3f(ka * dy1, dl, dh);
    const float pmax = fmaxf(power2(a, b, c, dx1, dys), power2(a, b, c, dxs, dy1));
    return pmax + __builtin_amdgcn_logf(op) + (7.994353436858858f + 0.02f) >= 0.0f;
}

// Bounding box of the set bits of an 8x8 quadrant mask (lane = 8 * row + column): scalar bit tricks, no vector work.
__device__ __forceinline__ void mask_bbox8(unsigned long long m, int& c0, int& c1, int& r0, int& r1) {
    uint32_t cols = (uint32_t)(m | (m >> 32));
    cols |= cols >> 16;
    cols |= cols >> 8;
    cols &= 0xFFu;                                   // bit c: some row has column c set
    unsigned long long t = m | (m >> 4);
    t |= t >> 2;
    t |= t >> 1;
    t &= 0x0101010101010101ull;                      // bit 8 r: row r is non-empty
    const uint32_t rows = (uint32_t)((t * 0x0102040810204080ull) >> 56);
    c0 = __builtin_ctz(cols); c1 = 31 - __builtin_clz(cols);
    r0 = __builtin_ctz(rows); r1 = 31 - __builtin_clz(rows);
}

// The same test for a whole 16x16 tile — the binning-time cull of S360_FLAG_LEAN_LISTS (k_preprocess counts and k_emit
// places a (Gaussian, tile) instance only when it passes; both evaluate THIS function on the stored record, so the
// histogram and the emission agree).  lop = v_log_f32(opacity), hoisted by the caller.  The margin is twice the quadrants'
// (0.04 against 0.02 log2 units): a tile-culled entry then fails every quadrant's own test by more than the rounding error
// of either evaluation (a quadrant's box lies inside the tile's, so its maximum cannot be larger) — the quadrants' survivor
// sets, and with them the backward's group composition, are exactly those of the unculled list: images AND gradients stay
// bit-identical to the upstream-compatible lists, only tiles_touched / list / n_contrib positions change.
__device__ __forceinline__ bool tile_hit(float x, float y, float a, float b, float c, float lop, float ka, float kb, int tx, int ty) {
    const float dh = x - (float)(16 * tx), dl = dh - 15.0f;
    const float eh = y - (float)(16 * ty), el = eh - 15.0f;
    const float dx1 = __builtin_amdgcn_fmed3f(0.0f, dl, dh);
    const float dy1 = __builtin_amdgcn_fmed3f(0.0f, el, eh);
    const float dys = __builtin_amdgcn_fmed3f(kb * dx1, el, eh);
    const float dxs = __builtin_amdgcn_fmed3f(ka * dy1, dl, dh);
    const float pmax = fmaxf(power2(a, b, c, dx1, dys), power2(a, b, c, dxs, dy1));
    return !(pmax + lop + (7.994353436858858f + 0.04f) < 0.0f);   // NaN (degenerate conic) keeps the instance
}

// Per-Gaussian "colour" of the fused depth map for the reference's DepthRenderingMode
// (cuda_splatting.py:244-251; z = camera-space depth in unscaled units).  The "log" mode reproduces the
// reference's swapped clamp (z.minimum(near).maximum(far)) as is.
__device__ __forceinline__ float depth_value(float z, float nearp, float farp, int mode) {
    if (mode == 1) return 1.0f / z;
    if (mode == 2) {
        const float eps = 1e-10f;
        const float disp_near = 1.0f / (nearp + eps), disp_far = 1.0f / (farp + eps), disp = 1.0f / (z + eps);
        return 1.0f - (disp - disp_far) / (disp_near - disp_far + eps);
    }
    if (mode == 3) return __logf(fmaxf(fminf(z, nearp), farp));
    return z;
}

// Tile rectangle of a splat (same float expression everywhere it is needed: preprocess, emit and
// the backward instance index all recompute it from the stored centre and integer radius).
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& minx, int& miny,
                                          int& maxx, int& maxy) {
    const float rr = (float)radius;
    minx = min(gx, max(0, (int)((px - rr) / 16.0f)));
    miny = min(gy, max(0, (int)((py - rr) / 16.0f)));
    maxx = min(gx, max(0, (int)((px + rr + 15.0f) / 16.0f)));
    maxy = min(gy, max(0, (int)((py + rr + 15.0f) / 16.0f)));
}

// ---- wave64 primitives -------------------------------------------------------------------
// broadcast lane `l` (wave-uniform) of a VGPR into an SGPR
__device__ __forceinline__ float rl(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Wave64 sum with plain DPP adds; total valid in lane 63.
__device__ __forceinline__ float wave_sum1_lane63(float a) {
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(a));
    return a;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Single workgroup: order[] = work-unit ids in descending order of their work estimate (backward: unit =
// tile*4 + quadrant, weight = the forward's replay length; forward: unit = tile, weight = list length)
// (64-bucket counting sort on weight / max weight).  Units are dispatched in index order round-robin over
// the CUs / SIMDs, so dealing them heavy-first gives every SIMD a similar mix (LPT-style balancing of
// the sequential per-quadrant chains, which cannot be split).
// The same launch also clears the backward's validity words (valid[instance*4 + quadrant] = 1 when that
// quadrant's wave wrote a partial record): block 0 orders, blocks 1.. zero (one 32-bit word = 4 flags per instance).
template <int THREADS>
__device__ __forceinline__ void order_units_body(const uint32_t* __restrict__ weight, uint32_t* __restrict__ order, int n) {
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_cnt[64];
    __shared__ uint32_t s_base[64];
    if (threadIdx.x == 0) s_max = 1;
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int i = threadIdx.x; i < n; i += THREADS) mx = max(mx, weight[i]);
    mx = wave_max_u32(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t wmax = s_max;
    for (int i = threadIdx.x; i < n; i += THREADS) atomicAdd(&s_cnt[63 - (uint32_t)(((uint64_t)weight[i] * 63) / wmax)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 64; ++b) {
            s_base[b] = run;
            run += s_cnt[b];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const uint32_t b = 63 - (uint32_t)(((uint64_t)weight[i] * 63) / wmax);
        order[atomicAdd(&s_base[b], 1u)] = (uint32_t)i;
    }
}

// Backward variant: units are (tile, quadrant) waves; TILES are ordered heavy first (weight = the tile's longest
// quadrant replay) and the four quadrant waves of a tile are dealt to launch slots r, r + 8, r + 16, r + 24 of a 32-slot
// group — workgroup b runs on XCD b % 8 (observed placement; used for speed only), so the four walks over the same
// depth-sorted list run at about the same time behind the SAME 4-MB L2 and the list's records are fetched from the
// fabric once instead of four times.
template <int THREADS>
__device__ __forceinline__ void order_quadrants_body(const uint32_t* __restrict__ weight, uint32_t* __restrict__ order, int nt) {
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_cnt[64];
    __shared__ uint32_t s_base[64];
    if (threadIdx.x == 0) s_max = 1;
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int i = threadIdx.x; i < nt; i += THREADS) {
        const uint4 w = reinterpret_cast<const uint4*>(weight)[i];
        mx = max(mx, max(max(w.x, w.y), max(w.z, w.w)));
    }
    mx = wave_max_u32(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t wmax = s_max;
    for (int i = threadIdx.x; i < nt; i += THREADS) {
        const uint4 w = reinterpret_cast<const uint4*>(weight)[i];
        atomicAdd(&s_cnt[63 - (uint32_t)(((uint64_t)max(max(w.x, w.y), max(w.z, w.w)) * 63) / wmax)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 64; ++b) {
            s_base[b] = run;
            run += s_cnt[b];
        }
    }
    __syncthreads();
    const uint32_t full = (uint32_t)nt / 8u * 8u, tail = (uint32_t)nt - full;
    for (int i = threadIdx.x; i < nt; i += THREADS) {
        const uint4 w = reinterpret_cast<const uint4*>(weight)[i];
        const uint32_t b = 63 - (uint32_t)(((uint64_t)max(max(w.x, w.y), max(w.z, w.w)) * 63) / wmax);
        const uint32_t r = atomicAdd(&s_base[b], 1u);
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t slot = r < full ? (r / 8u) * 32u + q * 8u + (r & 7u) : full * 4u + q * tail + (r - full);
            order[slot] = 4u * (uint32_t)i + q;
        }
    }
}

// Final reduction of the loss epilogue (the reference computes the loss as separate torch ops right after the decoder: LossMse,
// src/loss/loss_mse.py:30-31; compute_psnr, src/evaluation/metrics.py:11-21).  One 1 024-thread workgroup; fixed assignment of
// partials to threads, fixed shuffle tree, fixed wave order: deterministic.  Runs as its own launch at the end of the forward
// (k_mse_finish) or — S360_FLAG_DEFER_LOSS — inside the backward's first launch (k_order_units), where its chain of ~2-us memory
// round trips hides behind the unit ordering and the validity clear.
constexpr int MSE_BLOCK = 1024;
__device__ __forceinline__ void mse_finish_body(const float* __restrict__ partials, int n_per_view, int V, float loss_scale,
                                                float inv_elems, float* __restrict__ out) {
    constexpr int VG = 8;  // views per sweep: their loads are independent, one round trip per 1024 partials of each
    __shared__ float s_w[MSE_BLOCK / 64][VG][2];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    float total = 0.f;
    for (int v0 = 0; v0 < V; v0 += VG) {
        float a[VG], b[VG];
#pragma unroll
        for (int j = 0; j < VG; ++j) a[j] = b[j] = 0.f;
        for (int i = threadIdx.x; i < n_per_view; i += MSE_BLOCK) {
            float2 q[VG];
#pragma unroll
            for (int j = 0; j < VG; ++j)
                q[j] = v0 + j < V ? reinterpret_cast<const float2*>(partials)[(size_t)(v0 + j) * n_per_view + i] : make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < VG; ++j) {
                a[j] += q[j].x;
                b[j] += q[j].y;
            }
        }
#pragma unroll
        for (int j = 0; j < VG; ++j) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a[j] += __shfl_xor(a[j], o);
                b[j] += __shfl_xor(b[j], o);
            }
            if (lane == 0) {
                s_w[wave][j][0] = a[j];
                s_w[wave][j][1] = b[j];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int j = 0; j < VG && v0 + j < V; ++j) {
                float sa = 0.f, sb = 0.f;
                for (int w = 0; w < MSE_BLOCK / 64; ++w) {
                    sa += s_w[w][j][0];
                    sb += s_w[w][j][1];
                }
                total += sa;
                out[1 + v0 + j] = sb * inv_elems;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = total * loss_scale;
}


// header words (uint32 index) through which a S360_FLAG_DEFER_LOSS forward tells the backward where its loss goes
#define S360_HDR_LOSS 16   /* [16,17] partials pointer, [18,19] loss_out pointer, [20] n_per_view, [21] V, [22] loss_scale bits, [23] inv_elems bits */

// S360_FLAG_SPLIT_LISTS state for the backward composite (see s360_bwd_em.h), parked in front of the segment launch list
struct SegBwdPod {
    const uint32_t* seg_flag;
    const uint32_t* chunk_start;
    const float4* seg_c;
    const float* seg_t;
    const uint32_t* seg_cnt;
    const uint2* seg_info;
    const uint32_t* seg_list;
    uint32_t n_seg_blocks;
    uint32_t dbg_base;
};
static_assert(sizeof(SegBwdPod) <= 32 * 4, "parked in the 32 words in front of seg_list");

static __global__ __launch_bounds__(1024) void k_order_units(const uint32_t* __restrict__ weight, uint32_t* __restrict__ order, int n,
                                                     uint32_t* __restrict__ valid_words, const uint32_t* __restrict__ header,
                                                     uint32_t cap, float4* __restrict__ pairgrad_atomic,
                                                     const uint8_t* __restrict__ vis_mask, int P, int V,
                                                     const uint32_t* __restrict__ seg_cnt, const uint32_t* __restrict__ chunk_start,
                                                     uint32_t* __restrict__ seg_list, const uint2* __restrict__ seg_info, SegBwdPod sbpod) {
    if (blockIdx.x == 2 && seg_list) {
        if (threadIdx.x == 0) *reinterpret_cast<SegBwdPod*>(seg_list - 32) = sbpod;   // the backward composite reads it through one pointer
        // S360_FLAG_SPLIT_LISTS: the segment units that hold survivor records, compacted (any order: every unit is self-contained) —
        // the backward composite's first workgroups take them grid-stride
        __shared__ uint32_t s_n;
        if (threadIdx.x == 0) s_n = 0u;
        __syncthreads();
        const uint32_t n = header[S360_HDR_SEGWORK];   // the forward's work items: (tile, segment << 2 | quadrant)
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            uint2 w = seg_info[i];
            w.y &= ~S360_SEG_CLAIM;
            if (seg_cnt[((size_t)SEG_PER_CHUNK * chunk_start[w.x] + (w.y >> 2)) * 4 + (w.y & 3u)]) seg_list[1u + atomicAdd(&s_n, 1u)] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) seg_list[0] = s_n;
    }
    if (blockIdx.x == 1) {   // S360_FLAG_DEFER_LOSS: the forward left its loss reduction to this launch
        const uint64_t pp = (uint64_t)header[S360_HDR_LOSS] | ((uint64_t)header[S360_HDR_LOSS + 1] << 32);
        const uint64_t po = (uint64_t)header[S360_HDR_LOSS + 2] | ((uint64_t)header[S360_HDR_LOSS + 3] << 32);
        if (pp && po)
            mse_finish_body(reinterpret_cast<const float*>(pp), (int)header[S360_HDR_LOSS + 4], (int)header[S360_HDR_LOSS + 5],
                            __uint_as_float(header[S360_HDR_LOSS + 6]), __uint_as_float(header[S360_HDR_LOSS + 7]), reinterpret_cast<float*>(po));
    }
    if (blockIdx.x > 0) {
        const size_t stride = (size_t)(gridDim.x - 1) * 1024;
        if (pairgrad_atomic) {   // S360_FLAG_ATOMIC_GRADS: the composite ADDS into the visible pairs' records — start them at zero
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (size_t g = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; g < (size_t)P; g += stride)
                for (uint32_t m = vis_mask[g]; m; m &= m - 1) {
                    float4* r = pairgrad_atomic + 3 * ((size_t)__builtin_ctz(m) * P + g);
                    r[0] = z; r[1] = z; r[2] = z;
                }
            return;
        }
        const size_t nv = (size_t)min(header[0], cap);
        for (size_t i = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < nv; i += stride) valid_words[i] = 0u;
        return;
    }
    if (!order) return;
    order_quadrants_body<1024>(weight, order, n / 4);
}


}  // namespace s360
