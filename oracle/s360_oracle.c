/*
 * s360_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Scalar restatement of the tile-based perspective Gaussian-splat rasteriser that
 * thucz/splatter360 calls at  src/model/decoder/cuda_splatting.py:99-124
 * (`GaussianRasterizationSettings` / `GaussianRasterizer`), forward and backward.
 *
 * PARITY UNPINNED.  The rasteriser is the un-vendored, un-pinned pip dependency
 * `git+https://github.com/dcharatan/diff-gaussian-rasterization-modified`
 * (/root/reference/requirements.txt:17).  Its source is NOT under /root/reference, it cannot be
 * built or imported here, and the reference has no tests or golden vectors for it.  This file
 * restates the published 3DGS rasteriser algorithm and is anchored on the
 * reference's call site (argument layout, matrix conventions — pinned by tests/golden/ captures
 * of cuda_splatting.py) plus float64 finite differences and an independent PyTorch autograd
 * restatement (oracle/torch_ref.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: see oracle/Makefile.  -ffp-contract=off is REQUIRED (the HIP geometry kernels are
 * compiled contraction-free with the same expression order so that all integer intermediates
 * — radii, rects, tiles_touched, offsets, sorted keys/values, tile ranges — are bit-exact).
 * REAL=float  -> liboracle_f32.so  (parity authority)
 * REAL=double -> liboracle_f64.so  (finite-difference validation of the analytic backward)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define R(x) ((real)(x))

static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_ceil(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_min(real a, real b) { return a < b ? a : b; }
/* float->int with the saturating semantics of the GPU conversion instruction (NaN -> 0). */
static inline int r2i(real x) {
    if (x != x) return 0;
    if (x >= R(2147483647.0)) return 2147483647;
    if (x <= R(-2147483648.0)) return (-2147483647 - 1);
    return (int)x;
}
static inline int i_max(int a, int b) { return a > b ? a : b; }
static inline int i_min(int a, int b) { return a < b ? a : b; }

#define TILE 16

/* ---- SH constants: SURVEY.md Appendix A.3 (degree <= 3 = public 3DGS; degree 4 = standard
 *      real-SH table; the fork's exact degree-4 form is unverifiable here). ---- */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};
static const double SH_C4[9] = {2.5033429417967046, -1.7701307697799304, 0.9461746957575601,
                                -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
                                0.47308734787878004, -1.7701307697799304, 0.6258357354491761};

typedef struct {
    int32_t P;          /* number of Gaussians */
    int32_t H, W;       /* image size */
    int32_t sh_degree;  /* active degree (0..4); ignored when colours are precomputed */
    int32_t M;          /* SH coefficients stored per Gaussian per channel (array stride) */
    int32_t use_sh;     /* 1: shs given, 0: colors_precomp given */
    real tanfovx, tanfovy;
    real bg[3];
    real viewmatrix[16]; /* as handed over by cuda_splatting.py:86 (transposed w2c, flat) */
    real projmatrix[16]; /* cuda_splatting.py:87 */
    real campos[3];      /* cuda_splatting.py:109 */
    int32_t spherical;   /* 0: perspective (the reference's path); 1: native equirectangular splat mode (no
                            reference counterpart — specified by this file, see geo_sph) */
} OrcParams;

typedef struct {
    OrcParams prm;
    int32_t NP;  /* rasterised pairs: P (perspective) or 2P (spherical: Gaussian g and its seam ghost P + g) */
    /* inputs (borrowed copies) */
    real *means, *cov6, *opac, *shs, *colors_in;
    /* per-Gaussian forward state */
    int32_t *radii;
    uint32_t *tiles_touched, *offsets; /* offsets = inclusive scan */
    int32_t *rect;                     /* minx,miny,maxx,maxy */
    real *xy, *depth, *conic_op, *rgb;
    uint8_t *clamped;                  /* 3 per Gaussian */
    /* binning */
    uint64_t L;
    uint64_t *keys;   /* sorted: tile<<32 | float32 bits of depth */
    uint32_t *vals;   /* sorted Gaussian indices */
    uint32_t *ranges; /* 2 per tile: start,end */
    /* image state */
    real *image, *final_T;
    uint32_t *n_contrib;
    /* per-Gaussian raster gradients (backward intermediates) */
    real *g_xy, *g_conic, *g_op, *g_rgb;
} Orc;

/* x' = V p with V(i,j) = m[j*4+i]  (row-vector convention: memory is the transposed matrix) */
static inline void xform43(const real* m, const real* p, real* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform44(const real* m, const real* p, real* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* SH basis values Y[0..24] at unit direction (x,y,z). */
static void sh_basis(int deg, real x, real y, real z, real* Y) {
    Y[0] = R(SH_C0);
    if (deg > 0) {
        Y[1] = -R(SH_C1) * y;
        Y[2] = R(SH_C1) * z;
        Y[3] = -R(SH_C1) * x;
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = R(SH_C2[0]) * xy;
            Y[5] = R(SH_C2[1]) * yz;
            Y[6] = R(SH_C2[2]) * (R(2) * zz - xx - yy);
            Y[7] = R(SH_C2[3]) * xz;
            Y[8] = R(SH_C2[4]) * (xx - yy);
            if (deg > 2) {
                Y[9] = R(SH_C3[0]) * y * (R(3) * xx - yy);
                Y[10] = R(SH_C3[1]) * xy * z;
                Y[11] = R(SH_C3[2]) * y * (R(4) * zz - xx - yy);
                Y[12] = R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy);
                Y[13] = R(SH_C3[4]) * x * (R(4) * zz - xx - yy);
                Y[14] = R(SH_C3[5]) * z * (xx - yy);
                Y[15] = R(SH_C3[6]) * x * (xx - R(3) * yy);
                if (deg > 3) {
                    Y[16] = R(SH_C4[0]) * xy * (xx - yy);
                    Y[17] = R(SH_C4[1]) * yz * (R(3) * xx - yy);
                    Y[18] = R(SH_C4[2]) * xy * (R(7) * zz - R(1));
                    Y[19] = R(SH_C4[3]) * yz * (R(7) * zz - R(3));
                    Y[20] = R(SH_C4[4]) * (zz * (R(35) * zz - R(30)) + R(3));
                    Y[21] = R(SH_C4[5]) * xz * (R(7) * zz - R(3));
                    Y[22] = R(SH_C4[6]) * (xx - yy) * (R(7) * zz - R(1));
                    Y[23] = R(SH_C4[7]) * xz * (xx - R(3) * yy);
                    Y[24] = R(SH_C4[8]) * (xx * (xx - R(3) * yy) - yy * (R(3) * xx - yy));
                }
            }
        }
    }
}

/* d Y_k / d(x,y,z), polynomial forms as written above. */
static void sh_basis_grad(int deg, real x, real y, real z, real* dx, real* dy, real* dz) {
    int n = (deg + 1) * (deg + 1);
    for (int k = 0; k < n; ++k) dx[k] = dy[k] = dz[k] = R(0);
    if (deg > 0) {
        dy[1] = -R(SH_C1);
        dz[2] = R(SH_C1);
        dx[3] = -R(SH_C1);
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dx[4] = R(SH_C2[0]) * y;  dy[4] = R(SH_C2[0]) * x;
            dy[5] = R(SH_C2[1]) * z;  dz[5] = R(SH_C2[1]) * y;
            dx[6] = R(SH_C2[2]) * (-R(2) * x); dy[6] = R(SH_C2[2]) * (-R(2) * y); dz[6] = R(SH_C2[2]) * (R(4) * z);
            dx[7] = R(SH_C2[3]) * z;  dz[7] = R(SH_C2[3]) * x;
            dx[8] = R(SH_C2[4]) * (R(2) * x); dy[8] = R(SH_C2[4]) * (-R(2) * y);
            if (deg > 2) {
                dx[9] = R(SH_C3[0]) * (R(6) * xy); dy[9] = R(SH_C3[0]) * (R(3) * xx - R(3) * yy);
                dx[10] = R(SH_C3[1]) * yz; dy[10] = R(SH_C3[1]) * xz; dz[10] = R(SH_C3[1]) * xy;
                dx[11] = R(SH_C3[2]) * (-R(2) * xy); dy[11] = R(SH_C3[2]) * (R(4) * zz - xx - R(3) * yy); dz[11] = R(SH_C3[2]) * (R(8) * yz);
                dx[12] = R(SH_C3[3]) * (-R(6) * xz); dy[12] = R(SH_C3[3]) * (-R(6) * yz); dz[12] = R(SH_C3[3]) * (R(6) * zz - R(3) * xx - R(3) * yy);
                dx[13] = R(SH_C3[4]) * (R(4) * zz - R(3) * xx - yy); dy[13] = R(SH_C3[4]) * (-R(2) * xy); dz[13] = R(SH_C3[4]) * (R(8) * xz);
                dx[14] = R(SH_C3[5]) * (R(2) * xz); dy[14] = R(SH_C3[5]) * (-R(2) * yz); dz[14] = R(SH_C3[5]) * (xx - yy);
                dx[15] = R(SH_C3[6]) * (R(3) * xx - R(3) * yy); dy[15] = R(SH_C3[6]) * (-R(6) * xy);
                if (deg > 3) {
                    real xyz = xy * z;
                    dx[16] = R(SH_C4[0]) * y * (R(3) * xx - yy); dy[16] = R(SH_C4[0]) * x * (xx - R(3) * yy);
                    dx[17] = R(SH_C4[1]) * (R(6) * xyz); dy[17] = R(SH_C4[1]) * z * (R(3) * xx - R(3) * yy); dz[17] = R(SH_C4[1]) * y * (R(3) * xx - yy);
                    dx[18] = R(SH_C4[2]) * y * (R(7) * zz - R(1)); dy[18] = R(SH_C4[2]) * x * (R(7) * zz - R(1)); dz[18] = R(SH_C4[2]) * (R(14) * xyz);
                    dy[19] = R(SH_C4[3]) * z * (R(7) * zz - R(3)); dz[19] = R(SH_C4[3]) * y * (R(21) * zz - R(3));
                    dz[20] = R(SH_C4[4]) * (R(140) * zz * z - R(60) * z);
                    dx[21] = R(SH_C4[5]) * z * (R(7) * zz - R(3)); dz[21] = R(SH_C4[5]) * x * (R(21) * zz - R(3));
                    dx[22] = R(SH_C4[6]) * (R(2) * x) * (R(7) * zz - R(1)); dy[22] = -R(SH_C4[6]) * (R(2) * y) * (R(7) * zz - R(1)); dz[22] = R(SH_C4[6]) * (R(14) * z) * (xx - yy);
                    dx[23] = R(SH_C4[7]) * z * (R(3) * xx - R(3) * yy); dy[23] = -R(SH_C4[7]) * (R(6) * xyz); dz[23] = R(SH_C4[7]) * x * (xx - R(3) * yy);
                    dx[24] = R(SH_C4[8]) * (R(4) * xx * x - R(12) * x * yy); dy[24] = R(SH_C4[8]) * (R(4) * yy * y - R(12) * xx * y);
                }
            }
        }
    }
}

/* Geometry shared by forward and backward: view-space point, clamped tangent coords, the two
 * rows of M = J * Rot(w2c) and v0 = Sigma*M0, v1 = Sigma*M1.  (Appendix A.1 step 4) */
typedef struct {
    real t[3], txc, tyc, fx, fy;
    int xin, yin; /* 1 when t.x/t.z (t.y/t.z) is inside the 1.3*tanfov clamp */
    real J00, J02, J11, J12;
    real M0[3], M1[3], v0[3], v1[3];
    real a, b, c; /* cov2D incl. the +0.3 dilation */
} Geo;

static void geo_compute(const OrcParams* p, const real* mean, const real* c6, Geo* g) {
    const real* V = p->viewmatrix;
    xform43(V, mean, g->t);
    real limx = R(1.3) * p->tanfovx, limy = R(1.3) * p->tanfovy;
    real txtz = g->t[0] / g->t[2], tytz = g->t[1] / g->t[2];
    g->xin = !(txtz < -limx || txtz > limx);
    g->yin = !(tytz < -limy || tytz > limy);
    g->txc = r_min(limx, r_max(-limx, txtz)) * g->t[2];
    g->tyc = r_min(limy, r_max(-limy, tytz)) * g->t[2];
    g->fx = (real)p->W / (R(2) * p->tanfovx);
    g->fy = (real)p->H / (R(2) * p->tanfovy);
    real tz = g->t[2];
    g->J00 = g->fx / tz;
    g->J02 = -(g->fx * g->txc) / (tz * tz);
    g->J11 = g->fy / tz;
    g->J12 = -(g->fy * g->tyc) / (tz * tz);
    /* Rot(i,j) = V[j*4+i] */
    for (int j = 0; j < 3; ++j) {
        g->M0[j] = g->J00 * V[j * 4 + 0] + g->J02 * V[j * 4 + 2];
        g->M1[j] = g->J11 * V[j * 4 + 1] + g->J12 * V[j * 4 + 2];
    }
    real S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    for (int k = 0; k < 3; ++k) {
        g->v0[k] = S[k][0] * g->M0[0] + S[k][1] * g->M0[1] + S[k][2] * g->M0[2];
        g->v1[k] = S[k][0] * g->M1[0] + S[k][1] * g->M1[1] + S[k][2] * g->M1[2];
    }
    g->a = g->M0[0] * g->v0[0] + g->M0[1] * g->v0[1] + g->M0[2] * g->v0[2];
    g->b = g->M1[0] * g->v0[0] + g->M1[1] * g->v0[1] + g->M1[2] * g->v0[2];
    g->c = g->M1[0] * g->v1[0] + g->M1[1] * g->v1[1] + g->M1[2] * g->v1[2];
    g->a += R(0.3);
    g->c += R(0.3);
}

/* ====================================================================================================
 * Native equirectangular ("spherical") splat mode — SURVEY.md 8(f)-4.  NO reference counterpart (the reference only
 * ever renders cube faces); this is the specification the HIP kernels are tested against.
 *   camera frame = panorama frame of the encoder's ERP rays (src/geometry/utils360.py:93-104,148-153):
 *     theta = atan2(t.x, t.z), phi = atan2(t.y, rho), rho = sqrt(t.x^2 + t.z^2), r = |t|
 *     pixel (centres at integer coordinates): u = (0.5 - theta / 2pi) W - 0.5,  v = (0.5 - phi / pi) H - 0.5
 *   EWA with the Jacobian of (u, v) w.r.t. t; near the poles rho is clamped to 0.05 r INSIDE THE JACOBIAN only (the
 *   role the 1.3 tan(fov) clamp plays in the perspective path); +0.3 dilation, conic, radius, 16x16 tile rect, sort
 *   key (float bits of the RADIAL distance r), cull r <= 0.2 and the composite are those of the perspective path.
 *   Seam: a Gaussian is splatted at u in [0, W) ("main" pair g) and, when its radius < W/2, once more at u +- W
 *   ("ghost" pair P + g) so that footprints crossing the +-pi seam appear on both sides; both pairs go through the
 *   ordinary clipped tile rectangle, share the image, and their gradients add.
 *   atan2 is evaluated with IEEE add / mul / div / sqrt only (two half-angle reductions + a 5-term series, |err| < 1e-7
 *   rad) so that the float32 oracle and the HIP kernel agree bit for bit on every integer intermediate.
 * ==================================================================================================== */
static inline real s_atan_small(real z) { /* |z| <= tan(pi/16) */
    real z2 = z * z;
    return z * (R(1) + z2 * (R(-0.3333333333333333) + z2 * (R(0.2) + z2 * (R(-0.14285714285714285) + z2 * R(0.1111111111111111)))));
}
static inline real s_atan_unit(real z) { /* 0 <= z <= 1 */
    real z1 = z / (R(1) + r_sqrt(R(1) + z * z));
    real z2 = z1 / (R(1) + r_sqrt(R(1) + z1 * z1));
    return R(4) * s_atan_small(z2);
}
static inline real s_atan2(real y, real x) {
    real ax = x < R(0) ? -x : x, ay = y < R(0) ? -y : y;
    if (ax == R(0) && ay == R(0)) return R(0);
    int swap = ay > ax;
    real a = s_atan_unit(swap ? ax / ay : ay / ax);
    if (swap) a = R(1.5707963267948966) - a;
    if (x < R(0)) a = R(3.141592653589793) - a;
    return y < R(0) ? -a : a;
}

typedef struct {
    real t[3], r2, r, rho2, rho, rc;
    int clamped;
    real u, v;          /* pixel coordinates of the main copy */
    real J0[3], J1[3];  /* rows of d(u,v)/dt (J0[1] == 0) */
    real M0[3], M1[3], v0[3], v1[3];
    real a, b, c;
} GeoS;

static void geo_sph(const OrcParams* p, const real* mean, const real* c6, GeoS* g) {
    const real* V = p->viewmatrix;
    xform43(V, mean, g->t);
    const real t0 = g->t[0], t1 = g->t[1], t2 = g->t[2];
    g->rho2 = t0 * t0 + t2 * t2;
    g->r2 = g->rho2 + t1 * t1;
    g->r = r_sqrt(g->r2);
    g->rho = r_sqrt(g->rho2);
    g->clamped = g->rho < R(0.05) * g->r;
    g->rc = g->clamped ? R(0.05) * g->r : g->rho;
    const real theta = s_atan2(t0, t2), phi = s_atan2(t1, g->rho);
    g->u = (R(0.5) - theta / R(6.283185307179586)) * (real)p->W - R(0.5);
    g->v = (R(0.5) - phi / R(3.141592653589793)) * (real)p->H - R(0.5);
    const real c0 = -(real)p->W / R(6.283185307179586), c1 = -(real)p->H / R(3.141592653589793);
    const real A = R(1) / (g->rc * g->rc), Bq = R(1) / (g->r2 * g->rc), Cq = g->rc / g->r2;
    g->J0[0] = c0 * t2 * A; g->J0[1] = R(0); g->J0[2] = -(c0 * t0 * A);
    g->J1[0] = -(c1 * t0 * t1 * Bq); g->J1[1] = c1 * Cq; g->J1[2] = -(c1 * t2 * t1 * Bq);
    for (int j = 0; j < 3; ++j) {
        g->M0[j] = g->J0[0] * V[j * 4 + 0] + g->J0[2] * V[j * 4 + 2];
        g->M1[j] = g->J1[0] * V[j * 4 + 0] + g->J1[1] * V[j * 4 + 1] + g->J1[2] * V[j * 4 + 2];
    }
    real S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    for (int k = 0; k < 3; ++k) {
        g->v0[k] = S[k][0] * g->M0[0] + S[k][1] * g->M0[1] + S[k][2] * g->M0[2];
        g->v1[k] = S[k][0] * g->M1[0] + S[k][1] * g->M1[1] + S[k][2] * g->M1[2];
    }
    g->a = g->M0[0] * g->v0[0] + g->M0[1] * g->v0[1] + g->M0[2] * g->v0[2];
    g->b = g->M1[0] * g->v0[0] + g->M1[1] * g->v0[1] + g->M1[2] * g->v0[2];
    g->c = g->M1[0] * g->v1[0] + g->M1[1] * g->v1[1] + g->M1[2] * g->v1[2];
    g->a += R(0.3);
    g->c += R(0.3);
}

static void shade_one(Orc* o, int i, int gi) {  /* colour of pair i from Gaussian gi (shared by both projection modes) */
    const OrcParams* p = &o->prm;
    const real* mean = o->means + 3 * gi;
    if (p->use_sh) {
        real d[3] = {mean[0] - p->campos[0], mean[1] - p->campos[1], mean[2] - p->campos[2]};
        real inv = R(1) / r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        real x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
        real Y[25];
        sh_basis(p->sh_degree, x, y, z, Y);
        int n = (p->sh_degree + 1) * (p->sh_degree + 1);
        const real* sh = o->shs + (size_t)gi * p->M * 3;
        for (int c = 0; c < 3; ++c) {
            real acc = R(0);
            for (int k = 0; k < n; ++k) acc += Y[k] * sh[k * 3 + c];
            acc += R(0.5);
            o->clamped[3 * i + c] = acc < R(0);
            o->rgb[3 * i + c] = r_max(acc, R(0));
        }
    } else {
        for (int c = 0; c < 3; ++c) {
            o->rgb[3 * i + c] = o->colors_in[3 * gi + c];
            o->clamped[3 * i + c] = 0;
        }
    }
}

static void preprocess_one_sph(Orc* o, int i) {  /* i in [0, 2P): pair; ghost pairs are i >= P */
    const OrcParams* p = &o->prm;
    const int gi = i % p->P, ghost = i >= p->P;
    o->radii[i] = 0;
    o->tiles_touched[i] = 0;
    const real* mean = o->means + 3 * gi;
    GeoS g;
    geo_sph(p, mean, o->cov6 + 6 * gi, &g);
    if (g.r <= R(0.2)) return;
    real det = g.a * g.c - g.b * g.b;
    if (det == R(0)) return;
    real det_inv = R(1) / det;
    real conA = g.c * det_inv, conB = -g.b * det_inv, conC = g.a * det_inv;
    real mid = R(0.5) * (g.a + g.c);
    real sq = r_sqrt(r_max(R(0.1), mid * mid - det));
    real lam1 = mid + sq, lam2 = mid - sq;
    int radius = r2i(r_ceil(R(3) * r_sqrt(r_max(lam1, lam2))));
    real px = g.u, py = g.v;
    if (ghost) {
        if (radius >= p->W / 2) return; /* a footprint wider than half the panorama keeps its main copy only */
        px = px < R(0.5) * (real)p->W ? px + (real)p->W : px - (real)p->W;
    }
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
    real rr = (real)radius;
    int minx = i_min(gx, i_max(0, r2i((px - rr) / R(TILE))));
    int miny = i_min(gy, i_max(0, r2i((py - rr) / R(TILE))));
    int maxx = i_min(gx, i_max(0, r2i((px + rr + R(TILE - 1)) / R(TILE))));
    int maxy = i_min(gy, i_max(0, r2i((py + rr + R(TILE - 1)) / R(TILE))));
    if ((maxx - minx) * (maxy - miny) == 0) return;
    shade_one(o, i, gi);
    o->depth[i] = g.r;
    o->radii[i] = radius;
    o->xy[2 * i] = px;
    o->xy[2 * i + 1] = py;
    o->conic_op[4 * i + 0] = conA;
    o->conic_op[4 * i + 1] = conB;
    o->conic_op[4 * i + 2] = conC;
    o->conic_op[4 * i + 3] = o->opac[gi];
    o->rect[4 * i + 0] = minx;
    o->rect[4 * i + 1] = miny;
    o->rect[4 * i + 2] = maxx;
    o->rect[4 * i + 3] = maxy;
    o->tiles_touched[i] = (uint32_t)((maxx - minx) * (maxy - miny));
}

static void preprocess_one(Orc* o, int i) {
    const OrcParams* p = &o->prm;
    o->radii[i] = 0;
    o->tiles_touched[i] = 0;
    const real* mean = o->means + 3 * i;
    real pv[3];
    xform43(p->viewmatrix, mean, pv);
    if (pv[2] <= R(0.2)) return; /* near cull (A.1 step 2) */
    real ph[4];
    xform44(p->projmatrix, mean, ph);
    real pw = R(1) / (ph[3] + R(0.0000001));
    real prx = ph[0] * pw, pry = ph[1] * pw;
    Geo g;
    geo_compute(p, mean, o->cov6 + 6 * i, &g);
    real det = g.a * g.c - g.b * g.b;
    if (det == R(0)) return;
    real det_inv = R(1) / det;
    real conA = g.c * det_inv, conB = -g.b * det_inv, conC = g.a * det_inv;
    real mid = R(0.5) * (g.a + g.c);
    real sq = r_sqrt(r_max(R(0.1), mid * mid - det));
    real lam1 = mid + sq, lam2 = mid - sq;
    int radius = r2i(r_ceil(R(3) * r_sqrt(r_max(lam1, lam2))));
    real px = ((prx + R(1)) * (real)p->W - R(1)) * R(0.5);
    real py = ((pry + R(1)) * (real)p->H - R(1)) * R(0.5);
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
    real rr = (real)radius;
    int minx = i_min(gx, i_max(0, r2i((px - rr) / R(TILE))));
    int miny = i_min(gy, i_max(0, r2i((py - rr) / R(TILE))));
    int maxx = i_min(gx, i_max(0, r2i((px + rr + R(TILE - 1)) / R(TILE))));
    int maxy = i_min(gy, i_max(0, r2i((py + rr + R(TILE - 1)) / R(TILE))));
    if ((maxx - minx) * (maxy - miny) == 0) return;
    if (p->use_sh) {
        real d[3] = {mean[0] - p->campos[0], mean[1] - p->campos[1], mean[2] - p->campos[2]};
        real inv = R(1) / r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        real x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
        real Y[25];
        sh_basis(p->sh_degree, x, y, z, Y);
        int n = (p->sh_degree + 1) * (p->sh_degree + 1);
        const real* sh = o->shs + (size_t)i * p->M * 3;
        for (int c = 0; c < 3; ++c) {
            real acc = R(0);
            for (int k = 0; k < n; ++k) acc += Y[k] * sh[k * 3 + c];
            acc += R(0.5);
            o->clamped[3 * i + c] = acc < R(0);
            o->rgb[3 * i + c] = r_max(acc, R(0));
        }
    } else {
        for (int c = 0; c < 3; ++c) {
            o->rgb[3 * i + c] = o->colors_in[3 * i + c];
            o->clamped[3 * i + c] = 0;
        }
    }
    o->depth[i] = pv[2];
    o->radii[i] = radius;
    o->xy[2 * i] = px;
    o->xy[2 * i + 1] = py;
    o->conic_op[4 * i + 0] = conA;
    o->conic_op[4 * i + 1] = conB;
    o->conic_op[4 * i + 2] = conC;
    o->conic_op[4 * i + 3] = o->opac[i];
    o->rect[4 * i + 0] = minx;
    o->rect[4 * i + 1] = miny;
    o->rect[4 * i + 2] = maxx;
    o->rect[4 * i + 3] = maxy;
    o->tiles_touched[i] = (uint32_t)((maxx - minx) * (maxy - miny));
}

/* ---- stable merge sort of (tile, depth, emission index) ---- */
typedef struct {
    uint32_t tile;
    real depth;
    uint32_t val;
} Inst;

static int inst_less(const Inst* a, const Inst* b) {
    if (a->tile != b->tile) return a->tile < b->tile;
    return a->depth < b->depth;
}
static void merge_sort(Inst* a, Inst* tmp, size_t n) {
    if (n < 2) return;
    size_t h = n / 2;
    merge_sort(a, tmp, h);
    merge_sort(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = inst_less(&a[j], &a[i]) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(Inst));
}

static void bin_and_sort(Orc* o) {
    const OrcParams* p = &o->prm;
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
    uint64_t run = 0;
    for (int i = 0; i < o->NP; ++i) {
        run += o->tiles_touched[i];
        o->offsets[i] = (uint32_t)run;
    }
    o->L = run;
    free(o->keys); free(o->vals);
    o->keys = (uint64_t*)malloc((run ? run : 1) * sizeof(uint64_t));
    o->vals = (uint32_t*)malloc((run ? run : 1) * sizeof(uint32_t));
    Inst* inst = (Inst*)malloc((run ? run : 1) * sizeof(Inst));
    Inst* tmp = (Inst*)malloc((run ? run : 1) * sizeof(Inst));
    size_t k = 0;
    for (int i = 0; i < o->NP; ++i) {
        if (o->radii[i] <= 0) continue;
        const int32_t* r = o->rect + 4 * i;
        for (int y = r[1]; y < r[3]; ++y)
            for (int x = r[0]; x < r[2]; ++x) {
                inst[k].tile = (uint32_t)(y * gx + x);
                inst[k].depth = o->depth[i];
                inst[k].val = (uint32_t)i;
                ++k;
            }
    }
    merge_sort(inst, tmp, (size_t)run);
    memset(o->ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (size_t j = 0; j < run; ++j) {
        float df = (float)inst[j].depth;
        uint32_t bits;
        memcpy(&bits, &df, 4);
        o->keys[j] = ((uint64_t)inst[j].tile << 32) | bits;
        o->vals[j] = inst[j].val;
        if (j == 0 || inst[j - 1].tile != inst[j].tile) o->ranges[2 * inst[j].tile] = (uint32_t)j;
        if (j + 1 == run || inst[j + 1].tile != inst[j].tile) o->ranges[2 * inst[j].tile + 1] = (uint32_t)(j + 1);
    }
    free(inst);
    free(tmp);
}

static void render_forward(Orc* o) {
    const OrcParams* p = &o->prm;
    int gx = (p->W + TILE - 1) / TILE;
    int H = p->H, W = p->W;
#pragma omp parallel for schedule(dynamic, 64)
    for (int pix = 0; pix < H * W; ++pix) {
        int py = pix / W, px = pix % W;
        int tile = (py / TILE) * gx + (px / TILE);
        uint32_t s = o->ranges[2 * tile], e = o->ranges[2 * tile + 1];
        real T = R(1), C[3] = {R(0), R(0), R(0)};
        uint32_t contributor = 0, last = 0;
        real pxf = (real)px, pyf = (real)py;
        for (uint32_t j = s; j < e; ++j) {
            ++contributor;
            uint32_t id = o->vals[j];
            real dx = o->xy[2 * id] - pxf, dy = o->xy[2 * id + 1] - pyf;
            const real* co = o->conic_op + 4 * id;
            real power = -R(0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > R(0)) continue;
            real alpha = r_min(R(0.99), co[3] * r_exp(power));
            if (alpha < R(1.0) / R(255.0)) continue;
            real test_T = T * (R(1) - alpha);
            if (test_T < R(0.0001)) break;
            for (int c = 0; c < 3; ++c) C[c] += o->rgb[3 * id + c] * alpha * T;
            T = test_T;
            last = contributor;
        }
        o->final_T[pix] = T;
        o->n_contrib[pix] = last;
        for (int c = 0; c < 3; ++c) o->image[(size_t)c * H * W + pix] = C[c] + T * p->bg[c];
    }
}

/* ------------------------------------------------------------------ API */
#define EXPORT __attribute__((visibility("default")))

EXPORT int orc_real_bytes(void) { return (int)sizeof(real); }
EXPORT int orc_params_bytes(void) { return (int)sizeof(OrcParams); }

EXPORT Orc* orc_create(const OrcParams* prm, const real* means, const real* cov6, const real* opac,
                       const real* shs, const real* colors) {
    Orc* o = (Orc*)calloc(1, sizeof(Orc));
    o->prm = *prm;
    int P = prm->P;
    o->NP = prm->spherical ? 2 * P : P;
    size_t np = (size_t)(P ? P : 1);
    o->means = (real*)malloc(np * 3 * sizeof(real)); memcpy(o->means, means, (size_t)P * 3 * sizeof(real));
    o->cov6 = (real*)malloc(np * 6 * sizeof(real)); memcpy(o->cov6, cov6, (size_t)P * 6 * sizeof(real));
    o->opac = (real*)malloc(np * sizeof(real)); memcpy(o->opac, opac, (size_t)P * sizeof(real));
    if (prm->use_sh) {
        size_t n = (size_t)P * prm->M * 3;
        o->shs = (real*)malloc((n ? n : 1) * sizeof(real)); memcpy(o->shs, shs, n * sizeof(real));
    } else {
        o->colors_in = (real*)malloc(np * 3 * sizeof(real)); memcpy(o->colors_in, colors, (size_t)P * 3 * sizeof(real));
    }
    np = (size_t)(o->NP ? o->NP : 1); /* per-pair state from here on */
    o->radii = (int32_t*)calloc(np, sizeof(int32_t));
    o->tiles_touched = (uint32_t*)calloc(np, sizeof(uint32_t));
    o->offsets = (uint32_t*)calloc(np, sizeof(uint32_t));
    o->rect = (int32_t*)calloc(np * 4, sizeof(int32_t));
    o->xy = (real*)calloc(np * 2, sizeof(real));
    o->depth = (real*)calloc(np, sizeof(real));
    o->conic_op = (real*)calloc(np * 4, sizeof(real));
    o->rgb = (real*)calloc(np * 3, sizeof(real));
    o->clamped = (uint8_t*)calloc(np * 3, 1);
    int gx = (prm->W + TILE - 1) / TILE, gy = (prm->H + TILE - 1) / TILE;
    o->ranges = (uint32_t*)calloc((size_t)gx * gy * 2 + 2, sizeof(uint32_t));
    size_t npix = (size_t)prm->H * prm->W;
    o->image = (real*)calloc(npix * 3 + 1, sizeof(real));
    o->final_T = (real*)calloc(npix + 1, sizeof(real));
    o->n_contrib = (uint32_t*)calloc(npix + 1, sizeof(uint32_t));
    o->g_xy = (real*)calloc(np * 2, sizeof(real));
    o->g_conic = (real*)calloc(np * 3, sizeof(real));
    o->g_op = (real*)calloc(np, sizeof(real));
    o->g_rgb = (real*)calloc(np * 3, sizeof(real));
    return o;
}

EXPORT void orc_destroy(Orc* o) {
    if (!o) return;
    free(o->means); free(o->cov6); free(o->opac); free(o->shs); free(o->colors_in);
    free(o->radii); free(o->tiles_touched); free(o->offsets); free(o->rect); free(o->xy);
    free(o->depth); free(o->conic_op); free(o->rgb); free(o->clamped); free(o->keys);
    free(o->vals); free(o->ranges); free(o->image); free(o->final_T); free(o->n_contrib);
    free(o->g_xy); free(o->g_conic); free(o->g_op); free(o->g_rgb);
    free(o);
}

EXPORT uint64_t orc_forward(Orc* o) {
#pragma omp parallel for schedule(static, 1024)
    for (int i = 0; i < o->NP; ++i) {
        if (o->prm.spherical) preprocess_one_sph(o, i); else preprocess_one(o, i);
    }
    bin_and_sort(o);
    render_forward(o);
    return o->L;
}

/* accessors (pointers stay valid until the next orc_forward / orc_destroy) */
EXPORT const int32_t* orc_radii(Orc* o) { return o->radii; }
EXPORT const uint32_t* orc_tiles_touched(Orc* o) { return o->tiles_touched; }
EXPORT const uint32_t* orc_offsets(Orc* o) { return o->offsets; }
EXPORT const int32_t* orc_rect(Orc* o) { return o->rect; }
EXPORT const real* orc_xy(Orc* o) { return o->xy; }
EXPORT const real* orc_depth(Orc* o) { return o->depth; }
EXPORT const real* orc_conic_opacity(Orc* o) { return o->conic_op; }
EXPORT const real* orc_rgb(Orc* o) { return o->rgb; }
EXPORT const uint8_t* orc_clamped(Orc* o) { return o->clamped; }
EXPORT uint64_t orc_num_rendered(Orc* o) { return o->L; }
EXPORT const uint64_t* orc_keys(Orc* o) { return o->keys; }
EXPORT const uint32_t* orc_values(Orc* o) { return o->vals; }
EXPORT const uint32_t* orc_ranges(Orc* o) { return o->ranges; }
EXPORT const real* orc_image(Orc* o) { return o->image; }
EXPORT const real* orc_final_T(Orc* o) { return o->final_T; }
EXPORT const uint32_t* orc_n_contrib(Orc* o) { return o->n_contrib; }
EXPORT const real* orc_grad_xy_pix(Orc* o) { return o->g_xy; }
EXPORT const real* orc_grad_conic(Orc* o) { return o->g_conic; }
EXPORT const real* orc_grad_opacity_raster(Orc* o) { return o->g_op; }
EXPORT const real* orc_grad_rgb(Orc* o) { return o->g_rgb; }

/* Per-Gaussian backward of the spherical mode: raster gradients of the main pair i and the ghost pair P + i add (the
 * ghost's centre is the main one shifted by a constant), then chain through geo_sph.  d_means2D = (dL/du, dL/dv, 0)
 * in PIXEL units. */
static void backward_one_sph(Orc* o, int i, real* dm, real* dm2, real* dc, real* dsh, real* dcol, real* dop) {
    const OrcParams* p = &o->prm;
    const int P = p->P, j2 = P + i;
    const int vis0 = o->radii[i] > 0, vis1 = o->radii[j2] > 0;
    if (!vis0 && !vis1) return;
    real gxy[2] = {R(0), R(0)}, gcon[3] = {R(0), R(0), R(0)}, grgb[3] = {R(0), R(0), R(0)}, gop = R(0);
    const int ids[2] = {i, j2};
    for (int q = 0; q < 2; ++q) {
        if (!(q ? vis1 : vis0)) continue;
        const int id = ids[q];
        gxy[0] += o->g_xy[2 * id]; gxy[1] += o->g_xy[2 * id + 1];
        for (int k = 0; k < 3; ++k) gcon[k] += o->g_conic[3 * id + k];
        for (int k = 0; k < 3; ++k) grgb[k] += o->clamped[3 * id + k] && p->use_sh ? R(0) : o->g_rgb[3 * id + k];
        gop += o->g_op[id];
    }
    *dop = gop;
    dm2[0] = gxy[0]; dm2[1] = gxy[1]; dm2[2] = R(0);
    const real* mean = o->means + 3 * i;
    GeoS g;
    geo_sph(p, mean, o->cov6 + 6 * i, &g);
    real a = g.a, b = g.b, c = g.c;
    real det = a * c - b * b;
    real d2inv = R(1) / (det * det + R(0.0000001));
    real gA = gcon[0], gB = gcon[1], gC = gcon[2];
    real dL_da = R(0), dL_db = R(0), dL_dc = R(0);
    if (d2inv != R(0)) {
        dL_da = d2inv * (-c * c * gA + b * c * gB + (det - a * c) * gC);
        dL_dc = d2inv * (-a * a * gC + a * b * gB + (det - a * c) * gA);
        dL_db = d2inv * (R(2) * b * c * gA - (det + R(2) * b * b) * gB + R(2) * a * b * gC);
        const real *M0 = g.M0, *M1 = g.M1;
        dc[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
        dc[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
        dc[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
        dc[1] = R(2) * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + R(2) * M1[0] * M1[1] * dL_dc;
        dc[2] = R(2) * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + R(2) * M1[0] * M1[2] * dL_dc;
        dc[4] = R(2) * M0[1] * M0[2] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + R(2) * M1[1] * M1[2] * dL_dc;
    }
    real dM0[3], dM1[3];
    for (int j = 0; j < 3; ++j) {
        dM0[j] = R(2) * dL_da * g.v0[j] + dL_db * g.v1[j];
        dM1[j] = R(2) * dL_dc * g.v1[j] + dL_db * g.v0[j];
    }
    const real* V = p->viewmatrix;
    real dJ00 = R(0), dJ02 = R(0), dJ10 = R(0), dJ11 = R(0), dJ12 = R(0);
    for (int j = 0; j < 3; ++j) {
        dJ00 += dM0[j] * V[j * 4 + 0];
        dJ02 += dM0[j] * V[j * 4 + 2];
        dJ10 += dM1[j] * V[j * 4 + 0];
        dJ11 += dM1[j] * V[j * 4 + 1];
        dJ12 += dM1[j] * V[j * 4 + 2];
    }
    const real t0 = g.t[0], t1 = g.t[1], t2 = g.t[2], rc = g.rc, r2 = g.r2;
    const real c0 = -(real)p->W / R(6.283185307179586), c1 = -(real)p->H / R(3.141592653589793);
    const real A = R(1) / (rc * rc), Bq = R(1) / (r2 * rc);
    /* J00 = c0 t2 A, J02 = -c0 t0 A, J10 = -c1 t0 t1 Bq, J11 = c1 rc / r2, J12 = -c1 t2 t1 Bq */
    real dt[3];
    dt[0] = -(c0 * A) * dJ02 - (c1 * t1 * Bq) * dJ10;
    dt[1] = -(c1 * Bq) * (t0 * dJ10 + t2 * dJ12);
    dt[2] = (c0 * A) * dJ00 - (c1 * t1 * Bq) * dJ12;
    const real dA = c0 * (dJ00 * t2 - dJ02 * t0);
    const real dB = -(c1 * t1) * (dJ10 * t0 + dJ12 * t2);
    const real dC = c1 * dJ11;
    const real drc = dA * (-R(2) / (rc * rc * rc)) + dB * (-R(1) / (r2 * rc * rc)) + dC / r2;
    const real dr2 = dB * (-R(1) / (r2 * r2 * rc)) + dC * (-rc / (r2 * r2));
    for (int k = 0; k < 3; ++k) dt[k] += R(2) * g.t[k] * dr2;
    if (g.clamped) {
        for (int k = 0; k < 3; ++k) dt[k] += drc * (R(0.05) * g.t[k] / g.r);
    } else {
        dt[0] += drc * (t0 / g.rho);
        dt[2] += drc * (t2 / g.rho);
    }
    /* centre: (u, v) with the TRUE rho */
    const real iu = c0 / g.rho2, iv = c1 / (r2 * g.rho);
    dt[0] += gxy[0] * (iu * t2) + gxy[1] * (-(iv * t0 * t1));
    dt[1] += gxy[1] * (c1 * g.rho / r2);
    dt[2] += gxy[0] * (-(iu * t0)) + gxy[1] * (-(iv * t2 * t1));
    for (int j = 0; j < 3; ++j) dm[j] = V[j * 4 + 0] * dt[0] + V[j * 4 + 1] * dt[1] + V[j * 4 + 2] * dt[2];
    if (!p->use_sh) {
        if (dcol) for (int cc = 0; cc < 3; ++cc) dcol[cc] = grgb[cc];
    } else if (dsh) {
        real d[3] = {mean[0] - p->campos[0], mean[1] - p->campos[1], mean[2] - p->campos[2]};
        real len = r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        real inv = R(1) / len;
        real x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
        real Y[25], bx[25], by[25], bz[25];
        sh_basis(p->sh_degree, x, y, z, Y);
        sh_basis_grad(p->sh_degree, x, y, z, bx, by, bz);
        int n = (p->sh_degree + 1) * (p->sh_degree + 1);
        const real* sh = o->shs + (size_t)i * p->M * 3;
        real ddir[3] = {R(0), R(0), R(0)};
        for (int k = 0; k < n; ++k) {
            real s_ = R(0);
            for (int cc = 0; cc < 3; ++cc) {
                dsh[k * 3 + cc] = Y[k] * grgb[cc];
                s_ += sh[k * 3 + cc] * grgb[cc];
            }
            ddir[0] += bx[k] * s_;
            ddir[1] += by[k] * s_;
            ddir[2] += bz[k] * s_;
        }
        real dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
        dm[0] += (ddir[0] - x * dot) * inv;
        dm[1] += (ddir[1] - y * dot) * inv;
        dm[2] += (ddir[2] - z * dot) * inv;
    }
}

/*
 * Backward (Appendix A.4).  dL_dimage is [3,H,W].  Outputs (caller-allocated):
 *   d_means3D[P,3], d_means2D[P,3] (NDC-scaled: pixel gradient * (0.5W, 0.5H), z = 0),
 *   d_cov6[P,6], d_sh[P,M,3] (or NULL), d_colors[P,3] (or NULL), d_opacity[P].
 * Per-pixel contributions are summed in pixel-index order (deterministic).
 */
/* 0 (default): serial, pixel-index summation order (deterministic, used by the parity tests).
 * 1: OpenMP over pixels with atomic accumulation (used only to TIME the CPU baseline). */
static int g_parallel_backward = 0;
EXPORT void orc_set_parallel_backward(int on) { g_parallel_backward = on; }

EXPORT void orc_backward(Orc* o, const real* dL_dimage, real* d_means3D, real* d_means2D,
                         real* d_cov6, real* d_sh, real* d_colors, real* d_opacity) {
    const OrcParams* p = &o->prm;
    int P = p->P, H = p->H, W = p->W;
    int gx = (W + TILE - 1) / TILE;
    memset(o->g_xy, 0, sizeof(real) * 2 * (size_t)o->NP);
    memset(o->g_conic, 0, sizeof(real) * 3 * (size_t)o->NP);
    memset(o->g_op, 0, sizeof(real) * (size_t)o->NP);
    memset(o->g_rgb, 0, sizeof(real) * 3 * (size_t)o->NP);
    /* --- render backward: back-to-front replay per pixel --- */
    const int par = g_parallel_backward;
#define ACC(dst, val) do { real v_ = (val); if (par) { _Pragma("omp atomic") dst += v_; } else dst += v_; } while (0)
#pragma omp parallel for schedule(dynamic, 64) if (par)
    for (int pix = 0; pix < H * W; ++pix) {
        int py = pix / W, px = pix % W;
        int tile = (py / TILE) * gx + (px / TILE);
        uint32_t s = o->ranges[2 * tile];
        real T_final = o->final_T[pix];
        real T = T_final;
        uint32_t last = o->n_contrib[pix];
        real dpix[3] = {dL_dimage[pix], dL_dimage[(size_t)H * W + pix], dL_dimage[(size_t)2 * H * W + pix]};
        real bg_dot = p->bg[0] * dpix[0] + p->bg[1] * dpix[1] + p->bg[2] * dpix[2];
        real accum[3] = {R(0), R(0), R(0)}, last_color[3] = {R(0), R(0), R(0)};
        real last_alpha = R(0);
        real pxf = (real)px, pyf = (real)py;
        for (uint32_t k = last; k-- > 0;) {
            uint32_t id = o->vals[s + k];
            real dx = o->xy[2 * id] - pxf, dy = o->xy[2 * id + 1] - pyf;
            const real* co = o->conic_op + 4 * id;
            real power = -R(0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > R(0)) continue;
            real G = r_exp(power);
            real alpha = r_min(R(0.99), co[3] * G);
            if (alpha < R(1.0) / R(255.0)) continue;
            T = T / (R(1) - alpha);
            real dchannel_dcolor = alpha * T;
            real dL_dalpha = R(0);
            for (int c = 0; c < 3; ++c) {
                real col = o->rgb[3 * id + c];
                accum[c] = last_alpha * last_color[c] + (R(1) - last_alpha) * accum[c];
                last_color[c] = col;
                dL_dalpha += (col - accum[c]) * dpix[c];
                ACC(o->g_rgb[3 * id + c], dchannel_dcolor * dpix[c]);
            }
            dL_dalpha *= T;
            last_alpha = alpha;
            dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot;
            real dL_dG = co[3] * dL_dalpha;
            real gdx = G * dx, gdy = G * dy;
            real dG_ddelx = -gdx * co[0] - gdy * co[1];
            real dG_ddely = -gdy * co[2] - gdx * co[1];
            ACC(o->g_xy[2 * id], dL_dG * dG_ddelx);
            ACC(o->g_xy[2 * id + 1], dL_dG * dG_ddely);
            ACC(o->g_conic[3 * id + 0], -R(0.5) * gdx * dx * dL_dG);
            ACC(o->g_conic[3 * id + 1], -gdx * dy * dL_dG); /* true d/dB (no half factor) */
            ACC(o->g_conic[3 * id + 2], -R(0.5) * gdy * dy * dL_dG);
            ACC(o->g_op[id], G * dL_dalpha);
        }
    }
    /* --- per-Gaussian backward --- */
#pragma omp parallel for schedule(static, 1024) if (par)
    for (int i = 0; i < P; ++i) {
        real* dm = d_means3D + 3 * i;
        dm[0] = dm[1] = dm[2] = R(0);
        d_means2D[3 * i] = d_means2D[3 * i + 1] = d_means2D[3 * i + 2] = R(0);
        for (int k = 0; k < 6; ++k) d_cov6[6 * i + k] = R(0);
        d_opacity[i] = R(0);
        if (d_sh) for (int k = 0; k < p->M * 3; ++k) d_sh[(size_t)i * p->M * 3 + k] = R(0);
        if (d_colors) d_colors[3 * i] = d_colors[3 * i + 1] = d_colors[3 * i + 2] = R(0);
        if (p->spherical) {
            backward_one_sph(o, i, dm, d_means2D + 3 * i, d_cov6 + 6 * i, d_sh ? d_sh + (size_t)i * p->M * 3 : NULL,
                             d_colors ? d_colors + 3 * i : NULL, d_opacity + i);
            continue;
        }
        if (o->radii[i] <= 0) continue;
        const real* mean = o->means + 3 * i;
        d_opacity[i] = o->g_op[i];
        /* cov2D / conic chain */
        Geo g;
        geo_compute(p, mean, o->cov6 + 6 * i, &g);
        real a = g.a, b = g.b, c = g.c;
        real det = a * c - b * b;
        real d2inv = R(1) / (det * det + R(0.0000001));
        real gA = o->g_conic[3 * i], gB = o->g_conic[3 * i + 1], gC = o->g_conic[3 * i + 2];
        real dL_da = R(0), dL_db = R(0), dL_dc = R(0);
        if (d2inv != R(0)) {
            dL_da = d2inv * (-c * c * gA + b * c * gB + (det - a * c) * gC);
            dL_dc = d2inv * (-a * a * gC + a * b * gB + (det - a * c) * gA);
            dL_db = d2inv * (R(2) * b * c * gA - (det + R(2) * b * b) * gB + R(2) * a * b * gC);
            const real *M0 = g.M0, *M1 = g.M1;
            real* dc = d_cov6 + 6 * i;
            dc[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
            dc[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
            dc[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
            dc[1] = R(2) * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + R(2) * M1[0] * M1[1] * dL_dc;
            dc[2] = R(2) * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + R(2) * M1[0] * M1[2] * dL_dc;
            dc[4] = R(2) * M0[1] * M0[2] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + R(2) * M1[1] * M1[2] * dL_dc;
        }
        /* dL/dM rows, then J, then t, then mean */
        real dM0[3], dM1[3];
        for (int j = 0; j < 3; ++j) {
            dM0[j] = R(2) * dL_da * g.v0[j] + dL_db * g.v1[j];
            dM1[j] = R(2) * dL_dc * g.v1[j] + dL_db * g.v0[j];
        }
        const real* V = p->viewmatrix;
        real dJ00 = R(0), dJ02 = R(0), dJ11 = R(0), dJ12 = R(0);
        for (int j = 0; j < 3; ++j) {
            dJ00 += dM0[j] * V[j * 4 + 0];
            dJ02 += dM0[j] * V[j * 4 + 2];
            dJ11 += dM1[j] * V[j * 4 + 1];
            dJ12 += dM1[j] * V[j * 4 + 2];
        }
        real tz = R(1) / g.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        real dt[3];
        dt[0] = (g.xin ? R(1) : R(0)) * (-g.fx * tz2 * dJ02);
        dt[1] = (g.yin ? R(1) : R(0)) * (-g.fy * tz2 * dJ12);
        dt[2] = -g.fx * tz2 * dJ00 - g.fy * tz2 * dJ11 + (R(2) * g.fx * g.txc) * tz3 * dJ02 +
                (R(2) * g.fy * g.tyc) * tz3 * dJ12;
        for (int j = 0; j < 3; ++j) dm[j] = V[j * 4 + 0] * dt[0] + V[j * 4 + 1] * dt[1] + V[j * 4 + 2] * dt[2];
        /* projection chain: dL/dmean2D in NDC-scaled units */
        real mx = o->g_xy[2 * i] * (R(0.5) * (real)W), my = o->g_xy[2 * i + 1] * (R(0.5) * (real)H);
        d_means2D[3 * i] = mx;
        d_means2D[3 * i + 1] = my;
        const real* Pm = p->projmatrix;
        real mh[4];
        xform44(Pm, mean, mh);
        real mw = R(1) / (mh[3] + R(0.0000001));
        real mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
        dm[0] += (Pm[0] * mw - Pm[3] * mul1) * mx + (Pm[1] * mw - Pm[3] * mul2) * my;
        dm[1] += (Pm[4] * mw - Pm[7] * mul1) * mx + (Pm[5] * mw - Pm[7] * mul2) * my;
        dm[2] += (Pm[8] * mw - Pm[11] * mul1) * mx + (Pm[9] * mw - Pm[11] * mul2) * my;
        /* colour chain */
        if (!p->use_sh) {
            if (d_colors) for (int cc = 0; cc < 3; ++cc) d_colors[3 * i + cc] = o->g_rgb[3 * i + cc];
        } else if (d_sh) {
            real dRGB[3];
            for (int cc = 0; cc < 3; ++cc) dRGB[cc] = o->clamped[3 * i + cc] ? R(0) : o->g_rgb[3 * i + cc];
            real d[3] = {mean[0] - p->campos[0], mean[1] - p->campos[1], mean[2] - p->campos[2]};
            real len = r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            real inv = R(1) / len;
            real x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
            real Y[25], bx[25], by[25], bz[25];
            sh_basis(p->sh_degree, x, y, z, Y);
            sh_basis_grad(p->sh_degree, x, y, z, bx, by, bz);
            int n = (p->sh_degree + 1) * (p->sh_degree + 1);
            const real* sh = o->shs + (size_t)i * p->M * 3;
            real* dsh = d_sh + (size_t)i * p->M * 3;
            real ddir[3] = {R(0), R(0), R(0)};
            for (int k = 0; k < n; ++k) {
                real s = R(0);
                for (int cc = 0; cc < 3; ++cc) {
                    dsh[k * 3 + cc] = Y[k] * dRGB[cc];
                    s += sh[k * 3 + cc] * dRGB[cc];
                }
                ddir[0] += bx[k] * s;
                ddir[1] += by[k] * s;
                ddir[2] += bz[k] * s;
            }
            real dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
            dm[0] += (ddir[0] - x * dot) * inv;
            dm[1] += (ddir[1] - y * dot) * inv;
            dm[2] += (ddir[2] - z * dot) * inv;
        }
    }
}
