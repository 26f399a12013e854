/*
 * s360.h — C ABI of the MI355X-native panoramic Gaussian-splat rasteriser (libs360.so).
 *
 * Drop-in boundary for the one native dependency of thucz/splatter360's render path: the
 * `diff_gaussian_rasterization` extension imported at
 *     /root/reference/src/model/decoder/cuda_splatting.py:5-8
 * and called at cuda_splatting.py:99-124 (forward) / by autograd (backward).  The reference
 * defines no C ABI itself (its extension is a pybind11 module that is not vendored); each entry
 * point below names the upstream pybind function / reference call site it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is DEVICE memory (HIP, gfx950) unless
 *     the name ends in _host;  all tensors are contiguous float32 / int32 / uint32;
 *   - the caller owns every buffer (outputs and workspaces); sizes come from s360_layout();
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs NO host
 *     synchronisation and allocates nothing; kernels are re-entrant across streams/devices;
 *   - return value: 0 on success, negative S360_E_* otherwise; nothing throws across the ABI;
 *   - a call renders V views of ONE Gaussian cloud (V = 1 reproduces one reference rasteriser
 *     call; V = 6 renders the six cube faces of an equirectangular view in one fused pass,
 *     replacing the per-face Python loop at src/model/decoder/decoder_splatting_cuda.py:47-59).
 */
#ifndef S360_H
#define S360_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S360_ABI_VERSION 22
#define S360_MAX_VIEWS 8
#define S360_TILE 16

enum {
    S360_OK = 0,
    S360_E_BADARG = -1,    /* null pointer / non-positive size / V > S360_MAX_VIEWS */
    S360_E_WORKSPACE = -2, /* workspace smaller than s360_layout() reports */
    S360_E_LAUNCH = -3,    /* hipGetLastError() != hipSuccess after a launch */
    S360_E_UNSUPPORTED = -4
};

/* flags */
#define S360_FLAG_SHARED_CAMPOS 1u    /* all V views share campos (and scale): SH->RGB evaluated once per Gaussian */
#define S360_FLAG_FORWARD_ONLY 8u     /* inference: skip the instance-slot tables (only s360_backward needs them);
                                         s360_backward returns S360_E_BADARG on a workspace rendered with this flag */
#define S360_FLAG_COV9 2u             /* covariances given (and their gradient returned) as [P,3,3] row-major, the
                                         reference's Gaussians.covariances layout (src/model/types.py:9); only the
                                         upper triangle is read / receives gradient, exactly like the
                                         cov[:, row, col] gather at cuda_splatting.py:115,123 */
#define S360_FLAG_SH_CHANNEL_MAJOR 4u /* SH given (and gradient returned) as [P,3,M], the reference's
                                         Gaussians.harmonics layout (types.py:10) — no "b g xyz n -> b g n xyz"
                                         rearrange copy (cuda_splatting.py:75) */

/*
 * One camera: the per-call fields of GaussianRasterizationSettings (cuda_splatting.py:99-112)
 * plus the scale-invariant factor of cuda_splatting.py:64-71 and the view's near/far.  44 floats, DEVICE memory (array
 * of V).  `scale` multiplies means3D (and scale^2 the covariances) inside the kernels — pass the
 * UNSCALED cloud and scale = 1/near to fuse the reference's three full-size rescale copies; pass
 * 1.0 when the cloud is already scaled (drop-in rasteriser call).
 * viewmatrix / projmatrix are the flat [4,4] tensors handed over at cuda_splatting.py:86-87
 * (row-vector convention: element [r][c] of the transposed matrix at index 4*r+c).
 */
#define S360_FLAG_SH_DEG4_IGNORED 16u  /* treat sh_degree 4 as degree 3: coefficients 16..24 are ignored (zero gradient).
                                         The reference always passes sh_degree = 4 (config/model/encoder/costvolume.yaml:16)
                                         to a rasteriser fork whose degree-4 table cannot be verified here;
                                         default (flag clear) = the standard real-SH degree-4 table, this flag = the
                                         behaviour of a fork that stops at the upstream 3DGS degree 3. */

#define S360_FLAG_SPHERICAL 32u        /* native equirectangular splat mode (NO reference counterpart — the
                                         reference only ever renders cube faces — specified by oracle/s360_oracle.c geo_sph):
                                         every image is an H x W equirectangular panorama rendered with the spherical
                                         projection of the encoder's ERP ray convention (src/geometry/utils360.py:93-104,
                                         148-153) and its Jacobian; views come in PAIRS (2i = the panorama camera, 2i+1 = its
                                         seam ghost: the same camera, splat centres shifted by +-W), V must be even, the call
                                         renders V/2 images [V/2,3,H,W]; viewmatrix = inverse(panorama c2w)^T, projmatrix
                                         and tanfov are ignored; sort key and near cull use the RADIAL distance;
                                         d_means2D is in pixel units */

#define S360_FLAG_LEAN_LISTS 64u       /* binning-time exact cull: a (Gaussian, tile) instance is counted, emitted, sorted and given an
                                         instance slot only if the splat can reach alpha >= 1/255 on some pixel of that 16x16 tile
                                         (the tile-sized instance of the composites' own quadrant test; upstream bins the whole 3-sigma
                                         rectangle; rectangles of more than 32 tiles are binned whole).  Skipped
                                         instances contribute to no pixel, so images, final_T, radii and every gradient are
                                         BIT-IDENTICAL to a call without the flag; tiles_touched, the sorted lists, num_instances and
                                         n_contrib (list positions) differ.  Flag clear = the upstream-compatible lists (what the
                                         integer-state parity tests compare with the oracle). */

#define S360_FLAG_DEFER_LOSS 128u       /* s360_forward_mse on a training call with loss_out != NULL: do not launch the loss reduction at
                                         the end of the forward; the first launch of s360_backward / _split / _composite on this
                                         workspace performs it (same arithmetic, same fixed order).  loss_out / partials must stay
                                         allocated until then; loss_out holds the loss only after that backward.  For training loops
                                         that read the scalar after the backward (logging) — one launch less per step. */

#define S360_FLAG_ATOMIC_GRADS 256u     /* opt-in, training calls: the backward composite adds its per-(entry, quadrant) sums straight
                                         into the pair's raster-gradient record with return-less float32 atomics instead of leaving
                                         one partial record per (instance slot, quadrant) for a deterministic gather.  No partial
                                         slots, no validity flags, no gather launch: s360_layout's backward_bytes drops from
                                         ~256 B per instance of capacity to 48 B per (view, Gaussian) pair.  The summation ORDER is
                                         then run-dependent: gradients are NOT bit-reproducible (upstream's backward is atomic and
                                         non-deterministic too).  Default (flag clear): no float atomics anywhere. */

#define S360_FLAG_SPLIT_LISTS 512u      /* long tile lists are composited SEGMENT-PARALLEL (forward and backward).  Upstream's composite (and this
                                         library's without the flag) walks a tile's depth-sorted list front to back as ONE sequential
                                         chain per pixel block; a tile list of 10-17 K entries whose pixels do not saturate — the pole
                                         clumps of a panorama: the 1 024 Gaussians of every polar ERP row land on a few face pixels,
                                         /root/reference/src/geometry/utils360.py:93-104 — then costs one wave hundreds of microseconds
                                         while the rest of the chip idles.  With the flag, an 8x8 quadrant that after the first 1 024
                                         entries of a list of more than 2 048 still holds a pixel far from saturating (T >= 1/16) hands
                                         the rest over in segments of 512 entries, one wave each: phase 1 forms every segment's own
                                         transmittance T_k per pixel, phase 2 composites the segment with the sequential rule from the
                                         pixel's true incoming transmittance T_head T_2 ... T_(k-1) (pixels that stopped earlier are
                                         skipped), a combine adds the contributions in list order.  The backward composites the same
                                         segments in parallel from the forward's transmittance behind each segment and the colour
                                         accumulated behind it.  What changes: floating-point association — a pixel's transmittance
                                         at a segment start is a product of segment products (images within 1e-6 of the sequential
                                         composite; north_star's bar is 1e-5) — and with it a stop decision where a product lands
                                         within rounding of 1e-4.  Every (pixel, entry) pair is evaluated with the same arithmetic and
                                         the same stop rule in list order.  Quadrants that do not split — every list up to 2 048
                                         entries, every quadrant that is (nearly) saturated after 1 024 — are bit-identical with and
                                         without the flag. */

#define S360_FLAG_RAW_INPUTS 1024u      /* the call is s360_forward_raw / s360_backward_raw (the workspace also keeps the 7 raw geometry
                                         words per Gaussian for the backward); required by those two, rejected by the others */

#define S360_FLAG_COOP_WALK 2048u       /* binning variant for clouds with many LARGE footprints: a rectangle of more than 32 tiles is
                                         counted (k_preprocess) and emitted (k_emit) by all 64 lanes of its wave, one tile per lane,
                                         instead of by the one lane that owns the Gaussian while 63 wait (a near splat of a uniform
                                         random cloud covers a whole 16x16-tile face).  Separately compiled instances of the two
                                         kernels: the default ones carry none of it.  Every result of the call is bit-identical with
                                         and without the flag (same counts, same slots; the order of a tile's unsorted bucket is
                                         irrelevant).  header_mirror word 2 reports how many such rectangles a call had. */

typedef struct S360View {
    float viewmatrix[16];
    float projmatrix[16];
    float campos[3];
    float tanfovx, tanfovy;
    float bg[3];
    float scale;
    float near_plane, far_plane; /* UNSCALED near / far of the view (only read when a depth map is requested) */
    float _pad;
} S360View;

typedef struct S360Params {
    int32_t P;              /* Gaussians */
    int32_t V;              /* views rendered by this call (1..S360_MAX_VIEWS) */
    int32_t H, W;           /* image size of every view */
    int32_t sh_degree;      /* active SH degree 0..4 (settings.sh_degree) */
    int32_t M;              /* SH coefficients stored per Gaussian per channel (shs.shape[1]); 0 = colors_precomp */
    uint32_t flags;         /* S360_FLAG_* */
    uint32_t max_instances; /* capacity of the binning buffers ((Gaussian,tile) pairs, "num_rendered") */
    uint32_t max_segments;  /* S360_FLAG_SPLIT_LISTS: segment slots to reserve in the workspace (44 x 256 B each); 0 = enough for every
                               list of the binning capacity to be long (max_instances / 256 slots).  A quadrant whose segments
                               do not fit is composited sequentially (never an error).  header_mirror reports how many a call used. */
    uint32_t _reserved;
    void* header_mirror;    /* NULL, or a HOST-visible (pinned / mapped) 8-byte aligned address: the forward also stores
                               (num_instances | overflow flag << 32 | sort chunks of the long lists << 33) there as one 64-bit word,
                               1 into the low half of the NEXT 64-bit word when some 8x8 quadrant of the call was worth
                               splitting (or was split) under S360_FLAG_SPLIT_LISTS's criterion — callers that set that flag
                               adaptively clear the word before a call and read it after — and the number of (Gaussian, view)
                               pairs binned over more than 32 tiles into the THIRD word (24 bytes in all; what a caller sets
                               S360_FLAG_COOP_WALK by) — the caller can size its next
                               call from the previous call's count without a device synchronisation (upstream reads the count
                               back synchronously inside every forward; this library's callers may run without that read,
                               and this is how they learn the count and the overflow flag anyway) */
} S360Params;

/* Byte offsets of every array inside the forward workspace (state kept for backward and exposed
 * for parity tests: upstream's geomBuffer / binningBuffer / imgBuffer). */
typedef struct S360Layout {
    size_t total_bytes;         /* forward workspace size */
    size_t header;              /* uint32[64]: [0]=num_instances [1]=overflow flag [2]=max tile list length [3]=merge passes needed
                                   [4]=pairs with more than 32 instance slots [5]=split (tile, quadrant) units of this call
                                   [6]=their segments (work items in seg_info) */
    size_t tiles_touched;       /* uint32[V*P] */
    size_t vis_mask;            /* uint8[P]  bit v set: Gaussian visible in view v (V <= 8).  tiles_touched is written for
                                   visible pairs only; the kernels test visibility on this byte, not on V words */
    size_t slot_base;           /* uint32[V*P][2]  word 0 (training calls): first instance slot of a visible pair (it owns
                                   `tiles_touched` consecutive slots; upstream's point_offsets scan is replaced by a block-wise
                                   reservation, so which range a pair gets is run-dependent — the ranges tile [0, num_instances));
                                   word 1 (S360_FLAG_LEAN_LISTS): bit i set = tile i of the pair's rectangle (scan order, at most
                                   32 tiles) holds an instance; the pair's slots follow its set bits */
    /* one 48-byte record per pair, stride 48 B: rec_b / rec_c = rec_a + 16 / + 32 (one cache line per gather) */
    size_t rec_a;               /* float4  x, y, -log2(e)/2 * conic.a, -log2(e) * conic.b */
    size_t rec_b;               /* float4  -log2(e)/2 * conic.c, opacity, r, g */
    size_t rec_c;               /* float4  b, radius (int32 bits), conservative cull half-extents wx, wy */
    size_t clamped;             /* uint8[V*P]   bit c set: colour channel c was clamped at 0 */
    size_t depths;              /* float[V*P]   view-space z of visible pairs (sort key) */
    size_t tile_count;          /* uint32[V*T] */
    size_t slot_ticket;         /* per-image instance-slot tickets, 256 B apart (cleared together with tile_count) */
    size_t merge_done;          /* uint32[V*T][4] completion counters of the global merge passes of the long lists
                                   (cleared together with tile_count) */
    size_t seg_flag;            /* uint32[V*T*4] S360_FLAG_SPLIT_LISTS: 1 = this (tile, quadrant) handed the rest of its list over to
                                   segment waves after SEG_HEAD entries (cleared together with tile_count) */
    size_t seg_arrive;          /* uint32[V*T*4] segment waves of a split quadrant that have delivered their phase-1 result (the
                                   segment's own transmittance); cleared together with tile_count */
    size_t seg_arrive2;         /* uint32[V*T*4 + 1] ... their phase-2 result (the segment composited from its true incoming
                                   transmittance): the last one to arrive runs the per-pixel combine; + the ticket counter of the
                                   work queue; cleared together with tile_count */
    size_t tile_start;          /* uint32[V*T+1] exclusive scan (upstream ranges: [start[t], start[t+1])) */
    size_t tile_cursor;         /* uint32[V*T] */
    size_t chunk_start;         /* uint32[V*T+1] number of 4096-key sort chunks of long lists before tile t */
    size_t tile_order;          /* uint32[V*T] tile ids, longest list first (dispatch order of the composite) */
    size_t keys;                /* uint64[max_instances]  (depth bits << 32 | pair index), sorted per tile */
    size_t keys_alt;            /* uint64[max_instances]  ping-pong buffer of the long-list merge passes */
    size_t list;                /* uint32[max_instances]  sorted pair indices p = v*P + g (upstream point_list) */
    size_t final_T;             /* float[V*H*W] */
    size_t n_contrib;           /* uint32[V*H*W] */
    size_t tile_max_contrib;    /* uint32[V*T] */
    size_t strip_last;          /* uint32[V*T*4] max n_contrib of each of the four 8x8 quadrants of a tile */
    size_t slot_pair;           /* uint32[max_instances] pair index p of every instance slot (training calls): lets the backward
                                   sum the per-(instance, quadrant) partial gradients slot-parallel.  Bit 31 is set on the slots
                                   of a pair that owns more than 32 of them (a long_pairs entry): the slot-parallel pass skips
                                   those — the wave-parallel pass reads them (V * P < 2^31 is required) */
    size_t long_pairs;          /* uint32[max_instances/32 + 1]  training calls: the pairs that own more than 32 instance slots
                                   (header[4] of them, in no particular order): the backward sums their slots wave-parallel */
    size_t rgbc;                /* float4[P]  SH colour of every Gaussian (r, g, b, clamp bits) when the views share a camera
                                   centre: evaluated once per call by a streaming kernel ahead of the geometry pass */
    size_t sh_jac;              /* float[P,3,3] d(rgb_c)/d(mean) through the view direction (training calls only): lets the
                                   backward add that term to dL/dmean without re-reading the 300-byte SH slab */
    size_t surv;                /* training calls: 48-byte records of the list entries that survive the exact cull of each
                                   8x8 quadrant, in list order, written by the forward composite (centre, conic, opacity,
                                   colour, radius, list position, pair): unit (tile t, quadrant q) owns records
                                   [4 start_t + q n_t, ... + n_t), n_t = the tile's list length.  The backward composite
                                   streams them back to front instead of walking and culling the tile list a second time */
    size_t surv_count;          /* uint32[V*T*4] survivor records of a unit that lie in front of its last contributor */
    /* S360_FLAG_SPLIT_LISTS state.  Segment SLOT s = 8 * chunk_start[t] + k is segment k (list positions [512 k, 512 (k+1)))
     * of long tile t; NSEG = S360Params.max_segments slots.  Slots k = 0 of a split quadrant hold what its head wave
     * published (the exact sequential state after the head); k >= 2 the segment waves' results. */
    size_t part_c;              /* float4[NSEG*4*64]  per (slot, quadrant, pixel): the segment's colour contribution (r, g, b, depth) */
    size_t part_t;              /* float [NSEG*4*64]  ... transmittance of the segment alone (phase 1; 0: the pixel stops inside it) */
    size_t part_e;              /* float [NSEG*4*64]  ... transmittance behind the segment (phase 2) */
    size_t part_l;              /* uint32[NSEG*4*64]  ... last contributing list position + 1 (0: none); slot 0: | done << 31 */
    size_t part_n;              /* uint32[NSEG*4]     survivor records the segment appended (slot k = 1: the head's count when no
                                   later segment contributes) */
    size_t seg_c;               /* float4[NSEG*4*64]  after the combine: colour accumulated BEHIND the segment (what the backward
                                   starts its colour-behind sum from) */
    size_t seg_t;               /* float [NSEG*4*64]  after the combine: transmittance behind the segment's last entry */
    size_t seg_cnt;             /* uint32[NSEG*4]     survivor records of the segment the backward replays (0: none / not split) */
    size_t seg_info;            /* uint32[NSEG*4][2]  the segment work items of the call (header[6] of them): (tile, segment k << 2 | quadrant) */
    size_t geo7;                /* float[P,7]  S360_FLAG_RAW_INPUTS training calls: the raw scale logits + quaternion of every Gaussian */
    size_t backward_bytes;      /* size of the separate backward scratch workspace */
} S360Layout;

/*
 * The encoder's raw per-pixel outputs, as GaussianAdapterERP.forward receives them
 * (/root/reference/src/model/encoder/common/gaussian_adapter_erp.py:50-61): what s360_forward_raw / s360_backward_raw consume instead
 * of the adapter's materialised means / covariances / harmonics.  All pointers DEVICE memory.
 */
typedef struct S360RawInputs {
    const float* extrinsics;    /* [n_views,4,4] context-panorama camera-to-world (row-major) */
    const float* depths;        /* [n_views*per_view] */
    const float* raw_gaussians; /* [n_views*per_view, 82]: 3 scale logits, quaternion (x,y,z,w), 3 x 25 SH coefficients channel-major */
    const float* sh_rotation;   /* [n_views,25,25] block-diagonal Wigner-D matrices of rotate_sh (s360_sh_rotation_blocks) or NULL = identity */
    int32_t n_views, per_view;  /* P = n_views * per_view Gaussians, view-major */
    int32_t H, W, per_ray;      /* context panorama size; per_view = H*W*per_ray, ray-major */
    int32_t erp_convention;     /* as s360_adapter_forward */
    float scale_min, scale_max, eps;
} S360RawInputs;

/* ABI version of the loaded library (== S360_ABI_VERSION). */
int s360_abi_version(void);
const char* s360_error_string(int code);

/* Workspace layout / sizes for the given problem (pure host arithmetic). */
int s360_layout(const S360Params* prm, S360Layout* out);

/*
 * Forward: replaces upstream `rasterize_gaussians(...)` as reached from
 * GaussianRasterizer.forward (call site cuda_splatting.py:117-124).
 *   views[V] (device)  means3D[P,3]  cov6[P,6] (cov3D_precomp, order 00,01,02,11,12,22; [P,3,3] with
 *   S360_FLAG_COV9)  opacities[P]  shs[P,M,3] ([P,3,M] with S360_FLAG_SH_CHANNEL_MAJOR) or NULL
 *   colors_precomp[P,3] or NULL (exactly one of shs / colors_precomp non-NULL)
 * Outputs: images[V,3,H,W], radii[V,P] (int32), plus state in `workspace`.
 */
int s360_forward(const S360Params* prm, const S360View* views, const float* means3D,
                 const float* cov6, const float* opacities, const float* shs,
                 const float* colors_precomp, float* images, int32_t* radii, void* workspace,
                 size_t workspace_bytes, void* stream);

/*
 * Forward with a fused depth map: colour AND the alpha-premultiplied expected depth of
 * render_depth_cuda (cuda_splatting.py:226-269: camera-space z of every Gaussian composited with the
 * colour weights, background 0, not normalised) in ONE pass instead of a second full rasterisation
 * (decoder_splatting_cuda.py:72-97 renders every face twice when depth is wanted).
 *   depth_mode: S360_DEPTH_* (the reference's DepthRenderingMode); depth_maps[V,H,W].
 * The depth map is differentiable: pass dL_ddepth to s360_backward — the reference's
 * training step can request depth (model_wrapper_erp.py:228, train_cfg.depth_mode) and LossDepth
 * (src/loss/loss_depth.py:37-60) back-propagates through it into means, covariances and opacities.
 */
#define S360_DEPTH_DEPTH 0
#define S360_DEPTH_DISPARITY 1
#define S360_DEPTH_RELATIVE_DISPARITY 2
#define S360_DEPTH_LOG 3
int s360_forward_depth(const S360Params* prm, const S360View* views, const float* means3D,
                       const float* cov6, const float* opacities, const float* shs,
                       const float* colors_precomp, float* images, float* depth_maps, int32_t depth_mode,
                       int32_t* radii, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Forward with the cube-face loss epilogue fused into the composite store: replaces the
 * torch ops the reference runs on the rendered faces right after the decoder —
 *   LossMse.forward       src/loss/loss_mse.py:30-31   weight * mean((color - target)^2)
 *   compute_psnr          src/evaluation/metrics.py:11-21   -10 log10(mean((clip01(gt) - clip01(pred))^2))
 * and the loss's autograd seed.  target[V,3,H,W]; d_images[V,3,H,W] = grad_scale * (image - target) (pass
 * grad_scale = 2*weight/N, N = elements averaged over); partials[V*tiles*4, 2] = per 8x8 quadrant of every tile (in
 * tile order) the sums of squared differences, plain and clipped — summed in a fixed order
 * by the caller, or (loss_out != NULL) by one more small launch of the same call: loss_out[0] = (grad_scale / 2) * sum
 * of all plain partials (= weight * mean over the N elements), loss_out[1 + v] = clipped MSE of view v.  Deterministic
 * either way.  depth_maps may be null (no depth channel).
 */
int s360_forward_mse(const S360Params* prm, const S360View* views, const float* means3D,
                     const float* cov6, const float* opacities, const float* shs,
                     const float* colors_precomp, float* images, float* depth_maps, int32_t depth_mode,
                     int32_t* radii, const float* target, float grad_scale, float* d_images, float* partials,
                     float* loss_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward: replaces upstream `rasterize_gaussians_backward(...)` (autograd backward of the
 * call above).  `workspace` is the forward workspace, unmodified since s360_forward.
 *   dL_dimages[V,3,H,W]; dL_dimages_scale: NULL, or a DEVICE pointer to one float every element of dL_dimages is
 *   multiplied by as it is read (the scalar autograd hands back for a loss whose seed is already stored — the
 *   d_images of s360_forward_mse — or an AMP loss scale: no separate elementwise pass, no host read)
 *   dL_ddepth[V,H,W] / depth_mode: gradient of the fused depth map of s360_forward_depth (same depth_mode as the
 *   forward); NULL: no depth channel (e.g. the forward was s360_forward).
 * Outputs (all written, no accumulation into caller data):
 *   d_means3D[P,3]  d_means2D[V,P,3] (NDC-scaled screen-space gradient, z = 0)
 *   d_cov6[P,6]  d_opacities[P]  d_shs[P,M,3] or NULL  d_colors[P,3] or NULL  (d_cov6 / d_shs in the
 *   layouts selected by the flags; gradients are w.r.t. the UNSCALED inputs).  d_shs == NULL (harmonics
 *   frozen) still propagates dRGB/d(view direction) into d_means3D, as upstream does.
 * Gradients are summed over the V views with a fixed (deterministic) order; no float atomics anywhere.
 */
int s360_backward(const S360Params* prm, const S360View* views, const float* means3D,
                  const float* cov6, const float* opacities, const float* shs,
                  const float* colors_precomp, const void* workspace, size_t workspace_bytes,
                  const float* dL_dimages, const float* dL_dimages_scale,
                  const float* dL_ddepth, int32_t depth_mode, float* d_means3D, float* d_means2D,
                  float* d_cov6, float* d_opacities, float* d_shs, float* d_colors, void* bwd_workspace,
                  size_t bwd_workspace_bytes, void* stream);

/*
 * Split backward for multi-GPU view sharding (no reference counterpart: the reference all-reduces encoder
 * parameters under DDP, src/main.py:117-130).  dL/dSH of one rendered panorama is, per Gaussian, the rank-1
 * product Y(dir) (x) dL/dRGB; summing it over N ranks by all-reducing N full 300-byte slabs moves 25x more
 * bytes than exchanging the factors.  s360_backward_split() is s360_backward() for views sharing one camera
 * centre (S360_FLAG_SHARED_CAMPOS) WITHOUT the dL/dSH pass: it returns d_means3D (complete, including this rank's
 * view-direction term — the forward keeps d(rgb)/d(mean) per Gaussian, S360Layout.sh_jac), d_cov6, d_opacities and
 * d_rgb_sum[P,4] = (clamp-masked sum of dL/dRGB, index of the first view that saw the Gaussian or -1 as int32
 * bits).  After all-gathering d_rgb_sum (with .w rewritten to the index of the owning rank's representative view,
 * or -1) and one S360View per rank, s360_sh_backward() writes the summed dL/dSH = sum_ranks Y(dir_rank) (x) dRGB_rank
 * (it reads neither the SH coefficients nor any gradient buffer: 16 B in per Gaussian and rank, 300 B out).
 */
int s360_backward_split(const S360Params* prm, const S360View* views, const float* means3D,
                        const float* cov6, const float* opacities, const float* shs,
                        const void* workspace, size_t workspace_bytes,
                        const float* dL_dimages, const float* dL_dimages_scale, const float* dL_ddepth,
                        int32_t depth_mode, float* d_means3D, float* d_means2D, float* d_cov6,
                        float* d_opacities, float* d_rgb_sum, void* bwd_workspace,
                        size_t bwd_workspace_bytes, void* stream);
int s360_sh_backward(const S360Params* prm, int32_t n_groups, const S360View* views,
                     const float* means3D, const float* d_rgb_sums /* [n_groups,P,4] */, float* d_shs,
                     void* stream);

/*
 * The backward in two parts, for the chunked multi-GPU gradient exchange (no reference counterpart: the reference leaves the
 * gradient exchange to Lightning DDP, src/main.py:117-130; here the per-Gaussian gradients of one panorama per rank are summed
 * over the ranks INSIDE the rasteriser's backward, and the exchange of one Gaussian range runs while the next range is still
 * being computed):
 *   s360_backward_composite   every (tile, quadrant) replay + the per-pair sum -> state in bwd_workspace (no outputs);
 *   s360_backward_gaussians   the per-Gaussian chains of Gaussians [g_begin, g_begin + g_count) from that state (views sharing
 *                             one camera centre, SH given): d_packed[P,10] rows (d_mean 3 | the 6 unique d_covariance entries |
 *                             d_opacity — the unit one all-reduce moves; complete d_mean incl. this rank's view-direction term),
 *                             d_rgb_sum[P,4] rows (clamp-masked sum of dL/dRGB; .w = rank_stamp as int32 bits where the Gaussian
 *                             was visible to this call, else -1 — what s360_sh_backward expects after the all-gather),
 *                             optional d_means2D[V,P,3].  s360_backward_split == composite + gaussians over the whole cloud.
 *   s360_unpack_gradients     packed[P,10] (after the all-reduce) -> d_means3D[P,3], d_cov ([P,3,3] upper triangle with cov9 != 0,
 *                             else [P,6]), d_opacities[P].
 */
int s360_backward_composite(const S360Params* prm, const S360View* views, const void* workspace, size_t workspace_bytes,
                            const float* dL_dimages, const float* dL_dimages_scale, const float* dL_ddepth, int32_t depth_mode,
                            void* bwd_workspace, size_t bwd_workspace_bytes, void* stream);
int s360_backward_gaussians(const S360Params* prm, const S360View* views, const float* means3D, const float* cov6,
                            const float* shs, const void* workspace, size_t workspace_bytes, int32_t with_depth,
                            int32_t depth_mode, int32_t g_begin, int32_t g_count, int32_t rank_stamp, float* d_packed,
                            float* d_means2D, float* d_rgb_sum, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream);
int s360_unpack_gradients(const float* packed, int32_t P, int32_t cov9, float* d_means3D, float* d_cov, float* d_opacities,
                          void* stream);
/* The all-gather form of the exchange (up to two ranks: one collective per Gaussian range instead of two): packed_blocks holds n_blocks
 * gathered [g_count,10] row blocks, one per rank, back to back; their sum (rank order: the same bits on every rank) is written to rows
 * [g_begin, g_begin + g_count) of d_means3D / d_cov / d_opacities — the local reduction and s360_unpack_gradients in one pass. */
int s360_reduce_unpack_gradients(const float* packed_blocks, int32_t n_blocks, int32_t g_begin, int32_t g_count, int32_t cov9,
                                 float* d_means3D, float* d_cov, float* d_opacities, void* stream);

/*
 * Camera records of a call in one launch: replaces the reference's per-call camera glue
 * (cuda_splatting.py:64-71 scale-invariant rescale, :80-84 get_fov / get_projection_matrix, :85-87 the two
 * transposed matrices; src/geometry/projection.py:233-247) — ~60 tiny torch / rocSOLVER launches for six cameras.
 *   extrinsics[N,4,4] camera-to-world (OpenCV), intrinsics[N,3,3] normalised, near / far[N],
 *   background[3] (background_per_view = 0) or [N,3] (1), scale_invariant as in render_cuda.
 *   views_out[N]: S360View records (near_plane / far_plane = the UNSCALED planes).
 * Same formulas in the same order as the reference; the two matrix inverses are Gauss-Jordan eliminations
 * with partial pivoting, so the records agree with the torch glue to a few ulp, not bit for bit.
 */
int s360_pack_views(const float* extrinsics, const float* intrinsics, const float* near_planes,
                    const float* far_planes, const float* background, int32_t background_per_view,
                    int32_t n_views, int32_t scale_invariant, S360View* views_out, void* stream);

/*
 * Gaussian-adapter tail: the producer of the per-Gaussian buffers this library rasterises — replaces
 * GaussianAdapterERP.forward (/root/reference/src/model/encoder/common/gaussian_adapter_erp.py:50-119: scale map :63-78,
 * quaternion normalisation :82, sh_mask :38-47,86, world covariance :89-92 with build_covariance of
 * common/gaussians.py:33-44, sphere un-projection src/geometry/sphere_projection.py:6-86 in the ERP convention of the dataset,
 * src/geometry/utils360.py:37-153: erp_convention 0 = 'hm3d' / 'replica' (the reference's configs), 1 = 'm3d',
 * 2 = 'residential', 3 = 'CoffeeArea' / 'outdoor_colmap') in one launch.
 *   extrinsics[V,4,4] context-panorama camera-to-world; depths[V,Gv]; raw_gaussians[V,Gv,7+3*d_sh] = 3 scale logits,
 *   quaternion (x,y,z,w), 3*d_sh SH coefficients channel-major; Gv = H*W*per_ray Gaussians per view, ray-major;
 *   sh_rotation[V,d_sh,d_sh] or NULL (= identity): per-view SH rotation, only the (2l+1)x(2l+1) diagonal blocks are
 *   read — the Wigner-D matrices rotate_sh (src/misc/sh_rotation.py:10-30) obtains from e3nn, built by the caller.
 * Outputs: means[V*Gv,3], covariances[V*Gv,3,3] (cov9 != 0) or [V*Gv,6] upper triangle (the rasteriser's
 * cov3D_precomp layout), harmonics[V*Gv,3,d_sh]; optional scales_out[V*Gv,3] / rotations_out[V*Gv,4] (the adapter's
 * export-only fields).  Opacities pass through the adapter unchanged and are not touched here.
 * s360_adapter_backward: gradients of (means, covariances, harmonics) -> d_depths[V,Gv], d_raw_gaussians (same layout).
 *   d_means == NULL (the reference's behaviour): the means are detached — the reference un-projects under torch.no_grad()
 *   (src/geometry/sphere_projection.py:14-86), so depth receives gradient through the scales only; a non-NULL d_means adds
 *   the un-projection's own term (this project's opt-in deviation).
 */
/*
 * The matrices of rotate_sh (src/misc/sh_rotation.py:10-30: per degree l, e3nn.o3.wigner_D(l, *matrix_to_angles(R)), applied to
 * the SH coefficient blocks at gaussian_adapter_erp.py:113 with R = the context view's camera-to-world rotation).
 *   rotations: n_views row-major 3x3 matrices, row_major_stride = 9 ([n,3,3]) or 16 (the rotation part of [n,4,4] poses);
 *   sh_rotation_out[n_views, d_sh, d_sh]: block-diagonal, block l = D^l(R) with Y^l(R d) = D^l(R) Y^l(d) in e3nn's real basis
 *   (polar axis y, azimuth from z towards x, m = -l..l, no Condon-Shortley phase; D^1 = R) — the `sh_rotation` argument of
 *   s360_adapter_forward / backward.  e3nn itself is not available where this was written: the convention is restated from its
 *   documentation and pinned by properties only (oracle/adapter_ref.py).
 */
/*
 * The adapter tail FUSED into the rasteriser: replaces GaussianAdapterERP.forward
 * (gaussian_adapter_erp.py:63-119) + the decoder's rasteriser call on its output (cuda_splatting.py:99-124) for views sharing one
 * camera centre, at the reference's configuration (degree-4 harmonics).  The call's first kernel reads the raw records once
 * (328 B/Gaussian), writes means_out[P,3] and cov6_out[P,6] (the adapter's values: what the geometry pass then reads, and what a
 * caller may keep) and evaluates the view-dependent colour with the SH basis carried through sh_mask and the per-view rotation —
 * the [P,3,25] harmonics and [P,3,3] covariances are never materialised (340 B/Gaussian written + 340 read by the two-step path).
 * prm: P = raw->n_views * raw->per_view, M = 25, sh_degree = 4, flags must hold S360_FLAG_RAW_INPUTS | S360_FLAG_SHARED_CAMPOS and not
 * S360_FLAG_COV9.  target == NULL: no loss epilogue (d_images / partials / loss_out ignored); depth_maps == NULL: no depth channel.
 * s360_backward_raw: the backward of that call down to the encoder's outputs — d_depths[P], d_raw_gaussians[P,82] (dL/d(SH
 * coefficient) formed as (mask . D^T Y) (x) dL/dRGB: the [P,3,25] dL/dSH round trip of the two-step path does not exist),
 * d_opacities[P]; d_means3D[P,3], d_cov6[P,6], d_rgb_sum[P,4] are caller-provided intermediates (valid results themselves).
 * differentiable_means = 0: the reference's detached means (sphere_projection.py:14-86 runs under torch.no_grad()); 1 adds the
 * un-projection's own depth term.  means / cov6: what s360_forward_raw wrote.
 */
int s360_forward_raw(const S360Params* prm, const S360View* views, const S360RawInputs* raw, const float* opacities,
                     float* means_out, float* cov6_out, float* images, float* depth_maps, int32_t depth_mode, int32_t* radii,
                     const float* target, float grad_scale, float* d_images, float* partials, float* loss_out, void* workspace,
                     size_t workspace_bytes, void* stream);
int s360_backward_raw(const S360Params* prm, const S360View* views, const S360RawInputs* raw, const float* means, const float* cov6,
                      const float* opacities, const void* workspace, size_t workspace_bytes, const float* dL_dimages,
                      const float* dL_dimages_scale, const float* dL_ddepth, int32_t depth_mode, int32_t differentiable_means,
                      float* d_means3D, float* d_cov6, float* d_opacities, float* d_rgb_sum, float* d_depths, float* d_raw_gaussians,
                      void* bwd_workspace, size_t bwd_workspace_bytes, void* stream);
/*
 * The last kernel of s360_backward_raw on its own, for the multi-GPU exchange on the raw path (no reference counterpart: the reference
 * leaves the gradient exchange to Lightning DDP, /root/reference/src/main.py:117-130): s360_backward_composite +
 * s360_backward_gaussians produce the rows the exchange moves (packed[P,10], d_rgb_sum[P,4]); after it, d_cov6 [, d_means3D] hold the
 * sums over the ranks and d_rgb_sums[n_groups,P,4] one clamp-masked dL/dRGB sum per rank (.w = index of that rank's record in
 * group_views[n_groups], int32 bits, or -1) — exactly k_raw_bwd's inputs: dL/d(SH coefficient) = sum over groups of
 * (mask . D^T Y(dir_group)) (x) dL/dRGB_group.  d_means3D == NULL: the reference's detached means.
 */
int s360_backward_raw_tail(const S360Params* prm, const S360View* group_views, int32_t n_groups, const S360RawInputs* raw,
                           const float* means, const void* workspace, size_t workspace_bytes, const float* d_means3D,
                           const float* d_cov6, const float* d_rgb_sums, float* d_depths, float* d_raw_gaussians, void* stream);
int s360_sh_rotation_blocks(const float* rotations, int32_t row_major_stride, int32_t n_views, int32_t d_sh,
                            float* sh_rotation_out, void* stream);
int s360_adapter_forward(const float* extrinsics, const float* depths, const float* raw_gaussians,
                         const float* sh_rotation, int32_t n_views, int32_t per_view, int32_t H, int32_t W,
                         int32_t per_ray, int32_t d_sh, float scale_min, float scale_max, float eps, float* means,
                         float* covariances, int32_t cov9, float* harmonics, float* scales_out,
                         float* rotations_out, int32_t erp_convention, void* stream);
int s360_adapter_backward(const float* extrinsics, const float* depths, const float* raw_gaussians,
                          const float* sh_rotation, int32_t n_views, int32_t per_view, int32_t H, int32_t W,
                          int32_t per_ray, int32_t d_sh, float scale_min, float scale_max, float eps,
                          const float* d_means, const float* d_covariances, int32_t cov9, const float* d_harmonics,
                          float* d_depths, float* d_raw_gaussians, int32_t erp_convention, void* stream);

/*
 * Cube -> equirectangular stitch: replaces Cube2Equirec.forward
 * (/root/reference/src/geometry/layers.py:108-116, F.grid_sample trilinear / border /
 * align_corners=True over the [C,6,fw,fw] face stack).
 *   faces[6,C,fw,fw] in Cube2Equirec's slot order (F R B L U D) when face_map == NULL, else
 *   slot s reads faces[face_map[s] & 7] and, when bit 3 of face_map[s] is set, flipped on both
 *   image axes — face_map = {3, 4, 1, 2, 0|8, 5|8} applies the reference's change_order()
 *   (src/model/model_wrapper_erp.py:135-145) to faces given in rendered order (U B L F R D)
 *   without materialising the permuted copy.  face_map_host is read on the HOST.
 *   grid[eh,ew,3] = Cube2Equirec.sample_grid (u, v, face-z);  erp[C,eh,ew].
 *   strides_host (HOST, 3 x int64, element units, may be NULL = dense [6,C,fw,fw]): strides between
 *   faces, channels and rows of `faces`; {fw, fw*6*fw, 6*fw} reads the reference's own
 *   [C,fw,6*fw] "faces side by side" input of Cube2Equirec.forward (layers.py:108-113) in place.
 */
int s360_cube2erp_forward(const float* faces, const float* grid, float* erp, int32_t channels,
                          int32_t face_w, int32_t equ_h, int32_t equ_w,
                          const int32_t* face_map_host, const int64_t* strides_host, void* stream);

/* Adjoint of the stitch: d_faces[6,C,fw,fw] (same face order / face_map as forward) is
 * ZEROED then accumulated with float atomics (non-deterministic summation order). */
int s360_cube2erp_backward(const float* d_erp, const float* grid, float* d_faces,
                           int32_t channels, int32_t face_w, int32_t equ_h, int32_t equ_w,
                           const int32_t* face_map_host, const int64_t* strides_host, void* stream);

/*
 * Optional measurement aid (no reference counterpart; the reference's Benchmarker is an
 * un-synchronised wall clock, src/misc/benchmarker.py:15-33).  While enabled, every kernel group
 * is bracketed by HIP events recorded on the launch stream; s360_profile_collect() synchronises
 * on those events and returns, per slot, the summed milliseconds and the number of launches since
 * the previous collect.  Arrays must hold s360_profile_slots() entries.
 */
/* Second measurement aid: counts[0] = (pixel, list entry) pairs that contribute to the images rendered into a TRAINING workspace
 * (alpha >= 1/255, in front of the pixel's last contributor), counts[1] = pairs the backward composite evaluates for them
 * (survivor records in front of each 8x8 quadrant's last contributor x 64 pixels).  counts: DEVICE uint64[2], written
 * asynchronously on `stream`.  bench.py derives the work-based VALU fraction of the composites from it. */
int s360_count_contributions(const S360Params* prm, const void* workspace, size_t workspace_bytes, uint64_t* counts, void* stream);
/* Second measurement aid: where the backward composite's evaluated (entry, pixel) slots go — counts[32]: [0] slots executed, [1] padding
 * lanes, [2] entry at / behind the pixel's last contributor, [3] splat does not reach the pixel (power > 0 or alpha < 1/255),
 * [4] contributing, [5] slots of skipped four-pixel runs, [6] units, [7] groups, [8..17] executed slots by the unit's contributing
 * fraction (deciles), [18..25] executed runs by how many of the group's records contribute to the run (0, 1-4, 5-8, 9-16, 17-24, 25-32,
 * 33-48, 49-64).  Training workspaces of unsplit calls only. */
int s360_count_backward_slots(const S360Params* prm, const void* workspace, size_t workspace_bytes, uint64_t* counts, void* stream);
int s360_profile_slots(void);
const char* s360_profile_slot_name(int slot);
int s360_profile_enable(int on);
int s360_profile_collect(float* total_ms, int32_t* calls);

#ifdef __cplusplus
}
#endif
#endif /* S360_H */
