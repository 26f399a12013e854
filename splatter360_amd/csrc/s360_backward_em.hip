// s360_backward_em.hip — translation unit of the entry-major backward composite (kernel in s360_bwd_em.h).
// Compiled with -fno-slp-vectorize (see the launcher comment in s360_bwd_em.h).  gfx950 / wave64 only.
#define S360_EM_KERNEL_TU 1
#include "s360_bwd_em.h"

namespace s360 {

void launch_render_bwd_em(bool with_depth, int n_units, hipStream_t st, const KParams& kp, const S360View* views,
                          const uint32_t* tile_start, const float4* surv, const uint32_t* surv_count, const uint2* slot_info,
                          const float* depths, const float* final_T, const uint32_t* n_contrib, const float* dL_dimages,
                          const float* dL_dimages_scale, const float* dL_ddepth, float4* part, uint8_t* valid, const uint32_t* order, int depth_mode,
                          float* pairgrad_atomic, uint32_t* dbg, const SegBwd* sbp_host, uint32_t n_seg_blocks) {
    const SegBwd sbv = sbp_host ? *sbp_host : SegBwd{};   // HOST struct, passed by value
#define S360_LAUNCH_BWD(WD, SG)                                                                                                              \
    hipLaunchKernelGGL((k_render_bwd_em<WD, SG>), dim3(n_units + (SG ? n_seg_blocks : 0u)), dim3(64), 0, st, kp, views, tile_start, surv, surv_count, \
                       slot_info, depths, final_T, n_contrib, dL_dimages, dL_dimages_scale, dL_ddepth, part, valid, order, depth_mode,          \
                       pairgrad_atomic, dbg, sbv, SG ? n_seg_blocks : 0u)
    const bool seg = sbp_host != nullptr;
    if (with_depth) { if (seg) S360_LAUNCH_BWD(true, true); else S360_LAUNCH_BWD(true, false); }
    else { if (seg) S360_LAUNCH_BWD(false, true); else S360_LAUNCH_BWD(false, false); }
#undef S360_LAUNCH_BWD
}

}  // namespace s360
