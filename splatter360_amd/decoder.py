"""Host-side mirror of the reference's decoder glue (src/model/decoder/cuda_splatting.py and
decoder_splatting_cuda.py) on top of the HIP rasteriser, plus the fused six-face path.

Drop-in layer (same names, argument meaning and error behaviour as the reference):
    render_cuda, render_depth_cuda, render_cuda_orthographic, get_projection_matrix, DecoderSplattingCUDA
Fused layer (what the MI355X design adds — one rasteriser call per panorama instead of six
Python-looped calls, shared per-Gaussian loads and SH evaluation, no host synchronisation):
    cube_cameras, pack_camera_views, render_views_fused, render_cube_faces
"""
from __future__ import annotations

from dataclasses import dataclass
from math import isqrt
from typing import Literal, Optional

import torch
from torch import Tensor

from . import cameras, rasterizer
from .cameras import get_fov, get_projection_matrix  # noqa: F401  (re-exported, reference names)

DepthRenderingMode = Literal["depth", "disparity", "relative_disparity", "log"]


def depth_to_relative_disparity(depth: Tensor, near: Tensor, far: Tensor, eps: float = 1e-10) -> Tensor:
    """src/model/encoder/costvolume/conversions.py (relative disparity in [0,1])."""
    disp_near = 1 / (near + eps)
    disp_far = 1 / (far + eps)
    disp = 1 / (depth + eps)
    return 1 - (disp - disp_far) / (disp_near - disp_far + eps)


def _triu_cov6(cov: Tensor) -> Tensor:
    row, col = torch.triu_indices(3, 3)
    return cov[..., row, col]


def render_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple,
                background_color: Tensor, gaussian_means: Tensor, gaussian_covariances: Tensor,
                gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor, scale_invariant: bool = True,
                use_sh: bool = True) -> Tensor:
    """Same contract as the reference's render_cuda (cuda_splatting.py:47-127): batch of b
    (camera, cloud) pairs -> [b,3,h,w].  One HIP rasteriser call per batch item, no .item() sync."""
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    vs, calls = rasterizer_boundary(extrinsics, intrinsics, near, far, gaussian_means, gaussian_covariances,
                                    gaussian_sh_coefficients, gaussian_opacities, scale_invariant, use_sh)
    h, w = image_shape
    images = []
    for i, kw in enumerate(calls):
        views = rasterizer.pack_views(vs["view_matrix"][i], vs["full_projection"][i], vs["campos"][i],
                                      vs["tan_fov_x"][i:i + 1], vs["tan_fov_y"][i:i + 1], background_color[i])
        img, _ = rasterizer.rasterize_views(
            kw["means3D"], kw["cov3D_precomp"], kw["opacities"], kw["shs"], kw["colors_precomp"], views=views,
            image_height=h, image_width=w, sh_degree=kw["sh_degree"], shared_campos=True, want_radii=False)
        images.append(img[0])
    return torch.stack(images)


def rasterizer_boundary(extrinsics, intrinsics, near, far, gaussian_means, gaussian_covariances,
                        gaussian_sh_coefficients, gaussian_opacities, scale_invariant=True, use_sh=True):
    """The tensors render_cuda hands to the rasteriser, per batch item (cuda_splatting.py:64-75,
    115-123): camera setup dict + [{means3D, cov3D_precomp, opacities, shs | colors_precomp,
    sh_degree}].  Pure torch (runs on CPU too): pinned against tests/golden/boundary_render_cuda.npz."""
    vs = cameras.view_setup(extrinsics, intrinsics, near, far, scale_invariant)
    scale = vs["scale"]
    if scale_invariant:
        gaussian_covariances = gaussian_covariances * (scale[:, None, None, None] ** 2)
        gaussian_means = gaussian_means * scale[:, None, None]
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    shs = gaussian_sh_coefficients.transpose(2, 3).contiguous()  # "b g xyz n -> b g n xyz"
    calls = []
    for i in range(extrinsics.shape[0]):
        calls.append(dict(means3D=gaussian_means[i], cov3D_precomp=_triu_cov6(gaussian_covariances[i]),
                          opacities=gaussian_opacities[i, ..., None], shs=shs[i] if use_sh else None,
                          colors_precomp=None if use_sh else shs[i, :, 0, :], sh_degree=degree))
    return vs, calls


def orthographic_setup(extrinsics: Tensor, width: Tensor, height: Tensor, near: Tensor, far: Tensor,
                       fov_degrees: float = 0.1):
    """Camera half of render_cuda_orthographic (cuda_splatting.py:155-179): a fake orthographic view
    made of a tiny field of view and a camera pulled back so that `width` spans the image."""
    dev = extrinsics.device
    b = extrinsics.shape[0]
    fov_x = torch.tensor(fov_degrees, device=dev).deg2rad()
    tan_fov_x = (0.5 * fov_x).tan()
    distance_to_near = (0.5 * width) / tan_fov_x
    tan_fov_y = 0.5 * height / distance_to_near
    fov_y = (2 * tan_fov_y).atan()
    near = near + distance_to_near
    far = far + distance_to_near
    move_back = torch.eye(4, dtype=torch.float32, device=dev)
    move_back[2, 3] = -distance_to_near
    extrinsics = extrinsics @ move_back
    proj = cameras.get_projection_matrix(near, far, fov_x.expand(b), fov_y).transpose(1, 2)
    view = cameras._inverse_nosync(extrinsics).transpose(1, 2)
    return dict(extrinsics=extrinsics, fov_x=fov_x, fov_y=fov_y, near=near, far=far, tan_fov_x=tan_fov_x,
                tan_fov_y=tan_fov_y, view_matrix=view.contiguous(), full_projection=(view @ proj).contiguous())


def render_cuda_orthographic(extrinsics: Tensor, width: Tensor, height: Tensor, near: Tensor, far: Tensor,
                             image_shape: tuple, background_color: Tensor, gaussian_means: Tensor,
                             gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor,
                             fov_degrees: float = 0.1, use_sh: bool = True, dump: Optional[dict] = None) -> Tensor:
    """Same contract as the reference's render_cuda_orthographic (cuda_splatting.py:130-220; used by
    src/visualization/validation_in_3d.py:68-81 and the figure scripts).  As in the reference the
    pull-back distance is a single value, i.e. batch size 1 (move_back[2, 3] = -distance_to_near)."""
    h, w = image_shape
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    shs = gaussian_sh_coefficients.transpose(2, 3).contiguous()
    o = orthographic_setup(extrinsics, width, height, near, far, fov_degrees)
    if dump is not None:
        for k in ("extrinsics", "fov_x", "fov_y", "near", "far"):
            dump[k] = o[k]
    images = []
    for i in range(extrinsics.shape[0]):
        views = rasterizer.pack_views(o["view_matrix"][i], o["full_projection"][i], o["extrinsics"][i, :3, 3],
                                      o["tan_fov_x"].reshape(1), o["tan_fov_y"].reshape(-1)[i:i + 1], background_color[i])
        img, _ = rasterizer.rasterize_views(
            gaussian_means[i], _triu_cov6(gaussian_covariances[i]), gaussian_opacities[i, ..., None],
            shs[i] if use_sh else None, None if use_sh else shs[i, :, 0, :], views=views, image_height=h,
            image_width=w, sh_degree=degree, shared_campos=True, want_radii=False)
        images.append(img[0])
    return torch.stack(images)


def _depth_colors(extrinsics, means, near, far, mode):
    cam = torch.einsum("bij,bgj->bgi", cameras._inverse_nosync(extrinsics),   # same LU as .inverse(), without its info read-back
                       torch.cat([means, torch.ones_like(means[..., :1])], dim=-1))
    z = cam[..., 2]
    if mode == "disparity":
        z = 1 / z
    elif mode == "relative_disparity":
        z = depth_to_relative_disparity(z, near[:, None], far[:, None])
    elif mode == "log":
        z = z.minimum(near[:, None]).maximum(far[:, None]).log()  # reference quirk kept (:251)
    return z


def render_depth_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple,
                      gaussian_means: Tensor, gaussian_covariances: Tensor, gaussian_opacities: Tensor,
                      scale_invariant: bool = True, mode: DepthRenderingMode = "depth") -> Tensor:
    """Reference's render_depth_cuda (cuda_splatting.py:226-269): alpha-premultiplied expected
    depth -> [b,h,w]."""
    z = _depth_colors(extrinsics, gaussian_means, near, far, mode)
    b = z.shape[0]
    result = render_cuda(extrinsics, intrinsics, near, far, image_shape,
                         torch.zeros((b, 3), dtype=z.dtype, device=z.device), gaussian_means,
                         gaussian_covariances, z[:, :, None, None].expand(-1, -1, 3, 1), gaussian_opacities,
                         scale_invariant=scale_invariant, use_sh=False)
    return result.mean(dim=1)


# ----------------------------------------------------------------------------- fused path
def cube_cameras(pano_c2w: Tensor, near, far):
    """[4,4] panorama pose -> (extrinsics[6,4,4], intrinsics[6,3,3], near[6], far[6]) of its six
    face cameras — what the reference's dataset hands the decoder for one target panorama
    (dataset_hm3d.py:280-314, "extrinsics_cubes" / "intrinsics_cubes" / "near_cubes")."""
    dev = pano_c2w.device
    ext = cameras.cube_face_extrinsics(pano_c2w[None])[0]
    k = cameras.cube_face_intrinsics(1, device=dev)[0]
    # materialised like the data loader's tensors (an expanded scalar would cost a copy kernel in every packing call)
    n = torch.as_tensor(near, dtype=torch.float32, device=dev).reshape(-1).expand(6).contiguous()
    f = torch.as_tensor(far, dtype=torch.float32, device=dev).reshape(-1).expand(6).contiguous()
    return ext, k, n, f


def pack_camera_views_torch(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                            scale_invariant: bool = True) -> Tensor:
    """The reference's camera glue as torch ops (cameras.view_setup: two LU inverses, fov, projection, one bmm —
    ~60 tiny launches for six cameras), packed into views[V,44].  This is what the drop-in render_cuda uses and
    what the golden captures pin on CPU; the fused path uses the one-kernel form below."""
    vs = cameras.view_setup(extrinsics, intrinsics, near, far, scale_invariant)
    return rasterizer.pack_views(vs["view_matrix"], vs["full_projection"], vs["campos"], vs["tan_fov_x"],
                                 vs["tan_fov_y"], background, scale=vs["scale"], near=near, far=far)


def pack_camera_views(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                      scale_invariant: bool = True, glue: str = "native") -> Tensor:
    """V cameras (extrinsics[V,4,4] c2w, normalised intrinsics[V,3,3], near/far[V], background [3] or
    [V,3]) -> views[V,44] for one rasteriser call: the camera half of render_cuda
    (cuda_splatting.py:64-71,80-87) for all V views at once; the cloud half of the scale-invariant
    rescale happens inside the kernels (S360View.scale).  glue="native": ONE kernel launch (s360_pack_views);
    glue="torch": the reference's own torch ops (a few ulp apart: LU vs Gauss-Jordan inverses).  No host sync."""
    if glue == "torch":
        return pack_camera_views_torch(extrinsics, intrinsics, near, far, background, scale_invariant)
    return rasterizer.pack_views_native(extrinsics, intrinsics, near, far, background, scale_invariant)


def render_views_fused(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple,
                       background: Tensor, gaussian_means: Tensor, gaussian_covariances: Tensor,
                       gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor, *, shared_campos: Optional[bool] = None,
                       max_instances: Optional[int] = None, check: str = "sync", views: Optional[Tensor] = None, glue: str = "native",
                       depth_mode: Optional[DepthRenderingMode] = None, defer_sh: bool = False,
                       mse_target: Optional[Tensor] = None, mse_weight: float = 1.0, mse_count: Optional[int] = None,
                       exchange=None, lean: Optional[bool] = None, mse_defer: bool = False, atomic_grads: Optional[bool] = None,
                       split_lists: Optional[bool] = None):
    """V <= 8 views of ONE cloud in one fused rasteriser call: means[G,3], covariances[G,3,3],
    harmonics[G,3,d_sh] (the reference's Gaussians layout, src/model/types.py:7-12, read in place),
    opacities[G] -> [V,3,h,w].  With the six face cameras of a panorama (one camera centre) and glue="torch" (the
    reference's own camera ops) this is bit-for-bit the result of six reference-style render_cuda calls; the default one-kernel
    glue agrees with it to a few ulp of the camera records (Gauss-Jordan instead of LU inverses).
    shared_campos: True = all views share one camera centre and near plane (SH colours evaluated once per
    Gaussian); False = per-view evaluation; None (default) = decided from `extrinsics` / `near`
    (views_share_camera_centre: one small synchronising read when they live on the device; with
    pre-packed `views` and no extrinsics the views are taken to be independent, False).
    lean: None = rasterizer.LEAN_LISTS (on: tile lists hold only the instances that can reach a pixel; results bit-identical).
    split_lists: None = rasterizer.SPLIT_LONG_LISTS ("auto": adaptive; True: long lists whose pixels do not saturate are composited segment-parallel).
    exchange: distributed.ExchangeConfig — multi-GPU, the gradients come back summed over the ranks (rasterize_views).
    Host synchronisation: check="sync" (default) reads the binning-overflow flag back after the forward, like
    upstream's own scan read-back, and re-renders with the exact capacity if needed; check="lazy" together with an
    explicit shared_campos never synchronises (validate later with rasterizer.last_state().overflowed()).  With depth_mode set,
    returns (colour, depth[V,h,w]): the depth maps of render_depth_cuda from the SAME pass (the reference
    rasterises every face a second time for them, decoder_splatting_cuda.py:72-97).
    With mse_target[V,3,h,w] (the supervising cube faces) the L2 loss / PSNR epilogue is fused into the render
    (rasterizer.FusedMse appended as the last result)."""
    if views is None:  # callers may pass pre-packed views
        views = pack_camera_views(extrinsics, intrinsics, near, far, background, glue=glue)
    if shared_campos is None:
        shared_campos = views_share_camera_centre(extrinsics, near) if extrinsics is not None else int(views.shape[0]) == 1
    n = gaussian_sh_coefficients.shape[-1]
    h, w = image_shape
    out = rasterizer.rasterize_views(
        gaussian_means, gaussian_covariances, gaussian_opacities, gaussian_sh_coefficients, None, views=views,
        image_height=h, image_width=w, sh_degree=isqrt(n) - 1, shared_campos=shared_campos, want_radii=False,
        max_instances=max_instances, check=check, cov9=True, sh_channel_major=True, depth_mode=depth_mode,
        defer_sh=defer_sh, mse_target=mse_target, mse_weight=mse_weight, mse_count=mse_count, exchange=exchange, lean=lean,
        mse_defer=mse_defer, atomic_grads=atomic_grads, split_lists=split_lists)
    res = (out[0],) if depth_mode is None else (out[0], out[2])
    if mse_target is not None:
        res = res + (out[-1],)
    return res[0] if len(res) == 1 else res


def views_share_camera_centre(extrinsics: Tensor, near: Tensor) -> bool:
    """True when all views have one camera centre and one near plane (the six faces of a panorama): the
    condition under which SH colours may be evaluated once per Gaussian (S360_FLAG_SHARED_CAMPOS).  For device tensors
    the comparison costs one small synchronising read per call (no cache: a tensor's address / version / shape do not identify
    its contents — the caching allocator hands the next batch's cameras the same block).  Pass shared_campos explicitly on
    latency-critical paths."""
    if extrinsics.shape[0] <= 1:
        return True
    t = extrinsics[..., :3, 3]
    return bool(((t == t[:1]).all() & (near == near.reshape(-1)[0]).all()).item())


def render_cube_faces(pano_c2w: Tensor, near: Tensor, far: Tensor, face_w: int, background: Tensor,
                      gaussian_means: Tensor, gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor,
                      gaussian_opacities: Tensor, *, max_instances: Optional[int] = None, check: str = "sync",
                      glue: str = "native") -> Tensor:
    """Convenience: panorama pose -> faces[6,3,fw,fw] in the reference's rendered order (top, front,
    left, back, right, bottom)."""
    ext, k, n, f = cube_cameras(pano_c2w, near, far)
    return render_views_fused(ext, k, n, f, (face_w, face_w), background, gaussian_means, gaussian_covariances,
                              gaussian_sh_coefficients, gaussian_opacities, max_instances=max_instances, check=check, glue=glue,
                              shared_campos=True)   # six faces of one panorama: one camera centre by construction


def render_erp_spherical(pano_c2w: Tensor, near, image_shape: tuple, background: Tensor, gaussian_means: Tensor,
                         gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor, *,
                         max_instances: Optional[int] = None, check: str = "sync") -> Tensor:
    """Native equirectangular splatting: n <= 4 panorama poses [n,4,4] of ONE cloud (the reference's
    Gaussians layouts) -> [n,3,H,W] rendered directly in ERP space, no cube faces and no stitch.  This is what
    BASELINE.json's north star literally describes; the reference itself never does it (it renders six pinhole faces,
    model_wrapper_erp.py:221-229), so results are NOT comparable with the reference — the mode is specified and pinned by
    the oracle (oracle/s360_oracle.c geo_sph, tests/test_oracle_spherical.py).  Differentiable."""
    h, w = image_shape
    n = gaussian_sh_coefficients.shape[-1]
    pose = pano_c2w.reshape(-1, 4, 4)
    nr = torch.as_tensor(near, dtype=torch.float32, device=pose.device).reshape(-1).expand(pose.shape[0])
    views = rasterizer.pack_views_spherical(pose, background, scale=1.0 / nr, near=nr)
    out = rasterizer.rasterize_views(gaussian_means, gaussian_covariances, gaussian_opacities, gaussian_sh_coefficients, None,
                                     views=views, image_height=h, image_width=w, sh_degree=isqrt(n) - 1, shared_campos=pose.shape[0] == 1,
                                     want_radii=False, max_instances=max_instances, check=check, cov9=True, sh_channel_major=True,
                                     spherical=True)
    return out[0]


@dataclass
class DecoderOutput:
    color: Tensor
    depth: Optional[Tensor]


class DecoderSplattingFused(torch.nn.Module):
    """Decoder with the reference's forward contract (decoder.py:37-48) on the fused path: the v views of
    every batch item are rendered `views_per_group` at a time (6 = the cube faces of one target panorama,
    which share a camera centre) in single rasteriser calls, colour and depth together."""

    def __init__(self, background_color=(0.0, 0.0, 0.0), views_per_group: int = 6, shared_campos: Optional[bool] = None,
                 check: str = "sync", glue: str = "native"):
        """shared_campos: None (default) = checked per group of views, with ONE small device->host read per
        forward() call (groups whose views do not share a camera centre and near plane are rendered with per-view
        SH evaluation instead of silently taking the first view's direction); True / False = trust the caller (no read).
        check: "sync" (overflow flag read back per rasteriser call, automatic re-render) or "lazy"."""
        super().__init__()
        self.register_buffer("background_color", torch.tensor(background_color, dtype=torch.float32), persistent=False)
        self.views_per_group, self.shared_campos = views_per_group, shared_campos
        self.check, self.glue = check, glue

    def forward(self, gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode=None) -> "DecoderOutput":
        b, v = extrinsics.shape[:2]
        colors, depths = [], []
        groups = [(i, slice(s, min(v, s + self.views_per_group))) for i in range(b) for s in range(0, v, self.views_per_group)]
        if self.shared_campos is None:   # one comparison kernel chain + one read for all groups of the call
            first = (torch.arange(v, device=extrinsics.device) // self.views_per_group) * self.views_per_group
            t = extrinsics[..., :3, 3]
            same = ((t == t[:, first]).all(-1) & (near == near[:, first])).cpu()
            shared = {(i, e.start): bool(same[i, e].all()) for i, e in groups}
        else:
            shared = {(i, e.start): bool(self.shared_campos) for i, e in groups}
        # all camera records of the call in ONE launch
        bgv = self.background_color
        allv = pack_camera_views(extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3), near.reshape(-1),
                                 far.reshape(-1), bgv, glue=self.glue).view(b, v, -1)
        packed = {(i, e.start): allv[i, e] for i, e in groups}
        # Gaussians whose means / covariances / harmonics are still lazy.LazyFields of one adapter call (splatter360_amd.install(adapter=True)
        # replaced the reference's GaussianAdapterERP): groups of views sharing a camera centre are rendered straight from the encoder's
        # raw outputs (s360_forward_raw / s360_backward_raw) — the [G,3,25] harmonics and [G,3,3] covariances are never materialised.
        # Any other group indexes the fields, which materialises them once with the stand-alone adapter kernels.
        from . import lazy as _lazy
        bundle = _lazy.bundle_of(gaussians)
        for i in range(b):
            cs, ds = [], []
            for s in range(0, v, self.views_per_group):
                e = slice(s, min(v, s + self.views_per_group))
                if bundle is not None and shared[(i, s)]:
                    m = bundle.module
                    nvc = bundle.shape5[1]
                    rot = bundle.sh_rotation()
                    out = rasterizer.rasterize_raw(
                        bundle.depths[i].reshape(-1), gaussians.opacities[i].reshape(-1), bundle.raw[i].reshape(-1, bundle.raw.shape[-1]),
                        bundle.extrinsics[i], views=packed[(i, s)], image_height=int(image_shape[0]), image_width=int(image_shape[1]),
                        context_shape=bundle.image_shape, scale_min=float(m.cfg.gaussian_scale_min), scale_max=float(m.cfg.gaussian_scale_max),
                        sh_rotation=None if rot is None else rot[i * nvc:(i + 1) * nvc], per_ray=bundle.per_ray, eps=bundle.eps,
                        erp_convention=_lazy._adapter.ERP_CONVENTIONS[bundle.dataset_name], differentiable_means=m._s360.differentiable_means,
                        check=self.check, depth_mode=depth_mode)
                    cs.append(out[0])
                    if depth_mode is not None:
                        ds.append(out[3])
                    continue
                out = render_views_fused(extrinsics[i, e], intrinsics[i, e], near[i, e], far[i, e], image_shape,
                                         self.background_color, gaussians.means[i], gaussians.covariances[i],
                                         gaussians.harmonics[i], gaussians.opacities[i], shared_campos=shared[(i, s)],
                                         depth_mode=depth_mode, views=packed[(i, s)], check=self.check)
                if depth_mode is None:
                    cs.append(out)
                else:
                    cs.append(out[0])
                    ds.append(out[1])
            colors.append(torch.cat(cs))
            if depth_mode is not None:
                depths.append(torch.cat(ds))
        return DecoderOutput(torch.stack(colors), None if depth_mode is None else torch.stack(depths))


class DecoderSplattingCUDA(torch.nn.Module):
    """Mirror of the reference decoder (decoder_splatting_cuda.py:19-97): forward(gaussians,
    extrinsics[b,v,4,4], intrinsics[b,v,3,3], near[b,v], far[b,v], (h,w), depth_mode)."""

    def __init__(self, background_color=(0.0, 0.0, 0.0)):
        super().__init__()
        self.register_buffer("background_color", torch.tensor(background_color, dtype=torch.float32), persistent=False)

    def forward(self, gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode=None) -> DecoderOutput:
        b, v, _, _ = extrinsics.shape
        colors = torch.zeros((b, v, 3, *image_shape), dtype=torch.float32, device=extrinsics.device)
        bg = self.background_color[None].expand(b, 3)
        for view_idx in range(v):
            colors[:, view_idx] = render_cuda(
                extrinsics[:, view_idx], intrinsics[:, view_idx], near[:, view_idx], far[:, view_idx], image_shape, bg,
                gaussians.means, gaussians.covariances, gaussians.harmonics, gaussians.opacities)
        depth = None if depth_mode is None else self.render_depth(gaussians, extrinsics, intrinsics, near, far,
                                                                  image_shape, depth_mode)
        return DecoderOutput(colors, depth)

    def render_depth(self, gaussians, extrinsics, intrinsics, near, far, image_shape, mode="depth") -> Tensor:
        b, v, _, _ = extrinsics.shape
        depths = torch.zeros((b, v, *image_shape), dtype=torch.float32, device=extrinsics.device)
        for view_idx in range(v):
            depths[:, view_idx] = render_depth_cuda(
                extrinsics[:, view_idx], intrinsics[:, view_idx], near[:, view_idx], far[:, view_idx], image_shape,
                gaussians.means, gaussians.covariances, gaussians.opacities, mode=mode)
        return depths
