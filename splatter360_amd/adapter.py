"""The Gaussian-adapter tail of the splatter360 encoder — the step that PRODUCES the 340 B/Gaussian the
rasteriser reads — as one fused HIP kernel pair (forward + backward) and, beside it, a plain
torch restatement that runs on CPU and is pinned against a golden capture of the reference module.

Reference behaviour mirrored (file:line under /root/reference):
  * GaussianAdapterERP.forward            src/model/encoder/common/gaussian_adapter_erp.py:50-119
      scale map :63-78, quaternion normalisation :82, sh_mask :38-47,86, world covariance :89-92
  * build_covariance / quaternion_to_matrix  src/model/encoder/common/gaussians.py:8-44  (xyzw order)
  * sphere un-projection                  src/geometry/sphere_projection.py:6-86 with the 'hm3d' / 'replica' ERP
                                          convention of src/geometry/utils360.py:93-104,148-153
  * rotate_sh                             src/misc/sh_rotation.py:10-30 — block-diagonal Wigner-D product.  e3nn (which
                                          builds the D matrices) is not installed in this image, so the per-view
                                          matrices are an INPUT here (`sh_rotation[V, d_sh, d_sh]`, only the
                                          (2l+1)x(2l+1) diagonal blocks are read; None = identity); with e3nn present,
                                          wigner_blocks_e3nn() produces them exactly as the reference does.

What the fused form saves: ~20 elementwise / matmul launches with their intermediates, and — with cov6=True — the
[G,3,3] covariance materialisation (the rasteriser reads the 6 unique entries).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from . import _lib


@dataclass
class AdapterGaussians:
    """Same fields as the reference's Gaussians container of the adapter (gaussian_adapter_erp.py:16-22)."""
    means: Tensor
    covariances: Tensor
    scales: Tensor
    rotations: Tensor
    harmonics: Tensor
    opacities: Tensor


def sh_mask(d_sh: int) -> Tensor:
    """1 for DC, 0.1 * 0.25^degree for the higher bands (gaussian_adapter_erp.py:38-47)."""
    deg = math.isqrt(d_sh) - 1
    m = torch.ones(d_sh, dtype=torch.float32)
    for l in range(1, deg + 1):
        m[l * l:(l + 1) * (l + 1)] = 0.1 * 0.25 ** l
    return m


def erp_directions(h: int, w: int, device=None) -> Tensor:
    """[h*w,3] unit rays of the ERP pixel centres, 'hm3d'/'replica' convention (utils360.py:93-104,148-153)."""
    x = torch.linspace(0, w - 1, w, device=device)
    y = torch.linspace(0, h - 1, h, device=device)
    theta = (0.5 - (x + 0.5) / w) * 2 * math.pi
    phi = -((y + 0.5) / h - 0.5) * math.pi
    phi, theta = torch.meshgrid(phi, theta, indexing="ij")
    return torch.stack([torch.cos(phi) * torch.sin(theta), torch.sin(phi), torch.cos(phi) * torch.cos(theta)], -1).reshape(-1, 3)


def quaternion_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    """gaussians.py:8-31 (xyzw order, normalised by 2 / (|q|^2 + eps))."""
    i, j, k, r = torch.unbind(q, dim=-1)
    two_s = 2 / ((q * q).sum(dim=-1) + eps)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(*q.shape[:-1], 3, 3)


def rotate_sh_blocks(sh: Tensor, rot: Optional[Tensor]) -> Tensor:
    """sh[..., d_sh] -> block-diagonal product with rot[..., d_sh, d_sh] (only the (2l+1)^2 diagonal blocks are used)."""
    if rot is None:
        return sh
    d_sh = sh.shape[-1]
    out = []
    for l in range(math.isqrt(d_sh)):
        s = slice(l * l, (l + 1) * (l + 1))
        out.append(torch.einsum("...ij,...j->...i", rot[..., s, s], sh[..., s]))
    return torch.cat(out, dim=-1)


def adapter_tail_torch(extrinsics: Tensor, depths: Tensor, opacities: Tensor, raw_gaussians: Tensor, image_shape,
                       scale_min: float, scale_max: float, sh_rotation: Optional[Tensor] = None, eps: float = 1e-8,
                       per_ray: int = 1) -> AdapterGaussians:
    """Plain-torch restatement of GaussianAdapterERP.forward on flat tensors: extrinsics[V,4,4] (context panorama
    c2w), depths / opacities[V,Gv] (Gv = h*w*per_ray, ray-major), raw_gaussians[V,Gv,7+3*d_sh] = (3 scale logits,
    4 quaternion xyzw, 3*d_sh SH as (xyz d_sh)).  Returns tensors with leading dims [V,Gv]."""
    h, w = image_shape
    v, gv = depths.shape
    d_sh = (raw_gaussians.shape[-1] - 7) // 3
    scales, rot, sh = raw_gaussians.split((3, 4, 3 * d_sh), dim=-1)
    scales = scale_min + (scale_max - scale_min) * scales.sigmoid()
    scales = scales * depths[..., None] * (1 / max(w, h))
    rot = rot / (rot.norm(dim=-1, keepdim=True) + eps)
    sh = sh.reshape(v, gv, 3, d_sh) * sh_mask(d_sh).to(sh.device)
    r = quaternion_to_matrix(rot)
    s = scales.diag_embed()
    cov = r @ s @ s.transpose(-1, -2) @ r.transpose(-1, -2)
    c2w = extrinsics[:, None, :3, :3]
    cov = c2w @ cov @ c2w.transpose(-1, -2)
    dirs = erp_directions(h, w, depths.device).repeat_interleave(per_ray, 0)          # [Gv,3]
    pts = dirs[None] * depths[..., None]
    means = torch.einsum("vij,vgj->vgi", extrinsics[:, :3, :3], pts) + extrinsics[:, None, :3, 3]
    harm = rotate_sh_blocks(sh, None if sh_rotation is None else sh_rotation[:, None, None])
    return AdapterGaussians(means, cov, scales, rot, harm, opacities)


def wigner_blocks_e3nn(c2w_rotations: Tensor, d_sh: int) -> Tensor:
    """[V,3,3] -> [V,d_sh,d_sh] block-diagonal Wigner-D matrices exactly as rotate_sh builds them
    (sh_rotation.py:19-24: matrix_to_angles + wigner_D per degree).  Needs e3nn (absent from this image)."""
    from e3nn.o3 import matrix_to_angles, wigner_D  # noqa: PLC0415
    alpha, beta, gamma = matrix_to_angles(c2w_rotations)
    out = torch.zeros((*c2w_rotations.shape[:-2], d_sh, d_sh), dtype=c2w_rotations.dtype, device=c2w_rotations.device)
    for l in range(math.isqrt(d_sh)):
        s = slice(l * l, (l + 1) * (l + 1))
        out[..., s, s] = wigner_D(l, alpha, beta, gamma).type(c2w_rotations.dtype).to(c2w_rotations.device)
    return out


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _AdapterTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, extrinsics, depths, raw, sh_rot, cfg):
        h, w, per_ray, smin, smax, eps, cov6 = cfg
        if not depths.is_cuda:
            raise RuntimeError("the fused adapter tail runs on the GPU only (adapter_tail_torch is the CPU restatement)")
        ext = extrinsics.detach().float().contiguous()
        dep = depths.detach().float().contiguous()
        rw = raw.detach().float().contiguous()
        rot = None if sh_rot is None else sh_rot.detach().float().contiguous()
        v, gv = dep.shape
        d_sh = (rw.shape[-1] - 7) // 3
        if gv != h * w * per_ray or rw.shape[-1] != 7 + 3 * d_sh or tuple(ext.shape) != (v, 4, 4):
            raise RuntimeError("adapter tail: inconsistent shapes")
        dev = dep.device
        means = torch.empty((v, gv, 3), dtype=torch.float32, device=dev)
        cov = torch.empty((v, gv, 6) if cov6 else (v, gv, 3, 3), dtype=torch.float32, device=dev)
        harm = torch.empty((v, gv, 3, d_sh), dtype=torch.float32, device=dev)
        scales = torch.empty((v, gv, 3), dtype=torch.float32, device=dev)
        rots = torch.empty((v, gv, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = _lib.lib().s360_adapter_forward(_ptr(ext), _ptr(dep), _ptr(rw), _ptr(rot), v, gv, h, w, per_ray, d_sh,
                                                 C.c_float(smin), C.c_float(smax), C.c_float(eps), _ptr(means), _ptr(cov),
                                                 int(not cov6), _ptr(harm), _ptr(scales), _ptr(rots), stream)
        _lib.check(rc, "s360_adapter_forward")
        ctx.cfg = (h, w, per_ray, smin, smax, eps, cov6, v, gv, d_sh)
        ctx.save_for_backward(ext, dep, rw, rot)
        ctx.mark_non_differentiable(scales, rots)
        return means, cov, harm, scales, rots

    @staticmethod
    def backward(ctx, d_means, d_cov, d_harm, _ds, _dr):
        ext, dep, rw, rot = ctx.saved_tensors
        h, w, per_ray, smin, smax, eps, cov6, v, gv, d_sh = ctx.cfg
        dev = dep.device
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else t.detach().float().contiguous()
        d_means = z(d_means, (v, gv, 3))
        d_cov = z(d_cov, (v, gv, 6) if cov6 else (v, gv, 3, 3))
        d_harm = z(d_harm, (v, gv, 3, d_sh))
        d_dep = torch.empty_like(dep)
        d_raw = torch.empty_like(rw)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = _lib.lib().s360_adapter_backward(_ptr(ext), _ptr(dep), _ptr(rw), _ptr(rot), v, gv, h, w, per_ray, d_sh,
                                                  C.c_float(smin), C.c_float(smax), C.c_float(eps), _ptr(d_means), _ptr(d_cov),
                                                  int(not cov6), _ptr(d_harm), _ptr(d_dep), _ptr(d_raw), stream)
        _lib.check(rc, "s360_adapter_backward")
        return None, d_dep, d_raw, None, None


def adapter_tail(extrinsics: Tensor, depths: Tensor, opacities: Tensor, raw_gaussians: Tensor, image_shape,
                 scale_min: float, scale_max: float, sh_rotation: Optional[Tensor] = None, eps: float = 1e-8,
                 per_ray: int = 1, cov6: bool = False) -> AdapterGaussians:
    """Fused HIP form of adapter_tail_torch (same arguments / results; differentiable w.r.t. depths and raw_gaussians,
    opacities pass through).  cov6=True returns the covariance as its 6 unique entries (00,01,02,11,12,22) — the
    rasteriser's cov3D_precomp layout — instead of [.,3,3]."""
    h, w = image_shape
    means, cov, harm, scales, rots = _AdapterTail.apply(extrinsics, depths, raw_gaussians, sh_rotation,
                                                        (int(h), int(w), int(per_ray), float(scale_min), float(scale_max),
                                                         float(eps), bool(cov6)))
    return AdapterGaussians(means, cov, scales, rots, harm, opacities)


class GaussianAdapterERP(torch.nn.Module):
    """Drop-in for the reference module (gaussian_adapter_erp.py:31-137): same constructor fields and forward
    arguments (dataset_name, extrinsics[b,v,1,1,1,4,4], depths[b,v,r,srf,spp], opacities, raw_gaussians[b,v,r,srf,1,c],
    image_shape), same result container with the reference's shapes.  `sh_rotation`: "e3nn" (default: the
    reference's Wigner-D matrices, needs e3nn), "identity", or a callable c2w_rotations[V,3,3] -> [V,d_sh,d_sh]."""

    def __init__(self, gaussian_scale_min: float, gaussian_scale_max: float, sh_degree: int, sh_rotation="e3nn"):
        super().__init__()
        self.scale_min, self.scale_max, self.sh_degree, self.sh_rotation = gaussian_scale_min, gaussian_scale_max, sh_degree, sh_rotation
        self.register_buffer("sh_mask", sh_mask(self.d_sh), persistent=False)

    @property
    def d_sh(self) -> int:
        return (self.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh

    def forward(self, dataset_name, extrinsics, depths, opacities, raw_gaussians, image_shape, eps: float = 1e-8):
        if dataset_name not in ("hm3d", "replica"):
            raise Exception(f"ERP convention of dataset {dataset_name!r} is not implemented (utils360.py:93-104 'hm3d'/'replica' only)")
        b, v, r, srf, spp = depths.shape
        h, w = image_shape
        if srf != 1 and raw_gaussians.shape[4] == 1:
            raw_gaussians = raw_gaussians.expand(b, v, r, srf, spp, raw_gaussians.shape[-1])
        ext = extrinsics.reshape(b * v, 4, 4)
        if self.sh_rotation == "identity":
            rot = None
        elif self.sh_rotation == "e3nn":
            rot = wigner_blocks_e3nn(ext[:, :3, :3], self.d_sh)
        else:
            rot = self.sh_rotation(ext[:, :3, :3])
        raw = raw_gaussians.broadcast_to(b, v, r, srf, spp, self.d_in).reshape(b * v, r * srf * spp, self.d_in)
        fn = adapter_tail if depths.is_cuda else adapter_tail_torch
        g = fn(ext, depths.reshape(b * v, -1), opacities.reshape(b * v, -1), raw, (h, w), self.scale_min, self.scale_max,
               sh_rotation=rot, eps=eps, per_ray=srf * spp)
        sh5 = (b, v, r, srf, spp)
        return AdapterGaussians(g.means.reshape(*sh5, 3), g.covariances.reshape(*sh5, 3, 3), g.scales.reshape(*sh5, 3),
                                g.rotations.reshape(*sh5, 4), g.harmonics.reshape(*sh5, 3, self.d_sh), opacities)
