"""Plain-torch / numpy restatement of the GaussianAdapterERP tail and of rotate_sh.

TEST INFRASTRUCTURE: only tests/ may import this file.  It checks the fused HIP kernels of
splatter360_amd/csrc/s360_adapter.hip (s360_adapter_forward / backward, s360_sh_rotation_blocks); the product package never
imports it and has no CPU path of its own.

What it restates (file:line under /root/reference):
  * GaussianAdapterERP.forward   src/model/encoder/common/gaussian_adapter_erp.py:50-119 — pinned, values AND gradients,
    against tests/golden/adapter_erp_tail.npz (captured from the reference module itself with rotate_sh = identity);
  * the sphere un-projection runs under torch.no_grad() in the reference (src/geometry/sphere_projection.py:14-86, the
    return included): the means it hands on are DETACHED — depth receives gradient through the scales only.
    `differentiable_means=True` is this project's opt-in deviation;
  * rotate_sh                     src/misc/sh_rotation.py:10-30: per degree l, D^l = e3nn.o3.wigner_D(l, *matrix_to_angles(R)).
    e3nn is not installed here, so `wigner_blocks()` below builds the same matrices from their defining property
    Y^l(R d) = D^l(R) Y^l(d) in e3nn's real basis, restated from e3nn's documentation / generated formulas:
    polar axis y, azimuth from z towards x, m = -l..l, no Condon-Shortley phase, i.e.
        Y_{l,m}  = sqrt(2 (l-|m|)!/(l+|m|)!) * d^|m|P_l/dy^|m| * { Im (z+ix)^|m| (m<0) | Re (z+ix)^m (m>0) },  Y_{l,0} = P_l(y)
    (l = 1: (x, y, z); l = 2: sqrt3 xz, sqrt3 xy, y^2 - (x^2+z^2)/2, sqrt3 yz, sqrt3/2 (z^2-x^2) — e3nn's own l <= 2 forms; three
    l = 3 and four l = 4 members are checked against e3nn's generation recursion as recalled: tests/test_adapter_cpu.py).
    PARITY UNPINNED for this convention: nothing under /root/reference holds a rotated-harmonics vector and e3nn cannot be
    imported; tests pin the construction by properties (D^1 = R, orthogonality, homomorphism, equivariance at fresh directions).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
from torch import Tensor


def sh_mask(d_sh: int) -> Tensor:
    """gaussian_adapter_erp.py:38-47."""
    deg = math.isqrt(d_sh) - 1
    m = torch.ones(d_sh, dtype=torch.float32)
    for l in range(1, deg + 1):
        m[l * l:(l + 1) * (l + 1)] = 0.1 * 0.25 ** l
    return m


def erp_directions(h: int, w: int, device=None, dataset_name: str = "hm3d") -> Tensor:
    """utils360.py: get_xy_coords :21-35, equi_2_spherical :37-104, spherical_2_cartesian :106-153, per dataset name."""
    x = torch.linspace(0, w - 1, w, device=device)
    y = torch.linspace(0, h - 1, h, device=device)
    y, x = torch.meshgrid(y, x, indexing="ij")
    if dataset_name in ("hm3d", "replica"):
        theta = (0.5 - (x + 0.5) / w) * 2 * math.pi
        phi = -((y + 0.5) / h - 0.5) * math.pi
        d = (torch.cos(phi) * torch.sin(theta), torch.sin(phi), torch.cos(phi) * torch.cos(theta))
    elif dataset_name == "m3d":
        theta = x / (w - 1) * 2 * math.pi - 0.5 * math.pi
        phi = y / (h - 1) * math.pi
        d = (torch.sin(phi) * torch.cos(theta), torch.cos(phi), torch.sin(phi) * torch.sin(theta))
    elif dataset_name == "residential":
        theta = math.pi * (2 * x / (w - 1) - 1.5)
        phi = math.pi * (0.5 - y / (h - 1))
        d = (torch.cos(theta) * torch.cos(phi), torch.sin(phi), torch.sin(theta) * torch.cos(phi))
    elif dataset_name in ("CoffeeArea", "outdoor_colmap"):
        theta = (-2 * math.pi / (w - 1)) * x + 2 * math.pi
        phi = (math.pi / (h - 1)) * y
        d = (torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi))
    else:
        raise Exception(dataset_name)
    return torch.stack(d, -1).reshape(-1, 3)


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
                       per_ray: int = 1, differentiable_means: bool = False, dataset_name: str = "hm3d"):
    """GaussianAdapterERP.forward on flat tensors: extrinsics[V,4,4] (context panorama c2w), depths / opacities[V,Gv]
    (Gv = h*w*per_ray, ray-major), raw_gaussians[V,Gv,7+3*d_sh] = (3 scale logits, 4 quaternion xyzw, 3*d_sh SH as (xyz d_sh)).
    Returns a namespace with fields means, covariances, scales, rotations, harmonics, opacities ([V,Gv,...])."""
    from types import SimpleNamespace
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
    dirs = erp_directions(h, w, depths.device, dataset_name).repeat_interleave(per_ray, 0)          # [Gv,3]
    dm = depths if differentiable_means else depths.detach()   # sphere_projection.py:14: the reference computes means under no_grad
    pts = dirs[None] * dm[..., None]
    means = torch.einsum("vij,vgj->vgi", extrinsics[:, :3, :3], pts) + extrinsics[:, None, :3, 3]
    harm = rotate_sh_blocks(sh, None if sh_rotation is None else sh_rotation[:, None, None])
    return SimpleNamespace(means=means, covariances=cov, scales=scales, rotations=rot, harmonics=harm, opacities=opacities)


# ---------------------------------------------------------------------------------------------- rotate_sh (e3nn convention)
def _legendre_derivs(l: int, y: np.ndarray):
    """[d^m P_l / dy^m for m = 0..l] (explicit polynomials, l <= 4)."""
    o = np.ones_like(y)
    return {0: [o],
            1: [y, o],
            2: [(3 * y * y - 1) / 2, 3 * y, 3 * o],
            3: [(5 * y ** 3 - 3 * y) / 2, (15 * y * y - 3) / 2, 15 * y, 15 * o],
            4: [(35 * y ** 4 - 30 * y * y + 3) / 8, (35 * y ** 3 - 15 * y) / 2, (105 * y * y - 15) / 2, 105 * y, 105 * o]}[l]


def e3nn_real_sh(l: int, d: np.ndarray) -> np.ndarray:
    """[N,3] unit directions -> [N,2l+1] real spherical harmonics of degree l in e3nn's basis (see the header)."""
    d = np.asarray(d, np.float64)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    q = _legendre_derivs(l, y)
    zx = (z + 1j * x)
    out = np.zeros((d.shape[0], 2 * l + 1))
    out[:, l] = q[0]
    for m in range(1, l + 1):
        n = math.sqrt(2.0 * math.factorial(l - m) / math.factorial(l + m))
        e = zx ** m
        out[:, l - m] = n * q[m] * e.imag
        out[:, l + m] = n * q[m] * e.real
    return out


def wigner_blocks(rotations: np.ndarray, d_sh: int, n_dirs: int = 256, seed: int = 0) -> np.ndarray:
    """[V,3,3] rotations -> [V,d_sh,d_sh] block-diagonal D with Y^l(R d) = D^l(R) Y^l(d): least squares over n_dirs random
    directions, float64 (the kernel under test solves the same identity on 2l+1 fixed directions with tabulated inverses)."""
    rot = np.asarray(rotations, np.float64).reshape(-1, 3, 3)
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n_dirs, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = np.zeros((rot.shape[0], d_sh, d_sh))
    for l in range(math.isqrt(d_sh)):
        s = slice(l * l, (l + 1) ** 2)
        y0 = e3nn_real_sh(l, d)                                   # [N, 2l+1]
        for v in range(rot.shape[0]):
            y1 = e3nn_real_sh(l, d @ rot[v].T)                    # Y(R d)
            out[v, s, s] = np.linalg.lstsq(y0, y1, rcond=None)[0].T   # y0 D^T = y1
    return out
