"""PyTorch-CPU restatement of the rasteriser (vectorised, autograd backward).

TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline leg may import it).
PARITY UNPINNED for the rasteriser internals — see oracle/s360_oracle.c.  This is the
"PyTorch-CPU composite reference" of BASELINE.json configs[0] (the reference itself has no CPU
render path: SURVEY.md §0.4) and an independent check of the C oracle: same algorithm
(call site /root/reference/src/model/decoder/cuda_splatting.py:99-124),
different implementation style (per-tile dense alpha matrices + cumprod, autograd for gradients).

Upstream gradient conventions reproduced with straight-through tricks:
  * alpha = min(0.99, o*G) back-propagates as if unclamped (A.4 #1);
  * clamped tangent coordinates are constants for the backward (A.4 #7);
  * max(rgb, 0) has zero gradient where clamped (A.4 #9).
"""
from __future__ import annotations

import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]
C4 = [2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
      -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761]


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    """[P,3] unit dirs -> [P,(deg+1)^2]."""
    x, y, z = d.unbind(-1)
    Y = [torch.full_like(x, C0)]
    if deg > 0:
        Y += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        Y += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        Y += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    if deg > 3:
        Y += [C4[0] * xy * (xx - yy), C4[1] * yz * (3 * xx - yy), C4[2] * xy * (7 * zz - 1), C4[3] * yz * (7 * zz - 3),
              C4[4] * (zz * (35 * zz - 30) + 3), C4[5] * xz * (7 * zz - 3), C4[6] * (xx - yy) * (7 * zz - 1),
              C4[7] * xz * (xx - 3 * yy), C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(Y, -1)


def render(settings: dict, means3D, cov6, opacities, shs=None, colors_precomp=None, return_aux=False):
    """Differentiable [3,H,W] image.  `settings` as in oracle.rasterize (numpy / python values);
    tensors are torch CPU tensors (float32 or float64) that may require grad."""
    dt = means3D.dtype
    T_ = lambda a: torch.as_tensor(a, dtype=dt).reshape(-1)
    H, W = int(settings["image_height"]), int(settings["image_width"])
    V, Pm = T_(settings["viewmatrix"]), T_(settings["projmatrix"])
    campos, bg = T_(settings["campos"]), T_(settings["bg"])
    tfx, tfy = float(settings["tanfovx"]), float(settings["tanfovy"])
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    Vm, Pmm = V.reshape(4, 4), Pm.reshape(4, 4)  # row-vector convention: p_row @ M
    ph = torch.cat([means3D, ones], 1)
    t = (ph @ Vm)[:, :3]
    hom = ph @ Pmm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    tz = t[:, 2]
    limx, limy = 1.3 * tfx, 1.3 * tfy
    rx, ry = t[:, 0] / tz, t[:, 1] / tz
    txc = torch.where((rx < -limx) | (rx > limx), (rx.clamp(-limx, limx) * tz).detach(), rx * tz)
    tyc = torch.where((ry < -limy) | (ry > limy), (ry.clamp(-limy, limy) * tz).detach(), ry * tz)
    fx, fy = W / (2 * tfx), H / (2 * tfy)
    z0 = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, z0, -(fx * txc) / (tz * tz)], -1),
                     torch.stack([z0, fy / tz, -(fy * tyc) / (tz * tz)], -1)], 1)  # [P,2,3]
    R = Vm[:3, :3].T  # camera-from-world rotation
    M = J @ R
    i = [0, 1, 2, 1, 3, 4, 2, 4, 5]
    S = cov6[:, i].reshape(P, 3, 3)
    cov2 = M @ S @ M.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_inv = 1.0 / det
    conic = torch.stack([c * det_inv, -b * det_inv, a * det_inv], -1)
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3 * torch.sqrt(lam))
    px = ((ndc[:, 0] + 1) * W - 1) * 0.5
    py = ((ndc[:, 1] + 1) * H - 1) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        minx = ((px - radius) / 16).trunc().clamp(0, gx)
        miny = ((py - radius) / 16).trunc().clamp(0, gy)
        maxx = ((px + radius + 15) / 16).trunc().clamp(0, gx)
        maxy = ((py + radius + 15) / 16).trunc().clamp(0, gy)
        visible = (tz > 0.2) & (det != 0) & ((maxx - minx) * (maxy - miny) > 0)
    if shs is not None:
        deg = int(settings["sh_degree"])
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        Y = sh_basis(deg, d)
        rgb = torch.einsum("pk,pkc->pc", Y, shs[:, : (deg + 1) ** 2]) + 0.5
        rgb = torch.clamp(rgb, min=0.0)
    else:
        rgb = colors_precomp
    op = opacities.reshape(-1)

    image = torch.zeros(3, H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    final_T = torch.ones(H, W, dtype=dt)
    depth_sort = tz.detach()
    for ty in range(gy):
        for tx in range(gx):
            sel = visible & (minx <= tx) & (tx < maxx) & (miny <= ty) & (ty < maxy)
            ids = sel.nonzero().reshape(-1)
            ys = torch.arange(ty * 16, min(ty * 16 + 16, H))
            xs = torch.arange(tx * 16, min(tx * 16 + 16, W))
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            if ids.numel() == 0:
                image[:, yy, xx] = bg[:, None, None]
                continue
            order = torch.sort(depth_sort[ids], stable=True).indices
            ids = ids[order]
            pxf, pyf = xx.reshape(-1, 1).to(dt), yy.reshape(-1, 1).to(dt)
            dx = px[ids][None, :] - pxf
            dy = py[ids][None, :] - pyf
            con = conic[ids]
            power = -0.5 * (con[:, 0] * dx * dx + con[:, 2] * dy * dy) - con[:, 1] * dx * dy
            G = torch.exp(torch.clamp(power, max=0.0))
            raw = op[ids][None, :] * G
            alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # min(0.99, .) with unclamped gradient
            ok = (power <= 0) & (alpha >= 1.0 / 255.0)
            alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
            one_m = 1 - alpha
            Tincl = torch.cumprod(one_m, dim=1)  # T after entry j
            Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
            with torch.no_grad():
                stop = ok & (Tincl < 0.0001)
                stopped = torch.cumsum(stop.to(torch.int64), 1) > 0  # this entry and everything after is dropped
                keep = ok & ~stopped
            w = torch.where(keep, alpha * Texcl, torch.zeros_like(alpha))
            col = w @ rgb[ids]
            Tfin = torch.where(keep, one_m, torch.ones_like(one_m)).prod(dim=1)
            image[:, yy.reshape(-1), xx.reshape(-1)] = (col + Tfin[:, None] * bg[None, :]).T
            with torch.no_grad():
                idx1 = torch.arange(1, ids.numel() + 1)[None, :].expand_as(keep)
                n_contrib[yy.reshape(-1), xx.reshape(-1)] = torch.where(keep, idx1, torch.zeros_like(idx1)).max(dim=1).values
                final_T[yy.reshape(-1), xx.reshape(-1)] = Tfin
    if return_aux:
        return image, dict(radii=torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32),
                           n_contrib=n_contrib, final_T=final_T.detach(), xy=torch.stack([px, py], -1).detach(),
                           rgb=rgb.detach(), conic=conic.detach())
    return image
