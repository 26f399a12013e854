"""Shared scene builders for the tests (CPU-only; uses the product's camera helpers)."""
from __future__ import annotations

import numpy as np
import torch

from splatter360_amd import cameras, synthetic


def face_settings(face: int, h: int, w: int, near=0.1, far=10.0, position=(0.0, 0.0, 0.0), bg=(0.0, 0.0, 0.0)):
    """GaussianRasterizationSettings fields (numpy) of one cube face, built the way the reference
    builds them (cuda_splatting.py:64-112) for a panorama at `position` with identity rotation."""
    pano = torch.from_numpy(synthetic.target_pano_pose(position))[None]
    ext = cameras.cube_face_extrinsics(pano)[0, face][None]
    k = cameras.cube_face_intrinsics(1)[0, face][None]
    vs = cameras.view_setup(ext, k, torch.tensor([near]), torch.tensor([far]))
    return dict(image_height=h, image_width=w, tanfovx=float(vs["tan_fov_x"][0]),
                tanfovy=float(vs["tan_fov_y"][0]), bg=np.asarray(bg, np.float32),
                viewmatrix=vs["view_matrix"][0].numpy(), projmatrix=vs["full_projection"][0].numpy(),
                sh_degree=4, campos=vs["campos"][0].numpy(), scale=float(vs["scale"][0]))


def settings_from_views(views, i: int, h: int, w: int, sh_degree: int = 4):
    """The same GaussianRasterizationSettings fields, read back from row i of a packed S360View tensor [V,44] — so
    that the oracle sees bit-identical camera records to the HIP path whichever glue (torch ops or the one-kernel
    s360_pack_views) produced them."""
    r = views[i].detach().cpu().numpy().astype(np.float32)
    return dict(image_height=h, image_width=w, tanfovx=float(r[35]), tanfovy=float(r[36]), bg=r[37:40].copy(),
                viewmatrix=r[0:16].reshape(4, 4).copy(), projmatrix=r[16:32].reshape(4, 4).copy(), sh_degree=sh_degree,
                campos=r[32:35].copy(), scale=float(r[40]))


def boundary_tensors(cloud: dict, scale: float):
    """means3D, cov6, shs[G,n,3], opacities[G,1] exactly as render_cuda hands them over
    (cuda_splatting.py:68-75,115-123)."""
    means = cloud["means"] * np.float32(scale)
    cov = cloud["covariances"] * np.float32(scale) ** 2
    r, c = np.triu_indices(3)
    cov6 = np.ascontiguousarray(cov[:, r, c])
    shs = np.ascontiguousarray(cloud["harmonics"].transpose(0, 2, 1))
    return means.astype(np.float32), cov6.astype(np.float32), shs.astype(np.float32), cloud["opacities"][:, None].astype(np.float32)


def small_front_scene(n=40, seed=0, h=64, w=64, d_sh=25, spread=0.6, zrange=(2.0, 6.0), srange=(0.03, 0.25)):
    """A handful of well-conditioned Gaussians in front of an identity camera (tanfov 1), as
    already-scaled boundary tensors.  Used for finite-difference and known-answer tests."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(*zrange, n)
    xy = rng.uniform(-spread, spread, (n, 2)) * z[:, None]
    means = np.concatenate([xy, z[:, None]], 1)
    s = np.exp(rng.uniform(np.log(srange[0]), np.log(srange[1]), (n, 3)))
    r = synthetic._random_rotations(rng, n)
    cov = np.einsum("nij,nj,nkj->nik", r, s * s, r)
    rr, cc = np.triu_indices(3)
    cov6 = cov[:, rr, cc]
    shs = rng.standard_normal((n, d_sh, 3)) * synthetic.sh_band_mask(d_sh)[None, :, None] * 3
    shs[:, 0, :] = rng.uniform(0.2, 1.5, (n, 3))
    opac = rng.uniform(0.2, 0.95, (n, 1))
    near, far = 1.0, 100.0
    proj = cameras.get_projection_matrix(torch.tensor([near]), torch.tensor([far]), torch.tensor([np.pi / 2]), torch.tensor([np.pi / 2]))[0].numpy().astype(np.float64)
    view = np.eye(4)
    settings = dict(image_height=h, image_width=w, tanfovx=1.0, tanfovy=1.0, bg=np.array([0.1, 0.2, 0.3]),
                    viewmatrix=view.T.copy(), projmatrix=(view.T @ proj.T).copy(), sh_degree=int(round(np.sqrt(d_sh))) - 1,
                    campos=np.zeros(3))
    return settings, means, cov6, shs, opac


def check_instance_slots(slot_base, slot_pair, tiles_touched, L):
    """Training state that replaces upstream's point_offsets scan: every visible pair owns `touched` consecutive
    instance slots starting at slot_base[pair]; the ranges tile [0, L) exactly and slot_pair is their owner table.
    (Which range a pair gets is run-dependent: k_emit reserves block-wise with one atomic per block.)"""
    import numpy as np
    base = np.asarray(slot_base).reshape(-1).astype(np.int64) & 0xFFFFFFFF
    tt = np.asarray(tiles_touched).reshape(-1).astype(np.int64)
    vis = np.nonzero(tt > 0)[0]
    assert tt.sum() == L
    if L == 0:
        return
    order = np.argsort(base[vis], kind="stable")
    b, n = base[vis][order], tt[vis][order]
    assert b[0] == 0 and (b[1:] == (b + n)[:-1]).all() and (b + n)[-1] == L
    owner = np.repeat(vis[order], n)
    sp = np.asarray(slot_pair).reshape(-1)[:L].astype(np.int64) & 0xFFFFFFFF
    np.testing.assert_array_equal(sp & 0x7FFFFFFF, owner)
    np.testing.assert_array_equal(sp >> 31, (tt[owner] > 32).astype(np.int64))      # bit 31: a slot of a pair that owns more than 32
