"""The oracle's NATIVE EQUIRECTANGULAR splat mode.  There is no reference counterpart (the
reference renders cube faces only), so the mode is specified by oracle/s360_oracle.c (geo_sph) and pinned here by:
float64 finite differences of its analytic backward, closed-form pixel positions along the encoder's ERP ray
convention, the seam ghost, and agreement with the six-face cube render of the same cloud at matching resolution."""
import math

import numpy as np
import pytest

from oracle import oracle
from splatter360_amd import synthetic


def sph_settings(h, w, c2w=None, bg=(0.0, 0.0, 0.0), sh_degree=4, scale=1.0):
    c2w = np.eye(4) if c2w is None else np.asarray(c2w, np.float64)
    c2w = c2w.copy()
    c2w[:3, 3] *= scale
    w2c = np.linalg.inv(c2w)
    return dict(image_height=h, image_width=w, tanfovx=1.0, tanfovy=1.0, bg=np.asarray(bg, np.float64), viewmatrix=w2c.T.copy(),
                projmatrix=np.eye(4), sh_degree=sh_degree, campos=c2w[:3, 3].copy(), scale=scale)


def shell_cloud(n, seed, d_sh=25, rmin=1.0, rmax=4.0, smin=0.02, smax=0.15):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    means = d * rng.uniform(rmin, rmax, (n, 1))
    s = np.exp(rng.uniform(np.log(smin), np.log(smax), (n, 3)))
    r = synthetic._random_rotations(rng, n)
    cov = np.einsum("nij,nj,nkj->nik", r, s * s, r)
    rr, cc = np.triu_indices(3)
    shs = rng.standard_normal((n, d_sh, 3)) * synthetic.sh_band_mask(d_sh)[None, :, None] * 3
    shs[:, 0, :] = rng.uniform(0.2, 1.5, (n, 3))
    return means, cov[:, rr, cc], shs, rng.uniform(0.2, 0.95, (n, 1))


def test_pixel_positions_follow_the_encoder_erp_convention():
    """A point along the ray of ERP pixel (x, y) (utils360.py:93-104,148-153) lands exactly on pixel (x, y)."""
    h, w = 32, 64
    dirs = synthetic.erp_ray_directions(h, w)
    pts = []
    for (y, x) in ((0, 0), (5, 17), (16, 32), (31, 63), (20, 1)):
        pts.append(((y, x), dirs[y, x] * 2.5))
    means = np.array([p for _, p in pts])
    n = len(pts)
    cov6 = np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4]), (n, 1))
    f = oracle.rasterize(sph_settings(h, w, sh_degree=0), means3D=means, cov3D_precomp=cov6, opacities=np.full((n, 1), 0.9),
                         colors_precomp=np.ones((n, 3)), dtype=np.float64, spherical=True).forward()
    for k, ((y, x), _) in enumerate(pts):
        np.testing.assert_allclose(f["xy"][k], [x, y], atol=1e-5)
        np.testing.assert_allclose(f["depth"][k], 2.5, atol=1e-12)          # sort key = radial distance
    assert f["tiles_touched"].shape == (2 * n,)                                 # pairs: main + seam ghost


def test_seam_ghost_renders_the_far_side_of_a_footprint():
    """A Gaussian straight behind the panorama (theta = +-pi) straddles the seam: the left and right image edges
    both receive it, symmetrically."""
    h, w = 32, 64
    means = np.array([[0.0, 0.0, -2.0]])
    cov6 = np.array([[0.02, 0, 0, 0.02, 0, 0.02]])
    f = oracle.rasterize(sph_settings(h, w, sh_degree=0), means3D=means, cov3D_precomp=cov6, opacities=np.array([[0.9]]),
                         colors_precomp=np.array([[1.0, 0.5, 0.25]]), dtype=np.float64, spherical=True).forward()
    assert (f["radii"] > 0).all() and f["xy"][0][0] != f["xy"][1][0] and abs(abs(f["xy"][0][0] - f["xy"][1][0]) - w) < 1e-9
    img = f["image"][0]
    assert img[16, 0] > 0.1 and img[16, w - 1] > 0.1
    np.testing.assert_allclose(img[:, :4], img[:, ::-1][:, :4], atol=1e-6)       # symmetric about the seam
    assert img[16, w // 2] == 0.0


def _loss(S, m, c, s, o, wgt):
    r = oracle.rasterize(S, means3D=m, cov3D_precomp=c, opacities=o, shs=s, dtype=np.float64, spherical=True)
    return float((r.forward()["image"] * wgt).sum())


@pytest.mark.parametrize("seed", [0, 1])
def test_spherical_backward_matches_float64_finite_differences(seed):
    h, w = 32, 64
    means, cov6, shs, opac = shell_cloud(40, seed)
    if seed == 1:   # some Gaussians near the poles (rho clamp inside the Jacobian) and one across the seam
        means[:4] = np.array([[0.02, 2.0, 0.03], [-0.05, -1.5, 0.02], [0.3, 2.5, -0.2], [0.01, 0.0, -2.2]])
    q = np.array([0.9, 0.1, -0.3, 0.2]); q /= np.linalg.norm(q)
    ww, x, y, z = q
    c2w = np.eye(4)
    c2w[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - ww * z), 2 * (x * z + ww * y)], [2 * (x * y + ww * z), 1 - 2 * (x * x + z * z), 2 * (y * z - ww * x)],
                   [2 * (x * z - ww * y), 2 * (y * z + ww * x), 1 - 2 * (x * x + y * y)]]
    c2w[:3, 3] = [0.1, -0.2, 0.05]
    S = sph_settings(h, w, c2w, bg=(0.1, 0.2, 0.3))
    rng = np.random.default_rng(5 + seed)
    wgt = rng.standard_normal((3, h, w))
    r = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64, spherical=True)
    f = r.forward()
    assert (f["radii"][:40] > 0).sum() >= 30
    g = r.backward(wgt)
    args = dict(m=means, c=cov6, s=shs, o=opac)
    for name, key in (("m", "means3D"), ("c", "cov3D"), ("s", "shs"), ("o", "opacities")):
        arr = args[name]
        for i in list(rng.choice(arr.size, size=10, replace=False)) + ([0, 1, 2, 3, 4, 5, 9, 10, 11] if name == "m" and seed == 1 else []):
            eps = 1e-6
            hi, lo = arr.copy(), arr.copy()
            hi.reshape(-1)[i] += eps
            lo.reshape(-1)[i] -= eps
            num = (_loss(S, **{**args, name: hi}, wgt=wgt) - _loss(S, **{**args, name: lo}, wgt=wgt)) / (2 * eps)
            ana = g[key].reshape(-1)[i]
            assert abs(num - ana) <= 5e-5 * max(1e-3, abs(num), abs(ana)), (key, i, num, ana)


def test_spherical_render_agrees_with_the_cube_render_away_from_the_poles():
    """Same cloud, same panorama pose: the ERP splat and the six-face cube render + stitch are two discretisations of
    the same radiance; along the equator band (where both projections are near-isotropic) the images agree closely."""
    import torch
    from helpers import boundary_tensors, face_settings
    from splatter360_amd import stitch
    h, w, fw = 64, 128, 32
    means, cov6, shs, opac = shell_cloud(400, 3, rmin=2.0, rmax=3.0, smin=0.08, smax=0.2)
    f = oracle.rasterize(sph_settings(h, w), means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, spherical=True).forward()
    faces = []
    for face in range(6):
        S = face_settings(face, fw, fw, near=1.0, far=100.0)
        faces.append(oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs).forward()["image"])
    c = torch.tensor(np.stack(faces)).clone()           # change_order (model_wrapper_erp.py:135-145) + Cube2Equirec
    c[0] = torch.flip(c[0], dims=[-1, -2])
    c[5] = torch.flip(c[5], dims=[-1, -2])
    c = c[[3, 4, 1, 2, 0, 5]]
    vol = torch.stack(list(c), 1)[None]
    grid = stitch.Cube2Equirec(fw, h, w).sample_grid
    erp_cube = torch.nn.functional.grid_sample(vol, grid, padding_mode="border", align_corners=True)[0, :, 0].numpy()
    corr = lambda a, b: float(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    band = slice(h // 2 - 8, h // 2 + 8)
    a = f["image"]
    assert corr(a[:, band], erp_cube[:, band]) > 0.97 and corr(a, erp_cube) > 0.95          # measured 0.981 / 0.965
    # and it is THIS alignment: mirrored or shifted variants decorrelate
    assert corr(a[:, :, ::-1], erp_cube) < 0.5 and corr(a[:, ::-1], erp_cube) < 0.5
    assert all(corr(np.roll(a, sft, axis=2), erp_cube) < corr(a, erp_cube) - 0.05 for sft in (-1, 1))
