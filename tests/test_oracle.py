"""CPU tests of the oracle itself: float64 finite differences, independent PyTorch-autograd
restatement, analytic known-answer scenes.  The rasteriser internals are PARITY
UNPINNED by the reference (no source, no vectors): these tests are what pins the oracle."""
import math

import numpy as np
import pytest
import torch

from helpers import small_front_scene
from oracle import oracle, torch_ref


def _loss_f64(S, m, c, s, o, w):
    r = oracle.rasterize(S, means3D=m, cov3D_precomp=c, opacities=o, shs=s, dtype=np.float64)
    return float((r.forward()["image"] * w).sum())


def test_backward_matches_float64_finite_differences():
    S, means, cov6, shs, opac = small_front_scene(n=30, seed=1, h=48, w=48)
    rng = np.random.default_rng(5)
    w = rng.standard_normal((3, 48, 48))
    r = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
    r.forward()
    g = r.backward(w)
    args = dict(m=means, c=cov6, s=shs, o=opac)
    for name, key in (("m", "means3D"), ("c", "cov3D"), ("s", "shs"), ("o", "opacities")):
        arr = args[name]
        for i in rng.choice(arr.size, size=12, replace=False):
            eps = 1e-6
            hi, lo = arr.copy(), arr.copy()
            hi.reshape(-1)[i] += eps
            lo.reshape(-1)[i] -= eps
            num = (_loss_f64(S, **{**args, name: hi}, w=w) - _loss_f64(S, **{**args, name: lo}, w=w)) / (2 * eps)
            ana = g[key].reshape(-1)[i]
            assert abs(num - ana) <= 2e-5 * max(1e-3, abs(num), abs(ana)), (key, i, num, ana)


def test_clamped_sh_channels_get_no_gradient_and_fd_still_matches():
    S, means, cov6, shs, opac = small_front_scene(n=12, seed=3, h=32, w=32)
    shs = shs.copy()
    shs[:6, 0, 1] = -4.0  # green DC far below -0.5/C0 -> clamped at 0
    r = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
    f = r.forward()
    assert f["clamped"][:6, 1].all() and (f["rgb"][:6, 1] == 0).all()
    w = np.random.default_rng(0).standard_normal((3, 32, 32))
    g = r.backward(w)
    assert np.abs(g["shs"][:6, :, 1]).max() == 0.0
    assert np.abs(g["shs"][:6, :, 0]).max() > 0.0


@pytest.mark.parametrize("dtype", [torch.float64])
def test_oracle_matches_torch_autograd_restatement(dtype):
    S, means, cov6, shs, opac = small_front_scene(n=40, seed=2, h=48, w=64)
    npdt = np.float64
    r = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=npdt)
    f = r.forward()
    w = np.random.default_rng(1).standard_normal((3, 48, 64))
    g = r.backward(w)
    t = lambda a: torch.tensor(a, dtype=dtype, requires_grad=True)
    m, c, s, o = t(means), t(cov6), t(shs), t(opac)
    img, aux = torch_ref.render(S, m, c, o, shs=s, return_aux=True)
    assert np.abs(img.detach().numpy() - f["image"]).max() < 1e-9
    np.testing.assert_array_equal(aux["radii"].numpy(), f["radii"])
    np.testing.assert_array_equal(aux["n_contrib"].numpy(), f["n_contrib"])
    (img * torch.tensor(w, dtype=dtype)).sum().backward()
    for a, b in ((m.grad, g["means3D"]), (c.grad, g["cov3D"]), (s.grad, g["shs"]), (o.grad, g["opacities"])):
        b = np.asarray(b)
        assert np.abs(a.numpy() - b).max() <= 2e-6 * (np.abs(b).max() + 1e-9)


def test_float32_oracle_close_to_float64_oracle():
    S, means, cov6, shs, opac = small_front_scene(n=60, seed=7, h=64, w=64)
    f32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float32).forward()
    f64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64).forward()
    assert np.abs(f32["image"] - f64["image"]).mean() < 1e-6
    np.testing.assert_array_equal(f32["radii"], f64["radii"])


def _one(S_over=None, mean=(0, 0, 4.0), sigma=0.2, opacity=0.8, color=(0.3, 0.6, 0.9), h=64, w=64, bg=(0.1, 0.2, 0.3)):
    view = np.eye(4)
    import torch as _t
    from splatter360_amd import cameras
    proj = cameras.get_projection_matrix(_t.tensor([1.0]), _t.tensor([100.0]), _t.tensor([math.pi / 2]), _t.tensor([math.pi / 2]))[0].numpy().astype(np.float64)
    S = dict(image_height=h, image_width=w, tanfovx=1.0, tanfovy=1.0, bg=np.array(bg), viewmatrix=view.T.copy(),
             projmatrix=(view.T @ proj.T).copy(), sh_degree=0, campos=np.zeros(3))
    if S_over:
        S.update(S_over)
    cov6 = np.array([[sigma ** 2, 0, 0, sigma ** 2, 0, sigma ** 2]])
    return S, np.array([mean], dtype=np.float64), cov6, np.array([[opacity]]), np.array([color])


def test_known_answer_single_isotropic_gaussian():
    """Closed form: centre pixel position ((0+1)*W-1)/2, 2-D variance (f*sigma/z)^2 + 0.3,
    alpha(d) = o*exp(-d^2/(2 var)), out = c*alpha + (1-alpha)*bg, cut at alpha < 1/255."""
    S, m, c6, o, col = _one()
    r = oracle.rasterize(S, means3D=m, cov3D_precomp=c6, opacities=o, colors_precomp=col, dtype=np.float64)
    f = r.forward()
    var = (32.0 * 0.2 / 4.0) ** 2 + 0.3
    assert f["radii"][0] == math.ceil(3 * math.sqrt(var))
    np.testing.assert_allclose(f["xy"][0], [31.5, 31.5])
    yy, xx = np.mgrid[0:64, 0:64]
    d2 = (xx - 31.5) ** 2 + (yy - 31.5) ** 2
    alpha = np.minimum(0.99, 0.8 * np.exp(-d2 / (2 * var)))
    alpha[alpha < 1 / 255] = 0
    # pixels outside the splat's tile rectangle never see it
    rect = f["rect"][0]
    inside = (xx >= rect[0] * 16) & (xx < rect[2] * 16) & (yy >= rect[1] * 16) & (yy < rect[3] * 16)
    alpha[~inside] = 0
    bg = np.array([0.1, 0.2, 0.3])
    expect = col[0][:, None, None] * alpha + (1 - alpha) * bg[:, None, None]
    np.testing.assert_allclose(f["image"], expect, atol=1e-12)
    assert f["n_contrib"].max() == 1


def test_alpha_cap_and_near_cull_and_background():
    S, m, c6, o, col = _one(opacity=5.0)  # o*G > 0.99 at the centre -> capped
    f = oracle.rasterize(S, means3D=m, cov3D_precomp=c6, opacities=o, colors_precomp=col, dtype=np.float64).forward()
    centre = f["image"][:, 31, 31]
    # at the nearest pixel G = exp(-0.5/var); o*G still > 0.99 -> alpha = 0.99 exactly
    np.testing.assert_allclose(centre, 0.99 * col[0] + 0.01 * np.array([0.1, 0.2, 0.3]), atol=1e-12)
    for z, vis in ((0.2, False), (0.2000001, True)):
        S, m, c6, o, col = _one(mean=(0, 0, z), sigma=0.001)
        f = oracle.rasterize(S, means3D=m, cov3D_precomp=c6, opacities=o, colors_precomp=col, dtype=np.float64).forward()
        assert (f["radii"][0] > 0) == vis
        if not vis:
            np.testing.assert_allclose(f["image"], np.broadcast_to(np.array([0.1, 0.2, 0.3])[:, None, None], (3, 64, 64)))


def test_front_to_back_order_equal_depth_ties_and_early_stop():
    S, _, _, _, _ = _one()
    # three opaque-ish splats on the axis: the nearer one dominates; equal depths keep index order
    means = np.array([[0, 0, 5.0], [0, 0, 3.0], [0, 0, 3.0]])
    cov6 = np.tile(np.array([[0.09, 0, 0, 0.09, 0, 0.09]]), (3, 1))
    col = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]])
    o = np.full((3, 1), 0.9)
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=o, colors_precomp=col, dtype=np.float32).forward()
    t = (31 // 16) * 4 + 31 // 16
    s, e = f["ranges"][t]
    assert list(f["values"][s:e]) == [1, 2, 0]  # depth 3 (idx 1), depth 3 (idx 2, tie -> index order), depth 5
    px = f["image"][:, 31, 31]
    assert px[1] > px[2] > px[0]
    # early stop: many opaque layers -> T*(1-alpha) < 1e-4 stops the pixel, later splats are ignored
    n = 12
    means = np.stack([np.zeros(n), np.zeros(n), 3.0 + np.arange(n)], 1)
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=np.tile(cov6[:1], (n, 1)), opacities=np.full((n, 1), 0.95),
                         colors_precomp=np.tile(col[:1], (n, 1)), dtype=np.float32).forward()
    # alpha ~ 0.9 each near the centre: T = 1e-1, 1e-2, 1e-3 pass, the 4th would drop T below 1e-4
    assert f["n_contrib"][31, 31] == 3 and 1e-4 <= f["final_T"][31, 31] < 2e-3


def test_tile_rect_clamps_at_image_borders_and_offscreen_is_dropped():
    S, _, _, _, _ = _one(h=48, w=80)
    means = np.array([[-3.9, -3.9, 4.0], [3.9, 3.9, 4.0], [30.0, 0, 4.0]])  # two corners, one far off-screen
    cov6 = np.tile(np.array([[0.04, 0, 0, 0.04, 0, 0.04]]), (3, 1))
    S["image_height"], S["image_width"] = 48, 80
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=np.full((3, 1), 0.5),
                         colors_precomp=np.ones((3, 3)), dtype=np.float32).forward()
    gx, gy = 5, 3
    assert f["rect"][0][0] == 0 and f["rect"][0][1] == 0
    assert f["rect"][1][2] == gx and f["rect"][1][3] == gy
    assert f["radii"][2] == 0 and f["tiles_touched"][2] == 0
    assert f["num_rendered"] == f["tiles_touched"].sum() == f["offsets"][-1]


def test_empty_cloud():
    S, _, _, _, _ = _one()
    f = oracle.rasterize(S, means3D=np.zeros((0, 3)), cov3D_precomp=np.zeros((0, 6)), opacities=np.zeros((0, 1)),
                         colors_precomp=np.zeros((0, 3)), dtype=np.float32).forward()
    assert f["num_rendered"] == 0
    np.testing.assert_allclose(f["image"], np.broadcast_to(np.array([0.1, 0.2, 0.3], np.float32)[:, None, None], (3, 64, 64)))


def _settings_from_camera(c2w, k, near, far, h, w, bg=(0.0, 0.0, 0.0)):
    """GaussianRasterizationSettings fields built by the product's restatement of the reference glue
    (cuda_splatting.py:64-112; pinned bit-for-bit by tests/test_golden_glue.py)."""
    from splatter360_amd import cameras
    vs = cameras.view_setup(c2w[None], k[None], torch.tensor([near]), torch.tensor([far]))
    return dict(image_height=h, image_width=w, tanfovx=float(vs["tan_fov_x"][0]), tanfovy=float(vs["tan_fov_y"][0]),
                bg=np.asarray(bg, np.float64), viewmatrix=vs["view_matrix"][0].numpy(), projmatrix=vs["full_projection"][0].numpy(),
                sh_degree=4, campos=vs["campos"][0].numpy(), scale=float(vs["scale"][0]))


def test_reference_smoke_scene_one_unit_gaussian_band2_sh():
    """The scene of the reference's only rasteriser script (src/scripts/test_splatter.py:21-87): ONE Gaussian at the
    origin with unit covariance, opacity 1, red SH band 2 = 10, camera 10 units away, 90-degree fov, near 0.1 /
    far 20, 512x512 — rendered here from (0,0,-10) looking down +z (SH left unrotated).  Closed form after the
    1/near rescale: depth 100, 2-D variance (256*10/100)^2 + 0.3, radius ceil(3 sigma), centre alpha capped at
    0.99, red = 0.5 + 10*Y_6(0,0,1) = 0.5 + 10*0.31539*2 (unclamped above 1), green = blue = 0.5."""
    c2w = torch.eye(4)
    c2w[2, 3] = -10.0
    k = torch.eye(3)
    k[0, 0] = k[1, 1] = k[0, 2] = k[1, 2] = 0.5
    S = _settings_from_camera(c2w, k, 0.1, 20.0, 512, 512)
    assert S["scale"] == 10.0 and abs(S["tanfovx"] - 1.0) < 1e-6
    sh = np.zeros((1, 25, 3))
    sh[0, 4:9, 0] = 10.0
    means = np.zeros((1, 3)) * S["scale"]
    cov6 = np.array([[1.0, 0, 0, 1.0, 0, 1.0]]) * S["scale"] ** 2
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=np.ones((1, 1)), shs=sh, dtype=np.float64).forward()
    var = (256.0 * 10.0 / 100.0) ** 2 + 0.3
    assert f["radii"][0] == math.ceil(3.0 * math.sqrt(var)) == 77
    np.testing.assert_allclose(f["xy"][0], [255.5, 255.5], atol=1e-4)
    np.testing.assert_allclose(f["depth"][0], 100.0, rtol=1e-6)
    red = 0.5 + 10.0 * 0.31539156525252005 * 2.0
    np.testing.assert_allclose(f["rgb"][0], [red, 0.5, 0.5], rtol=1e-6)
    a = min(0.99, math.exp(-0.5 * (0.5 ** 2 + 0.5 ** 2) / var))        # nearest pixel centres are half a pixel off
    assert a == 0.99
    np.testing.assert_allclose(f["image"][:, 255, 255], 0.99 * np.array([red, 0.5, 0.5]), rtol=1e-6)
    # 40 px to the right of the centre the splat is still above the 1/255 cut; 100 px to the right it is below it
    np.testing.assert_allclose(f["image"][1, 255, 295], 0.5 * math.exp(-0.5 * ((295 - 255.5) ** 2 + 0.25) / var), rtol=1e-5)
    assert math.exp(-0.5 * ((355 - 255.5) ** 2 + 0.25) / var) < 1.0 / 255.0 and f["image"][1, 255, 355] == 0.0
    assert f["tiles_touched"][0] == (int((255.5 + 77 + 15) // 16) - int((255.5 - 77) // 16)) ** 2


def test_gaussian_on_a_cube_edge_is_rendered_on_both_faces():
    """A splat whose centre lies in the plane bisecting two adjacent 90-degree faces projects onto the shared image
    border of both (x_ndc = +1 in one, -1 in the other) and must be binned / rendered by both."""
    from helpers import face_settings
    from splatter360_amd import cameras, synthetic
    ext = cameras.cube_face_extrinsics(torch.from_numpy(synthetic.target_pano_pose((0, 0, 0)))[None])[0]
    fa, fb = 1, 2                                       # "front" and "left" of the rendered order
    axis = lambda f: ext[f][:3, 2].numpy().astype(np.float64)   # camera +z (viewing direction) in world space
    assert abs(float(axis(fa) @ axis(fb))) < 1e-6
    p = (axis(fa) + axis(fb)) / math.sqrt(2.0) * 2.0    # 2 units away, on the bisecting plane
    hits = []
    for f in (fa, fb):
        S = face_settings(f, 64, 64)
        means = p[None] * S["scale"]
        cov6 = np.array([[0.01, 0, 0, 0.01, 0, 0.01]]) * S["scale"] ** 2
        o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=np.full((1, 1), 0.9),
                             colors_precomp=np.array([[1.0, 0.5, 0.25]]), dtype=np.float64)
        r = o.forward()
        assert r["radii"][0] > 0 and r["tiles_touched"][0] > 0
        x = r["xy"][0][0]
        assert min(abs(x - 63.5), abs(x + 0.5)) < 1e-3            # ((+-1 + 1) * 64 - 1) / 2: exactly on a vertical border
        img = r["image"]
        col = 63 if abs(x - 63.5) < 1e-3 else 0
        assert img[0, :, col].max() > 0.3 and img[0, :, 63 - col].max() == 0.0   # visible at that border only
        hits.append(col)
    assert sorted(hits) == [0, 63]                                   # opposite borders in the two faces
