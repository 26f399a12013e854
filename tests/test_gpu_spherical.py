"""HIP kernels in the native equirectangular splat mode (S360_FLAG_SPHERICAL) against the oracle's
specification of that mode (oracle/s360_oracle.c geo_sph; pinned on CPU by tests/test_oracle_spherical.py).
No reference counterpart exists: the reference renders cube faces only."""
import numpy as np
import pytest
import torch

from helpers import boundary_tensors, check_instance_slots, settings_from_views
from oracle import oracle
from splatter360_amd import decoder, rasterizer, synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("parity_lists")]   # integer state is compared with the oracle: upstream-compatible lists


def _pose(pos=(0.1, -0.2, 0.05), rot=True):
    m = np.eye(4, dtype=np.float32)
    if rot:
        m[:3, :3] = synthetic._random_rotations(np.random.default_rng(4), 1)[0]
    m[:3, 3] = pos
    return m


@pytest.mark.parametrize("case", ["uniform", "encoder_like"])
def test_spherical_forward_and_backward_vs_oracle(gpu, case):
    if case == "uniform":
        cloud = synthetic.uniform_cloud(12_000, seed=3, extent=3.0, scale_range=(0.02, 0.25))
        h, w, near = 64, 128, 0.1
    else:
        cloud = synthetic.encoder_like_cloud(64, 128, seed=2)          # 16 384 Gaussians incl. the poles and the seam
        h, w, near = 128, 256, 0.1
    n = cloud["means"].shape[0]
    pose = torch.tensor(_pose(), device=gpu)
    ps = [torch.tensor(cloud[k], device=gpu, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu)
    views = rasterizer.pack_views_spherical(pose[None], bg, scale=1.0 / near, near=near)
    img, radii = rasterizer.rasterize_views(ps[0], ps[1], ps[3], ps[2], None, views=views, image_height=h, image_width=w, sh_degree=4,
                                            shared_campos=True, cov9=True, sh_channel_major=True, spherical=True)
    assert img.shape == (1, 3, h, w) and radii.shape == (2, n)
    st = rasterizer.last_state()
    t = st.tensors()
    S = settings_from_views(views, 0, h, w)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, spherical=True)
    f = orc.forward()
    # integer state bit-exact: radii, tiles_touched (pairs: main then seam ghost), sorted list, keys; instance slots consistent
    np.testing.assert_array_equal(radii.cpu().numpy().reshape(-1), f["radii"])
    np.testing.assert_array_equal(t["tiles_touched"].cpu().numpy().reshape(-1).astype(np.uint32), f["tiles_touched"])
    L = f["num_rendered"]
    check_instance_slots(t["slot_base"].cpu().numpy(), t["slot_pair"].cpu().numpy(), t["tiles_touched"].cpu().numpy(), L)
    assert st.num_rendered() == L and (f["radii"][n:] > 0).sum() > 0           # some seam ghosts exist
    np.testing.assert_array_equal(t["list"][:L].cpu().numpy().astype(np.uint32), f["values"])
    vis = f["radii"] > 0
    np.testing.assert_array_equal(t["rec_a"].reshape(-1, 4).cpu().numpy()[vis][:, :2], f["xy"][vis])
    np.testing.assert_array_equal(t["depths"].reshape(-1).cpu().numpy()[vis], f["depth"][vis])
    d = np.abs(img[0].detach().cpu().numpy() - f["image"])
    assert d.mean() <= 1e-5 and d.max() <= 2e-4, (d.mean(), d.max())
    assert (t["n_contrib"][0].cpu().numpy().astype(np.uint32) != f["n_contrib"]).mean() <= 2e-3
    # backward
    gimg = np.random.default_rng(1).standard_normal((3, h, w)).astype(np.float32)
    img.backward(torch.tensor(gimg, device=gpu)[None])
    g32 = orc.backward(gimg)
    o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, spherical=True, dtype=np.float64)
    o64.forward()
    g64 = o64.backward(gimg)
    sc = np.float64(S["scale"])
    r, c = np.triu_indices(3)
    got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c], shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1),
               opacities=ps[3].grad.cpu().numpy())
    fold = dict(means3D=sc, cov3D=sc * sc, shs=1.0, opacities=1.0)
    for k in got:
        w64 = np.asarray(g64[k], np.float64).reshape(got[k].shape) * fold[k]
        scale = np.abs(w64).max() + 1e-30
        e = np.abs(got[k] - w64).max() / scale
        e32 = np.abs(np.asarray(g32[k], np.float64).reshape(got[k].shape) * fold[k] - w64).max() / scale
        assert e <= max(5e-4, 3.0 * e32), (k, e, e32)


def test_render_erp_spherical_lands_context_pixels_on_themselves(gpu):
    """Decoder-level entry point: an encoder-like cloud (one Gaussian per ERP pixel of a context panorama at the origin)
    rendered from that same pose puts every Gaussian on the pixel that produced it."""
    hc, wc = 32, 64
    dirs = synthetic.erp_ray_directions(hc, wc).reshape(-1, 3)
    means = (dirs * 2.0).astype(np.float32)
    n = means.shape[0]
    cov = np.tile((np.eye(3) * 1e-4).astype(np.float32), (n, 1, 1))
    harm = np.zeros((n, 3, 25), np.float32)
    harm[:, 0, 0] = ((np.arange(n) % wc) / wc - 0.5) / 0.28209479177387814          # red encodes the source column
    harm[:, 1, 0] = ((np.arange(n) // wc) / hc - 0.5) / 0.28209479177387814         # green the source row
    opa = np.full((n,), 0.95, np.float32)
    t = lambda a: torch.tensor(a, device=gpu)
    img = decoder.render_erp_spherical(torch.eye(4, device=gpu), 0.1, (hc, wc), torch.zeros(3, device=gpu), t(means), t(cov), t(harm), t(opa))[0]
    cols = torch.arange(wc, device=gpu)[None, :].expand(hc, wc) / wc
    rows = torch.arange(hc, device=gpu)[:, None].expand(hc, wc) / hc
    inner = slice(4, hc - 4)   # near the poles neighbouring splats overlap (the Jacobian stretches them along the row)
    assert (img[0, inner] / img[0, inner].amax() - cols[inner] / cols[inner].amax()).abs().mean().item() < 0.05
    assert (img[1, inner] - rows[inner] * (img[1, inner].amax() / rows[inner].amax())).abs().mean().item() < 0.05
