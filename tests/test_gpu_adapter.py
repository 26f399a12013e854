"""Fused adapter-tail kernels (s360_adapter_forward / backward) against the golden capture of the reference's
GaussianAdapterERP (rotate_sh = identity, see tests/golden/make_golden_adapter.py) and against the torch restatement
(values and autograd gradients), with and without SH rotation blocks, [.,3,3] and 6-entry covariance layouts."""
from pathlib import Path

import numpy as np
import pytest
import torch

from splatter360_amd import adapter, decoder, synthetic

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def test_fused_tail_matches_reference_capture(gpu):
    g = np.load(G / "adapter_erp_tail.npz")
    t = lambda k: torch.tensor(g[k], device=gpu)
    b, v, r = g["depths"].shape[:3]
    h, w = (int(x) for x in g["image_shape"])
    mod = adapter.GaussianAdapterERP(float(g["scale_min"]), float(g["scale_max"]), 4, sh_rotation="identity").to(gpu)
    out = mod("hm3d", t("extrinsics")[:, :, None, None, None], t("depths"), t("opacities_in"), t("raw_gaussians"), (h, w))
    for name, want in (("means", g["means"]), ("covariances", g["covariances"]), ("harmonics", g["harmonics_unrotated"]),
                       ("scales", g["scales"]), ("rotations", g["rotations"]), ("opacities", g["opacities"])):
        got = getattr(out, name).cpu().numpy().reshape(want.shape)
        np.testing.assert_allclose(got, want, rtol=3e-6, atol=3e-6 * np.abs(want).max(), err_msg=name)


def _random_case(gpu, v, h, w, seed, with_rot):
    rng = np.random.default_rng(seed)
    ext = np.tile(np.eye(4, dtype=np.float32), (v, 1, 1))
    ext[:, :3, :3] = synthetic._random_rotations(rng, v)
    ext[:, :3, 3] = rng.uniform(-1, 1, (v, 3))
    gv = h * w
    dep = np.exp(rng.uniform(np.log(0.5), np.log(8.0), (v, gv))).astype(np.float32)
    raw = rng.standard_normal((v, gv, 82)).astype(np.float32)
    opa = rng.uniform(0.05, 0.95, (v, gv)).astype(np.float32)
    rot = None
    if with_rot:
        rot = np.zeros((v, 25, 25), np.float32)
        for l in range(5):
            s = slice(l * l, (l + 1) ** 2)
            q, _ = np.linalg.qr(rng.standard_normal((v, 2 * l + 1, 2 * l + 1)))
            rot[:, s, s] = q
    tt = lambda a: None if a is None else torch.tensor(a, device=gpu)
    return tt(ext), tt(dep), tt(opa), tt(raw), tt(rot)


@pytest.mark.parametrize("with_rot", [False, True])
@pytest.mark.parametrize("cov6", [False, True])
def test_fused_tail_values_and_gradients_match_torch_autograd(gpu, with_rot, cov6):
    v, h, w = 3, 12, 24
    ext, dep, opa, raw, rot = _random_case(gpu, v, h, w, 5 + with_rot, with_rot)
    res = []
    wm = torch.randn(v, h * w, 3, device=gpu)
    wc = torch.randn(v, h * w, 3, 3, device=gpu)
    wh = torch.randn(v, h * w, 3, 25, device=gpu)
    r_, c_ = torch.triu_indices(3, 3)
    for fused in (False, True):
        d = dep.clone().requires_grad_(True)
        rw = raw.clone().requires_grad_(True)
        if fused:
            out = adapter.adapter_tail(ext, d, opa, rw, (h, w), 0.5, 15.0, sh_rotation=rot, cov6=cov6)
            cov_term = (out.covariances * wc[:, :, r_, c_]).sum() if cov6 else (out.covariances * wc).sum()
        else:
            out = adapter.adapter_tail_torch(ext, d, opa, rw, (h, w), 0.5, 15.0, sh_rotation=rot)
            # the 6-entry layout reads the upper triangle only (cuda_splatting.py:115,123)
            cov_term = (out.covariances[:, :, r_, c_] * wc[:, :, r_, c_]).sum() if cov6 else (out.covariances * wc).sum()
        ((out.means * wm).sum() + cov_term + (out.harmonics * wh).sum()).backward()
        res.append((out, d.grad, rw.grad))
    (ot, dt, rt), (of, df, rf) = res
    cov_t = ot.covariances[:, :, r_, c_] if cov6 else ot.covariances
    for name, a, b in (("means", of.means, ot.means), ("cov", of.covariances, cov_t), ("harm", of.harmonics, ot.harmonics),
                       ("d_depth", df, dt), ("d_raw_scale", rf[..., :3], rt[..., :3]), ("d_raw_quat", rf[..., 3:7], rt[..., 3:7]),
                       ("d_raw_sh", rf[..., 7:], rt[..., 7:])):
        scale = b.abs().max().item() + 1e-20
        assert (a - b).abs().max().item() / scale <= 2e-5, (name, (a - b).abs().max().item() / scale)


def test_adapter_feeds_the_rasteriser_without_the_3x3_materialisation(gpu):
    """encoder tail -> rasteriser hand-off: 6-entry covariances straight into the multi-view rasteriser call give the same
    faces as the [.,3,3] route of the reference layouts."""
    from splatter360_amd import rasterizer
    v, h, w = 2, 32, 64
    ext, dep, opa, raw, _ = _random_case(gpu, v, h, w, 11, False)
    a9 = adapter.adapter_tail(ext, dep, opa, raw, (h, w), 0.5, 15.0)
    a6 = adapter.adapter_tail(ext, dep, opa, raw, (h, w), 0.5, 15.0, cov6=True)
    pose = torch.eye(4, device=gpu)
    e, K, n, f = decoder.cube_cameras(pose, 0.1, 10.0)
    bg = torch.zeros(3, device=gpu)
    views = decoder.pack_camera_views(e, K, n, f, bg)
    flat = lambda t, *s: t.reshape(-1, *s)
    ref = decoder.render_views_fused(e, K, n, f, (64, 64), bg, flat(a9.means, 3), flat(a9.covariances, 3, 3), flat(a9.harmonics, 3, 25),
                                     flat(a9.opacities), views=views, shared_campos=True)
    got, _ = rasterizer.rasterize_views(flat(a6.means, 3), flat(a6.covariances, 6), flat(a6.opacities), flat(a6.harmonics, 3, 25),
                                        views=views, image_height=64, image_width=64, sh_degree=4, shared_campos=True,
                                        sh_channel_major=True, want_radii=False)
    assert torch.equal(got, ref) and ref.abs().max().item() > 0
