"""Fused adapter-tail kernels (s360_adapter_forward / backward, s360_sh_rotation_blocks) against the golden capture of the
reference's GaussianAdapterERP (values and the reference module's own autograd gradients; rotate_sh = identity, see
tests/golden/make_golden_adapter.py), against the torch restatement in oracle/adapter_ref.py (values and autograd gradients,
detached and opt-in differentiable means, with and without SH rotation blocks, [.,3,3] and 6-entry covariance layouts) and,
for rotate_sh's matrices, against the oracle's float64 construction."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import adapter_ref
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


def test_fused_tail_gradients_match_the_reference_modules_autograd(gpu):
    """d_depths / d_raw_gaussians of the reference module itself (means detached: its un-projection runs under no_grad)."""
    g = np.load(G / "adapter_erp_tail.npz")
    t = lambda k: torch.tensor(g[k], device=gpu)
    h, w = (int(x) for x in g["image_shape"])
    mod = adapter.GaussianAdapterERP(float(g["scale_min"]), float(g["scale_max"]), 4, sh_rotation="identity").to(gpu)
    d = t("depths").requires_grad_(True)
    raw = t("raw_gaussians").requires_grad_(True)
    out = mod("hm3d", t("extrinsics")[:, :, None, None, None], d, t("opacities_in"), raw, (h, w))
    assert not out.means.requires_grad
    ((out.covariances * t("cot_covariances")).sum() + (out.harmonics * t("cot_harmonics")).sum()).backward()
    for got, want in ((d.grad, g["d_depths"]), (raw.grad, g["d_raw_gaussians"])):
        np.testing.assert_allclose(got.cpu().numpy().reshape(want.shape), want, rtol=1e-4, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("name", ["replica", "m3d", "residential", "CoffeeArea", "outdoor_colmap"])
def test_fused_tail_other_erp_conventions_match_reference_capture(gpu, name):
    g = np.load(G / "adapter_erp_tail.npz")
    t = lambda k: torch.tensor(g[k], device=gpu)
    h, w = (int(x) for x in g["image_shape"])
    mod = adapter.GaussianAdapterERP(float(g["scale_min"]), float(g["scale_max"]), 4, sh_rotation="identity", differentiable_means=True).to(gpu)
    d = t("depths").requires_grad_(True)
    out = mod(name, t("extrinsics")[:, :, None, None, None], d, t("opacities_in"), t("raw_gaussians"), (h, w))
    want = g["means_" + name]
    np.testing.assert_allclose(out.means.detach().cpu().numpy().reshape(want.shape), want, rtol=3e-6, atol=3e-6 * np.abs(want).max())
    # the opt-in mean gradient follows the same convention: d mean / d depth = C dir
    (out.means * t("cot_means")).sum().backward()
    b, v, r = g["depths"].shape[:3]
    dd = torch.tensor(g["depths"]).reshape(b * v, r).requires_grad_(True)
    ref = adapter_ref.adapter_tail_torch(torch.tensor(g["extrinsics"]).reshape(b * v, 4, 4), dd, torch.tensor(g["opacities_in"]).reshape(b * v, r),
                                         torch.tensor(g["raw_gaussians"]).reshape(b * v, r, -1), (h, w), 0.5, 15.0, differentiable_means=True,
                                         dataset_name=name)
    (ref.means * torch.tensor(g["cot_means"]).reshape(ref.means.shape)).sum().backward()
    np.testing.assert_allclose(d.grad.cpu().numpy().reshape(-1), dd.grad.numpy().reshape(-1), rtol=1e-4, atol=1e-5 * float(dd.grad.abs().max()))
    with pytest.raises(Exception):
        mod("re10k", t("extrinsics")[:, :, None, None, None], d, t("opacities_in"), t("raw_gaussians"), (h, w))


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


@pytest.mark.parametrize("diff_means", [False, True])
@pytest.mark.parametrize("with_rot", [False, True])
@pytest.mark.parametrize("cov6", [False, True])
def test_fused_tail_values_and_gradients_match_torch_autograd(gpu, with_rot, cov6, diff_means):
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
            out = adapter.adapter_tail(ext, d, opa, rw, (h, w), 0.5, 15.0, sh_rotation=rot, cov6=cov6, differentiable_means=diff_means)
            cov_term = (out.covariances * wc[:, :, r_, c_]).sum() if cov6 else (out.covariances * wc).sum()
        else:
            out = adapter_ref.adapter_tail_torch(ext, d, opa, rw, (h, w), 0.5, 15.0, sh_rotation=rot, differentiable_means=diff_means)
            # the 6-entry layout reads the upper triangle only (cuda_splatting.py:115,123)
            cov_term = (out.covariances[:, :, r_, c_] * wc[:, :, r_, c_]).sum() if cov6 else (out.covariances * wc).sum()
        assert out.means.requires_grad == diff_means
        ((out.means * wm).sum() * (1.0 if diff_means else 0.0) + cov_term + (out.harmonics * wh).sum()).backward()
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


def test_native_sh_rotation_blocks_match_the_float64_construction(gpu):
    """s360_sh_rotation_blocks (rotate_sh's matrices, sh_rotation.py:19-24) against oracle/adapter_ref.wigner_blocks — and, through
    the module's default sh_rotation="native", rotated harmonics against the torch restatement fed those float64 matrices."""
    from scipy.spatial.transform import Rotation
    R = Rotation.random(6, random_state=11).as_matrix().astype(np.float32)
    want = adapter_ref.wigner_blocks(R, 25)
    got = adapter.sh_rotation_blocks(torch.tensor(R, device=gpu), 25).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=3e-6)           # float32 rotation in, float64 inside, float32 out
    pose = np.tile(np.eye(4, dtype=np.float32), (6, 1, 1))
    pose[:, :3, :3] = R
    pose[:, :3, 3] = 1.5
    np.testing.assert_array_equal(adapter.sh_rotation_blocks(torch.tensor(pose, device=gpu), 25).cpu().numpy(), got)
    for d_sh in (1, 4, 9, 16):
        np.testing.assert_allclose(adapter.sh_rotation_blocks(torch.tensor(R, device=gpu), d_sh).cpu().numpy(), want[:, :d_sh, :d_sh], atol=3e-6)
    v, h, w = 6, 4, 8
    ext, dep, opa, raw, _ = _random_case(gpu, v, h, w, 21, False)
    ext[:, :3, :3] = torch.tensor(R, device=gpu)
    mod = adapter.GaussianAdapterERP(0.5, 15.0, 4).to(gpu)      # default: native rotate_sh
    out = mod("hm3d", ext[None, :, None, None, None], dep[None, :, :, None, None], opa[None, :, :, None, None],
              raw[None, :, :, None, None, :], (h, w))
    ref = adapter_ref.adapter_tail_torch(ext.cpu(), dep.cpu(), opa.cpu(), raw.cpu(), (h, w), 0.5, 15.0,
                                         sh_rotation=torch.tensor(want, dtype=torch.float32))
    np.testing.assert_allclose(out.harmonics.cpu().numpy().reshape(v, h * w, 3, 25), ref.harmonics.numpy(), atol=2e-6)


def test_native_sh_rotation_blocks_against_e3nn_where_it_exists(gpu):
    pytest.importorskip("e3nn")
    from scipy.spatial.transform import Rotation
    R = torch.tensor(Rotation.random(4, random_state=5).as_matrix(), dtype=torch.float32)
    want = adapter.wigner_blocks_e3nn(R, 25).numpy()
    np.testing.assert_allclose(adapter.sh_rotation_blocks(R.to(gpu), 25).cpu().numpy(), want, atol=1e-5)


def test_rotated_harmonics_against_the_reference_capture_when_it_exists(gpu):
    """tests/golden/adapter_erp_tail_rotated.npz is written by make_golden_adapter.py only where e3nn is importable (the
    reference's REAL rotate_sh, sh_rotation.py:10-30); no such environment existed in rounds 1-4, so the file is absent and this
    test skips.  Once committed it pins the native SH rotation end to end."""
    f = G / "adapter_erp_tail_rotated.npz"
    if not f.exists():
        pytest.skip("no rotated-harmonics capture (made only where e3nn is installed)")
    z, zr = np.load(G / "adapter_erp_tail.npz"), np.load(f)
    mod = adapter.GaussianAdapterERP(float(z["scale_min"]), float(z["scale_max"]), 4).to(gpu)
    t = lambda k: torch.tensor(z[k], device=gpu)
    h, w = (int(x) for x in z["image_shape"])
    out = mod("hm3d", t("extrinsics")[:, :, None, None, None], t("depths"), t("opacities_in"), t("raw_gaussians"), (h, w))
    np.testing.assert_allclose(out.harmonics.cpu().numpy(), zr["harmonics_rotated"], atol=5e-6)


def test_e3nn_selfcheck_is_a_noop_without_e3nn_and_raises_on_a_wrong_convention(gpu, monkeypatch):
    """adapter.selfcheck_sh_rotation_against_e3nn: None where e3nn is absent; with a stand-in for wigner_blocks_e3nn it passes on
    the right matrices and raises on transposed ones (the failure a wrong convention would produce)."""
    import sys
    import types
    adapter._E3NN_CHECKED.clear()
    if "e3nn" not in sys.modules:
        try:
            import e3nn  # noqa: F401
        except Exception:
            assert adapter.selfcheck_sh_rotation_against_e3nn(gpu, 25) is None
    adapter._E3NN_CHECKED.clear()
    fake = types.ModuleType("e3nn"); fake.o3 = types.ModuleType("e3nn.o3")
    monkeypatch.setitem(sys.modules, "e3nn", fake)
    monkeypatch.setitem(sys.modules, "e3nn.o3", fake.o3)
    good = lambda R, d: torch.tensor(adapter_ref.wigner_blocks(R.numpy(), d), dtype=torch.float32)
    monkeypatch.setattr(adapter, "wigner_blocks_e3nn", good)
    assert adapter.selfcheck_sh_rotation_against_e3nn(gpu, 25) is True
    adapter._E3NN_CHECKED.clear()
    monkeypatch.setattr(adapter, "wigner_blocks_e3nn", lambda R, d: good(R, d).transpose(1, 2).contiguous())
    with pytest.raises(RuntimeError, match="differ from e3nn"):
        adapter.selfcheck_sh_rotation_against_e3nn(gpu, 25)
    adapter._E3NN_CHECKED.clear()
