"""CPU side of the adapter tail: the torch restatement (oracle/adapter_ref.py) against the golden capture of the reference
module — values AND the reference's own autograd gradients (tests/golden/adapter_erp_tail.npz, made by
tests/golden/make_golden_adapter.py with rotate_sh = identity) — the numpy construction of rotate_sh's matrices against
their defining properties, and the product module's refusal of CPU tensors."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import adapter_ref
from splatter360_amd import adapter

G = Path(__file__).resolve().parent / "golden"


def _load():
    g = np.load(G / "adapter_erp_tail.npz")
    t = lambda k: torch.tensor(g[k])
    b, v, r = g["depths"].shape[:3]
    return g, t, (b, v, r)


def test_torch_restatement_matches_reference_capture():
    g, t, (b, v, r) = _load()
    h, w = (int(x) for x in g["image_shape"])
    out = adapter_ref.adapter_tail_torch(t("extrinsics").reshape(b * v, 4, 4), t("depths").reshape(b * v, r), t("opacities_in").reshape(b * v, r),
                                         t("raw_gaussians").reshape(b * v, r, -1), (h, w), float(g["scale_min"]), float(g["scale_max"]))
    np.testing.assert_array_equal(adapter_ref.sh_mask(25).numpy(), g["sh_mask"])
    np.testing.assert_array_equal(adapter.sh_mask(25).numpy(), g["sh_mask"])
    for name, want in (("means", g["means"]), ("covariances", g["covariances"]), ("harmonics", g["harmonics_unrotated"]),
                       ("scales", g["scales"]), ("rotations", g["rotations"]), ("opacities", g["opacities"])):
        got = getattr(out, name).numpy().reshape(want.shape)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6 * np.abs(want).max(), err_msg=name)


def test_torch_restatement_gradients_match_the_reference_modules_autograd():
    """The reference un-projects under torch.no_grad() (sphere_projection.py:14-86): its means are detached
    (`means_require_grad` in the capture is False) and depth receives gradient through the scales only."""
    g, t, (b, v, r) = _load()
    assert not bool(g["means_require_grad"])
    h, w = (int(x) for x in g["image_shape"])
    d = t("depths").reshape(b * v, r).requires_grad_(True)
    raw = t("raw_gaussians").reshape(b * v, r, -1).requires_grad_(True)
    out = adapter_ref.adapter_tail_torch(t("extrinsics").reshape(b * v, 4, 4), d, t("opacities_in").reshape(b * v, r), raw, (h, w),
                                         float(g["scale_min"]), float(g["scale_max"]))
    assert not out.means.requires_grad
    ((out.covariances * t("cot_covariances").reshape(out.covariances.shape)).sum()
     + (out.harmonics * t("cot_harmonics").reshape(out.harmonics.shape)).sum()).backward()
    for got, want in ((d.grad, g["d_depths"]), (raw.grad, g["d_raw_gaussians"])):
        np.testing.assert_allclose(got.numpy().reshape(want.shape), want, rtol=1e-4, atol=2e-5 * np.abs(want).max())
    # the opt-in deviation really differs: with differentiable means the un-projection adds its own depth term
    d2 = t("depths").reshape(b * v, r).requires_grad_(True)
    o2 = adapter_ref.adapter_tail_torch(t("extrinsics").reshape(b * v, 4, 4), d2, t("opacities_in").reshape(b * v, r), raw.detach(), (h, w),
                                        float(g["scale_min"]), float(g["scale_max"]), differentiable_means=True)
    (o2.means * t("cot_means").reshape(o2.means.shape)).sum().backward()
    assert d2.grad.abs().max() > 0


def test_every_erp_convention_of_the_reference_matches_its_capture():
    """utils360.py knows five ray conventions (hm3d / replica, m3d, residential, CoffeeArea / outdoor_colmap); the capture holds
    the reference module's means for each."""
    g, t, (b, v, r) = _load()
    h, w = (int(x) for x in g["image_shape"])
    for name in ("hm3d", "replica", "m3d", "residential", "CoffeeArea", "outdoor_colmap"):
        out = adapter_ref.adapter_tail_torch(t("extrinsics").reshape(b * v, 4, 4), t("depths").reshape(b * v, r), t("opacities_in").reshape(b * v, r),
                                             t("raw_gaussians").reshape(b * v, r, -1), (h, w), float(g["scale_min"]), float(g["scale_max"]),
                                             dataset_name=name)
        want = g["means"] if name == "hm3d" else g["means_" + name]
        np.testing.assert_allclose(out.means.numpy().reshape(want.shape), want, rtol=2e-6, atol=2e-6 * np.abs(want).max(), err_msg=name)
    assert adapter.ERP_CONVENTIONS == {"hm3d": 0, "replica": 0, "m3d": 1, "residential": 2, "CoffeeArea": 3, "outdoor_colmap": 3}


def test_product_module_has_the_reference_signature_and_refuses_cpu_tensors():
    g, t, (b, v, r) = _load()
    h, w = (int(x) for x in g["image_shape"])
    mod = adapter.GaussianAdapterERP(float(g["scale_min"]), float(g["scale_max"]), 4, sh_rotation="identity")
    assert mod.d_sh == 25 and mod.d_in == 82
    with pytest.raises(RuntimeError, match="GPU only"):
        mod("hm3d", t("extrinsics")[:, :, None, None, None], t("depths"), t("opacities_in"), t("raw_gaussians"), (h, w))
    with pytest.raises(RuntimeError, match="GPU only"):
        adapter.adapter_tail(t("extrinsics").reshape(b * v, 4, 4), t("depths").reshape(b * v, r), t("opacities_in").reshape(b * v, r),
                             t("raw_gaussians").reshape(b * v, r, -1), (h, w), 0.5, 15.0)
    with pytest.raises(RuntimeError, match="GPU only"):
        adapter.sh_rotation_blocks(torch.eye(3)[None], 25)


def test_block_rotation_equals_dense_block_diagonal_product():
    rng = np.random.default_rng(0)
    sh = torch.tensor(rng.standard_normal((5, 3, 25)), dtype=torch.float32)
    rot = torch.zeros(5, 25, 25)
    for l in range(5):
        s = slice(l * l, (l + 1) ** 2)
        rot[:, s, s] = torch.tensor(rng.standard_normal((5, 2 * l + 1, 2 * l + 1)), dtype=torch.float32)
    got = adapter_ref.rotate_sh_blocks(sh, rot[:, None])
    want = torch.einsum("vij,vcj->vci", rot, sh)
    assert torch.allclose(got, want, atol=1e-5)


def _rotations(n, seed):
    from scipy.spatial.transform import Rotation
    return Rotation.random(n, random_state=seed).as_matrix()


def test_rotate_sh_matrices_have_their_defining_properties():
    """rotate_sh (sh_rotation.py:10-30) = e3nn's wigner_D per degree.  e3nn is absent: the construction is pinned by what
    defines it — D^1 = R in e3nn's (x, y, z) = m (-1, 0, 1) basis, orthogonality, the group law, and 'rotated coefficients
    evaluated at d equal the unrotated ones at R^-1 d' — not against e3nn itself (unpinned, said in oracle/adapter_ref.py)."""
    R = _rotations(4, 3)
    D = adapter_ref.wigner_blocks(R, 25)
    rng = np.random.default_rng(1)
    d = rng.standard_normal((40, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for v in range(4):
        np.testing.assert_allclose(D[v, 0, 0], 1.0, atol=1e-12)
        np.testing.assert_allclose(D[v, 1:4, 1:4], R[v], atol=1e-12)
        np.testing.assert_allclose(D[v] @ D[v].T, np.eye(25), atol=1e-10)
        for l in range(5):
            s = slice(l * l, (l + 1) ** 2)
            c = rng.standard_normal(2 * l + 1)
            lhs = adapter_ref.e3nn_real_sh(l, d) @ (D[v, s, s] @ c)            # rotated coefficients at d
            rhs = adapter_ref.e3nn_real_sh(l, d @ R[v]) @ c                    # unrotated coefficients at R^-1 d  (d @ R = (R^T d^T)^T)
            np.testing.assert_allclose(lhs, rhs, atol=1e-10)
    np.testing.assert_allclose(adapter_ref.wigner_blocks((R[0] @ R[1])[None], 25)[0], D[0] @ D[1], atol=1e-10)
    np.testing.assert_allclose(adapter_ref.wigner_blocks(np.eye(3)[None], 25)[0], np.eye(25), atol=1e-12)
    # a rotation about the polar axis (e3nn's y) by a mixes only the +-m pairs with cos / sin(m a)
    a = 0.7
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Dy = adapter_ref.wigner_blocks(Ry[None], 25)[0]
    for l in range(1, 5):
        blk = Dy[l * l:(l + 1) ** 2, l * l:(l + 1) ** 2]
        np.testing.assert_allclose(blk[l, l], 1.0, atol=1e-12)
        for m in range(1, l + 1):
            np.testing.assert_allclose(abs(blk[l + m, l + m]), abs(np.cos(m * a)), atol=1e-10)
            np.testing.assert_allclose(abs(blk[l + m, l - m]), abs(np.sin(m * a)), atol=1e-10)


def test_e3nn_basis_is_the_documented_one_for_low_degrees():
    """e3nn's generated formulas for l <= 2 (and three l = 3 members) as recalled from its source: x, y, z | sqrt3 xz, sqrt3 xy,
    y^2 - (x^2+z^2)/2, sqrt3 yz, sqrt3/2 (z^2 - x^2) | sqrt(30)/6 (Y2_0 z + Y2_4 x), sqrt5 Y2_0 y, y (2y^2 - 3(x^2+z^2))/2."""
    rng = np.random.default_rng(2)
    d = rng.standard_normal((30, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d.T
    np.testing.assert_allclose(adapter_ref.e3nn_real_sh(1, d), d, atol=1e-14)
    y2 = np.stack([np.sqrt(3) * x * z, np.sqrt(3) * x * y, y * y - 0.5 * (x * x + z * z), np.sqrt(3) * y * z, np.sqrt(3) / 2 * (z * z - x * x)], 1)
    np.testing.assert_allclose(adapter_ref.e3nn_real_sh(2, d), y2, atol=1e-14)
    y3 = adapter_ref.e3nn_real_sh(3, d)
    np.testing.assert_allclose(y3[:, 0], np.sqrt(30) / 6 * (y2[:, 0] * z + y2[:, 4] * x), atol=1e-14)
    np.testing.assert_allclose(y3[:, 1], np.sqrt(5) * y2[:, 0] * y, atol=1e-14)
    np.testing.assert_allclose(y3[:, 3], 0.5 * y * (2 * y * y - 3 * (x * x + z * z)), atol=1e-14)


def test_e3nn_degree4_members_follow_the_recalled_recursion_up_to_one_factor():
    """e3nn generates degree l from degree l-1 (its `_spherical_harmonics` source, as recalled):
    sh_4_0 = 3/4 sqrt2 (sh_3_0 z + sh_3_6 x), sh_4_1 = 3/4 sh_3_0 y + 3/8 sqrt6 (sh_3_1 z + sh_3_5 x),
    sh_4_4 = -3/28 sqrt42 (sh_3_2 x + sh_3_4 z) + 3/7 sqrt7 sh_3_3 y, sh_4_8 = 3/4 sqrt2 (sh_3_6 z - sh_3_0 x).
    The closed form used here must give the same four functions up to ONE common (per-degree) normalisation factor — order and
    signs of the degree-4 members are what the rotation matrices depend on."""
    rng = np.random.default_rng(4)
    d = rng.standard_normal((25, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d.T
    s3 = adapter_ref.e3nn_real_sh(3, d).T
    y4 = adapter_ref.e3nn_real_sh(4, d)
    rec = {0: 0.75 * np.sqrt(2) * (s3[0] * z + s3[6] * x),
           1: 0.75 * s3[0] * y + 0.375 * np.sqrt(6) * (s3[1] * z + s3[5] * x),
           4: -3 / 28 * np.sqrt(42) * (s3[2] * x + s3[4] * z) + 3 / 7 * np.sqrt(7) * s3[3] * y,
           8: 0.75 * np.sqrt(2) * (s3[6] * z - s3[0] * x)}
    factor = np.sqrt(7.0) / 3.0          # e3nn's degree-3 -> degree-4 step carries sqrt(9/7) in its own normalisation
    for m, f in rec.items():
        np.testing.assert_allclose(f * factor, y4[:, m], atol=1e-13, err_msg=str(m))
