"""The literal drop-in: `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer` — the
module name the reference imports (/root/reference/src/model/decoder/cuda_splatting.py:5-8) — driven with the reference's own
call pattern (cuda_splatting.py:91-124: a zero `means2D` with retain_grad(), keyword settings with python-float tanfov,
keyword call, `opacities[..., None]`, `cov3D_precomp` gathered with triu_indices, radii discarded) on the arguments the
reference's render_cuda really handed its rasteriser for the six faces of a panorama (tests/golden/boundary_render_cuda.npz,
captured from the reference's Python with a recording fake extension).  Images and gradients against the CPU oracle."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("face", range(6))
def test_reference_call_pattern_through_the_dropin_module(gpu, face):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer   # the reference's import line
    cap = np.load(G / "boundary_render_cuda.npz")
    c = lambda k: cap[f"f{face}_{k}"]
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu)
    h, w = int(c("image_height")), int(c("image_width"))
    gaussian_means = t(c("means3D")).requires_grad_(True)
    shs = t(c("shs")).requires_grad_(True)
    gaussian_opacities = t(c("opacities")[:, 0]).requires_grad_(True)
    cov_full = np.zeros((48, 3, 3), np.float32)
    r, cc = np.triu_indices(3)
    cov_full[:, r, cc] = c("cov3D_precomp")
    gaussian_covariances = t(cov_full).requires_grad_(True)

    # ---- cuda_splatting.py:91-124, per batch element ----
    mean_gradients = torch.zeros_like(gaussian_means, requires_grad=True)
    try:
        mean_gradients.retain_grad()
    except Exception:
        pass
    settings = GaussianRasterizationSettings(
        image_height=h,
        image_width=w,
        tanfovx=float(c("tanfovx")),
        tanfovy=float(c("tanfovy")),
        bg=t(c("bg")),
        scale_modifier=1.0,
        viewmatrix=t(c("viewmatrix")),
        projmatrix=t(c("projmatrix")),
        sh_degree=int(c("sh_degree")),
        campos=t(c("campos")),
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(settings)
    row, col = torch.triu_indices(3, 3)
    image, radii = rasterizer(
        means3D=gaussian_means,
        means2D=mean_gradients,
        shs=shs,
        colors_precomp=None,
        opacities=gaussian_opacities[..., None],
        cov3D_precomp=gaussian_covariances[:, row, col],
    )
    assert image.shape == (3, h, w) and radii.shape == (48,) and radii.dtype == torch.int32
    gimg = np.random.default_rng(face).standard_normal((3, h, w)).astype(np.float32)
    image.backward(t(gimg))

    S = dict(image_height=h, image_width=w, tanfovx=float(c("tanfovx")), tanfovy=float(c("tanfovy")), bg=c("bg"),
             viewmatrix=c("viewmatrix"), projmatrix=c("projmatrix"), sh_degree=int(c("sh_degree")), campos=c("campos"))
    o = oracle.rasterize(S, means3D=c("means3D"), cov3D_precomp=c("cov3D_precomp"), opacities=c("opacities"), shs=c("shs"))
    f = o.forward()
    g = o.backward(gimg)
    np.testing.assert_array_equal(radii.cpu().numpy(), f["radii"])
    per_px = np.abs(image.detach().cpu().numpy().astype(np.float64) - f["image"]).mean(0)
    assert per_px.max() <= 1e-5, per_px.max()
    got = dict(means3D=gaussian_means.grad, means2D=mean_gradients.grad, shs=shs.grad, opacities=gaussian_opacities.grad,
               cov3D=gaussian_covariances.grad[:, row.to(gpu), col.to(gpu)])
    for k, v in got.items():
        assert v is not None, k
        want = np.asarray(g[k], np.float64).reshape(-1)
        err = np.abs(v.cpu().numpy().astype(np.float64).reshape(-1) - want).max() / (np.abs(want).max() + 1e-12)
        assert err <= 2e-4, (k, err)
    # the covariance gather's adjoint leaves the lower triangle without gradient, exactly like autograd through [:, row, col]
    assert float(gaussian_covariances.grad[:, 1, 0].abs().max()) == 0.0
