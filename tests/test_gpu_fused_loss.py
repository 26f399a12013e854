"""Loss epilogue fused into the composite store (s360_forward_mse) against the reference's formulation in
torch: LossMse = weight * mean((color - target)^2) (src/loss/loss_mse.py:30-31) and compute_psnr
(src/evaluation/metrics.py:11-21), values and gradients."""
import pytest
import torch

from splatter360_amd import decoder, rasterizer, synthetic

pytestmark = pytest.mark.gpu


def _setup(gpu, w=128, face=64, seed=1):
    cloud = synthetic.encoder_like_cloud(w // 2, w, seed=seed)
    ps = [torch.tensor(cloud[k], device=gpu).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
    cams = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    g = torch.Generator(device="cpu").manual_seed(seed)
    gt = (torch.rand((6, 3, face, face), generator=g) * 1.4 - 0.2).to(gpu)   # some values outside [0,1]: exercises the clip
    return ps, cams, gt


def _psnr_ref(gt, pred):       # metrics.py:11-21 restated
    gt = gt.clip(min=0, max=1)
    pred = pred.clip(min=0, max=1)
    mse = ((gt - pred) ** 2).mean(dim=(1, 2, 3))
    mse[mse == 0.0] = 1e-10
    return -10 * mse.log10()


@pytest.mark.parametrize("weight,with_depth", [(1.0, False), (0.37, True)])
def test_fused_mse_matches_torch_ops(gpu, weight, with_depth):
    ps, (ext, K, near, far), gt = _setup(gpu)
    bg = torch.tensor([0.1, 0.0, 0.3], device=gpu)
    faces = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps)
    loss = weight * ((faces - gt) ** 2).mean()
    loss.backward()
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    res = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, mse_target=gt, mse_weight=weight,
                                     depth_mode="depth" if with_depth else None)
    faces2, fm = res[0], res[-1]
    assert torch.equal(faces2, faces.detach())
    if with_depth:
        _, dep = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *[p.detach() for p in ps], depth_mode="depth")
        assert torch.equal(res[1], dep)
    assert abs(fm.loss.item() - loss.item()) <= 2e-6 * abs(loss.item())
    torch.testing.assert_close(fm.psnr(), _psnr_ref(gt, faces.detach()), rtol=1e-5, atol=1e-5)
    (fm.loss * 1.0).backward()
    for p, w in zip(ps, want):
        scale = w.abs().max().item() + 1e-20
        assert (p.grad - w).abs().max().item() / scale <= 2e-6


def test_fused_mse_scaled_and_combined_with_image_gradient(gpu):
    ps, (ext, K, near, far), gt = _setup(gpu, seed=2)
    bg = torch.zeros(3, device=gpu)
    extra = torch.randn(6, 3, 64, 64, generator=torch.Generator(device="cpu").manual_seed(7)).to(gpu) * 1e-4
    faces = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps)
    (2.5 * ((faces - gt) ** 2).mean() + (faces * extra).sum()).backward()
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    faces2, fm = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, mse_target=gt)
    (2.5 * fm.loss + (faces2 * extra).sum()).backward()
    for p, w in zip(ps, want):
        scale = w.abs().max().item() + 1e-20
        assert (p.grad - w).abs().max().item() / scale <= 2e-5   # g + seed is rounded once more than autograd's sum


def test_fused_mse_loss_scalar_applied_inside_the_backward(gpu):
    """A scaled loss alone (no image gradient): the scalar autograd hands back reaches the composite through
    dL_dimages_scale (s360.h) — bit-identical to multiplying the stored seed in a separate elementwise pass."""
    ps, (ext, K, near, far), gt = _setup(gpu, seed=3)
    bg = torch.zeros(3, device=gpu)
    faces, fm = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, mse_target=gt)
    seed = rasterizer.last_state().d_images.clone()
    (3.25 * fm.loss).backward()
    got = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    faces2 = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps)
    faces2.backward(seed * 3.25)
    for p, g in zip(ps, got):
        assert torch.equal(p.grad, g)


def test_fused_mse_ragged_image_and_no_grad(gpu):
    """40x24 faces (partial tiles): strips outside the image contribute nothing; forward-only call works."""
    cloud = synthetic.uniform_cloud(3000, seed=4)
    ps = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    gt = torch.rand(6, 3, 24, 40, device=gpu)
    with torch.no_grad():
        faces, fm = decoder.render_views_fused(ext, K, near, far, (24, 40), torch.zeros(3, device=gpu), *ps, mse_target=gt)
    want = ((faces - gt) ** 2).mean()
    assert abs(fm.loss.item() - want.item()) <= 2e-6 * want.item()
    torch.testing.assert_close(fm.psnr(), _psnr_ref(gt, faces), rtol=1e-5, atol=1e-5)


def test_deferred_loss_is_reduced_by_the_backward(gpu):
    """mse_defer=True (S360_FLAG_DEFER_LOSS): no reduction launch at the end of the forward; the backward's first launch writes
    the same loss / clipped MSE bit for bit, and the gradients do not change."""
    ps, (ext, K, near, far), gt = _setup(gpu, seed=3)
    bg = torch.tensor([0.0, 0.2, 0.1], device=gpu)
    faces, fm = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, mse_target=gt, mse_weight=0.8)
    fm.loss.backward()
    want_loss, want_clip = fm.loss.detach().clone(), fm.clipped_mse.clone()
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    faces2, fm2 = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, mse_target=gt, mse_weight=0.8, mse_defer=True)
    assert torch.equal(faces2, faces)
    # before the backward the deferred scalars read as NaN, never as uninitialised memory
    assert bool(torch.isnan(fm2.loss.detach()).all()) and bool(torch.isnan(fm2.clipped_mse).all())
    fm2.loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(fm2.loss.detach(), want_loss) and torch.equal(fm2.clipped_mse, want_clip)
    for p, w in zip(ps, want):
        assert torch.equal(p.grad, w)
    # an inference call ignores the flag: the loss is there without any backward
    with torch.no_grad():
        _, fm3 = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *[p.detach() for p in ps], mse_target=gt, mse_weight=0.8,
                                            mse_defer=True)
    assert torch.equal(fm3.loss, want_loss)
