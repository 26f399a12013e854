"""s360_forward_raw / s360_backward_raw (the Gaussian-adapter tail FUSED into the rasteriser's first and last kernels)
against the two-step path it replaces — adapter.adapter_tail (s360_adapter_forward / backward: values and gradients golden-pinned
against the reference's GaussianAdapterERP, tests/test_gpu_adapter.py) followed by the fused six-face render (oracle-checked):

  * means and 6-entry covariances: the adapter's own expressions -> bit-identical to adapter_tail's;
  * images / depth maps / fused loss: round 6 rotates the coefficients exactly like the adapter kernel (same helper, same order) and
    evaluates colours with k_sh_eval3_jac's own code -> bit-identical (round 5 carried the basis through the transform: 1e-6);
  * gradients w.r.t. depths, opacities and raw_gaussians (scale logits, quaternion, all 75 SH coefficients): the rank-1 form
    (mask . D^T Y) (x) dL/dRGB instead of the [G,3,25] dL/dSH round trip;
with and without the per-view SH rotation (rotate_sh, /root/reference/src/misc/sh_rotation.py:10-30), detached (the reference's) and
differentiable means, colour + depth + fused L2 loss, ragged sizes (a workgroup straddling two context views).
Reference path replaced: gaussian_adapter_erp.py:63-119 -> decoder_splatting_cuda.py:47-59 -> cuda_splatting.py:99-124."""
import numpy as np
import pytest
import torch

from splatter360_amd import adapter, decoder, rasterizer

pytestmark = pytest.mark.gpu


def _inputs(dev, nv, h, w, seed, per_ray=1):
    g = torch.Generator().manual_seed(seed)
    n = h * w * per_ray
    depths = torch.exp(torch.empty(nv, n).uniform_(np.log(0.8), np.log(6.0), generator=g))
    opac = torch.sigmoid(torch.randn(nv, n, generator=g))
    raw = torch.randn(nv, n, 82, generator=g)
    raw[..., 7:] *= 0.8
    ext = torch.eye(4).repeat(nv, 1, 1)
    from scipy.spatial.transform import Rotation
    ext[:, :3, :3] = torch.tensor(Rotation.random(nv, random_state=seed).as_matrix(), dtype=torch.float32)
    ext[:, :3, 3] = torch.tensor([[-0.4, 0.0, 0.1], [0.4, 0.05, -0.1], [0.0, 0.3, 0.4]][:nv])
    return [t.to(dev) for t in (depths, opac, raw, ext)]


def _two_step(depths, opac, raw, ext, rot, cams, fw, hw, diff_means, depth_mode, target):
    d, o, r = (t.clone().requires_grad_(True) for t in (depths, opac, raw))
    nv = ext.shape[0]
    g = adapter.adapter_tail(ext, d, o, r, hw, 0.5, 15.0, sh_rotation=rot, differentiable_means=diff_means)
    e, K, near, far = cams
    out = decoder.render_views_fused(e, K, near, far, (fw, fw), torch.zeros(3, device=d.device), g.means.reshape(-1, 3), g.covariances.reshape(-1, 3, 3),
                                     g.harmonics.reshape(-1, 3, 25), g.opacities.reshape(-1), shared_campos=True, depth_mode=depth_mode, mse_target=target)
    return out, (d, o, r), g


def _raw(depths, opac, raw, ext, rot, cams, fw, hw, diff_means, depth_mode, target):
    d, o, r = (t.clone().requires_grad_(True) for t in (depths, opac, raw))
    e, K, near, far = cams
    views = decoder.pack_camera_views(e, K, near, far, torch.zeros(3, device=d.device))
    out = rasterizer.rasterize_raw(d.reshape(-1), o.reshape(-1), r.reshape(-1, 82), ext, views=views, image_height=fw, image_width=fw, context_shape=hw,
                                   scale_min=0.5, scale_max=15.0, sh_rotation=rot, differentiable_means=diff_means, depth_mode=depth_mode,
                                   mse_target=target)
    return out, (d, o, r)


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("rotate,diff_means,hw,nv", [(True, False, (32, 64), 2), (False, False, (32, 64), 2), (True, True, (24, 40), 3), (True, False, (9, 14), 3)])
def test_raw_entry_equals_adapter_then_render(gpu, rotate, diff_means, hw, nv):
    fw = 64
    depths, opac, raw, ext = _inputs(gpu, nv, hw[0], hw[1], seed=5 + nv)
    rot = adapter.sh_rotation_blocks(ext, 25) if rotate else None
    cams = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    gen = torch.Generator().manual_seed(1)
    target = torch.rand((6, 3, fw, fw), generator=gen).to(gpu)
    wd = torch.randn((6, fw, fw), generator=gen).to(gpu)
    (img_a, dep_a, fm_a), ins_a, g = _two_step(depths, opac, raw, ext, rot, cams, fw, hw, diff_means, "depth", target)
    (img_b, means_b, cov_b, dep_b, fm_b), ins_b = _raw(depths, opac, raw, ext, rot, cams, fw, hw, diff_means, "depth", target)
    # geometry: the adapter's own values
    assert torch.equal(means_b, g.means.reshape(-1, 3))
    r_, c_ = torch.triu_indices(3, 3)
    assert torch.equal(cov_b, g.covariances.reshape(-1, 3, 3)[:, r_, c_])
    # round 6: the raw kernel rotates the COEFFICIENTS with the adapter kernel's own expression (sh_rotate_coefs25) and evaluates
    # colours / jacobian with k_sh_eval3_jac's code on them -> the same bits all the way to the pixels
    assert torch.equal(img_a.detach(), img_b.detach())
    assert torch.equal(dep_a.detach(), dep_b.detach())
    assert float(fm_a.loss.detach()) == float(fm_b.loss.detach())
    (fm_a.loss + 0.01 * (dep_a * wd).mean()).backward()
    (fm_b.loss + 0.01 * (dep_b * wd).mean()).backward()
    for name, a, b in zip(("depths", "opacities", "raw"), ins_a, ins_b):
        assert a.grad is not None and b.grad is not None and bool(torch.isfinite(b.grad).all()), name
        assert _rel(b.grad, a.grad) <= 2e-4, (name, _rel(b.grad, a.grad))
    ra, rb = ins_a[2].grad, ins_b[2].grad
    for lo, hi, what in ((0, 3, "scale logits"), (3, 7, "quaternion"), (7, 82, "harmonics")):
        assert _rel(rb[..., lo:hi].reshape(-1), ra[..., lo:hi].reshape(-1)) <= 2e-4, what
    assert float(ra[..., 7:].abs().max()) > 0 and float(ins_a[0].grad.abs().max()) > 0


def test_raw_entry_inference_call_and_plain_images(gpu):
    """No gradient required: the inference form (no backward state), plain call without loss / depth."""
    hw, nv, fw = (32, 64), 2, 48
    depths, opac, raw, ext = _inputs(gpu, nv, *hw, seed=3)
    rot = adapter.sh_rotation_blocks(ext, 25)
    cams = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    with torch.no_grad():
        img_a = _two_step(depths, opac, raw, ext, rot, cams, fw, hw, False, None, None)[0]
        views = decoder.pack_camera_views(*cams, torch.zeros(3, device=gpu))
        img_b, _, _ = rasterizer.rasterize_raw(depths.reshape(-1), opac.reshape(-1), raw.reshape(-1, 82), ext, views=views, image_height=fw, image_width=fw,
                                               context_shape=hw, scale_min=0.5, scale_max=15.0, sh_rotation=rot)
    assert torch.equal(img_a, img_b)
    with pytest.raises(RuntimeError):
        rasterizer.rasterize_raw(depths.reshape(-1), opac.reshape(-1), raw.reshape(-1, 82)[:, :40], ext, views=views, image_height=fw, image_width=fw,
                                 context_shape=hw, scale_min=0.5, scale_max=15.0)


_MFMA_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import test_gpu_raw_entry as T
from splatter360_amd import adapter, decoder
dev = torch.device("cuda:0")
hw, nv, fw = (32, 64), 2, 64
depths, opac, raw, ext = T._inputs(dev, nv, hw[0], hw[1], seed=11)
rot = adapter.sh_rotation_blocks(ext, 25)
cams = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
target = torch.rand((6, 3, fw, fw), generator=torch.Generator().manual_seed(2)).to(dev)
(img, means, cov, dep, fm), ins = T._raw(depths, opac, raw, ext, rot, cams, fw, hw, False, "depth", target)
fm.loss.backward()
np.savez(sys.argv[2], img=img.detach().cpu().numpy(), loss=fm.loss.detach().cpu().numpy(), **{"g%d" % i: t.grad.cpu().numpy() for i, t in enumerate(ins)})
"""


def test_raw_eval_mfma_variant_is_bit_identical(gpu, tmp_path):
    """S360_RAW_MFMA=1 (k_raw_eval<.., MFMA>: the per-view coefficient rotation on v_mfma_f32_16x16x4_f32, DESIGN.md section 4 "MFMA")
    against the default SGPR-operand form: the switch is read once per process, so each variant renders the same seeded raw cloud in
    its own process; images, loss and every raw gradient must be the same bits."""
    import os, subprocess, sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    outs = []
    for flag in ("0", "1"):
        out = tmp_path / f"raw_mfma_{flag}.npz"
        env = dict(os.environ, S360_RAW_MFMA=flag)
        r = subprocess.run([sys.executable, "-c", _MFMA_CHILD, root, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["img"]).all() and float(np.abs(a["g2"]).max()) > 0.0
