"""The decoder class `splatter360_amd.install()` registers in the reference's registry, run on the GPU against a capture of the
reference's own decoder call (tests/golden/decoder_forward_call.npz, made by tests/golden/make_golden_decoder.py from
/root/reference/src/model/decoder/__init__.py:5-13 + decoder_splatting_cuda.py:19-97 with a recording rasteriser):
  * the 24 rasteriser invocations the reference made (12 colour, 12 depth) are replayed, with exactly the settings / tensors it
    handed over, through this repository's drop-in `diff_gaussian_rasterization` — that is the unchanged reference on this
    library, INTEGRATION.md section 1;
  * the fused decoder (one rasteriser call per panorama, colour + depth together) must reproduce those images from the call's
    raw inputs, and its gradients must equal the drop-in mirror's."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import diff_gaussian_rasterization as dgr
from splatter360_amd import decoder, plugin

pytestmark = pytest.mark.gpu
FIX = Path(__file__).resolve().parent / "golden" / "decoder_forward_call.npz"


def _replay(z, dev):
    n = int(z["n_calls"])
    imgs = []
    for i in range(n):
        t = lambda k: torch.tensor(z[f"c{i}_{k}" if f"c{i}_{k}" in z else f"c0_{k}"], device=dev)
        use_sh = bool(z[f"c{i}_use_sh"])
        st = dgr.GaussianRasterizationSettings(
            image_height=int(z[f"c{i}_image_height"]), image_width=int(z[f"c{i}_image_width"]), tanfovx=float(z[f"c{i}_tanfovx"]),
            tanfovy=float(z[f"c{i}_tanfovy"]), bg=t("bg"), scale_modifier=float(z[f"c{i}_scale_modifier"]), viewmatrix=t("viewmatrix"),
            projmatrix=t("projmatrix"), sh_degree=int(z[f"c{i}_sh_degree"]), campos=t("campos"), prefiltered=False, debug=False)
        m3 = t("means3D")
        img, _ = dgr.GaussianRasterizer(st)(means3D=m3, means2D=torch.zeros_like(m3), shs=t("shs") if use_sh else None,
                                            colors_precomp=None if use_sh else t("colors_precomp"), opacities=t("opacities"),
                                            cov3D_precomp=t("cov3D_precomp"))
        imgs.append(img)
    return imgs


def _inputs(z, dev, grad=False):
    g = SimpleNamespace(**{k: torch.tensor(z[k], device=dev).requires_grad_(grad) for k in ("means", "covariances", "harmonics", "opacities")})
    cams = [torch.tensor(z[k], device=dev) for k in ("extrinsics", "intrinsics", "near", "far")]
    return g, cams, tuple(int(x) for x in z["image_shape"])


def test_registered_decoder_reproduces_the_reference_call(gpu):
    z = np.load(FIX)
    imgs = _replay(z, gpu)
    n = len(imgs) // 2
    want_color = torch.stack(imgs[:n])[None]                       # [b, v, 3, h, w]
    want_depth = torch.stack([im.mean(0) for im in imgs[n:]])[None]   # render_depth_cuda: mean over the three equal channels
    cls = plugin.make_decoder_class(torch.nn.Module, decoder.DecoderOutput)
    dec = cls(SimpleNamespace(name="splatting_cuda"), SimpleNamespace(background_color=z["background_color"].tolist())).to(gpu)
    g, (ext, K, near, far), shape = _inputs(z, gpu)
    out = dec.forward(g, ext, K, near, far, shape, depth_mode="depth")
    assert isinstance(out, decoder.DecoderOutput) and out.color.shape == want_color.shape and out.depth.shape == want_depth.shape
    # camera records: the reference's CPU LU inverses (captured) vs one Gauss-Jordan kernel: a few ulp -> pixels to ~1e-6
    assert (out.color - want_color).abs().max().item() <= 5e-6
    assert ((out.depth - want_depth).abs() / (want_depth.abs() + 1e-3)).max().item() <= 1e-5
    assert float(want_color.std()) > 0.05 and float(want_depth.max()) > 0.1      # the capture is not a blank render
    assert torch.equal(dec.render_depth(g, ext, K, near, far, shape, mode="depth"), out.depth)


def test_registered_decoder_gradients_equal_the_dropin_mirror(gpu):
    z = np.load(FIX)
    bgc = z["background_color"].tolist()
    w_c = torch.randn(1, 12, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(gpu)
    w_d = torch.randn(1, 12, 32, 32, generator=torch.Generator().manual_seed(4)).to(gpu) * 0.1
    grads = []
    for which in ("fused", "mirror"):
        g, (ext, K, near, far), shape = _inputs(z, gpu, grad=True)
        if which == "fused":
            dec = plugin.make_decoder_class(torch.nn.Module, decoder.DecoderOutput, glue="torch")(
                SimpleNamespace(name="splatting_cuda"), SimpleNamespace(background_color=bgc)).to(gpu)
        else:
            dec = decoder.DecoderSplattingCUDA(background_color=bgc).to(gpu)      # per-face drop-in calls, like the reference's loop
        out = dec.forward(g, ext, K, near, far, shape, depth_mode="depth")
        ((out.color * w_c).sum() + (out.depth * w_d).sum()).backward()
        grads.append((out.color.detach(), out.depth.detach(), [g.means.grad, g.covariances.grad, g.harmonics.grad, g.opacities.grad]))
    (c0, d0, g0), (c1, d1, g1) = grads
    assert (c0 - c1).abs().max().item() <= 2e-6 and ((d0 - d1).abs() / (d1.abs() + 1e-3)).max().item() <= 1e-5
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-12)
