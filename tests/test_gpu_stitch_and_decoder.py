"""GPU tests of the cube->ERP stitch kernel, the decoder-level drop-in API and the depth path."""
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import boundary_tensors, face_settings
from oracle import oracle
from splatter360_amd import cameras, decoder, stitch, synthetic

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("size", [(32, 64, 128), (64, 128, 256)])
def test_cube2equirec_matches_reference_golden(gpu, size):
    fw, eh, ew = size
    g = np.load(G / f"cube2equirec_{fw}_{eh}_{ew}.npz")
    mod = stitch.Cube2Equirec(fw, eh, ew).to(gpu)
    erp = mod(torch.tensor(g["cube"], device=gpu)[None])[0].cpu().numpy()
    # the reference's own output (torch CPU grid_sample): same taps and weights, different summation order
    np.testing.assert_allclose(erp, g["erp"], rtol=0, atol=2e-6)


def test_stitch_rendered_equals_change_order_then_cube2equirec(gpu):
    fw, eh, ew = 32, 64, 128
    mod = stitch.Cube2Equirec(fw, eh, ew).to(gpu)
    faces = torch.randn(6, 3, fw, fw, device=gpu)
    got = mod.stitch_rendered(faces)
    # reference pipeline (model_wrapper_erp.py:393-400) with torch ops: flip faces 0 and 5, permute, concat along width
    c = faces.clone()
    c[0] = torch.flip(c[0], dims=[-1, -2])
    c[5] = torch.flip(c[5], dims=[-1, -2])
    c = c[[3, 4, 1, 2, 0, 5]]
    cube = torch.cat(list(c), dim=-1)[None]  # [1,3,fw,6fw]
    want = mod(cube)[0]
    assert torch.equal(got, want)
    vol = torch.stack(list(c), 1)[None].cpu()
    ref = torch.nn.functional.grid_sample(vol, mod.sample_grid.cpu(), padding_mode="border", align_corners=True)[0, :, 0]
    assert (got.cpu() - ref).abs().max() <= 2e-6


def test_stitch_backward_is_the_adjoint(gpu):
    fw, eh, ew = 32, 64, 128
    mod = stitch.Cube2Equirec(fw, eh, ew).to(gpu)
    faces = torch.randn(6, 3, fw, fw, device=gpu, requires_grad=True)
    w = torch.randn(3, eh, ew, device=gpu)
    (mod.stitch_rendered(faces) * w).sum().backward()
    u = torch.randn(6, 3, fw, fw, device=gpu)
    lhs = (mod.stitch_rendered(u) * w).sum()          # <A u, w>
    rhs = (u * faces.grad).sum()                      # <u, A^T w>
    assert abs(lhs.item() - rhs.item()) <= 1e-3 * (abs(lhs.item()) + 1.0)


def test_render_depth_cuda_vs_oracle(gpu):
    cloud = synthetic.uniform_cloud(6000, seed=5, extent=3.0, scale_range=(0.02, 0.3))
    fw, face = 64, 2
    ext = cameras.cube_face_extrinsics(torch.eye(4)[None])[0, face][None].to(gpu)
    k = cameras.cube_face_intrinsics(1)[0, face][None].to(gpu)
    near, far = torch.tensor([0.1], device=gpu), torch.tensor([10.0], device=gpu)
    t = lambda key: torch.tensor(cloud[key], device=gpu)[None]
    depth = decoder.render_depth_cuda(ext, k, near, far, (fw, fw), t("means"), t("covariances"), t("opacities"))[0].cpu().numpy()
    S = face_settings(face, fw, fw)
    means, cov6, _, opac = boundary_tensors(cloud, S["scale"])
    z = decoder._depth_colors(ext.cpu(), t("means").cpu(), near.cpu(), far.cpu(), "depth")[0].numpy()
    colors = np.repeat(z[:, None], 3, 1)
    S["bg"] = np.zeros(3, np.float32)
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, colors_precomp=colors).forward()
    assert np.abs(depth - f["image"].mean(0)).mean() <= 1e-5 * max(1.0, float(np.abs(f["image"]).max()))


def test_decoder_module_forward_shapes_and_values(gpu):
    from types import SimpleNamespace
    cloud = synthetic.uniform_cloud(3000, seed=6, extent=3.0, scale_range=(0.02, 0.3))
    fw = 32
    gs = SimpleNamespace(**{k: torch.tensor(v, device=gpu)[None] for k, v in cloud.items()})
    ext = cameras.cube_face_extrinsics(torch.eye(4)[None]).to(gpu)          # [1,6,4,4]
    k = cameras.cube_face_intrinsics(1).to(gpu)
    near = torch.full((1, 6), 0.1, device=gpu)
    far = torch.full((1, 6), 10.0, device=gpu)
    dec = decoder.DecoderSplattingCUDA((0.0, 0.0, 0.0)).to(gpu)
    out = dec(gs, ext, k, near, far, (fw, fw), depth_mode="depth")
    assert out.color.shape == (1, 6, 3, fw, fw) and out.depth.shape == (1, 6, fw, fw)
    fused = decoder.render_views_fused(ext[0], k[0], near[0], far[0], (fw, fw), torch.zeros(3, device=gpu), gs.means[0],
                                       gs.covariances[0], gs.harmonics[0], gs.opacities[0], glue="torch")
    assert torch.equal(out.color[0], fused)


def test_scales_rotations_input_form(gpu):
    """Upstream also accepts (scales, rotations) instead of cov3D_precomp."""
    from test_gpu_parity import _settings_to_torch
    from helpers import small_front_scene
    from splatter360_amd import rasterizer
    S, means, cov6, shs, opac = small_front_scene(n=20, seed=3, h=32, w=32)
    rng = np.random.default_rng(0)
    scales = rng.uniform(0.05, 0.3, (20, 3)).astype(np.float32)
    q = rng.standard_normal((20, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu)
    rast = rasterizer.GaussianRasterizer(_settings_to_torch(S, gpu))
    img_a, _ = rast(means3D=t(means), opacities=t(opac), shs=t(shs), scales=t(scales), rotations=t(q))
    cov = rasterizer._cov6_from_scale_rotation(t(scales), t(q), 1.0)
    img_b, _ = rast(means3D=t(means), opacities=t(opac), shs=t(shs), cov3D_precomp=cov)
    assert torch.equal(img_a, img_b)
    with pytest.raises(Exception):
        rast(means3D=t(means), opacities=t(opac), shs=t(shs))
    with pytest.raises(Exception):
        rast(means3D=t(means), opacities=t(opac), shs=t(shs), colors_precomp=t(means), cov3D_precomp=cov)


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity", "log"])
def test_fused_depth_equals_second_pass_depth_render(gpu, mode):
    """Colour + depth from ONE pass (s360_forward_depth) vs the reference-style second rasterisation
    (render_depth_cuda) for every DepthRenderingMode, on the six faces of a panorama."""
    from types import SimpleNamespace
    cloud = synthetic.uniform_cloud(8000, seed=21, extent=3.0, scale_range=(0.02, 0.3))
    fw = 64
    gs = SimpleNamespace(**{k: torch.tensor(v, device=gpu)[None] for k, v in cloud.items()})
    ext = cameras.cube_face_extrinsics(torch.tensor(synthetic.target_pano_pose((0.1, 0.0, -0.2)))[None]).to(gpu)
    k = cameras.cube_face_intrinsics(1).to(gpu)
    near = torch.full((1, 6), 0.1, device=gpu)
    far = torch.full((1, 6), 10.0, device=gpu)
    ref = decoder.DecoderSplattingCUDA().to(gpu)(gs, ext, k, near, far, (fw, fw), depth_mode=mode)
    fused = decoder.DecoderSplattingFused(glue="torch").to(gpu)(gs, ext, k, near, far, (fw, fw), depth_mode=mode)
    assert torch.equal(fused.color, ref.color)
    scale = ref.depth.abs().max().item() + 1e-12
    assert (fused.depth - ref.depth).abs().max().item() <= 2e-5 * scale
    assert (fused.depth - ref.depth).abs().mean().item() <= 2e-6 * scale


def test_fused_decoder_eval_shape_equals_dropin_decoder(gpu):
    """3 panoramas x 6 faces through the decoder modules (evaluation shape): with the reference's torch camera glue the
    fused decoder is bit-identical to the per-face drop-in decoder; with the one-kernel native glue (default) the camera
    records differ by a few ulp (Gauss-Jordan vs LU inverses), the images by <= 1e-6."""
    from types import SimpleNamespace
    cloud = synthetic.uniform_cloud(4000, seed=5, extent=3.0, scale_range=(0.02, 0.3))
    gs = SimpleNamespace(**{k: torch.tensor(v, device=gpu)[None] for k, v in cloud.items()})
    panos = torch.stack([torch.tensor(synthetic.target_pano_pose((0.2 * j, 0.0, -0.1 * j))) for j in range(3)])
    ext = cameras.cube_face_extrinsics(panos).reshape(1, 18, 4, 4).to(gpu)
    k = cameras.cube_face_intrinsics(3).reshape(1, 18, 3, 3).to(gpu)
    near = torch.full((1, 18), 0.1, device=gpu)
    far = torch.full((1, 18), 10.0, device=gpu)
    ref = decoder.DecoderSplattingCUDA().to(gpu)(gs, ext, k, near, far, (32, 32))
    assert torch.equal(decoder.DecoderSplattingFused(glue="torch").to(gpu)(gs, ext, k, near, far, (32, 32)).color, ref.color)
    nat = decoder.DecoderSplattingFused().to(gpu)(gs, ext, k, near, far, (32, 32)).color
    assert (nat - ref.color).abs().max().item() <= 1e-6


def test_fused_decoder_detects_views_with_different_camera_centres(gpu):
    """a view list that is NOT six faces of one panorama (mixed camera centres inside a group) must not be
    rendered with the first view's SH direction: the fused decoder checks the centres and falls back to per-view SH."""
    from types import SimpleNamespace
    cloud = synthetic.uniform_cloud(4000, seed=8, extent=3.0, scale_range=(0.02, 0.3))
    gs = SimpleNamespace(**{k: torch.tensor(v, device=gpu)[None] for k, v in cloud.items()})
    panos = torch.stack([torch.tensor(synthetic.target_pano_pose((0.3 * j, 0.1 * j, -0.2 * j))) for j in range(2)])
    e12 = cameras.cube_face_extrinsics(panos)                      # [2,6,4,4]
    ext = torch.cat([e12[0, :3], e12[1, :3], e12[1, 3:], e12[0, 3:]])[None].to(gpu)   # groups of 6 mix both centres
    k = cameras.cube_face_intrinsics(2).reshape(1, 12, 3, 3).to(gpu)
    near = torch.full((1, 12), 0.1, device=gpu)
    far = torch.full((1, 12), 10.0, device=gpu)
    ref = decoder.DecoderSplattingCUDA().to(gpu)(gs, ext, k, near, far, (32, 32))
    auto = decoder.DecoderSplattingFused(glue="torch").to(gpu)(gs, ext, k, near, far, (32, 32))
    assert torch.equal(auto.color, ref.color)
    forced = decoder.DecoderSplattingFused(shared_campos=True, glue="torch").to(gpu)(gs, ext, k, near, far, (32, 32))
    assert not torch.equal(forced.color, ref.color)                # what the unchecked flag would have produced
    assert decoder.views_share_camera_centre(ext[0, :3], near[0, :3]) and not decoder.views_share_camera_centre(ext[0, :6], near[0, :6])


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity", "log"])
def test_fused_depth_map_is_differentiable_like_the_second_pass(gpu, mode):
    """the reference's training step may request depth (model_wrapper_erp.py:228) and LossDepth
    (loss_depth.py:37-60) back-propagates through it.  Gradients of (colour loss + depth loss) through the fused
    one-pass decoder must equal those through the reference-style decoder (separate depth rasterisation with
    colours_precomp = per-Gaussian depth value, gradient reaching the means through the einsum at cuda_splatting.py:239-242)."""
    from types import SimpleNamespace
    cloud = synthetic.uniform_cloud(6000, seed=31, extent=3.0, scale_range=(0.03, 0.3))
    fw = 48
    ext = cameras.cube_face_extrinsics(torch.tensor(synthetic.target_pano_pose((0.1, -0.1, 0.2)))[None]).to(gpu)
    k = cameras.cube_face_intrinsics(1).to(gpu)
    near = torch.full((1, 6), 0.1, device=gpu)
    far = torch.full((1, 6), 10.0, device=gpu)
    wc = torch.randn(1, 6, 3, fw, fw, device=gpu)
    wd = torch.randn(1, 6, fw, fw, device=gpu)
    grads = []
    for dec in (decoder.DecoderSplattingCUDA().to(gpu), decoder.DecoderSplattingFused().to(gpu)):
        gs = SimpleNamespace(**{key: torch.tensor(v, device=gpu)[None].requires_grad_(True) for key, v in cloud.items()})
        out = dec(gs, ext, k, near, far, (fw, fw), depth_mode=mode)
        ((out.color * wc).sum() + (out.depth * wd).sum()).backward()
        grads.append({key: getattr(gs, key).grad for key in cloud})
    for key in cloud:
        a, b = grads[0][key], grads[1][key]
        scale = a.abs().max().item() + 1e-20
        assert (a - b).abs().max().item() / scale <= 2e-4, (mode, key, (a - b).abs().max().item() / scale)
    if mode != "log":   # "log" keeps the reference's swapped clamp: its value does not depend on the Gaussian at all
        gs = SimpleNamespace(**{key: torch.tensor(v, device=gpu)[None].requires_grad_(True) for key, v in cloud.items()})
        out = decoder.DecoderSplattingFused().to(gpu)(gs, ext, k, near, far, (fw, fw), depth_mode=mode)
        (out.depth * wd).sum().backward()       # depth-only loss: used to raise / give zero gradient
        assert gs.means.grad.abs().max().item() > 0 and gs.harmonics.grad.abs().max().item() == 0


@pytest.mark.parametrize("fw,eh,ew", [(256, 512, 1024), (512, 1024, 2048)])
def test_cube2equirec_values_at_headline_sizes_match_reference_golden(gpu, fw, eh, ew):
    """ERP VALUES of the stitch at the headline (256 -> 1024x512) and configs[4] (512 -> 2048x1024) sizes against strided rows /
    columns of the reference's own Cube2Equirec.forward output on a seeded cube (/root/reference/src/geometry/layers.py:108-116;
    fixture: tests/golden/make_golden_decoder.py) — the smaller fixtures pin the grid only up to 64-px faces."""
    z = np.load(G / f"cube2equirec_{fw}_{eh}_{ew}_erp.npz")
    cube = np.random.default_rng(int(z["seed"])).standard_normal((1, 3, fw, 6 * fw)).astype(np.float32)
    assert float(cube.astype(np.float64).sum()) == float(z["cube_sum64"])      # the regenerated input IS the fixture's input
    c2e = stitch.Cube2Equirec(fw, eh, ew).to(gpu)
    erp = c2e(torch.tensor(cube, device=gpu))[0].cpu().numpy()
    rs, cs = int(z["row_stride"]), int(z["col_stride"])
    # trilinear weights in float32: the HIP kernel and torch's grid_sample order the eight products differently
    np.testing.assert_allclose(erp[:, ::rs], z["rows"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(erp[:, :, ::cs], z["cols"], rtol=0, atol=2e-6)
    assert abs(float(erp.astype(np.float64).sum()) - float(z["erp_sum64"])) <= 1e-6 * erp.size
