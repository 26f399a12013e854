"""GPU parity: HIP rasteriser (through the C ABI) vs the CPU oracle on identical inputs.

Bars: integer intermediates bit-exact; pixels <= 1e-5 mean per-pixel L1;
gradients rel 1e-4 (float summation order differs: DPP tree vs pixel order)."""
import numpy as np
import pytest
import torch

from helpers import boundary_tensors, check_instance_slots, face_settings, settings_from_views, small_front_scene
from oracle import oracle
from splatter360_amd import rasterizer, synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("parity_lists")]   # integer state is compared with the oracle: upstream-compatible lists


def _settings_to_torch(S, dev):
    return rasterizer.GaussianRasterizationSettings(
        image_height=S["image_height"], image_width=S["image_width"], tanfovx=S["tanfovx"], tanfovy=S["tanfovy"],
        bg=torch.tensor(np.asarray(S["bg"], np.float32), device=dev), scale_modifier=1.0,
        viewmatrix=torch.tensor(np.asarray(S["viewmatrix"], np.float32), device=dev),
        projmatrix=torch.tensor(np.asarray(S["projmatrix"], np.float32), device=dev), sh_degree=S["sh_degree"],
        campos=torch.tensor(np.asarray(S["campos"], np.float32), device=dev), prefiltered=False, debug=False)


def run_hip(S, means, cov6, shs, opac, dev, colors=None, grad_image=None):
    t = lambda a: None if a is None else torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=True)
    m, c, s, o, col = t(means), t(cov6), t(shs), t(opac), t(colors)
    m2 = torch.zeros_like(m, requires_grad=True)
    rast = rasterizer.GaussianRasterizer(_settings_to_torch(S, dev))
    img, radii = rast(means3D=m, means2D=m2, shs=s, colors_precomp=col, opacities=o, cov3D_precomp=c)
    st = rasterizer.last_state()
    out = dict(image=img.detach().cpu().numpy(), radii=radii.cpu().numpy(), state={k: v.cpu().numpy() for k, v in st.tensors().items()},
               num_rendered=st.num_rendered())
    if grad_image is not None:
        img.backward(torch.tensor(np.asarray(grad_image, np.float32), device=dev))
        out["grads"] = dict(means3D=m.grad.cpu().numpy(), means2D=m2.grad.cpu().numpy(), cov3D=c.grad.cpu().numpy(),
                            opacities=o.grad.cpu().numpy(), shs=None if s is None else s.grad.cpu().numpy(),
                            colors_precomp=None if col is None else col.grad.cpu().numpy())
    return out


def check_forward(h, f, P, H, W):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st = h["state"]
    assert h["num_rendered"] == f["num_rendered"]
    np.testing.assert_array_equal(h["radii"], f["radii"])
    np.testing.assert_array_equal(st["tiles_touched"][0].astype(np.uint32), f["tiles_touched"])
    assert f["offsets"][-1] == f["num_rendered"] if len(f["offsets"]) else True   # upstream's scan total == our slot count
    check_instance_slots(st["slot_base"][0], st["slot_pair"], st["tiles_touched"][0], f["num_rendered"])
    vis = f["radii"] > 0
    np.testing.assert_array_equal(st["rec_a"][0][vis][:, :2], f["xy"][vis])          # pixel centres: bit-exact
    np.testing.assert_array_equal(st["depths"][0][vis], f["depth"][vis])               # depths: bit-exact
    np.testing.assert_array_equal(st["rec_c"][0][vis][:, 1].view(np.int32), f["radii"][vis])
    # conic is stored pre-scaled by one float multiply: (-log2(e)/2) a, (-log2(e)) b, (-log2(e)/2) c — still bit-exact
    kd, ko = np.float32(-0.5 * 1.4426950408889634), np.float32(-1.4426950408889634)
    np.testing.assert_array_equal(st["rec_a"][0][vis][:, 2], f["conic_opacity"][vis][:, 0] * kd)
    np.testing.assert_array_equal(st["rec_a"][0][vis][:, 3], f["conic_opacity"][vis][:, 1] * ko)
    np.testing.assert_array_equal(st["rec_b"][0][vis][:, 0], f["conic_opacity"][vis][:, 2] * kd)
    L = f["num_rendered"]
    ts = st["tile_start"].astype(np.uint32)
    # upstream-equivalent sorted (key, value) list and tile ranges
    np.testing.assert_array_equal(st["list"][:L].astype(np.uint32), f["values"])
    tile_of = np.repeat(np.arange(gx * gy, dtype=np.uint64), np.diff(ts.astype(np.int64)))
    keys_up = (tile_of << np.uint64(32)) | (st["keys"][:L].view(np.uint64) >> np.uint64(32))
    np.testing.assert_array_equal(keys_up, f["keys"])
    nonempty = f["ranges"][:, 1] > f["ranges"][:, 0]
    np.testing.assert_array_equal(ts[:-1][nonempty], f["ranges"][nonempty, 0])
    np.testing.assert_array_equal(ts[1:][nonempty], f["ranges"][nonempty, 1])
    # colours (SH evaluated with the oracle's rounding) and clamp flags
    rgb = np.stack([st["rec_b"][0][:, 2], st["rec_b"][0][:, 3], st["rec_c"][0][:, 0]], 1)
    np.testing.assert_allclose(rgb[vis], f["rgb"][vis], rtol=0, atol=2e-6)
    # north_star: <= 1e-5 per-pixel L1.  Asserted on the maximum over the pixels whose n_contrib agrees with the oracle's; a
    # pixel where exp() lands within an ulp of the 1/255 or 1e-4 threshold on opposite sides in v_exp_f32 and libm takes a
    # different accept / stop decision (a legitimate O(alpha) difference): such pixels are counted, not hidden — at most
    # one per 20 000 pixels (none on any committed case)
    per_px = np.abs(h["image"].astype(np.float64) - f["image"]).mean(0)
    nc = st["n_contrib"][0].astype(np.uint32)
    same = nc == f["n_contrib"]
    assert per_px.mean() <= 1e-6, per_px.mean()
    assert per_px[same].max(initial=0.0) <= 1e-5, per_px[same].max()
    assert (~same).sum() <= same.size // 20000, int((~same).sum())
    # T = prod(1 - alpha): one factor with alpha near its 0.99 clamp amplifies the ulp of exp() a hundredfold (measured 5.5e-6
    # on the fuzz scenes, whose opacities reach 1; 2e-6 holds at the headline sizes and is asserted there)
    np.testing.assert_allclose(st["final_T"][0][same], f["final_T"][same], rtol=0, atol=1e-5)


def check_grads(hg, og, rtol=2e-4):
    for k in ("means3D", "means2D", "cov3D", "opacities", "shs", "colors_precomp"):
        if og.get(k) is None:
            continue
        a, b = hg[k].reshape(-1), np.asarray(og[k], np.float32).reshape(-1)
        scale = np.abs(b).max() + 1e-12
        err = np.abs(a - b).max() / scale
        assert err <= rtol, (k, err)


@pytest.mark.parametrize("seed", [0, 1])
def test_small_scene_forward_backward(gpu, seed):
    S, means, cov6, shs, opac = small_front_scene(n=60, seed=seed, h=64, w=80)
    rng = np.random.default_rng(seed)
    gimg = rng.standard_normal((3, 64, 80)).astype(np.float32)
    orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    f = orc.forward()
    og = orc.backward(gimg)
    h = run_hip(S, means, cov6, shs, opac, gpu, grad_image=gimg)
    check_forward(h, f, 60, 64, 80)
    check_grads(h["grads"], og)


@pytest.mark.parametrize("face", range(6))
def test_config0_faces_vs_oracle(gpu, face):
    """BASELINE config 0 shape: 10k Gaussians, 256x128 ERP -> 64x64 faces."""
    cloud = synthetic.uniform_cloud(10_000, seed=3, extent=3.0, scale_range=(0.02, 0.3))
    S = face_settings(face, 64, 64)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    rng = np.random.default_rng(face)
    gimg = rng.standard_normal((3, 64, 64)).astype(np.float32)
    orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    f = orc.forward()
    og = orc.backward(gimg)
    h = run_hip(S, means, cov6, shs, opac, gpu, grad_image=gimg)
    check_forward(h, f, 10_000, 64, 64)
    check_grads(h["grads"], og, rtol=5e-4)


def test_colors_precomp_path(gpu):
    S, means, cov6, shs, opac = small_front_scene(n=50, seed=4, h=48, w=48)
    colors = np.random.default_rng(1).uniform(0, 1, (50, 3)).astype(np.float32)
    gimg = np.random.default_rng(2).standard_normal((3, 48, 48)).astype(np.float32)
    orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, colors_precomp=colors)
    f = orc.forward()
    og = orc.backward(gimg)
    h = run_hip(S, means, cov6, None, opac, gpu, colors=colors, grad_image=gimg)
    check_forward(h, f, 50, 48, 48)
    check_grads(h["grads"], og)


def _cloud_tensors(cloud, dev, grad=False):
    return [torch.tensor(cloud[k], device=dev, requires_grad=grad) for k in ("means", "covariances", "harmonics", "opacities")]


def test_fused_cube6_equals_six_dropin_calls(gpu):
    """render_cube_faces (one V=6 call, zero-copy layouts, in-kernel 1/near rescale) must equal six
    reference-style render_cuda calls bit for bit in the forward; gradients agree to float-sum order."""
    from splatter360_amd import cameras, decoder
    cloud = synthetic.uniform_cloud(20_000, seed=7, extent=2.5, scale_range=(0.02, 0.25))
    fw = 64
    pose = torch.tensor(synthetic.target_pano_pose((0.1, -0.2, 0.05)), device=gpu)
    near, far = torch.tensor(0.1, device=gpu), torch.tensor(10.0, device=gpu)
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu)
    rng = np.random.default_rng(0)
    gimg = torch.tensor(rng.standard_normal((6, 3, fw, fw)).astype(np.float32), device=gpu)

    ins_f = _cloud_tensors(cloud, gpu, True)
    faces = decoder.render_cube_faces(pose, near, far, fw, bg, *ins_f, glue="torch")   # same camera records as the drop-in calls
    faces.backward(gimg)

    ins_d = _cloud_tensors(cloud, gpu, True)
    ext = cameras.cube_face_extrinsics(pose[None])  # [1,6,4,4]
    k = cameras.cube_face_intrinsics(1, device=gpu)
    outs = []
    for f in range(6):
        outs.append(decoder.render_cuda(ext[:, f], k[:, f], near[None], far[None], (fw, fw), bg[None], ins_d[0][None],
                                        ins_d[1][None], ins_d[2][None], ins_d[3][None])[0])
    ref = torch.stack(outs)
    ref.backward(gimg)
    assert torch.equal(faces, ref)
    for a, b, name in zip(ins_f, ins_d, ("means", "covariances", "harmonics", "opacities")):
        scale = b.grad.abs().max().item() + 1e-12
        err = (a.grad - b.grad).abs().max().item() / scale
        assert err <= 2e-5, (name, err)
    # lower triangle of the covariance receives no gradient (adjoint of the triu gather)
    assert ins_f[1].grad[:, 1, 0].abs().max().item() == 0.0


def test_fused_cube6_at_736px_faces_uses_the_per_view_histogram_and_equals_six_calls(gpu):
    """Six 736x736 faces = 6 x 2116 tiles: the all-images tile histogram (50.8 KB) no longer fits the geometry kernel's LDS budget
    and it switches to the one-image-at-a-time form (flushed after every view, barriers inside the view loop, a ragged last
    block); the six single-face calls still use the all-images form.  Same images bit for bit, same integer state."""
    from splatter360_amd import cameras, decoder
    cloud = synthetic.uniform_cloud(5_003, seed=11, extent=2.5, scale_range=(0.02, 0.25))     # 5 003: a ragged last workgroup
    fw = 736
    pose = torch.tensor(synthetic.target_pano_pose((0.1, -0.2, 0.05)), device=gpu)
    near, far = torch.tensor(0.1, device=gpu), torch.tensor(10.0, device=gpu)
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu)
    ins = _cloud_tensors(cloud, gpu, False)
    faces = decoder.render_cube_faces(pose, near, far, fw, bg, *ins, glue="torch")
    tt = rasterizer.last_state().tensors()["tiles_touched"].clone()
    ext = cameras.cube_face_extrinsics(pose[None])
    k = cameras.cube_face_intrinsics(1, device=gpu)
    for f in range(6):
        ref = decoder.render_cuda(ext[:, f], k[:, f], near[None], far[None], (fw, fw), bg[None], ins[0][None], ins[1][None],
                                  ins[2][None], ins[3][None])[0]
        assert torch.equal(faces[f], ref)
        assert torch.equal(tt[f], rasterizer.last_state().tensors()["tiles_touched"][0])


def test_fused_cube6_vs_oracle(gpu):
    from splatter360_amd import decoder
    cloud = synthetic.uniform_cloud(10_000, seed=11, extent=3.0, scale_range=(0.02, 0.3))
    fw = 64
    pose = torch.tensor(synthetic.target_pano_pose(), device=gpu)
    near, far = torch.tensor(0.1, device=gpu), torch.tensor(10.0, device=gpu)
    bg = torch.zeros(3, device=gpu)
    ext, K, nr, fr = decoder.cube_cameras(pose, near, far)
    views = decoder.pack_camera_views(ext, K, nr, fr, bg)
    faces = decoder.render_views_fused(ext, K, nr, fr, (fw, fw), bg, *_cloud_tensors(cloud, gpu), views=views, shared_campos=True).cpu().numpy()
    for face in range(6):
        S = settings_from_views(views, face, fw, fw)
        means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
        f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs).forward()
        assert np.abs(faces[face] - f["image"]).mean() <= 1e-5
