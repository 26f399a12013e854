"""Oracle parity AT the BASELINE.json headline sizes (VERDICT r01 "next round" item 1):

  configs[1]/[2]  1 048 576 Gaussians, six 256x256 faces: all six faces forward (integer state bit-exact, pixel mean /
                  p99.9 / max reported), backward on a polar face (13 K-key lists, chunk + merge-pass sort) and a side
                  face against the float32 AND float64 oracle, and the fused six-face gradient against the sum of six
                  oracle backwards;
  configs[4]      4 194 304 Gaussians (what two 2048x1024 context panoramas emit), one 512x512 face, forward + backward;
  configs[3]      evaluation shape: 3 target panoramas x 6 faces, colour + fused depth in one pass, every
                  DepthRenderingMode against the oracle's colours_precomp = depth render.

The measured error statistics are written to gpurun_out/parity_report.json (quoted in DESIGN.md).
PARITY UNPINNED: the oracle restates the un-vendored upstream extension (see oracle/s360_oracle.c header)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import boundary_tensors, face_settings, settings_from_views
from oracle import oracle
from splatter360_amd import cameras, decoder, rasterizer, synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("parity_lists")]   # integer state is compared with the oracle: upstream-compatible lists
ROOT = Path(__file__).resolve().parent.parent
REPORT = {}


def _report(key, **vals):
    REPORT[key] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        prev = {}
        f = out / "parity_report.json"
        if f.exists():
            prev = json.loads(f.read_text())
        prev.update(REPORT)
        f.write_text(json.dumps(prev, indent=1, sort_keys=True))
    except OSError:
        pass


def _pixel_stats(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))          # [3,H,W]
    per_px = d.mean(0).reshape(-1)                                     # per-pixel L1 (mean over channels)
    return dict(mean=per_px.mean(), p999=np.quantile(per_px, 0.999), max=per_px.max(), max_channel=d.max())


def _face_state(t, v, P, T):
    """Per-face slices of a fused V-view workspace, renumbered like a stand-alone call of that face."""
    ts = t["tile_start"].to(torch.int64)
    lo, hi = int(ts[v * T]), int(ts[(v + 1) * T])
    lst = (t["list"][lo:hi].to(torch.int64) & 0xFFFFFFFF) - v * P
    keys = t["keys"][lo:hi]
    return dict(tile_start=(ts[v * T:(v + 1) * T + 1] - lo).cpu().numpy(), list=lst.cpu().numpy().astype(np.uint32),
                depth_bits=((keys >> 32) & 0xFFFFFFFF).cpu().numpy().astype(np.uint64),
                tiles_touched=t["tiles_touched"][v].cpu().numpy().astype(np.uint32),
                n_contrib=t["n_contrib"][v].cpu().numpy().astype(np.uint32), final_T=t["final_T"][v].cpu().numpy())


def _check_face_forward(fs, img, f, tag, gx_gy):
    """Integer state bit-exact (tiles_touched, sorted list, sort keys, tile ranges); pixels by mean / p99.9 / max."""
    np.testing.assert_array_equal(fs["tiles_touched"], f["tiles_touched"])
    L = f["num_rendered"]
    assert fs["list"].shape[0] == L
    np.testing.assert_array_equal(fs["list"], f["values"])
    tile_of = np.repeat(np.arange(gx_gy, dtype=np.uint64), np.diff(fs["tile_start"]))
    np.testing.assert_array_equal((tile_of << np.uint64(32)) | fs["depth_bits"], f["keys"])
    nonempty = f["ranges"][:, 1] > f["ranges"][:, 0]
    np.testing.assert_array_equal(fs["tile_start"][:-1][nonempty], f["ranges"][nonempty, 0])
    np.testing.assert_array_equal(fs["tile_start"][1:][nonempty], f["ranges"][nonempty, 1])
    st = _pixel_stats(img, f["image"])
    mism = float((fs["n_contrib"] != f["n_contrib"]).mean())
    amax = float(np.abs(f["image"]).max())
    _report(tag, n_contrib_mismatch=mism, longest_list=int(np.diff(fs["tile_start"]).max()), num_rendered=int(L), image_absmax=amax, **st)
    # north_star: "<= 1e-5 per-pixel L1" (of images in [0, 1]) — asserted on the MAXIMUM over pixels, relative to the image's
    # own range where the synthetic SH colours exceed 1 (measured: mean 2.8e-8, max 3.6e-7 at 1 M, no n_contrib mismatch at
    # any of these sizes; 1.2e-5 on one pixel of a 4 M polar face behind an 18.8 K-key list); a regression of one order of
    # magnitude fails here
    per_px = np.abs(img.astype(np.float64) - f["image"]).mean(0)
    over = per_px > 1e-5 * max(1.0, amax)
    # a pixel may sit behind an entry whose alpha lands within an ulp of 1/255 on opposite sides in v_exp_f32 and libm: one
    # accepts it, the other does not (n_contrib, the LAST contributor, need not change) — an O(alpha T) = 1e-5-sized legitimate
    # difference.  Seen once: one pixel of 1.57 M on the 4 M polar face (1.17e-5).  Allowed: one such pixel per 200 000, <= 5e-5.
    assert int(over.sum()) <= per_px.size // 200_000 and st["max"] <= 5e-5, (tag, st, int(over.sum()))
    assert st["p999"] <= 1e-6 and st["mean"] <= 1e-7, (tag, st, amax)
    assert mism <= 1e-5, (tag, mism)
    dT = np.abs(fs["final_T"].astype(np.float64) - f["final_T"])
    assert int((dT > 2e-6).sum()) <= per_px.size // 200_000 and dT.max() <= 2e-5, (tag, float(dT.max()), int((dT > 2e-6).sum()))


def _grad_err(got, want64, want32=None):
    got = np.asarray(got, np.float64).reshape(-1)
    w = np.asarray(want64, np.float64).reshape(-1)
    scale = np.abs(w).max() + 1e-30
    e = np.abs(got - w).max() / scale
    e32 = None if want32 is None else np.abs(np.asarray(want32, np.float64).reshape(-1) - w).max() / scale
    return e, e32


@pytest.fixture(scope="module")
def cloud1m():
    return synthetic.encoder_like_cloud(512, 1024, seed=0)


@pytest.fixture(scope="module")
def params1m(gpu, cloud1m):
    return [torch.tensor(cloud1m[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]


def _single_face_call(params, face, fw, dev, grad_image=None, position=(0.0, 0.0, 0.0), depth_mode=None):
    ps = [p.clone().requires_grad_(grad_image is not None) for p in params]
    pose = torch.tensor(synthetic.target_pano_pose(position), device=dev)
    ext, K, near, far = decoder.cube_cameras(pose, 0.1, 10.0)
    s = slice(face, face + 1)
    views = decoder.pack_camera_views(ext[s], K[s], near[s], far[s], torch.zeros(3, device=dev))
    out = decoder.render_views_fused(ext[s], K[s], near[s], far[s], (fw, fw), torch.zeros(3, device=dev), *ps,
                                     depth_mode=depth_mode, views=views)
    st = rasterizer.last_state()
    st.views = views      # the camera records the kernels saw: the oracle is given exactly these
    if grad_image is not None:
        (out if depth_mode is None else out[0]).backward(torch.tensor(grad_image, device=dev)[None])
    return out, st, ps


def test_1m_all_six_faces_forward_vs_oracle(gpu, cloud1m, params1m):
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    views = decoder.pack_camera_views(ext, K, near, far, torch.zeros(3, device=gpu))
    faces = decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=gpu), *params1m, views=views)
    t = rasterizer.last_state().tensors()
    faces = faces.cpu().numpy()
    P = cloud1m["means"].shape[0]
    for face in range(6):
        S = settings_from_views(views, face, 256, 256)
        means, cov6, shs, opac = boundary_tensors(cloud1m, S["scale"])
        f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs).forward()
        _check_face_forward(_face_state(t, face, P, 256), faces[face], f, f"1m_face{face}_fwd", 256)


@pytest.mark.parametrize("face", range(6))   # 0 / 5 = polar faces (lists of up to ~13 K keys), 1..4 = side faces
def test_1m_backward_vs_oracle(gpu, cloud1m, params1m, face):
    rng = np.random.default_rng(100 + face)
    gimg = rng.standard_normal((3, 256, 256)).astype(np.float32)
    out, st, ps = _single_face_call(params1m, face, 256, gpu, grad_image=gimg)
    S = settings_from_views(st.views, 0, 256, 256)
    means, cov6, shs, opac = boundary_tensors(cloud1m, S["scale"])
    o32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    o32.forward()
    g32 = o32.backward(gimg)
    del o32
    o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
    o64.forward()
    g64 = o64.backward(gimg)
    del o64
    sc = np.float64(S["scale"])
    r, c = np.triu_indices(3)
    got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c],
               shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1), opacities=ps[3].grad.cpu().numpy())
    fold = dict(means3D=sc, cov3D=sc * sc, shs=1.0, opacities=1.0)       # oracle gradients are w.r.t. the scaled cloud
    rep = {}
    for k in got:
        e, e32 = _grad_err(got[k], np.asarray(g64[k]) * fold[k], np.asarray(g32[k], np.float64) * fold[k])
        rep[k], rep[k + "_oracle_f32"] = e, e32
        assert e <= max(1e-4, 1.1 * e32), (face, k, e, e32)    # measured: within 1.000x of the float32 oracle's own distance
    _report(f"1m_face{face}_bwd_rel_err_vs_f64_oracle", **rep)


def test_1m_face_end_to_end_with_independently_computed_camera_records(gpu, cloud1m, params1m):
    """Closes the loop at the headline size: the oracle is given camera records computed INDEPENDENTLY by the reference-pinned
    torch glue on the CPU (helpers.face_settings -> cameras.view_setup, golden-pinned against the reference's own
    cuda_splatting.py:64-112), not the records the kernel under test produced (settings_from_views, used by the other 1 M
    tests).  The HIP path gets the raw cameras and packs them with its one-kernel glue (s360_pack_views: Gauss-Jordan inverses,
    a few ulp from LU) — so the integer state may differ where a radius or a rectangle edge sits within an ulp of an integer:
    counted and bounded, not asserted bit-exact."""
    face, pos = 2, (0.05, -0.02, 0.03)
    rng = np.random.default_rng(7)
    gimg = rng.standard_normal((3, 256, 256)).astype(np.float32)
    out, st, ps = _single_face_call(params1m, face, 256, gpu, grad_image=gimg, position=pos)
    S = face_settings(face, 256, 256, near=0.1, far=10.0, position=pos)
    means, cov6, shs, opac = boundary_tensors(cloud1m, S["scale"])
    o32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    f = o32.forward()
    g32 = o32.backward(gimg)
    del o32
    t = st.tensors()
    tt = t["tiles_touched"][0].cpu().numpy().astype(np.uint32)
    mism = float((tt != f["tiles_touched"]).mean())
    img = out.detach().cpu().numpy()[0]
    px = _pixel_stats(img, f["image"])
    per_px = np.abs(img.astype(np.float64) - f["image"]).mean(0)
    sc = np.float64(S["scale"])
    r, c = np.triu_indices(3)
    got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c],
               shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1), opacities=ps[3].grad.cpu().numpy())
    fold = dict(means3D=sc, cov3D=sc * sc, shs=1.0, opacities=1.0)
    rep = {k: _grad_err(got[k], np.asarray(g32[k], np.float64) * fold[k])[0] for k in got}
    _report("1m_face2_end_to_end_independent_cameras", tiles_touched_mismatch=mism, num_rendered_hip=int(st.num_rendered()),
            num_rendered_oracle=int(f["num_rendered"]), **px, **{"grad_" + k: v for k, v in rep.items()})
    assert mism <= 1e-5 and abs(int(st.num_rendered()) - int(f["num_rendered"])) <= 50
    assert px["mean"] <= 2e-7 and px["p999"] <= 2e-6 and int((per_px > 1e-5).sum()) <= 8, px
    for k, e in rep.items():
        assert e <= 2e-3, (k, e)      # float32 oracle vs HIP at 1 M: the other tests measure 1.7e-3 ... 4e-6 against float64


def test_1m_fused_six_face_gradient_equals_sum_of_oracle_backwards(gpu, cloud1m, params1m):
    """The headline configuration itself: one fused V=6 forward+backward (L2 loss on the faces) against the sum of six
    float64 oracle backwards seeded with the same per-face pixel gradients."""
    ps = [p.clone().requires_grad_(True) for p in params1m]
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    views = decoder.pack_camera_views(ext, K, near, far, torch.zeros(3, device=gpu))
    faces = decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=gpu), *ps, views=views)
    gt = torch.full_like(faces, 0.5)
    ((faces - gt) ** 2).mean().backward()
    seed = (2.0 / faces.numel() * (faces.detach() - gt)).cpu().numpy()
    P = cloud1m["means"].shape[0]
    tot = dict(means3D=np.zeros((P, 3)), cov3D=np.zeros((P, 6)), shs=np.zeros((P, 25, 3)), opacities=np.zeros((P, 1)))
    for face in range(6):
        S = settings_from_views(views, face, 256, 256)
        means, cov6, shs, opac = boundary_tensors(cloud1m, S["scale"])
        o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
        o.forward()
        g = o.backward(seed[face])
        sc = np.float64(S["scale"])
        tot["means3D"] += g["means3D"] * sc
        tot["cov3D"] += g["cov3D"] * sc * sc
        tot["shs"] += g["shs"]
        tot["opacities"] += g["opacities"]
        del o
    r, c = np.triu_indices(3)
    got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c],
               shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1), opacities=ps[3].grad.cpu().numpy()[:, None])
    rep = {}
    for k in got:
        rep[k], _ = _grad_err(got[k], tot[k])
        assert rep[k] <= 2e-4, (k, rep[k])      # measured <= 7.7e-5
    _report("1m_fused6_l2loss_bwd_rel_err_vs_f64_oracle_sum", **rep)


def test_4m_512_face_forward_backward_vs_oracle(gpu):
    """BASELINE configs[4] single-rank shape: G = 4 194 304 (two 2048x1024 context panoramas), 512x512 faces."""
    cloud = synthetic.encoder_like_cloud(1024, 2048, seed=0)
    assert cloud["means"].shape[0] == 4_194_304
    params = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    face = 2
    rng = np.random.default_rng(7)
    gimg = rng.standard_normal((3, 512, 512)).astype(np.float32)
    out, st, ps = _single_face_call(params, face, 512, gpu, grad_image=gimg)
    assert not st.overflowed()
    S = settings_from_views(st.views, 0, 512, 512)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    f = o.forward()
    _check_face_forward(_face_state(st.tensors(), 0, 4_194_304, 1024), out[0].detach().cpu().numpy(), f, "4m_512_face2_fwd", 1024)
    g32 = o.backward(gimg)
    del o
    sc = np.float64(S["scale"])
    r, c = np.triu_indices(3)
    got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c],
               shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1), opacities=ps[3].grad.cpu().numpy())
    fold = dict(means3D=sc, cov3D=sc * sc, shs=1.0, opacities=1.0)
    rep = {}
    for k in got:   # float32 oracle only (its own rounding is part of the distance): looser bar than the f64 comparison
        rep[k], _ = _grad_err(got[k], np.asarray(g32[k], np.float64) * fold[k])
        assert rep[k] <= 5e-5, (k, rep[k])      # measured <= 2.5e-6
    _report("4m_512_face2_bwd_rel_err_vs_f32_oracle", **rep)


def test_4m_512_fused_six_faces_forward_backward_vs_oracle(gpu):
    """BASELINE configs[4], one rank's whole share: ONE fused call renders the six 512x512 faces of a 2048x1024 panorama from
    4 194 304 Gaussians, L2 loss on the faces, backward — every face's integer state and pixels against the oracle, the
    summed gradient against the sum of six float32-oracle backwards (the float64 oracle needs ~6 x 40 s at this size)."""
    cloud = synthetic.encoder_like_cloud(1024, 2048, seed=0)
    P = cloud["means"].shape[0]
    ps = [torch.tensor(cloud[k], device=gpu).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    views = decoder.pack_camera_views(ext, K, near, far, torch.zeros(3, device=gpu))
    faces = decoder.render_views_fused(ext, K, near, far, (512, 512), torch.zeros(3, device=gpu), *ps, views=views)
    st = rasterizer.last_state()
    assert not st.overflowed()
    gt = torch.full_like(faces, 0.5)
    ((faces - gt) ** 2).mean().backward()
    seed = (2.0 / faces.numel() * (faces.detach() - gt)).cpu().numpy()
    t = st.tensors()
    img = faces.detach().cpu().numpy()
    tot = dict(means3D=np.zeros((P, 3)), cov3D=np.zeros((P, 6)), shs=np.zeros((P, 25, 3)), opacities=np.zeros((P, 1)))
    for face in range(6):
        S = settings_from_views(views, face, 512, 512)
        means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
        o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
        f = o.forward()
        _check_face_forward(_face_state(t, face, P, 1024), img[face], f, f"4m_512_fused_face{face}_fwd", 1024)
        g = o.backward(seed[face])
        sc = np.float64(S["scale"])
        tot["means3D"] += np.asarray(g["means3D"], np.float64) * sc
        tot["cov3D"] += np.asarray(g["cov3D"], np.float64) * sc * sc
        tot["shs"] += np.asarray(g["shs"], np.float64)
        tot["opacities"] += np.asarray(g["opacities"], np.float64).reshape(P, 1)
        del o
    r, c = np.triu_indices(3)
    got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c],
               shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1), opacities=ps[3].grad.cpu().numpy().reshape(P, 1))
    rep = {}
    for k in got:
        rep[k], _ = _grad_err(got[k], tot[k])
        assert rep[k] <= 2e-4, (k, rep[k])
    _report("4m_512_fused6_l2loss_bwd_rel_err_vs_f32_oracle_sum", **rep)


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity", "log"])
def test_eval_shape_colour_and_depth_vs_oracle(gpu, cloud1m, params1m, mode):
    """BASELINE configs[3] shape: 3 target panoramas x 6 faces, colour + depth from ONE pass per panorama; per mode one
    (panorama, face) pair is compared with the oracle (colour via SH, depth via colours_precomp = the reference's
    per-Gaussian depth value, cuda_splatting.py:239-251, rendered with background 0 and channel-averaged)."""
    positions = [(0.0, 0.0, 0.0), (0.1, 0.0, -0.05), (0.2, 0.0, -0.1)]
    pick = {"depth": (0, 4), "disparity": (1, 1), "relative_disparity": (2, 3), "log": (1, 5)}[mode]
    outs = []
    for pos in positions:
        pose = torch.tensor(synthetic.target_pano_pose(pos), device=gpu)
        ext, K, near, far = decoder.cube_cameras(pose, 0.1, 10.0)
        views = decoder.pack_camera_views(ext, K, near, far, torch.zeros(3, device=gpu))
        col, dep = decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=gpu), *params1m,
                                              depth_mode=mode, views=views)
        assert col.shape == (6, 3, 256, 256) and dep.shape == (6, 256, 256)
        outs.append((col, dep, ext, near, far, views))
    pi, face = pick
    col, dep, ext, near, far, views = outs[pi]
    S = settings_from_views(views, face, 256, 256)
    means, cov6, shs, opac = boundary_tensors(cloud1m, S["scale"])
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs).forward()
    st = _pixel_stats(col[face].cpu().numpy(), f["image"])
    assert st["mean"] <= 1e-7 and st["max"] <= 1e-5, st
    z = decoder._depth_colors(ext[face:face + 1].cpu(), torch.tensor(cloud1m["means"])[None], near[face:face + 1].cpu(),
                              far[face:face + 1].cpu(), mode)[0].numpy()
    fd = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac,
                          colors_precomp=np.repeat(z[:, None], 3, 1).astype(np.float32)).forward()
    want = fd["image"].mean(0)
    got = dep[face].cpu().numpy()
    scale = max(1.0, float(np.abs(want).max()))
    d = np.abs(got - want)
    _report(f"eval_1m_{mode}_pano{pi}_face{face}", colour_mean=st["mean"], colour_max=st["max"], depth_mean_rel=float(d.mean() / scale),
            depth_max_rel=float(d.max() / scale), depth_scale=scale)
    assert d.mean() <= 1e-6 * scale and d.max() <= 1e-5 * scale, (mode, d.mean(), d.max(), scale)


def test_render_cuda_orthographic_vs_oracle(gpu):
    """the fake-orthographic camera (0.1 degree FOV, camera pulled back ~1e3 units,
    cuda_splatting.py:130-220) through the HIP kernels: tan(fov/2) ~ 8.7e-4, focal ~ 3.7e4 px."""
    cloud = synthetic.uniform_cloud(3000, seed=9, extent=1.0, scale_range=(0.01, 0.08))
    h, w = 64, 96
    t = lambda k: torch.tensor(cloud[k], device=gpu)[None]
    ext = torch.eye(4, device=gpu)[None].clone()
    ext[0, :3, 3] = torch.tensor([0.1, -0.05, -3.0], device=gpu)
    width, height = torch.tensor([2.4], device=gpu), torch.tensor([1.6], device=gpu)
    near, far = torch.tensor([0.0], device=gpu), torch.tensor([20.0], device=gpu)
    bg = torch.tensor([[0.05, 0.1, 0.15]], device=gpu)
    dump = {}
    img = decoder.render_cuda_orthographic(ext, width, height, near, far, (h, w), bg, t("means"), t("covariances"),
                                           t("harmonics"), t("opacities"), dump=dump)[0].cpu().numpy()
    o = decoder.orthographic_setup(ext, width, height, near, far)
    S = dict(image_height=h, image_width=w, tanfovx=float(o["tan_fov_x"]), tanfovy=float(o["tan_fov_y"].reshape(-1)[0]),
             bg=bg[0].cpu().numpy(), viewmatrix=o["view_matrix"][0].cpu().numpy(), projmatrix=o["full_projection"][0].cpu().numpy(),
             sh_degree=4, campos=o["extrinsics"][0, :3, 3].cpu().numpy())
    means, cov6, shs, opac = boundary_tensors(cloud, 1.0)
    f = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs).forward()
    assert f["num_rendered"] > 1000            # the cloud is actually in view
    st = _pixel_stats(img, f["image"])
    _report("orthographic_a9", **st, num_rendered=int(f["num_rendered"]))
    assert st["mean"] <= 1e-7 and st["max"] <= 1e-5, st


def test_sh_degree4_ignored_switch(gpu):
    """S360_FLAG_SH_DEG4_IGNORED: sh_degree = 4 behaves like a rasteriser whose table stops at degree 3 —
    compared with the oracle run at sh_degree = 3 on the same 25-coefficient tensors; coefficients 16..24 get zero gradient."""
    from test_gpu_parity import check_forward, check_grads, run_hip
    from helpers import small_front_scene
    S, means, cov6, shs, opac = small_front_scene(n=60, seed=5, h=64, w=64)
    gimg = np.random.default_rng(3).standard_normal((3, 64, 64)).astype(np.float32)
    S3 = dict(S, sh_degree=3)
    orc = oracle.rasterize(S3, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    f = orc.forward()
    og = orc.backward(gimg)
    old = rasterizer.SH_DEG4_IGNORED
    try:
        rasterizer.SH_DEG4_IGNORED = True
        h = run_hip(S, means, cov6, shs, opac, gpu, grad_image=gimg)
    finally:
        rasterizer.SH_DEG4_IGNORED = old
    check_forward(h, f, 60, 64, 64)
    check_grads(h["grads"], og)
    assert np.abs(h["grads"]["shs"][:, 16:, :]).max() == 0.0
    h4 = run_hip(S, means, cov6, shs, opac, gpu)      # default: degree-4 table active -> different colours
    assert np.abs(h4["image"] - h["image"]).max() > 1e-4


def test_means_gradient_keeps_sh_direction_term_when_harmonics_are_frozen(gpu):
    """with shs.requires_grad = False (d_shs == NULL at the ABI) dL/dmean must still contain dRGB/ddir."""
    cloud = synthetic.uniform_cloud(5000, seed=13, extent=3.0, scale_range=(0.02, 0.3))
    pose = torch.tensor(synthetic.target_pano_pose((0.05, 0.1, -0.1)), device=gpu)
    near, far = torch.tensor(0.1, device=gpu), torch.tensor(10.0, device=gpu)
    w = torch.randn(6, 3, 64, 64, device=gpu)
    grads = []
    for sh_grad in (True, False):
        ps = [torch.tensor(cloud[k], device=gpu, requires_grad=(k != "harmonics" or sh_grad))
              for k in ("means", "covariances", "harmonics", "opacities")]
        (decoder.render_cube_faces(pose, near, far, 64, torch.zeros(3, device=gpu), *ps) * w).sum().backward()
        grads.append(ps)
    assert grads[1][2].grad is None
    for a, b in zip(grads[0], grads[1]):
        if b.grad is not None:
            assert torch.equal(a.grad, b.grad)
    # and the per-view-campos kernel (views with different camera centres)
    ext = torch.cat([decoder.cube_cameras(torch.tensor(synthetic.target_pano_pose((0.2 * i, 0.0, 0.0)), device=gpu), 0.1, 10.0)[0][1:3]
                     for i in range(2)])
    K = cameras.cube_face_intrinsics(1, device=gpu)[0, :4]
    nr, fr = torch.full((4,), 0.1, device=gpu), torch.full((4,), 10.0, device=gpu)
    w4 = torch.randn(4, 3, 64, 64, device=gpu)
    res = []
    for sh_grad in (True, False):
        ps = [torch.tensor(cloud[k], device=gpu, requires_grad=(k != "harmonics" or sh_grad))
              for k in ("means", "covariances", "harmonics", "opacities")]
        (decoder.render_views_fused(ext, K, nr, fr, (64, 64), torch.zeros(3, device=gpu), *ps, shared_campos=False) * w4).sum().backward()
        res.append(ps[0].grad.clone())
    assert torch.equal(res[0], res[1])
