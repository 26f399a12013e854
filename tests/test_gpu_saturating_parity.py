"""Oracle parity in the regimes the headline cloud does not reach (VERDICT r04 "missing" #7, "weak" #3), at the bench's own sizes:

  surface-like cloud   1 048 576 Gaussians of a coherent depth field with opacity >= 0.9 (synthetic.surface_like_cloud — the bench's
                       `workloads.surface_like` leg): pixels SATURATE (the early-termination logic `T (1 - alpha) < 1e-4`, `done`,
                       the replay-length / survivor-count bookkeeping of the training forward) behind tile lists of 9 - 17 K entries
                       on the polar faces, of which a pole quadrant replays > 10 K;
  uniform cloud        1 048 576 Gaussians U[-5,5]^3 (the bench's `workloads.uniform` leg): thousands of footprints beyond 32 tiles —
                       rectangles binned whole, pairs summed by the wave-parallel phase of k_gather_slots.

Per face: integer state bit-exact on the upstream-compatible lists with list splitting off (tiles_touched, sorted list, keys,
ranges), pixels / final_T / n_contrib with the bars of tests/test_gpu_headline_parity.py, backward against the float32 AND float64
oracle.  Then the PRODUCT DEFAULT (lean lists, long lists composited segment-parallel) against the same oracle outputs: pixels
<= 1e-5 (north_star), n_contrib exact up to the legitimate borderline flips, gradients with the same bar.
PARITY UNPINNED: the oracle restates the un-vendored upstream extension (see oracle/s360_oracle.c header)."""
import numpy as np
import pytest
import torch

from helpers import boundary_tensors, settings_from_views
from oracle import oracle
from splatter360_amd import rasterizer, synthetic
from test_gpu_headline_parity import _face_state, _grad_err, _pixel_stats, _report, _single_face_call

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def surface1m():
    return synthetic.surface_like_cloud(512, 1024, seed=0)


@pytest.fixture(scope="module")
def uniform1m():
    return synthetic.uniform_cloud(1 << 20, seed=0, extent=5.0)


def _params(cloud, dev):
    return [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]


class _Mode:
    """Module switches of the rasteriser for the duration of a block: (lean lists, segment-parallel long lists)."""

    def __init__(self, lean, split):
        self.want = (lean, split)

    def __enter__(self):
        self.old = (rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS)
        rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS = self.want

    def __exit__(self, *a):
        rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS = self.old


def _check_forward_saturating(fs, img, f, tag, n_tiles):
    """test_gpu_headline_parity._check_face_forward for the saturating regime: integer state bit-exact as there; the pixel bars
    allow for STOP FLIPS — a pixel whose `T (1 - alpha) < 1e-4` test lands on opposite sides in v_exp_f32 and libm stops one
    entry earlier or later; with opacities of 0.9 - 0.99 the transmittance at that point is up to 1e-4 / (1 - alpha) = 1e-2, so
    the one extra contribution is worth up to ~1e-3 (the headline cloud, opacity ~0.5, has none).  There may be a handful of such
    pixels per face (counted, bounded), and every other pixel obeys the headline bars."""
    np.testing.assert_array_equal(fs["tiles_touched"], f["tiles_touched"])
    L = f["num_rendered"]
    assert fs["list"].shape[0] == L
    np.testing.assert_array_equal(fs["list"], f["values"])
    tile_of = np.repeat(np.arange(n_tiles, dtype=np.uint64), np.diff(fs["tile_start"]))
    np.testing.assert_array_equal((tile_of << np.uint64(32)) | fs["depth_bits"], f["keys"])
    nonempty = f["ranges"][:, 1] > f["ranges"][:, 0]
    np.testing.assert_array_equal(fs["tile_start"][:-1][nonempty], f["ranges"][nonempty, 0])
    np.testing.assert_array_equal(fs["tile_start"][1:][nonempty], f["ranges"][nonempty, 1])
    return _check_pixels_saturating(img, fs["n_contrib"], fs["final_T"], f, tag)


def _check_pixels_saturating(img, n_contrib, final_T, f, tag, same_lists=True):
    st = _pixel_stats(img, f["image"])
    per_px = np.abs(img.astype(np.float64) - f["image"]).mean(0)
    amax = float(np.abs(f["image"]).max())
    over = per_px > 1e-5 * max(1.0, amax)
    flips = n_contrib != f["n_contrib"] if same_lists else over
    _report(tag, n_contrib_mismatch=float(flips.mean()) if same_lists else None, pixels_over_1e5=int(over.sum()), image_absmax=amax, **st)
    # (an ACCEPT flip — alpha within an ulp of 1/255 — need not move n_contrib, the last contributor; at transmittance ~1 it is worth
    # up to 1/255 of a colour: the same 1e-3 scale.  Hence a count, not a per-pixel attribution.)
    assert int(over.sum()) <= 8 and st["max"] <= 2e-3, (tag, st, int(over.sum()))
    assert st["p999"] <= 1e-6 and st["mean"] <= 1e-7, (tag, st)
    if same_lists:
        assert int(flips.sum()) <= 16, (tag, int(flips.sum()))
    dT = np.abs(final_T.astype(np.float64) - f["final_T"])
    assert int((dT > 2e-6).sum()) <= 16 and dT[~flips].max(initial=0.0) <= 2e-5, (tag, float(dT.max()), int((dT > 2e-6).sum()))
    return st


def _got_grads(ps):
    r, c = np.triu_indices(3)
    return dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c],
                shs=ps[2].grad.cpu().numpy().transpose(0, 2, 1), opacities=ps[3].grad.cpu().numpy())


def _face_case(cloud, params, face, dev, tag, seed, floor=1e-4):
    """One 256x256 face of a 1 M cloud: (a) parity lists, no splitting — everything against the oracle, integers bit-exact;
    (b) the product default — observables against the same oracle results.

    The backward is compared on the pixels whose forward DECISIONS agree (round 6).  A stop / accept flip (counted and bounded by the
    forward checks: <= 16 + 8 pixels of 65 536) changes which entries a pixel composites, and with them that pixel's gradient to
    EVERY entry in front — for a face-sized splat at the front of 4 000-entry lists one flipped pixel is worth 100 ordinary pixels
    (scripts/uniform_err_tiles.py: the whole 3.9x of VERDICT r05 weak #2 was one pixel of tile (4, 9) on the uniform cloud's face 0,
    opacity gradient -1.29e-3 against -1.97e-3 for that tile, every other tile within 8e-7).  The image gradient is therefore zeroed
    on the flipped pixels for the oracle and for the HIP calls alike, and the bar is the headline's again: max(1e-4, 1.1 x the float32
    oracle's own distance from float64)."""
    rng = np.random.default_rng(seed)
    gimg = rng.standard_normal((3, 256, 256)).astype(np.float32)
    with _Mode(False, False):
        out, st, _ = _single_face_call(params, face, 256, dev)
    S = settings_from_views(st.views, 0, 256, 256)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    o32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
    f = o32.forward()
    P = cloud["means"].shape[0]
    img = out[0].detach().cpu().numpy()
    fs = _face_state(st.tensors(), 0, P, 256)
    _check_forward_saturating(fs, img, f, tag + "_fwd", 256)
    amax = max(1.0, float(np.abs(f["image"]).max()))
    flipped = (fs["n_contrib"] != f["n_contrib"]) | (np.abs(img.astype(np.float64) - f["image"]).mean(0) > 1e-5 * amax)
    # (b) the product default: lean lists, long lists split into depth segments composited in parallel
    with _Mode(True, True):
        out2, st2, _ = _single_face_call(params, face, 256, dev)
    img2 = out2[0].detach().cpu().numpy()
    t2 = st2.tensors()
    # (lean lists: n_contrib counts positions of shorter lists, so a flip shows as the pixel difference itself)
    px = _check_pixels_saturating(img2, t2["n_contrib"][0].cpu().numpy().astype(np.uint32), t2["final_T"][0].cpu().numpy(), f, tag + "_default_mode_fwd",
                                  same_lists=False)
    flipped |= np.abs(img2.astype(np.float64) - f["image"]).mean(0) > 1e-5 * amax
    assert int(flipped.sum()) <= 24, int(flipped.sum())
    gimg = gimg * (~flipped)[None].astype(np.float32)
    # "saturated": the stop test tripped.  final_T itself never drops below 1e-4 (the tripping entry is not applied); with these
    # opacities a pixel that stopped has final_T < 1e-4 / (1 - 0.99) = 1e-2, and one that did not is far above it
    sat = float((f["final_T"] < 1e-2).mean())
    g32 = o32.backward(gimg)
    del o32
    o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
    o64.forward()
    g64 = o64.backward(gimg)
    del o64
    with _Mode(False, False):
        _, _, ps = _single_face_call(params, face, 256, dev, grad_image=gimg)
    sc = np.float64(S["scale"])
    fold = dict(means3D=sc, cov3D=sc * sc, shs=1.0, opacities=1.0)       # oracle gradients are w.r.t. the scaled cloud
    got = _got_grads(ps)
    rep = dict(saturated_pixel_fraction=sat, longest_list=int(np.diff(f["ranges"].astype(np.int64), axis=1).max()),
               max_n_contrib=int(f["n_contrib"].max()), flipped_pixels_masked=int(flipped.sum()))
    bars = {}
    for k in got:
        e, e32 = _grad_err(got[k], np.asarray(g64[k]) * fold[k], np.asarray(g32[k], np.float64) * fold[k])
        rep[k], rep[k + "_oracle_f32"] = e, e32
        bars[k] = max(floor, 1.1 * e32)
        assert e <= bars[k], (tag, k, e, e32)
    with _Mode(True, True):
        _, st2, ps2 = _single_face_call(params, face, 256, dev, grad_image=gimg)
    rep["split_quadrants"] = int(st2.header()[5].item())
    assert st2.split_errors() == 0
    got2 = _got_grads(ps2)
    for k in got2:
        e, _ = _grad_err(got2[k], np.asarray(g64[k]) * fold[k])
        rep[k + "_default_mode"] = e
        assert e <= 1.5 * bars[k], (tag, k, e, bars[k])
        e_modes, _ = _grad_err(got2[k], got[k])
        rep[k + "_default_vs_parity_mode"] = e_modes
    rep.update({"default_mode_px_" + k: v for k, v in px.items()})
    _report(tag + "_bwd_rel_err_vs_f64_oracle", **rep)
    return rep


@pytest.mark.parametrize("face", [0, 5, 2])      # both polar faces (the pole clumps) and one side face
def test_surface_like_1m_face_vs_oracle(gpu, surface1m, face):
    rep = _face_case(surface1m, _params(surface1m, gpu), face, gpu, f"surface_like_face{face}", 300 + face)
    assert rep["saturated_pixel_fraction"] > 0.5            # the regime this test exists for
    if face != 2:
        assert rep["longest_list"] > 9000 and rep["max_n_contrib"] > 4096 and rep["split_quadrants"] > 10


@pytest.mark.parametrize("face", [0, 3])
def test_uniform_1m_face_vs_oracle(gpu, uniform1m, face):
    params = _params(uniform1m, gpu)
    # (round 5 ran this cloud with floor=3e-4 and blamed summation order; it was one stop-flipped pixel — see _face_case)
    rep = _face_case(uniform1m, params, face, gpu, f"uniform_face{face}", 400 + face)
    st = rasterizer.last_state()
    assert int(st.header()[4].item()) > 500                 # pairs with more than 32 instance slots: the wave-parallel gather


def test_surface_like_fused_six_faces_default_mode_is_deterministic_and_close_to_unsplit(gpu, surface1m):
    """The bench's `surface_like` leg itself (fused six-face training step, product defaults): bit-reproducible run to run, and
    within float rounding of the same step with list splitting off (which the per-face tests above put against the oracle)."""
    from splatter360_amd import decoder
    params = _params(surface1m, gpu)
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    bg = torch.zeros(3, device=gpu)
    res = []
    for split in (True, True, False):
        with _Mode(True, split):
            ps = [p.clone().requires_grad_(True) for p in params]
            faces = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *ps, shared_campos=True)
            ((faces - 0.5) ** 2).mean().backward()
            res.append([faces.detach()] + [p.grad for p in ps])
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    d = (res[0][0] - res[2][0]).abs()
    assert float(d.mean()) <= 1e-7 and int((d > 1e-5).sum()) <= d.numel() // 20_000, (float(d.mean()), float(d.max()))
    for a, b in zip(res[0][1:], res[2][1:]):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()), (float((a - b).abs().max()), float(b.abs().max()))
