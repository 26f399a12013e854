"""Edge cases of the HIP path against the oracle: empty / tiny clouds, ragged sizes (P not a multiple of the
block, image not a multiple of the tile), huge splats, tile lists beyond every LDS sort class, binning-capacity
overflow (retry and lazy flag), per-view camera centres (non-shared SH), opacity extremes."""
import numpy as np
import pytest
import torch

from helpers import small_front_scene
from oracle import oracle
from splatter360_amd import rasterizer
from test_gpu_parity import _settings_to_torch, check_forward, check_grads, run_hip

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("parity_lists")]   # integer state is compared with the oracle: upstream-compatible lists


def _orc(S, means, cov6, shs, opac, gimg=None, colors=None):
    o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, colors_precomp=colors)
    f = o.forward()
    return f, (o.backward(gimg) if gimg is not None else None)


def test_empty_cloud(gpu):
    S, means, cov6, shs, opac = small_front_scene(n=4, seed=0, h=32, w=48)
    z = lambda *s: np.zeros(s, np.float32)
    h = run_hip(S, z(0, 3), z(0, 6), z(0, 25, 3), z(0, 1), gpu)
    bg = np.asarray(S["bg"], np.float32)
    np.testing.assert_allclose(h["image"], np.broadcast_to(bg[:, None, None], (3, 32, 48)), atol=1e-7)
    assert h["num_rendered"] == 0


@pytest.mark.parametrize("n,h,w", [(1, 16, 16), (257, 40, 72), (1000, 33, 17), (513, 100, 260)])
def test_ragged_sizes(gpu, n, h, w):
    S, means, cov6, shs, opac = small_front_scene(n=n, seed=n, h=h, w=w, spread=0.9)
    gimg = np.random.default_rng(n).standard_normal((3, h, w)).astype(np.float32)
    f, og = _orc(S, means, cov6, shs, opac, gimg)
    hh = run_hip(S, means, cov6, shs, opac, gpu, grad_image=gimg)
    check_forward(hh, f, n, h, w)
    check_grads(hh["grads"], og, rtol=5e-4)


def test_image_with_more_tiles_than_the_lds_binning_holds(gpu):
    """2064 x 2056 pixels = 16 641 tiles: k_emit takes its global-atomic form (the per-block LDS histogram holds 8 192 tiles)
    including the block-wise instance-slot reservation; forward state and gradients against the oracle."""
    n, h, w = 700, 2056, 2064
    S, means, cov6, shs, opac = small_front_scene(n=n, seed=5, h=h, w=w, spread=0.95, srange=(0.01, 0.08))
    gimg = np.random.default_rng(5).standard_normal((3, h, w)).astype(np.float32)
    f, og = _orc(S, means, cov6, shs, opac, gimg)
    hh = run_hip(S, means, cov6, shs, opac, gpu, grad_image=gimg)
    check_forward(hh, f, n, h, w)
    check_grads(hh["grads"], og, rtol=5e-4)


def test_huge_splats_cover_every_tile(gpu):
    """A few hundred splats, each covering the whole 128x128 image: tiles_touched = 64 per splat."""
    S, means, cov6, shs, opac = small_front_scene(n=300, seed=2, h=128, w=128, srange=(1.0, 2.5), spread=0.3)
    opac = (opac * 0.1).astype(np.float64)
    gimg = np.random.default_rng(0).standard_normal((3, 128, 128)).astype(np.float32)
    f, og = _orc(S, means, cov6, shs, opac, gimg)
    assert (f["tiles_touched"] == 64).mean() > 0.5
    hh = run_hip(S, means, cov6, shs, opac, gpu, grad_image=gimg)
    check_forward(hh, f, 300, 128, 128)
    check_grads(hh["grads"], og, rtol=1e-3)


@pytest.mark.parametrize("n", [4500, 9000, 20000, 70000])
def test_tile_list_longer_than_every_lds_sort_class(gpu, n):
    """Thousands of splats on the same tiles: lists of 2 chunks (one merge pass), 3 chunks (unpaired tail run),
    5 chunks (3 passes, ping-pong parity odd) of the chunk-sort + global-merge path, and > 65 536 keys (beyond
    the pass budget: global-memory bitonic fallback).  Sorted keys / list must equal the oracle's bit for bit."""
    rng = np.random.default_rng(3)
    S, _, _, _, _ = small_front_scene(n=2, seed=0, h=32, w=32)
    z = rng.uniform(2.0, 30.0, n)
    means = np.stack([rng.uniform(-0.02, 0.02, n) * z, rng.uniform(-0.02, 0.02, n) * z, z], 1)
    cov6 = np.tile(np.array([[1e-4, 0, 0, 1e-4, 0, 1e-4]]), (n, 1)) * (z[:, None] ** 2)
    colors = rng.uniform(0, 1, (n, 3))
    opac = rng.uniform(0.001, 0.02, (n, 1))
    f, _ = _orc(S, means, cov6, None, opac, colors=colors)
    assert np.diff(f["ranges"], axis=1).max() > 0.9 * n
    hh = run_hip(S, means, cov6, None, opac, gpu, colors=colors)
    check_forward(hh, f, n, 32, 32)


def test_many_multichunk_tiles_merge_is_race_free_over_repeated_runs(gpu):
    """k_merge_all published a finished merge pass before its write-through stores had been waited on, so a
    workgroup behind another XCD could merge stale keys.  Stress: 64 tiles of ~9 500 keys each (3 chunks, 2 passes, hundreds of
    (pass, chunk) units in flight across every XCD), forty calls; the list of every call must equal the host's stable sort of
    the emitted (tile, depth, pair) keys — the first call is also put against the oracle."""
    rng = np.random.default_rng(11)
    n, hw = 9500, 128
    S, _, _, _, _ = small_front_scene(n=2, seed=0, h=hw, w=hw)
    z = rng.uniform(2.0, 30.0, n)
    means = np.stack([rng.uniform(-0.05, 0.05, n) * z, rng.uniform(-0.05, 0.05, n) * z, z], 1)
    cov6 = np.tile(np.array([[1.0, 0, 0, 1.0, 0, 1.0]]), (n, 1)) * (z[:, None] ** 2)   # 3-sigma radius > the image: every tile
    colors = rng.uniform(0, 1, (n, 3))
    opac = rng.uniform(0.001, 0.01, (n, 1))
    f, _ = _orc(S, means, cov6, None, opac, colors=colors)
    assert (np.diff(f["ranges"], axis=1) > 8192).sum() == 64
    hh = run_hip(S, means, cov6, None, opac, gpu, colors=colors)
    # lists and keys against the oracle (not check_forward's pixel bars: every pixel of this scene ends within rounding of the
    # 1e-4 stop threshold, by construction of a 9 500-entry list of faint splats)
    st0 = hh["state"]
    assert hh["num_rendered"] == f["num_rendered"]
    np.testing.assert_array_equal(st0["list"][:f["num_rendered"]].astype(np.uint32), f["values"])
    tile_of = np.repeat(np.arange(64, dtype=np.uint64), np.diff(st0["tile_start"].astype(np.int64)))
    np.testing.assert_array_equal((tile_of << np.uint64(32)) | (st0["keys"][:f["num_rendered"]].view(np.uint64) >> np.uint64(32)), f["keys"])
    assert np.abs(hh["image"].astype(np.float64) - f["image"]).mean() <= 1e-6
    want = torch.tensor(f["values"].astype(np.int64), device=gpu)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu)
    st = _settings_to_torch(S, gpu)
    views = rasterizer.pack_views(st.viewmatrix, st.projmatrix, st.campos, st.tanfovx, st.tanfovy, st.bg)
    args = (t(means), t(cov6), t(opac), None, t(colors))
    for it in range(40):
        rasterizer.rasterize_views(*args, views=views, image_height=hw, image_width=hw, sh_degree=0, shared_campos=True)
        got = rasterizer.last_state().tensors()["list"][: want.numel()].to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, want), it


def test_equal_depth_ties_keep_index_order(gpu):
    S, _, _, _, _ = small_front_scene(n=2, seed=0, h=32, w=32)
    n = 600
    rng = np.random.default_rng(5)
    means = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), np.full(n, 4.0)], 1)  # all at depth 4
    cov6 = np.tile(np.array([[0.01, 0, 0, 0.01, 0, 0.01]]), (n, 1))
    colors = rng.uniform(0, 1, (n, 3))
    opac = rng.uniform(0.05, 0.5, (n, 1))
    f, _ = _orc(S, means, cov6, None, opac, colors=colors)
    hh = run_hip(S, means, cov6, None, opac, gpu, colors=colors)
    check_forward(hh, f, n, 32, 32)


def test_capacity_overflow_retry_and_lazy_flag(gpu):
    S, means, cov6, shs, opac = small_front_scene(n=400, seed=7, h=64, w=64, srange=(0.3, 0.8))
    f, _ = _orc(S, means, cov6, shs, opac)
    L = f["num_rendered"]
    assert L > 2000
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu)
    st = _settings_to_torch(S, gpu)
    views = rasterizer.pack_views(st.viewmatrix, st.projmatrix, st.campos, st.tanfovx, st.tanfovy, st.bg)
    kw = dict(views=views, image_height=64, image_width=64, sh_degree=4, shared_campos=True)
    img, _ = rasterizer.rasterize_views(t(means), t(cov6), t(opac), t(shs), None, max_instances=L // 3, check="sync", **kw)
    assert np.abs(img[0].cpu().numpy() - f["image"]).mean() <= 1e-5          # transparently re-run with the exact size
    assert rasterizer.last_state().prm.max_instances == L
    rasterizer.rasterize_views(t(means), t(cov6), t(opac), t(shs), None, max_instances=L // 3, check="lazy", **kw)
    s = rasterizer.last_state()
    assert s.overflowed() and s.num_rendered() == L                          # flagged, memory-safe, no sync taken
    torch.cuda.synchronize()


def test_views_with_different_camera_centres(gpu):
    """V=3 cameras with different centres in one call (SH evaluated per view) == three single-view calls,
    forward bit-for-bit; gradients to summation order."""
    import math
    from splatter360_amd import cameras
    S, means, cov6, shs, opac = small_front_scene(n=300, seed=9, h=48, w=48)
    t = lambda a, g=False: torch.tensor(np.asarray(a, np.float32), device=gpu, requires_grad=g)
    ext = torch.eye(4).repeat(3, 1, 1)
    ext[0, :3, 3] = torch.tensor([0.2, 0.0, 0.0])
    ext[1, :3, 3] = torch.tensor([-0.1, 0.1, 0.3])
    ext[2, :3, 3] = torch.tensor([0.0, -0.2, -0.2])
    K = cameras.cube_face_intrinsics(1)[0, :3]
    vs = cameras.view_setup(ext, K, torch.ones(3), torch.full((3,), 100.0), scale_invariant=False)
    views = rasterizer.pack_views(vs["view_matrix"], vs["full_projection"], vs["campos"], vs["tan_fov_x"], vs["tan_fov_y"],
                                  torch.tensor([0.1, 0.2, 0.3])).to(gpu)
    gimg = torch.randn(3, 3, 48, 48, device=gpu)
    a = [t(means, True), t(cov6, True), t(opac, True), t(shs, True)]
    imgs, radii = rasterizer.rasterize_views(a[0], a[1], a[2], a[3], None, views=views, image_height=48, image_width=48,
                                             sh_degree=4, shared_campos=False)
    imgs.backward(gimg)
    b = [t(means, True), t(cov6, True), t(opac, True), t(shs, True)]
    outs = []
    for v in range(3):
        o, _ = rasterizer.rasterize_views(b[0], b[1], b[2], b[3], None, views=views[v:v + 1], image_height=48, image_width=48,
                                          sh_degree=4, shared_campos=True)
        outs.append(o[0])
    ref = torch.stack(outs)
    ref.backward(gimg)
    assert torch.equal(imgs, ref) and radii.shape == (3, 300)
    for x, y in zip(a, b):
        assert (x.grad - y.grad).abs().max().item() <= 2e-5 * (y.grad.abs().max().item() + 1e-12)


def test_opacity_extremes_and_degree_below_four(gpu):
    S, means, cov6, shs, opac = small_front_scene(n=200, seed=11, h=48, w=48, d_sh=25)
    opac = opac.copy()
    opac[:50] = 0.0          # never contributes (255*o <= 1): culled everywhere
    opac[50:100] = 1.0
    opac[100:120] = 0.003    # below 1/255 even at the centre
    for deg in (0, 1, 2, 3):
        S2 = dict(S, sh_degree=deg)   # 25 coefficients stored, only (deg+1)^2 active
        gimg = np.random.default_rng(deg).standard_normal((3, 48, 48)).astype(np.float32)
        f, og = _orc(S2, means, cov6, shs, opac, gimg)
        hh = run_hip(S2, means, cov6, shs, opac, gpu, grad_image=gimg)
        check_forward(hh, f, 200, 48, 48)
        check_grads(hh["grads"], og, rtol=5e-4)
        assert np.abs(hh["grads"]["shs"][:, (deg + 1) ** 2:]).max() == 0.0


def test_concurrent_calls_from_two_host_threads_on_two_streams(gpu):
    """The library is re-entrant across host threads / streams of one device (the reference is called from the
    Lightning main thread and from autograd's backward thread): two threads render + backpropagate different
    clouds concurrently on their own streams; results equal the serial ones bit for bit."""
    import threading
    from splatter360_amd import decoder, synthetic
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    bg = torch.zeros(3, device=gpu)
    views = decoder.pack_camera_views(ext, K, near, far, bg)
    clouds = [synthetic.uniform_cloud(30000 + 5000 * i, seed=40 + i, extent=2.5, scale_range=(0.02, 0.3)) for i in range(2)]

    def run(i, out, stream=None):
        ps = [torch.tensor(clouds[i][k], device=gpu, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(3):
                for p in ps:
                    p.grad = None
                f = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, views=views)
                (f * f).sum().backward()
            if stream is not None:
                stream.synchronize()
        out[i] = [f.detach().clone()] + [p.grad.clone() for p in ps]

    serial, conc = {}, {}
    for i in range(2):
        run(i, serial)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    th = [threading.Thread(target=run, args=(i, conc, streams[i])) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    for i in range(2):
        for a, b in zip(serial[i], conc[i]):
            assert torch.equal(a, b)
