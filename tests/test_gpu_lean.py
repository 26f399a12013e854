"""Lean tile lists (S360_FLAG_LEAN_LISTS, the product default) against the upstream-compatible lists (flag clear).

The binning-time cull drops a (Gaussian, tile) instance only when the splat cannot reach alpha >= 1/255 on any pixel of the tile,
so everything a caller can observe must come out BIT-IDENTICAL — images, depth maps, radii, the fused loss and every gradient —
while tiles_touched / the sorted lists / num_rendered shrink.  Checked here:
  * torch.equal on all outputs and gradients, from a few dozen Gaussians to the headline sizes (1 M x six 256^2 faces,
    4 M x six 512^2 faces), drop-in and fused calls, colour + depth + fused L2 loss, cube and native-spherical modes;
  * the lean list of every tile is an ordered subsequence of the upstream list;
  * every dropped instance really is invisible: float64 re-evaluation of alpha on all 256 pixels of its tile stays < 1/255;
  * capacity overflow in lean mode is flagged and memory-safe.
Reference semantics preserved: SURVEY.md App. A.2; call site /root/reference/src/model/decoder/cuda_splatting.py:113-124."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import boundary_tensors, face_settings, small_front_scene
from splatter360_amd import decoder, rasterizer, synthetic
from test_gpu_parity import _settings_to_torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(autouse=True)
def _sequential_lists():
    """Bit-identity between the two list modes is a statement about the sequential composite: a quadrant that SPLITS its list
    (rasterizer.SPLIT_LONG_LISTS) does so at a list position, and positions differ between the modes — there the two agree to
    float rounding only (tests/test_gpu_saturating_parity.py).  Everything here runs with splitting off."""
    old = rasterizer.SPLIT_LONG_LISTS
    rasterizer.SPLIT_LONG_LISTS = False
    yield
    rasterizer.SPLIT_LONG_LISTS = old


def _report(key, **vals):
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        f = out / "lean_report.json"
        prev = json.loads(f.read_text()) if f.exists() else {}
        prev[key] = vals
        f.write_text(json.dumps(prev, indent=1, sort_keys=True))
    except OSError:
        pass


def _dropin(S, means, cov6, shs, opac, dev, lean, gimg):
    old = rasterizer.LEAN_LISTS
    rasterizer.LEAN_LISTS = lean
    try:
        t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=True)
        m, c, s, o = t(means), t(cov6), t(shs), t(opac)
        m2 = torch.zeros_like(m, requires_grad=True)
        rast = rasterizer.GaussianRasterizer(_settings_to_torch(S, dev))
        img, radii = rast(means3D=m, means2D=m2, shs=s, opacities=o, cov3D_precomp=c)
        st = rasterizer.last_state()
        tens = {k: (None if v is None else v.clone()) for k, v in st.tensors().items()}
        L = st.num_rendered()
        img.backward(torch.tensor(gimg, device=dev))
        return dict(img=img.detach(), radii=radii, grads=[x.grad for x in (m, m2, c, o, s)], t=tens, L=L)
    finally:
        rasterizer.LEAN_LISTS = old


def _check_lists(par, lean, P, H, W):
    """lean lists = ordered subsequences of the upstream lists; dropped instances are invisible on every pixel of their tile."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tp, tl = par["t"], lean["t"]
    sp = tp["tile_start"].cpu().numpy().astype(np.int64)
    sl = tl["tile_start"].cpu().numpy().astype(np.int64)
    lp = tp["list"].cpu().numpy().astype(np.int64)[:par["L"]]
    ll = tl["list"].cpu().numpy().astype(np.int64)[:lean["L"]]
    assert lean["L"] <= par["L"] and sl[-1] == lean["L"] and sp[-1] == par["L"]
    ra, rb = tp["rec_a"].cpu().numpy()[0], tp["rec_b"].cpu().numpy()[0]
    dropped = 0
    ys, xs = np.mgrid[0:16, 0:16]
    for t in range(gx * gy):
        a, b = lp[sp[t]:sp[t + 1]], ll[sl[t]:sl[t + 1]]
        keep = np.isin(a, b)
        np.testing.assert_array_equal(a[keep], b)          # same entries in the same (depth, index) order
        for g in a[~keep]:
            x, y, ca, cb = ra[g].astype(np.float64)
            cc, op = rb[g, :2].astype(np.float64)
            dx = x - (16 * (t % gx) + xs)
            dy = y - (16 * (t // gx) + ys)
            power = ca * dx * dx + cb * dx * dy + cc * dy * dy          # log2 of the Gaussian weight (pre-scaled conic)
            assert (op * np.exp2(power)).max() < 1.0 / 255.0
            dropped += 1
    assert dropped == par["L"] - lean["L"]
    # per-pair tile counts: lean <= upstream, same visible radii
    assert bool((tl["tiles_touched"] <= tp["tiles_touched"]).all())
    return dropped


def _equal_all(par, lean):
    assert torch.equal(par["img"], lean["img"])
    assert torch.equal(par["radii"], lean["radii"])
    for a, b in zip(par["grads"], lean["grads"]):
        assert torch.equal(a, b)
    assert torch.equal(par["t"]["final_T"], lean["t"]["final_T"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lean_small_scene(gpu, seed):
    S, means, cov6, shs, opac = small_front_scene(n=120, seed=seed, h=96, w=112, srange=(0.02, 0.6))
    gimg = np.random.default_rng(seed).standard_normal((3, 96, 112)).astype(np.float32)
    par = _dropin(S, means, cov6, shs, opac, gpu, False, gimg)
    lean = _dropin(S, means, cov6, shs, opac, gpu, True, gimg)
    _equal_all(par, lean)
    _check_lists(par, lean, 120, 96, 112)


@pytest.mark.parametrize("face", [0, 1, 5])
def test_lean_config0_faces(gpu, face):
    """BASELINE config 0 shape (10 k Gaussians, 64x64 faces), polar and equatorial faces."""
    cloud = synthetic.uniform_cloud(10_000, seed=3, extent=3.0, scale_range=(0.02, 0.3))
    S = face_settings(face, 64, 64)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    gimg = np.random.default_rng(face).standard_normal((3, 64, 64)).astype(np.float32)
    par = _dropin(S, means, cov6, shs, opac, gpu, False, gimg)
    lean = _dropin(S, means, cov6, shs, opac, gpu, True, gimg)
    _equal_all(par, lean)
    d = _check_lists(par, lean, 10_000, 64, 64)
    assert d > 0      # the cull does something on this cloud


def test_lean_large_footprints(gpu):
    """Splats whose rectangles exceed 32 tiles (binned whole: no hit mask) and faint ones that reach no tile at all."""
    S, means, cov6, shs, opac = small_front_scene(n=80, seed=7, h=160, w=160, srange=(0.05, 2.5), zrange=(3.0, 6.0))
    opac = opac.copy()
    opac[::5] = 0.003          # alpha < 1/255 everywhere: visible by rectangle, on no tile list in lean mode
    gimg = np.random.default_rng(7).standard_normal((3, 160, 160)).astype(np.float32)
    par = _dropin(S, means, cov6, shs, opac, gpu, False, gimg)
    lean = _dropin(S, means, cov6, shs, opac, gpu, True, gimg)
    _equal_all(par, lean)
    _check_lists(par, lean, 80, 160, 160)
    tt_p, tt_l = par["t"]["tiles_touched"][0], lean["t"]["tiles_touched"][0]
    big = tt_p > 32
    assert int(big.sum()) > 5 and torch.equal(tt_l[big], tt_p[big])      # rectangles beyond the 32-bit hit mask are binned whole
    faint = torch.zeros(80, dtype=torch.bool, device=gpu)
    faint[::5] = True
    small_faint = faint & ~big & (tt_p > 0)
    assert int(small_faint.sum()) > 0 and int(tt_l[small_faint].sum()) == 0
    assert torch.equal(par["radii"], lean["radii"])      # radii are upstream's (the rectangle's), not the lists'


@pytest.mark.parametrize("seed", [0, 1])
def test_lean_thin_diagonal_splats_near_the_32_tile_limit(gpu, seed):
    """needle-like diagonal splats (axis ratios 20:1 ... 300:1) whose rectangles span close to 32 tiles — the
    exponent's cancelling terms a dx^2, b dx dy, c dy^2 reach 1e4 ... 1e5 there and float32 rounding approaches the lean cull's
    margin.  Beyond the bound in k_preprocess (lean_safe) such a splat is binned whole; everything observable stays bit-identical
    and every dropped instance is still invisible in float64."""
    rng = np.random.default_rng(100 + seed)
    n, hw = 160, 160
    S, _, _, shs, _ = small_front_scene(n=n, seed=seed, h=hw, w=hw)
    z = rng.uniform(3.0, 6.0, n)
    means = np.stack([rng.uniform(-0.45, 0.45, n) * z, rng.uniform(-0.45, 0.45, n) * z, z], 1)
    ang = np.pi / 4 + rng.uniform(-0.3, 0.3, n) + (rng.integers(0, 2, n) * np.pi / 2)
    major = rng.uniform(22.0, 47.0, n) / 3.0 / (hw / 2) * z          # 3-sigma half-length of 22 ... 47 pixels: up to 6 tiles a side
    minor = major / np.exp(rng.uniform(np.log(20.0), np.log(300.0), n))
    c, s_ = np.cos(ang), np.sin(ang)
    R = np.zeros((n, 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s_, s_, c, 1.0
    sc = np.stack([major, minor, minor], 1)
    cov = np.einsum("nij,nj,nkj->nik", R, sc * sc, R)
    rr, cc = np.triu_indices(3)
    cov6 = cov[:, rr, cc]
    opac = rng.uniform(0.3, 0.99, (n, 1))
    gimg = rng.standard_normal((3, hw, hw)).astype(np.float32)
    par = _dropin(S, means, cov6, shs, opac, gpu, False, gimg)
    lean = _dropin(S, means, cov6, shs, opac, gpu, True, gimg)
    tt, ttl = par["t"]["tiles_touched"][0], lean["t"]["tiles_touched"][0]
    ra, rb = (par["t"][k][0].double() for k in ("rec_a", "rec_b"))
    rad = par["t"]["rec_c"][0][:, 1].contiguous().view(torch.int32).double()
    mag = (ra[:, 2].abs() + ra[:, 3].abs() + rb[:, 0].abs()) * (rad + 16.0) ** 2          # the bound k_preprocess evaluates
    risky = (mag >= 1.0e4) & (tt > 0) & (tt <= 32)
    assert int(((tt > 12) & (tt <= 32)).sum()) > 20 and int(risky.sum()) > 3      # the regime the finding is about is populated
    assert torch.equal(ttl[risky], tt[risky])                                      # ... and those splats are binned whole
    _equal_all(par, lean)
    _check_lists(par, lean, n, hw, hw)


def _fused(params, cams, fw, dev, lean, depth_mode="depth", target=None, max_instances=None, check="sync"):
    ps = [p.clone().requires_grad_(True) for p in params]
    ext, K, near, far = cams
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    out = decoder.render_views_fused(ext, K, near, far, (fw, fw), bg, *ps, shared_campos=True, depth_mode=depth_mode, mse_target=target,
                                     lean=lean, max_instances=max_instances, check=check)
    st = rasterizer.last_state()
    L = st.num_rendered()
    vis = int((st.tensors()["tiles_touched"] > 0).sum())
    if target is not None:
        col, dep, fm = out
        (fm.loss + 0.01 * (dep * dep).mean()).backward()
        extra = [fm.loss.detach(), fm.clipped_mse.detach()]
    else:
        col, dep = out
        g = torch.Generator(device="cpu").manual_seed(5)
        w = torch.randn(col.shape, generator=g).to(dev)
        ((col * w).sum() + 0.01 * (dep * dep).mean()).backward()
        extra = []
    return dict(col=col.detach(), dep=dep.detach(), grads=[p.grad for p in ps], L=L, vis=vis, extra=extra)


def _fused_equal(a, b):
    assert torch.equal(a["col"], b["col"]) and torch.equal(a["dep"], b["dep"])
    for x, y in zip(a["extra"], b["extra"]):
        assert torch.equal(x, y)
    for x, y in zip(a["grads"], b["grads"]):
        assert torch.equal(x, y)


def test_lean_fused_six_faces_1m(gpu):
    """Headline shape: 1 048 576 Gaussians, six 256x256 faces, colour + depth + fused L2 loss, all four gradients."""
    cloud = synthetic.encoder_like_cloud(512, 1024, seed=0)
    params = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    cams = decoder.cube_cameras(torch.tensor(synthetic.target_pano_pose((0.02, -0.01, 0.03)), device=gpu), 0.1, 10.0)
    target = torch.full((6, 3, 256, 256), 0.5, device=gpu)
    par = _fused(params, cams, 256, gpu, False, target=target)
    lean = _fused(params, cams, 256, gpu, True, target=target)
    _fused_equal(par, lean)
    _report("1m_six_faces_256", num_rendered_parity=par["L"], num_rendered_lean=lean["L"], ratio=lean["L"] / par["L"],
            visible_pairs_parity=par["vis"], visible_pairs_lean=lean["vis"])
    assert lean["L"] < 0.9 * par["L"]
    # without the loss epilogue, random image gradient
    par = _fused(params, cams, 256, gpu, False)
    lean = _fused(params, cams, 256, gpu, True)
    _fused_equal(par, lean)


def test_lean_fused_six_faces_4m_512(gpu):
    """BASELINE configs[4] single-rank shape: 4 194 304 Gaussians, six 512x512 faces."""
    cloud = synthetic.encoder_like_cloud(1024, 2048, seed=0)
    params = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    del cloud
    cams = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    par = _fused(params, cams, 512, gpu, False)
    lean = _fused(params, cams, 512, gpu, True)
    _fused_equal(par, lean)
    _report("4m_six_faces_512", num_rendered_parity=par["L"], num_rendered_lean=lean["L"], ratio=lean["L"] / par["L"],
            visible_pairs_parity=par["vis"], visible_pairs_lean=lean["vis"])


def test_lean_inference_call(gpu):
    cloud = synthetic.uniform_cloud(50_000, seed=2, extent=3.0, scale_range=(0.02, 0.3))
    params = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    with torch.no_grad():
        a = decoder.render_views_fused(ext, K, near, far, (128, 128), torch.zeros(3, device=gpu), *params, shared_campos=True, lean=False)
        la = rasterizer.last_state().num_rendered()
        b = decoder.render_views_fused(ext, K, near, far, (128, 128), torch.zeros(3, device=gpu), *params, shared_campos=True, lean=True)
        lb = rasterizer.last_state().num_rendered()
    assert torch.equal(a, b) and lb < la


def test_lean_spherical(gpu):
    cloud = synthetic.uniform_cloud(20_000, seed=4, extent=3.0, scale_range=(0.02, 0.2))
    params = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    pose = torch.tensor(synthetic.target_pano_pose((0.1, 0.0, -0.1)), device=gpu)
    res = []
    for lean in (False, True):
        old = rasterizer.LEAN_LISTS
        rasterizer.LEAN_LISTS = lean
        try:
            ps = [p.clone().requires_grad_(True) for p in params]
            img = decoder.render_erp_spherical(pose, 0.1, (128, 256), torch.zeros(3, device=gpu), *ps)
            L = rasterizer.last_state().num_rendered()
            (img * torch.linspace(-1, 1, img.numel(), device=gpu).view_as(img)).sum().backward()
            res.append((img.detach(), [p.grad for p in ps], L))
        finally:
            rasterizer.LEAN_LISTS = old
    assert torch.equal(res[0][0], res[1][0]) and res[1][2] < res[0][2]
    for x, y in zip(res[0][1], res[1][1]):
        assert torch.equal(x, y)


def test_lean_overflow_is_flagged_and_resized(gpu):
    cloud = synthetic.uniform_cloud(30_000, seed=6, extent=3.0, scale_range=(0.05, 0.4))
    params = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    cams = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    full = _fused(params, cams, 128, gpu, True)
    L = full["L"]
    # lazy: too small a capacity is flagged (no fault, no hang)
    ps = [p.clone().requires_grad_(True) for p in params]
    col = decoder.render_views_fused(*cams, (128, 128), torch.zeros(3, device=gpu), *ps, shared_campos=True, lean=True,
                                     max_instances=max(1024, L // 3), check="lazy")
    col.sum().backward()
    torch.cuda.synchronize()
    st = rasterizer.last_state()
    assert st.overflowed() and st.num_rendered() == L
    # sync: re-rendered with the exact size, identical result
    again = _fused(params, cams, 128, gpu, True, max_instances=max(1024, L // 3))
    _fused_equal(full, again)


def test_large_footprint_pairs_are_gathered_wave_parallel_and_agree_with_the_oracle(gpu):
    """Pairs with more than 32 instance slots take k_gather_slots' second phase (a wave per pair, lane-strided sums + a fixed
    shuffle tree instead of the owner thread's serial chain): gradients against the float64 oracle, run-to-run bit-identical, and
    the same in both list modes (rectangles beyond 32 tiles are binned whole either way)."""
    from oracle import oracle
    S, means, cov6, shs, opac = small_front_scene(n=60, seed=12, h=176, w=160, srange=(0.4, 3.0), zrange=(2.5, 5.0))
    gimg = np.random.default_rng(12).standard_normal((3, 176, 160)).astype(np.float32)
    runs = [_dropin(S, means, cov6, shs, opac, gpu, lean, gimg) for lean in (True, True, False)]
    assert int((runs[0]["t"]["tiles_touched"] > 32).sum()) > 20
    _equal_all(runs[0], runs[1])
    _equal_all(runs[0], runs[2])
    o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
    o64.forward()
    g64 = o64.backward(gimg)
    got = dict(zip(("means3D", "means2D", "cov3D", "opacities", "shs"), runs[0]["grads"]))
    for k in ("means3D", "cov3D", "opacities", "shs"):
        w = np.asarray(g64[k], np.float64).reshape(-1)
        e = np.abs(got[k].cpu().numpy().astype(np.float64).reshape(-1) - w).max() / (np.abs(w).max() + 1e-30)
        assert e <= 2e-4, (k, e)


def test_count_contributions_matches_a_recount_from_the_workspace(gpu):
    """s360_count_contributions (the measurement aid behind bench.py's work-based VALU figure): contributing (pixel, entry) pairs
    recounted in numpy from the same workspace — sorted lists, splat records, n_contrib — with the composite's accept test
    (alpha >= 1/255, power <= 0, in front of the pixel's last contributor)."""
    S, means, cov6, shs, opac = small_front_scene(n=150, seed=4, h=48, w=64, srange=(0.03, 0.4))
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu, requires_grad=True)
    rast = rasterizer.GaussianRasterizer(_settings_to_torch(S, gpu))
    m = t(means)
    img, _ = rast(means3D=m, means2D=torch.zeros_like(m), shs=t(shs), opacities=t(opac), cov3D_precomp=t(cov6))
    st = rasterizer.last_state()
    got, evaluated = st.count_contributions()
    tt = st.tensors()
    ts = tt["tile_start"].cpu().numpy().astype(np.int64)
    lst = tt["list"].cpu().numpy().astype(np.int64)
    ra, rb = tt["rec_a"][0].cpu().numpy().astype(np.float64), tt["rec_b"][0].cpu().numpy().astype(np.float64)
    nc = tt["n_contrib"][0].cpu().numpy().astype(np.int64)
    gx = (64 + 15) // 16
    want = 0
    borderline = 0
    for tile in range(len(ts) - 1):
        ent = lst[ts[tile]:ts[tile + 1]]
        if ent.size == 0:
            continue
        ys, xs = np.mgrid[0:16, 0:16]
        py, px = 16 * (tile // gx) + ys, 16 * (tile % gx) + xs
        ok = (py < 48) & (px < 64)
        last = np.where(ok, nc[np.minimum(py, 47), np.minimum(px, 63)], 0)
        for pos, g in enumerate(ent):
            dx, dy = ra[g, 0] - px, ra[g, 1] - py
            power = ra[g, 2] * dx * dx + ra[g, 3] * dx * dy + rb[g, 0] * dy * dy
            alpha = np.minimum(0.99, rb[g, 1] * np.exp2(power))
            sel = ok & (pos < last) & (power <= 0)
            want += int((sel & (alpha >= 1.0 / 255.0)).sum())
            borderline += int((sel & (np.abs(alpha - 1.0 / 255.0) < 1e-6)).sum())
    assert abs(got - want) <= borderline + 1, (got, want, borderline)
    assert want > 1000 and evaluated >= got and evaluated % 64 == 0
