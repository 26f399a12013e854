"""s360_pack_views (one kernel for the camera records of a call) against the reference's torch camera glue
(cameras.view_setup, golden-pinned on CPU by tests/test_golden_glue.py)."""
import numpy as np
import pytest
import torch

from splatter360_amd import cameras, decoder, rasterizer, synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("parity_lists")]   # integer state is compared with the oracle: upstream-compatible lists


def _random_poses(n, seed):
    rng = np.random.default_rng(seed)
    r = synthetic._random_rotations(rng, n)
    m = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    m[:, :3, :3] = r
    m[:, :3, 3] = rng.uniform(-2, 2, (n, 3))
    return torch.tensor(m)


def _ulp_diff(a, b):
    """max |a - b| in units of the float32 spacing at max(|a|, |b|, tiny)."""
    a, b = a.double(), b.double()
    mag = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
    spacing = 2.0 ** (torch.floor(torch.log2(mag)) - 23)
    return ((a - b).abs() / spacing).max().item()


def test_native_views_match_torch_glue_on_random_cameras(gpu):
    n = 37
    ext = _random_poses(n, 0).to(gpu)
    rng = np.random.default_rng(1)
    K = torch.zeros(n, 3, 3)
    K[:, 0, 0] = torch.tensor(rng.uniform(0.3, 1.2, n), dtype=torch.float32)
    K[:, 1, 1] = torch.tensor(rng.uniform(0.3, 1.2, n), dtype=torch.float32)
    K[:, 0, 2] = torch.tensor(rng.uniform(0.4, 0.6, n), dtype=torch.float32)
    K[:, 1, 2] = torch.tensor(rng.uniform(0.4, 0.6, n), dtype=torch.float32)
    K[:, 2, 2] = 1
    K = K.to(gpu)
    near = torch.tensor(rng.uniform(0.05, 0.5, n), dtype=torch.float32, device=gpu)
    far = near * torch.tensor(rng.uniform(20, 200, n), dtype=torch.float32, device=gpu)
    bg = torch.rand(n, 3, device=gpu)
    for si in (True, False):
        want = decoder.pack_camera_views_torch(ext, K, near, far, bg, scale_invariant=si)
        got = rasterizer.pack_views_native(ext, K, near, far, bg, scale_invariant=si)
        assert got.shape == want.shape == (n, rasterizer.VIEW_FLOATS)
        # exact fields: campos, background, scale, near / far (pure copies or one multiply)
        for sl in (slice(32, 35), slice(37, 40), slice(40, 43)):
            assert torch.equal(got[:, sl], want[:, sl])
        # matrices: a few ulp of the row scale (each entry is a short sum of products of O(scale) terms)
        for sl in (slice(0, 16), slice(16, 32)):
            scale = want[:, sl].abs().amax(dim=1, keepdim=True)
            assert ((got[:, sl] - want[:, sl]).abs() / scale).max().item() <= 2e-6
        assert _ulp_diff(got[:, 35:37], want[:, 35:37]) <= 4          # tan(fov/2)
    one_bg = torch.tensor([0.1, 0.2, 0.3], device=gpu)
    got = rasterizer.pack_views_native(ext, K, near, far, one_bg)
    assert torch.equal(got[:, 37:40], one_bg[None].expand(n, 3))


def test_native_views_of_cube_faces_render_like_the_torch_glue(gpu):
    """Six 90-degree face cameras of translated panoramas: tan(fov/2) is exactly 1, the records agree to <= 2 ulp and the
    rendered faces to <= 1e-6 (same binning on this cloud)."""
    cloud = synthetic.uniform_cloud(20_000, seed=7, extent=2.5, scale_range=(0.02, 0.25))
    ps = [torch.tensor(cloud[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    for pos in ((0.0, 0.0, 0.0), (0.1, -0.2, 0.05), (1.3, 0.7, -2.1)):
        pose = torch.tensor(synthetic.target_pano_pose(pos), device=gpu)
        ext, K, near, far = decoder.cube_cameras(pose, 0.1, 10.0)
        bg = torch.tensor([0.1, 0.2, 0.3], device=gpu)
        vt = decoder.pack_camera_views(ext, K, near, far, bg, glue="torch")
        vn = decoder.pack_camera_views(ext, K, near, far, bg)
        assert torch.equal(vn[:, 35:37], torch.ones(6, 2, device=gpu)) and torch.equal(vt[:, 35:37], vn[:, 35:37])
        scale = vt[:, :32].abs().amax(dim=1, keepdim=True)
        assert ((vn[:, :32] - vt[:, :32]).abs() / scale).max().item() <= 5e-7
        a = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, views=vt, shared_campos=True)
        ta = rasterizer.last_state().tensors()["tiles_touched"].clone()
        b = decoder.render_views_fused(ext, K, near, far, (64, 64), bg, *ps, views=vn, shared_campos=True)
        tb = rasterizer.last_state().tensors()["tiles_touched"]
        assert (a - b).abs().max().item() <= 1e-6
        assert (ta != tb).float().mean().item() <= 1e-4


def test_pack_views_rejects_bad_arguments(gpu):
    ext = torch.eye(4, device=gpu)[None]
    K = cameras.cube_face_intrinsics(1, device=gpu)[0, :1]
    n, f = torch.tensor([0.1], device=gpu), torch.tensor([10.0], device=gpu)
    with pytest.raises(RuntimeError):
        rasterizer.pack_views_native(ext, K, n, f, torch.zeros(4, device=gpu))
    with pytest.raises(RuntimeError):
        rasterizer.pack_views_native(ext.cpu(), K, n, f, torch.zeros(3))
