"""S360_FLAG_COOP_WALK (include/s360.h): the binning kernels compiled for clouds of LARGE footprints — a rectangle of more than 32
tiles is counted (k_preprocess) and emitted (k_emit) by all 64 lanes of its wave.  Every observable of the call must be
bit-identical with and without the flag (VERDICT r04 "next" #6: `torch.equal` between the variants), the adaptive switch follows
the previous call's count of such rectangles (word 2 of the pinned header mirror), and the default kernels stay the default for
the encoder-like cloud."""
import pytest
import torch

from splatter360_amd import _lib, decoder, rasterizer, synthetic

pytestmark = pytest.mark.gpu


class _Switch:
    def __init__(self, coop, lean=True):
        self.want = (coop, lean)

    def __enter__(self):
        self.old = (rasterizer.COOP_WALK, rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS)
        rasterizer.COOP_WALK, rasterizer.LEAN_LISTS = self.want
        rasterizer.SPLIT_LONG_LISTS = False        # (its adaptive switch may flip between two calls: not what is compared here)

    def __exit__(self, *a):
        rasterizer.COOP_WALK, rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS = self.old


def _step(params, dev, face=128, train=True):
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
    ps = [p.clone().requires_grad_(train) for p in params]
    faces = decoder.render_views_fused(ext, K, near, far, (face, face), torch.zeros(3, device=dev), *ps, shared_campos=True)
    st = rasterizer.last_state()
    if train:
        ((faces - 0.5) ** 2).mean().backward()
    torch.cuda.synchronize()
    t = st.tensors()
    n = st.num_rendered()
    return dict(faces=faces.detach(), grads=[p.grad for p in ps] if train else [], st=st, n=n, tile_start=t["tile_start"].clone(),
                tiles_touched=t["tiles_touched"].clone(), list=t["list"][:n].clone(), keys=t["keys"][:n].clone(),
                final_T=t["final_T"].clone(), n_contrib=t["n_contrib"].clone(), wide=int(st.header()[4].item()))


@pytest.fixture(scope="module")
def near_cloud(gpu):
    """131 072 Gaussians U[-2,2]^3: thousands of splats close enough to the origin to cover dozens to all 64 tiles of a 128^2 face."""
    c = synthetic.uniform_cloud(1 << 17, seed=3, extent=2.0)
    return [torch.tensor(c[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("train", [True, False])
def test_cooperative_walk_is_bit_identical_to_the_default_binning(gpu, near_cloud, lean, train):
    with _Switch(False, lean):
        a = _step(near_cloud, gpu, train=train)
    with _Switch(True, lean):
        b = _step(near_cloud, gpu, train=train)
    assert not (a["st"].prm.flags & _lib.FLAG_COOP_WALK) and (b["st"].prm.flags & _lib.FLAG_COOP_WALK)
    assert a["wide"] == b["wide"] and a["wide"] > 1000, a["wide"]        # the regime the variant exists for
    assert a["n"] == b["n"]
    for k in ("tile_start", "tiles_touched", "list", "keys", "final_T", "n_contrib", "faces"):     # (sorted list + keys: the binning's result)
        assert torch.equal(a[k], b[k]), k
    for x, y in zip(a["grads"], b["grads"]):
        assert torch.equal(x, y)


def test_auto_switch_follows_the_previous_calls_wide_rectangle_count(gpu, near_cloud):
    enc = synthetic.encoder_like_cloud(128, 256)
    enc = [torch.tensor(enc[k], device=gpu) for k in ("means", "covariances", "harmonics", "opacities")]
    with _Switch("auto"):
        for d in (rasterizer._MIRRORS,):
            d.clear()
        first = _step(near_cloud, gpu)                  # nothing known about the shape: the default kernels
        assert not (first["st"].prm.flags & _lib.FLAG_COOP_WALK)
        key = rasterizer._hint_key(gpu, first["st"].prm.P, 6, 128, 128, True)
        assert int(rasterizer._MIRRORS[key][2]) == first["wide"] >= rasterizer.AUTO_COOP_MIN_PAIRS     # the forward's report (word 2)
        second = _step(near_cloud, gpu)
        assert second["st"].prm.flags & _lib.FLAG_COOP_WALK
        assert torch.equal(first["faces"], second["faces"])
        for x, y in zip(first["grads"], second["grads"]):
            assert torch.equal(x, y)
        e1 = _step(enc, gpu)
        e2 = _step(enc, gpu)
        assert e1["wide"] < rasterizer.AUTO_COOP_MIN_PAIRS and not (e2["st"].prm.flags & _lib.FLAG_COOP_WALK)
