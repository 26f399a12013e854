"""Factored multi-GPU gradient exchange (s360_backward_split + s360_sh_backward,
distributed.sync_gradients_factored) against the plain "all-reduce everything" path: same gradients up to
float summation order, checked (a) in one process by playing both ranks, (b) with two gloo processes sharing
the one GPU of the box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

POSITIONS = ((0.0, 0.0, 0.0), (0.35, -0.1, 0.2))


def _cloud(dev, w=128, seed=3):
    from splatter360_amd import synthetic
    cloud = synthetic.encoder_like_cloud(w // 2, w, seed=seed)
    return [torch.tensor(cloud[k], device=dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]


def _cams(dev, pos):
    from splatter360_amd import decoder, synthetic
    pano = torch.from_numpy(synthetic.target_pano_pose(pos)).to(dev)
    return decoder.cube_cameras(pano, 0.1, 10.0)


def _render_backward(dev, ps, pos, seed, defer, exchange=None):
    from splatter360_amd import decoder, rasterizer
    ext, K, near, far = _cams(dev, pos)
    faces = decoder.render_views_fused(ext, K, near, far, (64, 64), torch.zeros(3, device=dev), *ps, defer_sh=defer, exchange=exchange,
                                       shared_campos=True)
    g = torch.Generator(device="cpu").manual_seed(seed)
    faces.backward(torch.randn(faces.shape, generator=g).to(dev))
    return rasterizer.last_deferred() if defer else None


def _close(a, b, tol=2e-5):
    scale = b.abs().max().item() + 1e-20
    assert (a - b).abs().max().item() / scale <= tol


def test_factored_equals_plain_sum_single_process(gpu):
    from splatter360_amd import rasterizer
    ps = _cloud(gpu)
    for r, pos in enumerate(POSITIONS):          # plain: autograd accumulates both "ranks"
        _render_backward(gpu, ps, pos, 10 + r, False)
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    defs = [_render_backward(gpu, ps, pos, 10 + r, True) for r, pos in enumerate(POSITIONS)]
    assert ps[2].grad is None                     # SH gradient deferred
    rgbs = []
    for r, d in enumerate(defs):
        rgb = d.d_rgb_sum.clone()
        w = rgb[:, 3].view(torch.int32)
        assert bool(((w >= -1) & (w < 6)).all())
        rgb[:, 3] = torch.where(w >= 0, torch.full_like(w, r), torch.full_like(w, -1)).view(torch.float32)
        rgbs.append(rgb)
    views = torch.stack([d.views[0] for d in defs])
    d_sh = rasterizer.finish_deferred_sh(defs[0].prm, views, defs[0].means3D, defs[0].shs, torch.stack(rgbs))
    _close(d_sh, want[2])
    _close(ps[0].grad, want[0])
    _close(ps[1].grad, want[1])
    _close(ps[3].grad, want[3])


def test_world_size_one_sync(gpu):
    from splatter360_amd import distributed as D
    ps = _cloud(gpu)
    _render_backward(gpu, ps, POSITIONS[1], 5, False)
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    d = _render_backward(gpu, ps, POSITIONS[1], 5, True)
    D.sync_gradients_factored(*ps, d)
    for p, w in zip(ps, want):
        _close(p.grad, w, 1e-6)


@pytest.mark.parametrize("n_chunks", [1, 4, 7])
def test_chunked_exchange_world_size_one_is_the_plain_backward_bit_for_bit(gpu, n_chunks):
    """exchange=ExchangeConfig() with one rank: composite + per-range tails + per-range SH rebuild + unpack must reproduce the
    one-call backward exactly (same kernels, same arithmetic, ranges only change which launch computes a Gaussian)."""
    from splatter360_amd import distributed as D
    ps = _cloud(gpu)
    assert ps[0].shape[0] == 16384      # 64 workgroups: ragged and uneven ranges for n_chunks = 7
    _render_backward(gpu, ps, POSITIONS[1], 5, False)
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    _render_backward(gpu, ps, POSITIONS[1], 5, False, exchange=D.ExchangeConfig(n_chunks=n_chunks))
    for p, w in zip(ps, want):
        assert torch.equal(p.grad, w)


def test_chunked_exchange_with_frozen_harmonics_skips_the_sh_exchange(gpu):
    """Harmonics that do not require grad: the exchange path neither gathers dL/dRGB nor rebuilds dL/dSH, the other
    three gradients are those of the plain backward — including the view-direction term of dL/dmean, which needs no dL/dSH."""
    from splatter360_amd import distributed as D
    ps = _cloud(gpu)
    ps[2] = ps[2].detach()                       # frozen harmonics
    _render_backward(gpu, ps, POSITIONS[0], 6, False)
    want = [p.grad.clone() if p.requires_grad else None for p in ps]
    for p in ps:
        p.grad = None
    _render_backward(gpu, ps, POSITIONS[0], 6, False, exchange=D.ExchangeConfig(n_chunks=3))
    assert ps[2].grad is None
    for p, w in zip(ps, want):
        if w is not None:
            assert torch.equal(p.grad, w)


def test_chunked_exchange_with_depth_gradient_and_upstream_layouts(gpu):
    """The in-backward exchange on the other input forms: the fused depth map's gradient (dL_ddepth through
    s360_backward_gaussians) and the upstream rasteriser layouts ([P,6] covariance, [P,25,3] harmonics) — world size 1, so the
    result must equal the one-call backward exactly."""
    from splatter360_amd import decoder, distributed as D, rasterizer
    ps = _cloud(gpu)
    ext, K, near, far = _cams(gpu, POSITIONS[1])
    gen = torch.Generator(device="cpu").manual_seed(3)
    wc, wd = torch.randn((6, 3, 64, 64), generator=gen).to(gpu), torch.randn((6, 64, 64), generator=gen).to(gpu)
    res = []
    for ex in (None, D.ExchangeConfig(n_chunks=5)):
        for p in ps:
            p.grad = None
        col, dep = decoder.render_views_fused(ext, K, near, far, (64, 64), torch.zeros(3, device=gpu), *ps, depth_mode="disparity",
                                              shared_campos=True, exchange=ex)
        ((col * wc).sum() + (dep * wd).sum()).backward()
        res.append([p.grad.clone() for p in ps])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # upstream layouts through rasterize_views
    views = decoder.pack_camera_views(ext, K, near, far, torch.zeros(3, device=gpu))
    r, c = torch.triu_indices(3, 3)
    res = []
    for ex in (None, D.ExchangeConfig(n_chunks=3)):
        m = ps[0].detach().clone().requires_grad_(True)
        c6 = ps[1].detach()[:, r, c].clone().requires_grad_(True)
        sh = ps[2].detach().transpose(1, 2).contiguous().requires_grad_(True)
        op = ps[3].detach().clone().requires_grad_(True)
        img, _ = rasterizer.rasterize_views(m, c6, op, sh, views=views, image_height=64, image_width=64, sh_degree=4,
                                            shared_campos=True, exchange=ex)
        (img * wc).sum().backward()
        res.append([t.grad.clone() for t in (m, c6, sh, op)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):      # per-view camera centres cannot use the factored exchange
        rasterizer.rasterize_views(ps[0], ps[1], ps[3], ps[2], views=views, image_height=64, image_width=64, sh_degree=4,
                                   shared_campos=False, cov9=True, sh_channel_major=True, exchange=D.ExchangeConfig())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), S360_DIST_BACKEND="gloo", S360_FORCE_DEVICE="0")
    from splatter360_amd import distributed as D
    D.init()
    dev = torch.device("cuda:0")
    ps = _cloud(dev)
    _render_backward(dev, ps, POSITIONS[rank], 10 + rank, False)
    D.allreduce_gradients([p.grad for p in ps])
    plain = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    d = _render_backward(dev, ps, POSITIONS[rank], 10 + rank, True)
    D.sync_gradients_factored(*ps, d)
    err = []
    for p, w in zip(ps, plain):
        err.append((p.grad - w).abs().max().item() / (w.abs().max().item() + 1e-20))
    # the chunked exchange INSIDE the rasteriser node's backward (s360_backward_composite / _gaussians / s360_sh_backward per
    # Gaussian range, ragged last range): .grad comes back already summed over the ranks
    for p in ps:
        p.grad = None
    _render_backward(dev, ps, POSITIONS[rank], 10 + rank, False, exchange=D.ExchangeConfig(n_chunks=3))
    for p, w in zip(ps, plain):
        err.append((p.grad - w).abs().max().item() / (w.abs().max().item() + 1e-20))
    # ... in its other form too (world 2 defaults to ONE coalesced all-gather per range; "reduce" = all-reduce + all-gather)
    for p in ps:
        p.grad = None
    _render_backward(dev, ps, POSITIONS[rank], 10 + rank, False, exchange=D.ExchangeConfig(n_chunks=2, mode="reduce"))
    for p, w in zip(ps, plain):
        err.append((p.grad - w).abs().max().item() / (w.abs().max().item() + 1e-20))
    # the RAW path: every rank renders its own target panorama from the same encoder outputs; rasterize_raw(exchange=...) returns the
    # gradients w.r.t. depths / opacities / raw records summed over the ranks (k_raw_bwd fed both ranks' dL/dRGB factors)
    from splatter360_amd import adapter, decoder, rasterizer
    gen = torch.Generator().manual_seed(21)
    hw, nv = (24, 48), 2
    n = hw[0] * hw[1]
    rdep = torch.exp(torch.empty(nv, n).uniform_(-0.2, 1.8, generator=gen)).to(dev)
    rop = torch.sigmoid(torch.randn(nv, n, generator=gen)).to(dev)
    rraw = torch.randn(nv, n, 82, generator=gen)
    rraw[..., 7:] *= 0.7
    rraw = rraw.to(dev)
    cext = torch.eye(4).repeat(nv, 1, 1)
    cext[0, :3, 3] = torch.tensor([-0.3, 0.0, 0.1]); cext[1, :3, 3] = torch.tensor([0.3, 0.05, -0.1])
    cext = cext.to(dev)
    rot = adapter.sh_rotation_blocks(cext, 25)
    e6, K6, n6, f6 = _cams(dev, POSITIONS[rank])
    views = decoder.pack_camera_views(e6, K6, n6, f6, torch.zeros(3, device=dev))
    gw = torch.randn((6, 3, 64, 64), generator=torch.Generator().manual_seed(40 + rank)).to(dev)

    def raw_step(exchange):
        leaves = [t.clone().requires_grad_(True) for t in (rdep, rop, rraw)]
        img = rasterizer.rasterize_raw(leaves[0].reshape(-1), leaves[1].reshape(-1), leaves[2].reshape(-1, 82), cext, views=views, image_height=64,
                                       image_width=64, context_shape=hw, scale_min=0.5, scale_max=15.0, sh_rotation=rot, exchange=exchange)[0]
        (img * gw).sum().backward()
        return [t.grad for t in leaves]

    plain_raw = raw_step(None)
    D.allreduce_gradients(plain_raw)
    for ex in (D.ExchangeConfig(), D.ExchangeConfig(n_chunks=2, mode="reduce")):
        for a, w in zip(raw_step(ex), plain_raw):
            err.append((a - w).abs().max().item() / (w.abs().max().item() + 1e-20))
    q.put((rank, err, float(plain[2].abs().sum().item())))
    torch.distributed.destroy_process_group()


def test_two_rank_factored_sync_gloo_on_one_gpu(gpu):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=600) for _ in range(world)]
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, err, mass in outs:
        assert mass > 0
        assert max(err) <= 2e-5, (rank, err)


def test_bench_two_ranks_on_one_gpu_end_to_end(gpu):
    """`python bench.py --gpus 2` started plainly on a one-GPU box (gloo, both ranks pinned to cuda:0): the self-launch, the
    chunked in-backward exchange through the real kernels, the timing protocol and the JSON line — small cloud, two steps."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(S360_DIST_BACKEND="gloo", S360_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-baseline", "0",
                        "--pano-h", "64", "--face", "64"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["rank_devices"] == [0, 0]
    assert d["value"] > 0 and d["scaling"] == "weak" and d["exchange"]["mode"] == "chunked" and d["forward_only"]["value"] > 0
    assert "chunked exchange inside the backward" in d["config"]["parallelism"]
