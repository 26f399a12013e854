"""world_size-2 gloo tests (CPU) of the N>1 path: view sharding + the gradient all-reduce.
The CPU oracle stands in for the renderer; the product's distributed helpers are what is tested."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import small_front_scene


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
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import oracle
    from splatter360_amd import distributed as D
    r, lr, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    S, means, cov6, shs, opac = small_front_scene(n=25, seed=9, h=32, w=32)
    n_views = 3
    mine = D.shard_views(n_views, rank, world)
    tot = [np.zeros_like(means), np.zeros_like(cov6), np.zeros_like(shs), np.zeros_like(opac)]
    for v in mine:  # "views" = the same camera with a per-view loss weight
        w_img = np.random.default_rng(100 + v).standard_normal((3, 32, 32))
        orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
        orc.forward()
        g = orc.backward(w_img)
        for t, k in zip(tot, ("means3D", "cov3D", "shs", "opacities")):
            t += g[k]
    grads = [torch.from_numpy(t.copy()) for t in tot] + [None]
    D.allreduce_gradients(grads)
    D.barrier()
    assert D.max_over_ranks(float(rank), "cpu") == world - 1
    q.put((rank, mine, [g.numpy() for g in grads[:4]]))
    torch.distributed.destroy_process_group()


def test_two_rank_view_sharding_and_gradient_allreduce():
    from oracle import oracle
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    outs.sort(key=lambda o: o[0])
    assert outs[0][1] == [0, 1] and outs[1][1] == [2]
    # single-process truth: sum over all three views
    S, means, cov6, shs, opac = small_front_scene(n=25, seed=9, h=32, w=32)
    want = [np.zeros_like(means), np.zeros_like(cov6), np.zeros_like(shs), np.zeros_like(opac)]
    for v in range(3):
        w_img = np.random.default_rng(100 + v).standard_normal((3, 32, 32))
        orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
        orc.forward()
        g = orc.backward(w_img)
        for t, k in zip(want, ("means3D", "cov3D", "shs", "opacities")):
            t += g[k]
    for r in range(world):
        for got, exp in zip(outs[r][2], want):
            np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-14)


def test_shard_views_partitions():
    from splatter360_amd.distributed import shard_views
    for n in (1, 3, 8, 18):
        for world in (1, 2, 4, 8):
            parts = [shard_views(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))


POS = ((0.0, 0.0, 0.0), (0.3, -0.1, 0.2))


def _rank_view(rank):
    """One 32x32 face view per rank from that rank's panorama centre; oracle gradients and the factors the split
    backward would export (d_rgb_sum = clamp-masked dL/dRGB, w = visibility)."""
    from helpers import boundary_tensors, face_settings
    from oracle import oracle
    from splatter360_amd import synthetic
    cloud = synthetic.uniform_cloud(300, seed=5, extent=2.0, scale_range=(0.05, 0.3))
    S = face_settings(1, 32, 32, position=POS[rank])
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64)
    f = o.forward()
    g = o.backward(np.random.default_rng(30 + rank).standard_normal((3, 32, 32)))
    vis = f["radii"] > 0
    drgb = np.where(f["clamped"].astype(bool), 0.0, g["raster_rgb"]) * vis[:, None]
    return cloud, S, g, drgb, vis


def _sh_pass_restated(prm, views, means3D, shs, d_rgb_sums):
    """torch restatement of what s360_sh_backward computes (dL/dSH = sum over groups of Y(dir) (x) dRGB; `shs` is only a
    shape template) — stands in for the HIP kernel in this CPU test of the exchange logic."""
    from oracle import torch_ref
    out = torch.zeros_like(shs)                       # [P, 25, 3]
    for j in range(d_rgb_sums.shape[0]):
        w = d_rgb_sums[j, :, 3].contiguous().view(torch.int32)
        assert bool(((w == j) | (w == -1)).all())     # .w was rewritten to the owning rank / group index
        campos, scale = views[j, 32:35], views[j, 40]
        d = means3D * scale - campos
        y = torch_ref.sh_basis(4, d / d.norm(dim=1, keepdim=True))           # [P, 25]
        out += (w >= 0).to(shs.dtype)[:, None, None] * y[:, :, None] * d_rgb_sums[j, :, None, :3]
    return out


def _factored_worker(rank, world, port, q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from splatter360_amd import distributed as D, rasterizer
    D.init(backend="gloo")
    rasterizer.finish_deferred_sh = _sh_pass_restated         # the HIP kernel's job, restated (no GPU here)
    cloud, S, g, drgb, vis = _rank_view(rank)
    t = lambda a: torch.tensor(np.asarray(a, np.float64))
    means, cov, op = t(cloud["means"]), t(cloud["covariances"]), t(cloud["opacities"])
    sh = torch.tensor(cloud["harmonics"], dtype=torch.float32).transpose(1, 2).contiguous()     # [P, 25, 3], f32 like the kernel's
    r, c = np.triu_indices(3)
    cov_grad = np.zeros((300, 3, 3)); cov_grad[:, r, c] = S["scale"] ** 2 * g["cov3D"]
    means.grad, cov.grad, op.grad = t(S["scale"] * g["means3D"]), t(cov_grad), t(g["opacities"].reshape(-1))
    sh.grad = None
    d_rgb_sum = torch.zeros((300, 4), dtype=torch.float32)
    d_rgb_sum[:, :3] = torch.tensor(drgb, dtype=torch.float32)
    d_rgb_sum[:, 3] = torch.where(torch.tensor(vis), torch.tensor(0, dtype=torch.int32), torch.tensor(-1, dtype=torch.int32)).view(torch.float32)
    views = torch.zeros((1, 44), dtype=torch.float32)
    views[0, 32:35] = torch.tensor(np.asarray(S["campos"], np.float32))
    views[0, 40] = S["scale"]
    deferred = rasterizer.DeferredSH(None, views, means.float(), sh, d_rgb_sum)
    D.sync_gradients_factored(means, cov, sh, op, deferred)
    q.put((rank, [x.grad.double().numpy() for x in (means, cov, sh, op)]))
    torch.distributed.destroy_process_group()


def test_two_rank_factored_exchange_reconstructs_the_summed_sh_gradient():
    """CPU (gloo, world size 2) test of distributed.sync_gradients_factored: the all-gathered (camera, dRGB) factors
    rebuild the SUM over ranks of the oracle's dL/dSH (the rank-1 structure the exchange relies on), the other
    gradients are all-reduced, and every rank ends with the same result."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_factored_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = [0.0, 0.0, 0.0, 0.0]
    r, c = np.triu_indices(3)
    for rank in range(world):
        cloud, S, g, drgb, vis = _rank_view(rank)
        cov_grad = np.zeros((300, 3, 3)); cov_grad[:, r, c] = S["scale"] ** 2 * g["cov3D"]
        # oracle gradients are w.r.t. the rescaled boundary tensors; SH is not rescaled
        want[0] = want[0] + S["scale"] * g["means3D"]
        want[1] = want[1] + cov_grad
        want[2] = want[2] + g["shs"]
        want[3] = want[3] + g["opacities"].reshape(-1)
    for rank, got in outs:
        for a, b, tol in zip(got, want, (1e-12, 1e-12, 2e-5, 1e-12)):   # SH goes through float32 factors
            scale = np.abs(b).max() + 1e-30
            assert np.abs(a - b).max() / scale <= tol


def _pipe_inputs(rank, step, p=257):
    """Deterministic per-(rank, step) gradients and factors of a made-up backward (float64 so that sums are exact enough)."""
    rng = np.random.default_rng(1000 * step + rank)
    cov = np.zeros((p, 3, 3)); r, c = np.triu_indices(3); cov[:, r, c] = rng.standard_normal((p, 6))
    vis = rng.uniform(size=p) < 0.7
    drgb = rng.standard_normal((p, 3)) * vis[:, None]
    campos = np.array([0.1 * rank, -0.2 * rank, 0.05 * step], np.float32)
    return dict(means=rng.standard_normal((p, 3)), cov=cov, op=rng.standard_normal(p), drgb=drgb, vis=vis, campos=campos)


def _pipe_cloud(p=257):
    rng = np.random.default_rng(77)
    return rng.uniform(-2, 2, (p, 3)).astype(np.float32), rng.standard_normal((p, 25, 3)).astype(np.float32)


def _pipelined_worker(rank, world, port, q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from splatter360_amd import distributed as D, rasterizer
    D.init(backend="gloo")
    rasterizer.finish_deferred_sh = _sh_pass_restated
    m3, shs = _pipe_cloud()
    means = torch.tensor(m3, dtype=torch.float64)
    cov = torch.zeros((257, 3, 3), dtype=torch.float64)
    op = torch.zeros(257, dtype=torch.float64)
    sh = torch.tensor(shs)
    results, pending = [], None
    for step in range(3):
        inp = _pipe_inputs(rank, step)
        # --- "forward of micro-batch `step`" would be enqueued here, while the previous exchange is in flight ---
        if pending is not None:
            # a backward that runs between start and finish accumulates into .grad — finish() must leave .grad alone,
            # or the reduced gradients would travel through the next exchange a second time
            sentinel = torch.full_like(means, 7.0)
            means.grad = sentinel
            results.append([g.double().numpy().copy() for g in pending.finish()])
            assert means.grad is sentinel and cov.grad is None and op.grad is None and sh.grad is None
        # --- "backward of micro-batch `step`": fills .grad and leaves the deferred SH factors ---
        means.grad, cov.grad, op.grad = torch.tensor(inp["means"]), torch.tensor(inp["cov"]), torch.tensor(inp["op"])
        d_rgb_sum = torch.zeros((257, 4), dtype=torch.float32)
        d_rgb_sum[:, :3] = torch.tensor(inp["drgb"], dtype=torch.float32)
        d_rgb_sum[:, 3] = torch.where(torch.tensor(inp["vis"]), torch.tensor(0, dtype=torch.int32), torch.tensor(-1, dtype=torch.int32)).view(torch.float32)
        views = torch.zeros((1, 44), dtype=torch.float32)
        views[0, 32:35] = torch.tensor(inp["campos"])
        views[0, 40] = 1.0
        pending = D.start_factored_exchange(means, cov, sh, op, rasterizer.DeferredSH(None, views, means.float(), sh, d_rgb_sum))
        assert means.grad is None and cov.grad is None and op.grad is None     # buffers now belong to the exchange
    results.append([g.double().numpy().copy() for g in pending.finish()])
    # reduce-scatter by Gaussian range (sharded consumer): rank r gets rows [r*per, (r+1)*per) of the sum
    full = [torch.tensor(_pipe_inputs(rank, 9)["means"]), torch.tensor(_pipe_inputs(rank, 9)["cov"])]
    mine = D.reduce_scatter_gradients(full)
    q.put((rank, results, [m.numpy() for m in mine]))
    torch.distributed.destroy_process_group()


def test_four_rank_pipelined_factored_exchange_and_reduce_scatter():
    """world size 4 (gloo): three micro-steps whose exchanges are started after each backward and finished only after the
    next micro-step's forward slot (the overlapped schedule of bench.py), the 40-byte packed all-reduce with the 6-entry
    covariance, and the reduce-scatter variant for a consumer sharded by Gaussian range."""
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    m3, shs = _pipe_cloud()
    for step in range(3):
        ins = [_pipe_inputs(r, step) for r in range(world)]
        want_m = sum(i["means"] for i in ins)
        want_c = sum(i["cov"] for i in ins)
        want_o = sum(i["op"] for i in ins)
        rgb = torch.zeros((world, 257, 4), dtype=torch.float32)
        views = torch.zeros((world, 44), dtype=torch.float32)
        for r, i in enumerate(ins):
            rgb[r, :, :3] = torch.tensor(i["drgb"], dtype=torch.float32)
            rgb[r, :, 3] = torch.where(torch.tensor(i["vis"]), torch.tensor(r, dtype=torch.int32), torch.tensor(-1, dtype=torch.int32)).view(torch.float32)
            views[r, 32:35] = torch.tensor(i["campos"]); views[r, 40] = 1.0
        want_sh = _sh_pass_restated(None, views, torch.tensor(m3), torch.tensor(shs), rgb).double().numpy()
        for rank, results, _ in outs:
            gm, gc, gs, go = results[step]
            np.testing.assert_allclose(gm, want_m, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(gc, want_c, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(go, want_o, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(gs, want_sh, rtol=1e-5, atol=1e-5)
    full_m = sum(_pipe_inputs(r, 9)["means"] for r in range(world))
    full_c = sum(_pipe_inputs(r, 9)["cov"] for r in range(world))
    per = (257 + world - 1) // world
    for rank, _, mine in outs:
        lo, hi = rank * per, min(257, (rank + 1) * per)
        np.testing.assert_allclose(mine[0], full_m[lo:hi], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(mine[1], full_c[lo:hi], rtol=1e-12, atol=1e-12)


def _chunk_inputs(rank, p):
    rng = np.random.default_rng(500 + rank)
    vis = rng.uniform(size=p) < 0.6
    return dict(packed=rng.standard_normal((p, 10)), drgb=(rng.standard_normal((p, 3)) * vis[:, None]).astype(np.float32), vis=vis,
                campos=np.array([0.1 * rank, 0.03 * rank, -0.07 * rank], np.float32))


def _chunked_worker(rank, world, port, q, p, n_chunks, two_groups, mode=None):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from splatter360_amd import distributed as D
    D.init(backend="gloo")
    gg = dist.new_group(list(range(world))) if two_groups else None
    m3, _ = _pipe_cloud(p)
    inp = _chunk_inputs(rank, p)
    packed = torch.zeros((p, 10), dtype=torch.float64)
    rgb = torch.zeros((p, 4), dtype=torch.float32)
    d_sh = torch.zeros((p, 25, 3), dtype=torch.float32)
    rep = torch.zeros(44, dtype=torch.float32)
    rep[32:35] = torch.tensor(inp["campos"]); rep[40] = 1.0
    produced, rebuilt = [], []

    def produce(lo, hi):                      # stands in for s360_backward_gaussians on rows [lo, hi)
        assert not produced or produced[-1][1] == lo
        produced.append((lo, hi))
        packed[lo:hi] = torch.tensor(inp["packed"][lo:hi])
        rgb[lo:hi, :3] = torch.tensor(inp["drgb"][lo:hi])
        rgb[lo:hi, 3] = torch.where(torch.tensor(inp["vis"][lo:hi]), torch.tensor(rank, dtype=torch.int32), torch.tensor(-1, dtype=torch.int32)).view(torch.float32)

    def rebuild_sh(lo, hi, rgb_all, rep_all):  # stands in for s360_sh_backward on rows [lo, hi)
        assert tuple(rgb_all.shape) == (world, hi - lo, 4) and tuple(rep_all.shape) == (world, 44)
        rebuilt.append((lo, hi))
        d_sh[lo:hi] = _sh_pass_restated(None, rep_all, torch.tensor(m3[lo:hi]), d_sh[lo:hi], rgb_all)

    frozen = n_chunks < 0        # harmonics frozen on every rank: no dL/dRGB gathers, no rebuild — only the packed all-reduces
    n_chunks = abs(n_chunks)
    tm = {}
    D.exchange_chunked(p, packed, rgb, rep, produce, None if frozen else rebuild_sh, n_chunks=n_chunks, group=None, group_gather=gg, mode=mode, timings=tm)
    want_mode = mode or ("gather" if world <= D.GATHER_ONLY_MAX_WORLD else "reduce")
    # collective calls of the step: one coalesced all-gather per range ("gather"), + one all-reduce per range ("reduce")
    assert tm["mode"] == want_mode and tm["collective_calls"] == len(produced) * ((1 if want_mode == "gather" else 2) - (1 if (frozen and want_mode == "reduce") else 0))
    assert produced == D.chunk_bounds(p, n_chunks) and produced[0][0] == 0 and produced[-1][1] == p
    assert rebuilt == ([] if frozen else produced)
    q.put((rank, packed.numpy(), d_sh.double().numpy(), len(produced)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,p,n_chunks,two_groups,mode", [(8, 1000, 4, False, None), (8, 1000, 4, True, None), (3, 700, 8, False, None), (2, 100, 4, False, None),
                                                              (3, 700, -3, False, None),   # negative: harmonics frozen (no SH exchange)
                                                              (3, 700, 2, False, "gather"), (2, 700, 1, False, "reduce"), (2, 700, -2, False, "gather")])
def test_chunked_exchange_sums_every_range_over_the_ranks(world, p, n_chunks, two_groups, mode):
    """distributed.exchange_chunked at world size 8 (gloo): ragged Gaussian ranges (1000 = 3 x 256 + 232), more ranges asked
    for than the cloud has workgroups, a cloud smaller than one workgroup, all-gathers on their own process group — every
    rank ends with the packed gradients summed over the ranks and dL/dSH rebuilt from all ranks' factors, range by range."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_chunked_worker, args=(r, world, port, q, p, n_chunks, two_groups, mode)) for r in range(world)]
    [pr.start() for pr in procs]
    outs = [q.get(timeout=300) for _ in range(world)]
    [pr.join(timeout=60) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    ins = [_chunk_inputs(r, p) for r in range(world)]
    want_packed = sum(i["packed"] for i in ins)
    m3, _ = _pipe_cloud(p)
    rgb = torch.zeros((world, p, 4), dtype=torch.float32)
    views = torch.zeros((world, 44), dtype=torch.float32)
    for r, i in enumerate(ins):
        rgb[r, :, :3] = torch.tensor(i["drgb"])
        rgb[r, :, 3] = torch.where(torch.tensor(i["vis"]), torch.tensor(r, dtype=torch.int32), torch.tensor(-1, dtype=torch.int32)).view(torch.float32)
        views[r, 32:35] = torch.tensor(i["campos"]); views[r, 40] = 1.0
    want_sh = _sh_pass_restated(None, views, torch.tensor(m3), torch.zeros((p, 25, 3)), rgb).double().numpy()
    from splatter360_amd.distributed import chunk_bounds
    for rank, packed, d_sh, n in outs:
        assert n == len(chunk_bounds(p, abs(n_chunks)))
        np.testing.assert_allclose(packed, want_packed, rtol=1e-12, atol=1e-12)
        if n_chunks < 0:
            assert not d_sh.any()         # frozen harmonics: nothing gathered, nothing rebuilt
        else:
            np.testing.assert_allclose(d_sh, want_sh, rtol=1e-5, atol=1e-5)


def test_exchange_plan_picks_ranges_from_the_cloud_and_the_form_from_the_ranks():
    """One range per 2 M Gaussians (every collective call is ~16 us on the critical path whatever it moves); one all-gather per range
    up to 2 ranks (56 B/Gaussian either way), all-reduce + all-gather beyond (182 instead of 392 B/Gaussian at 8 ranks)."""
    from splatter360_amd import distributed as D
    assert [len(D.exchange_plan(p, 8)[0]) for p in (1000, 1 << 20, 1 << 22, 1 << 24)] == [1, 1, 2, 4]
    assert [D.exchange_plan(1 << 20, w)[1] for w in (1, 2, 3, 8)] == ["gather", "gather", "reduce", "reduce"]
    assert D.exchange_plan(1 << 20, 8, n_chunks=3, mode="gather") == (D.chunk_bounds(1 << 20, 3), "gather")
    for n in (2, 4, 8):     # bytes received per Gaussian and rank: what the choice is made on
        gather, reduce_ = (n - 1) * 56, 2 * (n - 1) / n * 40 + (n - 1) * 16
        assert (gather <= reduce_) == (n <= D.GATHER_ONLY_MAX_WORLD)


def test_chunked_exchange_single_process_and_bounds():
    from splatter360_amd import distributed as D
    assert D.chunk_bounds(0, 4) == [] and D.chunk_bounds(5, 4) == [(0, 5)]
    for p, n in ((1 << 20, 4), (1000, 4), (1 << 20 | 3, 7), (257, 2)):
        b = D.chunk_bounds(p, n)
        assert b[0][0] == 0 and b[-1][1] == p and len(b) <= n and all(lo % 256 == 0 for lo, _ in b)
        assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    calls = []
    packed, rgb = torch.zeros((600, 10)), torch.zeros((600, 4))
    D.exchange_chunked(600, packed, rgb, torch.zeros(44), lambda lo, hi: calls.append(("p", lo, hi)),
                       lambda lo, hi, a, b: calls.append(("s", lo, hi, tuple(a.shape), tuple(b.shape))), n_chunks=3)
    assert calls == [("p", 0, 256), ("p", 256, 512), ("p", 512, 600), ("s", 0, 256, (1, 256, 4), (1, 44)),
                     ("s", 256, 512, (1, 256, 4), (1, 44)), ("s", 512, 600, (1, 88, 4), (1, 44))]


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus 2` with no torchrun environment (how the driver starts the N = 1 run): bench.py re-launches itself
    under torch.distributed.run, one process per rank, and rank 0 prints ONE JSON line that says what the communicator saw.
    --dry-run keeps the GPU work out (there is none here); backend gloo."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["S360_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-run", "1", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["allreduce_ok"] and len(d["rank_devices"]) == 2
    # N = 1 started plainly stays a single process (no launcher, no process group)
    r1 = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--dry-run", "1", "--steps", "1"], env=env,
                        capture_output=True, text=True, timeout=300)
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert r1.returncode == 0 and d1["n_gpus"] == 1 and d1["rccl_ranks"] == 1 and d1["backend"] is None
