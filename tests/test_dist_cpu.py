"""world_size-2 gloo tests (CPU) of the N>1 path: view sharding + the gradient all-reduce.
The CPU oracle stands in for the renderer; the product's distributed helpers are what is tested."""
import os
import socket

import numpy as np
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
