"""Multi-GPU sharding of the render path: one process per GPU, views of a replicated Gaussian
cloud sharded across ranks, ONE exchange step — an all-reduce(sum) of the per-Gaussian gradient
buffers (means 3 + cov 9 + SH 75 + opacity 1 floats per Gaussian = 352 B -> 369 MB at 1 M) over
RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" on CPU for the tests).

Reference behaviour: plain Lightning DDP, one sample per rank (/root/reference/src/main.py:
117-130); the renderer itself has no collective.  BASELINE.json's north star shards the target
views of one cloud one-per-GPU, which makes the gradient all-reduce the path's single exchange.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from MASTER_ADDR/MASTER_PORT when WORLD_SIZE > 1.  On a GPU
    machine the process is pinned to cuda:LOCAL_RANK *before* the communicator is created (RCCL binds
    to the current device).  S360_DIST_BACKEND / S360_FORCE_DEVICE override the backend / device index
    (used to exercise the N>1 code path with gloo on a single-GPU box)."""
    rank, local_rank, world = env_world()
    if torch.cuda.is_available():
        dev_index = int(os.environ.get("S360_FORCE_DEVICE", local_rank))
        torch.cuda.set_device(dev_index)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("S360_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_views(n_views: int, rank: int, world: int) -> List[int]:
    """Contiguous block partition of view indices; ranks beyond n_views get nothing."""
    per = (n_views + world - 1) // world
    return list(range(min(rank * per, n_views), min((rank + 1) * per, n_views)))


def allreduce_gradients(grads: Sequence[Optional[Tensor]], average: bool = False, group=None,
                        async_op: bool = False):
    """Sum (or mean) the per-Gaussian gradient tensors over all ranks, in place.  The four buffers
    are issued back to back (largest first so the ring is busy while the small ones queue);
    returns the work handles when async_op (average together with async_op is rejected: the division
    would have to run after handles the caller owns)."""
    if average and async_op:
        raise ValueError("allreduce_gradients: average=True cannot be combined with async_op=True "
                         "(divide after waiting on the returned handles)")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    world = dist.get_world_size(group)
    todo = sorted([g for g in grads if g is not None], key=lambda t: -t.numel())
    works = []
    for g in todo:
        works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
    if async_op:
        return works
    for w in works:
        w.wait()
    if average:
        for g in todo:
            g.div_(world)
    return []


def sync_gradients_factored(means: Tensor, covariances: Tensor, harmonics: Tensor, opacities: Tensor, deferred,
                            group=None) -> None:
    """Gradient exchange for views sharded one panorama per rank, exploiting that each rank's dL/dSH is the
    rank-1 product Y(dir_rank) (x) dL/dRGB_rank per Gaussian:
        all-reduce   means.grad / covariances.grad / opacities.grad      (52 B per Gaussian)
        all-gather   d_rgb_sum[P,4] and one camera record per rank        (16 B per Gaussian per rank)
    then every rank rebuilds the summed dL/dSH (and the view-direction part of dL/dmean) locally with
    s360_sh_backward.  At N = 8 and 1 M Gaussians a rank receives ~0.17 GB instead of the ~0.65 GB of a ring
    all-reduce of the full 369 MB gradient set; results equal the plain all-reduce up to float summation order.
    `deferred` = rasterizer.last_deferred() of a backward run with defer_sh=True.  Works for world size 1."""
    from . import rasterizer
    dev = harmonics.device
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    d = deferred
    if d is None:
        raise RuntimeError("sync_gradients_factored needs the DeferredSH of a backward run with defer_sh=True")
    # every rank must issue the same collectives: a tensor that received no gradient on this rank contributes zeros
    for t in (means, covariances, opacities):
        if t.grad is None:
            t.grad = torch.zeros_like(t)
    rgb = d.d_rgb_sum.clone()
    vis = rgb[:, 3].view(torch.int32) >= 0
    rgb[:, 3] = torch.where(vis, torch.full_like(rgb[:, 3].view(torch.int32), rank), torch.full_like(rgb[:, 3].view(torch.int32), -1)).view(torch.float32)
    rep = d.views[:1].contiguous()  # all views of the call share campos and scale
    ar = []
    if world > 1:
        # the communicator runs its work in issue order: the all-gathers first (the local SH pass waits for them),
        # the three all-reduces behind them, overlapping with that pass
        rgb_all = torch.empty((world * rgb.shape[0], 4), dtype=rgb.dtype, device=dev)   # dim-0 concat: gloo-compatible
        rep_all = torch.empty((world * rep.shape[1],), dtype=rep.dtype, device=dev)
        ag = [dist.all_gather_into_tensor(rgb_all, rgb, group=group, async_op=True),
              dist.all_gather_into_tensor(rep_all, rep.reshape(-1), group=group, async_op=True)]
        ar = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True)
              for g in (covariances.grad, means.grad, opacities.grad)]
        for w in ag:
            w.wait()
    else:
        rgb_all, rep_all = rgb, rep
    # the SH pass adds every rank's view-direction term of dL/dmean into a scratch buffer (means.grad is still
    # being all-reduced); it is folded in once the all-reduce has landed
    dm_extra = torch.zeros_like(means.grad)
    harmonics.grad = rasterizer.finish_deferred_sh(d.prm, rep_all.reshape(world, -1), d.means3D, d.shs,
                                                   rgb_all.view(world, -1, 4), dm_extra)
    for w in ar:
        w.wait()
    means.grad += dm_extra


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
