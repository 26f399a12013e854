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


_TRIU = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))


def _pack_small(d_means: Tensor, d_cov: Tensor, d_op: Tensor) -> Tensor:
    """means (3) + the 6 unique covariance entries + opacity (1) -> ONE contiguous [P,10] buffer (40 B / Gaussian): a single
    larger all-reduce instead of three, and the [P,3,3] covariance gradient (whose lower triangle is identically zero —
    the rasteriser reads the upper triangle only, cuda_splatting.py:115,123) does not travel as 9 floats.  RESTRICTION: a
    non-zero strict lower triangle of d_cov (possible only for gradients produced outside the rasteriser) does not travel;
    set S360_CHECK_TRIL=1 to assert its absence."""
    p = d_means.shape[0]
    if d_cov.dim() == 3 and __debug__ and os.environ.get("S360_CHECK_TRIL"):
        # only the upper triangle travels: a gradient in the strict lower triangle (a regulariser acting on the [P,3,3]
        # covariances outside the rasteriser) would be dropped silently — fold it into the upper entries before calling
        assert float(torch.tril(d_cov, -1).abs().max()) == 0.0, "covariance gradient has a non-zero strict lower triangle"
    c6 = d_cov if d_cov.dim() == 2 else torch.stack([d_cov[:, r, c] for r, c in _TRIU], dim=1)
    return torch.cat([d_means.reshape(p, 3), c6.reshape(p, 6), d_op.reshape(p, 1)], dim=1).contiguous()


def _unpack_small(buf: Tensor, cov_like: Tensor):
    d_means = buf[:, 0:3].contiguous()
    if cov_like.dim() == 2:
        d_cov = buf[:, 3:9].contiguous()
    else:
        d_cov = torch.zeros_like(cov_like)
        for k, (r, c) in enumerate(_TRIU):
            d_cov[:, r, c] = buf[:, 3 + k]
    return d_means, d_cov, buf[:, 9].contiguous().reshape(-1)


class FactoredExchange:
    """The gradient exchange of one step, split in two so that it can sit behind other GPU work:
    start_factored_exchange() issues the collectives (they wait, on the communicator's stream, for the backward kernels
    that produced the gradients and then run beside whatever the compute stream does next); finish() waits for the all-gathers,
    rebuilds the summed dL/dSH locally (s360_sh_backward over the N gathered factors), waits for the all-reduce and RETURNS the
    four summed gradients.  It does not touch the parameters' .grad: a .grad assigned here would be accumulated into by a
    backward that runs between start and finish and travel through the next exchange a second time; callers that
    want .grad semantics use sync_gradients_factored (blocking), which assigns after the exchange is complete."""

    def __init__(self, params, deferred, small, rgb_all, rep_all, works_ag, work_ar, world, means_snapshot):
        self.params, self.deferred, self.small = params, deferred, small
        self.rgb_all, self.rep_all, self.works_ag, self.work_ar, self.world = rgb_all, rep_all, works_ag, work_ar, world
        self.means_snapshot = means_snapshot
        self.result = None

    def finish(self):
        if self.result is not None:
            return self.result
        from . import rasterizer
        means, covariances, harmonics, opacities = self.params
        d = self.deferred
        for w in self.works_ag:
            w.wait()
        # dL/dSH = sum over ranks of Y(dir_rank) (x) dRGB_rank, rebuilt locally while the packed buffer is still being
        # all-reduced (every rank's view-direction term of dL/dmean is already inside its d_means: s360_backward_split).
        # Directions come from the means AS THEY WERE at the backward (snapshot taken by start: an optimiser step may have
        # moved the parameter since).
        d_sh = rasterizer.finish_deferred_sh(d.prm, self.rep_all.reshape(self.world, -1), self.means_snapshot, d.shs,
                                             self.rgb_all.view(self.world, -1, 4))
        if self.work_ar is not None:
            self.work_ar.wait()
        d_means, d_cov, d_op = _unpack_small(self.small, covariances)
        self.result = (d_means.reshape(means.shape), d_cov, d_sh, d_op.reshape(opacities.shape))
        return self.result


def start_factored_exchange(means: Tensor, covariances: Tensor, harmonics: Tensor, opacities: Tensor, deferred,
                            group=None, force_collectives: bool = False) -> FactoredExchange:
    """Gradient exchange for views sharded one panorama per rank, exploiting that each rank's dL/dSH is the
    rank-1 product Y(dir_rank) (x) dL/dRGB_rank per Gaussian (the pipelined form; `exchange_chunked` below is the one that
    hides the exchange INSIDE a step):
        all-reduce   [means | cov6 | opacity].grad packed as one [P,10] buffer      (40 B per Gaussian)
        all-gather   d_rgb_sum[P,4] and one camera record per rank                    (16 B per Gaussian per rank)
    then every rank rebuilds the summed dL/dSH (and the view-direction part of dL/dmean) locally with
    s360_sh_backward.  At N = 8 and 1 M Gaussians a rank receives ~0.19 GB instead of the ~0.65 GB of a ring
    all-reduce of the full 369 MB gradient set; results equal the plain all-reduce up to float summation order.
    `deferred` = rasterizer.last_deferred() of a backward run with defer_sh=True.  The .grad tensors are TAKEN from the
    parameters (set to None) so that the next step's backward cannot touch buffers the communicator is still reading.
    Works for world size 1.  All collectives are async; call .finish() on the result."""
    dev = harmonics.device
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    d = deferred
    if d is None:
        raise RuntimeError("start_factored_exchange needs the DeferredSH of a backward run with defer_sh=True")
    # every rank must issue the same collectives: a tensor that received no gradient on this rank contributes zeros
    grads = [t.grad if t.grad is not None else torch.zeros_like(t) for t in (means, covariances, opacities)]
    small = _pack_small(*grads)
    for t in (means, covariances, harmonics, opacities):
        t.grad = None
    rgb = d.d_rgb_sum.clone()
    vis = rgb[:, 3].view(torch.int32) >= 0
    rgb[:, 3] = torch.where(vis, torch.full_like(rgb[:, 3].view(torch.int32), rank), torch.full_like(rgb[:, 3].view(torch.int32), -1)).view(torch.float32)
    rep = d.views[:1].contiguous()  # all views of the call share campos and scale
    works_ag, work_ar = [], None
    if world > 1 or (force_collectives and dist.is_available() and dist.is_initialized()):
        # the communicator runs its work in issue order: the all-gathers first (the local SH pass waits for them),
        # the all-reduce behind them, overlapping with that pass
        rgb_all = torch.empty((world * rgb.shape[0], 4), dtype=rgb.dtype, device=dev)   # dim-0 concat: gloo-compatible
        rep_all = torch.empty((world * rep.shape[1],), dtype=rep.dtype, device=dev)
        works_ag = [dist.all_gather_into_tensor(rgb_all, rgb, group=group, async_op=True),
                    dist.all_gather_into_tensor(rep_all, rep.reshape(-1), group=group, async_op=True)]
        work_ar = dist.all_reduce(small, op=dist.ReduceOp.SUM, group=group, async_op=True)
    else:
        rgb_all, rep_all = rgb, rep
    snap = d.means3D.clone() if (world > 1 or works_ag) else d.means3D      # 12 B / Gaussian: see FactoredExchange.finish
    return FactoredExchange((means, covariances, harmonics, opacities), d, small, rgb_all, rep_all, works_ag, work_ar, world, snap)


def sync_gradients_factored(means: Tensor, covariances: Tensor, harmonics: Tensor, opacities: Tensor, deferred,
                            group=None, force_collectives: bool = False) -> None:
    """start_factored_exchange(...).finish(), then the summed gradients are stored in .grad of the four tensors: the blocking
    form (nothing can run between the two halves, so assigning .grad is safe here)."""
    g = start_factored_exchange(means, covariances, harmonics, opacities, deferred, group, force_collectives).finish()
    means.grad, covariances.grad, harmonics.grad, opacities.grad = g


def chunk_bounds(p: int, n_chunks: int, align: int = 256) -> List[tuple]:
    """[0, p) cut into at most n_chunks contiguous ranges whose starts are multiples of `align` (whole workgroups of the
    per-Gaussian kernels); the last range is ragged.  Fewer ranges come back when p is small."""
    n_chunks = max(1, int(n_chunks))
    per = max(align, ((p + n_chunks - 1) // n_chunks + align - 1) // align * align)
    return [(lo, min(p, lo + per)) for lo in range(0, p, per)] if p > 0 else []


GATHER_ONLY_MAX_WORLD = 2     # up to this many ranks the whole exchange is ONE all-gather per range (see exchange_plan)


def exchange_plan(p: int, world: int, n_chunks: Optional[int] = None, mode: Optional[str] = None):
    """(Gaussian ranges, mode) of one step's gradient exchange.

    mode "gather": every rank all-gathers its rows — packed[., 10] (40 B) and dRGB[., 4] (16 B) in ONE coalesced all-gather per range —
    and sums the packed rows of the N ranks locally (a fixed rank order: every rank holds the same bits).  Receives (N-1) x 56 B per
    Gaussian.  mode "reduce": all-reduce of the packed rows + all-gather of the dRGB rows: 2 (N-1)/N x 40 + (N-1) x 16 B per Gaussian,
    two collective calls per range.  At N = 2 both move 56 B (gather: one call instead of two); at N = 4: 168 against 108 B; at N = 8:
    392 against 182 B — so "gather" up to GATHER_ONLY_MAX_WORLD ranks, "reduce" beyond.

    n_chunks None: chosen from P — one range per 2 M Gaussians (1 M: a single range, 4 M: two, capped at four).  Every collective call
    costs ~16 us of host-side issue + launch on the step's critical path whatever it moves (profiles/r05_bench_1rank_nccl.json: nine
    calls, 0.145 ms exposed for an exchange of zero bytes), while what a further range can hide under is a share of the 45 + 65 us of
    per-Gaussian tail kernels at 1 M: ranges only pay when those kernels are long."""
    if mode is None:
        mode = "gather" if world <= GATHER_ONLY_MAX_WORLD else "reduce"
    if mode not in ("gather", "reduce"):
        raise ValueError("exchange mode must be 'gather' or 'reduce'")
    if n_chunks is None:
        n_chunks = max(1, min(4, p >> 21))
    return chunk_bounds(p, n_chunks), mode


def _coalesced_all_gather(outs, ins, group):
    """ONE collective call for several (output, input) pairs (c10d's fast-path coalescing: allgather_into_tensor_coalesced — a single
    ncclGroup on RCCL, implemented by gloo as well).  Returns the handle to wait on."""
    if ins[0].is_cuda and dist.get_backend(group) == "gloo":
        # gloo with device tensors (only the one-GPU test configuration): its coalesced entry point does not order itself behind the
        # producing kernels on the current stream the way its plain collectives do — issue the gathers one by one
        works = [dist.all_gather_into_tensor(o, i, group=group, async_op=True) for o, i in zip(outs, ins)]

        class _All:
            def wait(self):
                for w in works:
                    w.wait()
        return _All()
    from torch.distributed.distributed_c10d import _coalescing_manager
    with _coalescing_manager(group=group, async_ops=True) as cm:
        for o, i in zip(outs, ins):
            dist.all_gather_into_tensor(o, i, group=group)
    return cm


def exchange_chunked(p: int, packed: Tensor, rgb: Tensor, rep_view: Tensor, produce, rebuild_sh, n_chunks: Optional[int] = None, group=None,
                     group_gather=None, timings: Optional[dict] = None, force_collectives: bool = False, mode: Optional[str] = None,
                     reduce_rows=None) -> str:
    """The gradient exchange of ONE step inside that step's backward (no gradient accumulation assumed), in at most THREE collective
    calls for a cloud of up to 2 M Gaussians (round 5 issued nine: an all-gather and an all-reduce per range, four ranges, and a
    camera-record gather — 0.145 ms of exposed call overhead per 0.79-ms step before a byte moved):

        for each range c (exchange_plan: one per 2 M Gaussians):
            produce(lo, hi)                                  -> packed[lo:hi] (40 B/Gaussian), rgb[lo:hi] (16 B)
            mode "gather" (N <= 2):  ONE coalesced all-gather of (packed rows, rgb rows[, this rank's camera record with range 0])
            mode "reduce" (N  > 2):  all-reduce of the packed rows + ONE coalesced all-gather of (rgb rows[, camera record])
        for each range c:  wait;  "gather": packed[lo:hi] = sum over ranks of the gathered rows (rank order) — or, when the caller
                           passes reduce_rows(lo, hi, rows[world, hi-lo, 10]), that callback consumes the gathered rows itself (the
                           rasteriser: s360_reduce_unpack_gradients, the reduction fused with the unpack) and packed is left alone;
                           rebuild_sh(lo, hi, rgb_all_c[world, hi-lo, 4], rep_all[world, 44])
    Returns "local" (one rank, no collective: packed holds this rank's rows), "reduce" (packed holds the sums) or "gather" (sums in
    packed, or handed to reduce_rows).

    `produce` / `rebuild_sh` are the caller's kernels (rasterizer: s360_backward_gaussians / s360_sh_backward or, on the raw path,
    the single k_raw_bwd launch over all ranges; the CPU tests: numpy slices of oracle gradients).  rep_view[44]: this rank's
    representative camera record (all views of a call share the camera centre).  `group_gather`: a second process group for the
    all-gathers of mode "reduce", so that they do not queue behind the all-reduces of earlier ranges.  Works for world size 1 (no
    collectives).  rebuild_sh=None (harmonics frozen on EVERY rank): the dL/dRGB factors are neither gathered nor rebuilt.  The
    collectives are issued from inside the caller's backward: every rank of `group` must run this backward the same number of times
    with the same p / n_chunks / mode / rebuild_sh-or-None, or the ranks that did wait forever.
    force_collectives=True (with an initialised process group): a world of ONE rank also goes through the collective branch — how
    the RCCL path is exercised on a one-GPU box (tests/test_gpu_rccl_single_rank.py); the results are those of the short cut bit
    for bit."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    bounds, mode = exchange_plan(p, world, n_chunks, mode)
    if world == 1 and not (force_collectives and dist.is_available() and dist.is_initialized()):
        for lo, hi in bounds:
            produce(lo, hi)
        rep_all = rep_view.reshape(1, -1)
        if rebuild_sh is not None:
            for lo, hi in bounds:
                rebuild_sh(lo, hi, rgb[lo:hi].reshape(1, hi - lo, 4), rep_all)
        return "local"
    gg = group_gather if (group_gather is not None and mode == "reduce") else group
    pending = []
    rep_all = None
    for idx, (lo, hi) in enumerate(bounds):
        produce(lo, hi)
        n = hi - lo
        outs, ins = [], []
        pk_all = rgb_all = None
        if mode == "gather":
            pk_all = torch.empty((world * n, packed.shape[1]), dtype=packed.dtype, device=packed.device)      # dim-0 concat: gloo-compatible
            outs.append(pk_all); ins.append(packed[lo:hi])
        if rebuild_sh is not None:
            rgb_all = torch.empty((world * n, 4), dtype=rgb.dtype, device=rgb.device)
            outs.append(rgb_all); ins.append(rgb[lo:hi])
            if idx == 0:      # the camera records ride with the first range's gather
                rep_all = torch.empty((world * rep_view.numel(),), dtype=rep_view.dtype, device=rep_view.device)
                outs.append(rep_all); ins.append(rep_view.reshape(-1).contiguous())
        w_g = _coalesced_all_gather(outs, ins, gg) if outs else None
        w_r = dist.all_reduce(packed[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True) if mode == "reduce" else None
        pending.append((lo, hi, pk_all, rgb_all, w_g, w_r))
    for lo, hi, pk_all, rgb_all, w_g, w_r in pending:
        if w_g is not None:
            w_g.wait()
        if pk_all is not None:    # the local reduction of mode "gather": same order on every rank
            if reduce_rows is not None:
                reduce_rows(lo, hi, pk_all.view(world, hi - lo, packed.shape[1]))
            else:
                torch.sum(pk_all.view(world, hi - lo, packed.shape[1]), dim=0, out=packed[lo:hi])
        if rebuild_sh is not None:
            rebuild_sh(lo, hi, rgb_all.view(world, hi - lo, 4), rep_all.reshape(world, -1))
    for _, _, _, _, _, w_r in pending:
        if w_r is not None:
            w_r.wait()
    if timings is not None:
        timings["chunks"], timings["mode"] = len(bounds), mode
        timings["collective_calls"] = sum((w_g is not None) + (w_r is not None) for _, _, _, _, w_g, w_r in pending)
    return mode


class ExchangeConfig:
    """Hand this to rasterize_views(..., exchange=ExchangeConfig(...)) (views sharing one camera centre): the node's backward
    then returns the per-Gaussian gradients SUMMED over the ranks of `group` (every rank renders its own panorama of the same
    replicated cloud), exchanged range by range inside the backward itself — see exchange_chunked.  `group` must be the group
    (or the default group) all participating ranks initialised; EVERY rank of it must run the node's backward (a rank whose loss
    does not reach the render would leave the others waiting in the collectives).  Frozen harmonics (no gradient required on any
    rank) skip the dL/dRGB gathers and the dL/dSH rebuild."""

    def __init__(self, group=None, n_chunks: Optional[int] = None, group_gather=None, force_collectives: bool = False, mode: Optional[str] = None):
        """n_chunks / mode None: chosen by exchange_plan from the cloud's size / the number of ranks."""
        self.group, self.n_chunks, self.group_gather = group, (None if n_chunks is None else int(n_chunks)), group_gather
        self.force_collectives = bool(force_collectives)      # world size 1: still issue the collectives (see exchange_chunked)
        self.mode = mode

    def world(self) -> int:
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def rank(self) -> int:
        return dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0


def reduce_scatter_gradients(grads: Sequence[Tensor], group=None) -> List[Tensor]:
    """For a consumer that is itself sharded by Gaussian range (rank r owns Gaussians [r*ceil(P/N), ...)): every rank
    receives only ITS slice of the summed gradients — half the bytes of an all-reduce on a ring.
    Each tensor is [P, ...]; P is padded up to a multiple of the world size for the collective and the padding dropped.
    Returns the local slices (views of fresh buffers)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [g for g in grads]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    outs, works = [], []
    for g in grads:
        p = g.shape[0]
        per = (p + world - 1) // world
        flat = g.reshape(p, -1)
        if per * world != p:
            flat = torch.cat([flat, flat.new_zeros((per * world - p, flat.shape[1]))])
        out = torch.empty((per, flat.shape[1]), dtype=g.dtype, device=g.device)
        works.append(dist.reduce_scatter_tensor(out, flat.contiguous(), op=dist.ReduceOp.SUM, group=group, async_op=True))
        lo, hi = rank * per, min(p, (rank + 1) * per)
        outs.append((out, max(0, hi - lo), g.shape[1:]))
    for w in works:
        w.wait()
    return [o[:n].reshape(n, *tail) for o, n, tail in outs]


def rank_report(device) -> dict:
    """What the communicator actually is, for the bench line: backend, world size as the process group reports it, and the
    device index every rank is pinned to (gathered)."""
    if not (dist.is_available() and dist.is_initialized()):
        idx = torch.cuda.current_device() if (torch.cuda.is_available() and torch.device(device).type == "cuda") else -1
        return {"backend": None, "rccl_ranks": 1, "rank_devices": [idx]}
    world = dist.get_world_size()
    idx = torch.cuda.current_device() if torch.device(device).type == "cuda" else -1
    mine = torch.tensor([idx], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return {"backend": dist.get_backend(), "rccl_ranks": world, "rank_devices": [int(x) for x in out.cpu()]}


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
