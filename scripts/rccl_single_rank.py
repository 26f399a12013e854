#!/usr/bin/env python3
"""The N > 1 gradient exchange driven through RCCL on ONE GPU: a process group of world size 1 with backend
"nccl" (= RCCL on ROCm), `force_collectives=True` so that exchange_chunked / start_factored_exchange take their collective branch
instead of the world-size-1 short cut.  The all-reduces / all-gathers then really run as RCCL work on the communicator's streams
beside the ctypes-launched kernels on torch's current stream — in-place all-reduces of packed[lo:hi] while the next produce()
writes the neighbouring rows, the per-range dL/dSH rebuilds waiting on their all-gathers.  Results must equal the one-call
backward BIT FOR BIT (one rank: every sum has a single term).  Reference behaviour replaced: Lightning DDP's gradient all-reduce,
/root/reference/src/main.py:117-130.

    python scripts/rccl_single_rank.py            # prints "rccl single rank ok ..."
    rocprofv3 --kernel-trace --stats -d <dir> -- python scripts/rccl_single_rank.py     # the trace kept under profiles/
"""
import os
import socket
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from splatter360_amd import distributed as D, rasterizer
    from test_gpu_factored_sync import POSITIONS, _cloud, _render_backward
    # the comparisons below are bit-for-bit between calls: keep every call in ONE compositing mode (the adaptive default would run the
    # first call of a cloud with long unsaturated lists sequentially and split the following ones: equal to 1e-7, not to the bit)
    rasterizer.SPLIT_LONG_LISTS = False
    dev = torch.device("cuda:0")
    w = int(os.environ.get("S360_RCCL_TEST_W", "128"))
    ps = _cloud(dev, w=w)
    _render_backward(dev, ps, POSITIONS[1], 5, False)
    want = [p.grad.clone() for p in ps]
    checked = 0
    for n_chunks in (1, 3, 7):          # 7: ragged, uneven Gaussian ranges
        for rep in range(3):            # repeated: a stream-ordering bug shows up as run-to-run differences
            for p in ps:
                p.grad = None
            _render_backward(dev, ps, POSITIONS[1], 5, False, exchange=D.ExchangeConfig(n_chunks=n_chunks, force_collectives=True))
            for p, g in zip(ps, want):
                assert torch.equal(p.grad, g), (n_chunks, rep)
            checked += 1
    # both forms of the exchange (exchange_plan picks "gather" — ONE coalesced all-gather per range — up to two ranks, all-reduce +
    # all-gather beyond), and the default plan (one range for a cloud of this size)
    for ex_kw in (dict(mode="reduce", n_chunks=3), dict(mode="gather", n_chunks=3), dict()):
        for p in ps:
            p.grad = None
        _render_backward(dev, ps, POSITIONS[1], 5, False, exchange=D.ExchangeConfig(force_collectives=True, **ex_kw))
        for p, g in zip(ps, want):
            assert torch.equal(p.grad, g), ex_kw
        checked += 1
    # the RAW path (s360_forward_raw / s360_backward_raw): the exchange feeds k_raw_bwd the gathered dL/dRGB factors
    from splatter360_amd import adapter, decoder
    gen = torch.Generator().manual_seed(9)
    hw, nv = (w // 2, w), 2
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
    e6, K6, n6, f6 = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
    views = decoder.pack_camera_views(e6, K6, n6, f6, torch.zeros(3, device=dev))
    gw = torch.randn((6, 3, 64, 64), generator=gen).to(dev)

    def raw_step(exchange):
        leaves = [t.clone().requires_grad_(True) for t in (rdep, rop, rraw)]
        img = rasterizer.rasterize_raw(leaves[0].reshape(-1), leaves[1].reshape(-1), leaves[2].reshape(-1, 82), cext, views=views, image_height=64,
                                       image_width=64, context_shape=hw, scale_min=0.5, scale_max=15.0, sh_rotation=rot, exchange=exchange)[0]
        (img * gw).sum().backward()
        return [t.grad for t in leaves]

    want_raw = raw_step(None)
    for ex_kw in (dict(), dict(mode="reduce", n_chunks=3), dict(mode="gather", n_chunks=2)):
        for rep in range(2):
            got_raw = raw_step(D.ExchangeConfig(force_collectives=True, **ex_kw))
            for a, b in zip(got_raw, want_raw):
                assert torch.equal(a, b), ("raw", ex_kw, rep)
        checked += 1
    # a second communicator for the all-gathers (they then do not queue behind earlier ranges' all-reduces)
    gg = dist.new_group(ranks=[0], backend="nccl")
    for p in ps:
        p.grad = None
    _render_backward(dev, ps, POSITIONS[1], 5, False, exchange=D.ExchangeConfig(n_chunks=4, group_gather=gg, force_collectives=True))
    for p, g in zip(ps, want):
        assert torch.equal(p.grad, g)
    # frozen harmonics: only the packed all-reduces run
    ps2 = [ps[0], ps[1], ps[2].detach(), ps[3]]
    for p in ps:
        p.grad = None
    _render_backward(dev, ps2, POSITIONS[1], 5, False, exchange=D.ExchangeConfig(n_chunks=3, force_collectives=True))
    for i in (0, 1, 3):
        assert torch.equal(ps[i].grad, want[i])
    # the one-exchange-per-step form (start / finish around other work)
    for p in ps:
        p.grad = None
    d = _render_backward(dev, ps, POSITIONS[1], 5, True)
    ex = D.start_factored_exchange(*ps, d, force_collectives=True)
    assert ex.works_ag and ex.work_ar is not None       # the collectives were really issued
    junk = torch.randn(1 << 20, device=dev).sum()        # compute-stream work between start and finish
    got = ex.finish()
    for g, wgt in zip(got, want):
        assert (g.reshape(wgt.shape) - wgt).abs().max().item() <= 1e-6 * (wgt.abs().max().item() + 1e-20)
    torch.cuda.synchronize()
    float(junk)
    print(f"rccl single rank ok: backend {dist.get_backend()} world {dist.get_world_size()} gaussians {ps[0].shape[0]} "
          f"chunked runs {checked + 2} bit-identical, factored start/finish within 1e-6", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
