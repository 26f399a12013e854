#!/usr/bin/env python3
"""Two legs for rocprofv3 --kernel-trace --stats (scripts/collect_profiles.sh), 1 M Gaussians:
  adapter   the stand-alone adapter kernels (s360_adapter_forward / backward) + render, and the fused raw path (VERDICT r04 weak #8:
            no timing of the adapter kernels existed);
  dropin    the per-face drop-in training step of the unchanged reference's decoder loop (bench.py's `dropin_train`: weak #11)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import adapter, decoder, rasterizer, synthetic

dev = torch.device("cuda:0")
leg = sys.argv[1]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
gt = torch.full((6, 3, 256, 256), 0.5, device=dev)
if leg == "adapter":
    gen = torch.Generator().manual_seed(0)
    h, w, nv = 512, 1024, 2
    dep = torch.exp(torch.empty(nv, h * w).uniform_(-0.69, 2.08, generator=gen)).to(dev)
    op = torch.sigmoid(torch.randn(nv, h * w, generator=gen)).to(dev)
    raw = torch.randn(nv, h * w, 82, generator=gen)
    raw[..., 7:] *= 0.6
    raw = raw.to(dev)
    cext = torch.eye(4).repeat(nv, 1, 1)
    cext[0, :3, 3] = torch.tensor([-0.4, 0.0, 0.1])
    cext[1, :3, 3] = torch.tensor([0.4, 0.0, -0.1])
    cext = cext.to(dev)
    rot = adapter.sh_rotation_blocks(cext, 25)
    for it in range(4):
        d, o, r = dep.clone().requires_grad_(True), op.clone().requires_grad_(True), raw.clone().requires_grad_(True)
        g = adapter.adapter_tail(cext, d, o, r, (h, w), 0.5, 15.0, sh_rotation=rot)
        faces, fm = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, g.means.reshape(-1, 3), g.covariances.reshape(-1, 3, 3),
                                               g.harmonics.reshape(-1, 3, 25), g.opacities.reshape(-1), mse_target=gt, shared_campos=True)
        fm.loss.backward()
    for it in range(4):
        d, o, r = dep.clone().requires_grad_(True), op.clone().requires_grad_(True), raw.clone().requires_grad_(True)
        views = decoder.pack_camera_views(ext, K, near, far, bg)
        faces, _, _, fm = rasterizer.rasterize_raw(d.reshape(-1), o.reshape(-1), r.reshape(-1, 82), cext, views=views, image_height=256, image_width=256,
                                                   context_shape=(h, w), scale_min=0.5, scale_max=15.0, sh_rotation=rot, mse_target=gt)
        fm.loss.backward()
else:
    from types import SimpleNamespace
    cloud = synthetic.encoder_like_cloud(512, 1024)
    params = [torch.tensor(cloud[k], device=dev, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
    gs = SimpleNamespace(means=params[0][None], covariances=params[1][None], harmonics=params[2][None], opacities=params[3][None])
    dec = decoder.DecoderSplattingCUDA().to(dev)
    for it in range(3):
        for p in params:
            p.grad = None
        colors = dec(gs, ext[None], K[None], near[None], far[None], (256, 256)).color
        ((colors[0] - gt) ** 2).mean().backward()
torch.cuda.synchronize()
print(leg, "done")
