#!/usr/bin/env python3
"""The fused raw training step (s360_forward_raw / s360_backward_raw) at 1 M Gaussians, a few times, for rocprofv3
(--kernel-trace --stats, or one --pmc pass): what the k_raw_eval / k_raw_bwd figures of DESIGN.md section 6 come from.
Usage: raw_prof.py [steps] [rot=1|0]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import adapter, decoder, rasterizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
use_rot = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
gt = torch.full((6, 3, 256, 256), 0.5, device=dev)
gen = torch.Generator().manual_seed(0)
h, w, nv = 512, 1024, 2
dep = torch.exp(torch.empty(nv, h * w).uniform_(-0.69, 2.08, generator=gen)).to(dev).requires_grad_(True)
op = torch.sigmoid(torch.randn(nv, h * w, generator=gen)).to(dev).requires_grad_(True)
raw = torch.randn(nv, h * w, 82, generator=gen)
raw[..., 7:] *= 0.6
raw = raw.to(dev).requires_grad_(True)
cext = torch.eye(4).repeat(nv, 1, 1)
cext[0, :3, 3] = torch.tensor([-0.4, 0.0, 0.1])
cext[1, :3, 3] = torch.tensor([0.4, 0.0, -0.1])
cext = cext.to(dev)
rot = adapter.sh_rotation_blocks(cext, 25) if use_rot else None
for it in range(steps):
    for t in (dep, op, raw):
        t.grad = None
    views = decoder.pack_camera_views(ext, K, near, far, bg)
    faces, _, _, fm = rasterizer.rasterize_raw(dep.reshape(-1), op.reshape(-1), raw.reshape(-1, 82), cext, views=views, image_height=256, image_width=256,
                                               context_shape=(h, w), scale_min=0.5, scale_max=15.0, sh_rotation=rot, mse_target=gt)
    fm.loss.backward()
torch.cuda.synchronize()
print("raw_prof done", float(fm.loss))
