#!/usr/bin/env python3
"""Localise the uniform-cloud discrepancy of Gaussian 367556 (face 0): backward with the image gradient masked to one tile row /
one tile at a time, HIP against the float32 oracle, opacity and covariance gradient of that Gaussian."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from helpers import boundary_tensors, settings_from_views
from oracle import oracle
from splatter360_amd import rasterizer, synthetic
from test_gpu_headline_parity import _single_face_call

face, gi = 0, int(sys.argv[1]) if len(sys.argv) > 1 else 367556
dev = torch.device("cuda:0")
rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS = False, False
cloud = synthetic.uniform_cloud(1 << 20, seed=0, extent=5.0)
params = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
gfull = np.random.default_rng(400 + face).standard_normal((3, 256, 256)).astype(np.float32)
out, st, ps = _single_face_call(params, face, 256, dev, grad_image=gfull)
S = settings_from_views(st.views, 0, 256, 256)
means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
o32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
f = o32.forward()
t = st.tensors()
ncon = t["n_contrib"][0].cpu().numpy()
print("n_contrib mismatches", int((ncon != f["n_contrib"]).sum()))
def one(mask, tag):
    g = gfull * mask[None]
    _, _, p2 = _single_face_call(params, face, 256, dev, grad_image=g)
    go = o32.backward(g)
    h_op, o_op = float(p2[3].grad.reshape(-1)[gi]), float(np.asarray(go["opacities"]).reshape(-1)[gi])
    return h_op, o_op
rows = []
for ty in range(16):
    m = np.zeros((256, 256), np.float32); m[16 * ty:16 * ty + 16] = 1
    h, o = one(m, f"row {ty}")
    rows.append((abs(h - o), ty, h, o))
    print(f"tile row {ty:2d}: HIP {h:+.6e} oracle {o:+.6e} diff {h - o:+.3e}")
rows.sort(reverse=True)
for _, ty, _, _ in rows[:2]:
    for tx in range(16):
        m = np.zeros((256, 256), np.float32); m[16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16] = 1
        h, o = one(m, "")
        if abs(h - o) > 1e-6 * max(1e-3, abs(o)) + 3e-8:
            print(f"  tile ({tx:2d},{ty:2d}): HIP {h:+.6e} oracle {o:+.6e} diff {h - o:+.3e}  tile list length {int(np.diff(f['ranges'][ty * 16 + tx].astype(np.int64))[0])} "
                  f"max n_contrib {int(f['n_contrib'][16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16].max())}")
