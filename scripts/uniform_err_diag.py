#!/usr/bin/env python3
"""Where does the backward's distance from the float64 oracle come from on the uniform cloud (3.9x the float32
oracle's own distance on face 0)?  Per-Gaussian error of HIP and of the float32 oracle against float64, split by footprint."""
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

face = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS = False, False
cloud = synthetic.uniform_cloud(1 << 20, seed=0, extent=5.0)
params = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
gimg = np.random.default_rng(400 + face).standard_normal((3, 256, 256)).astype(np.float32)
out, st, ps = _single_face_call(params, face, 256, dev, grad_image=gimg)
S = settings_from_views(st.views, 0, 256, 256)
means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
o32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs); o32.forward(); g32 = o32.backward(gimg); del o32
o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64); o64.forward(); g64 = o64.backward(gimg); del o64
tt = st.tensors()["tiles_touched"][0].cpu().numpy()
sc = np.float64(S["scale"])
r, c = np.triu_indices(3)
got = dict(means3D=ps[0].grad.cpu().numpy(), cov3D=ps[1].grad.cpu().numpy()[:, r, c], opacities=ps[3].grad.cpu().numpy().reshape(-1, 1))
fold = dict(means3D=sc, cov3D=sc * sc, opacities=1.0)
for k in got:
    w = np.asarray(g64[k], np.float64).reshape(got[k].shape) * fold[k]
    w32 = np.asarray(g32[k], np.float64).reshape(got[k].shape) * fold[k]
    scale = np.abs(w).max()
    eh = np.abs(got[k] - w).max(1) / scale
    eo = np.abs(w32 - w).max(1) / scale
    print(f"== {k}: scale {scale:.3e}  HIP max {eh.max():.3e}  f32-oracle max {eo.max():.3e}")
    for lo, hi in ((1, 1), (2, 4), (5, 32), (33, 128), (129, 256)):
        m = (tt >= lo) & (tt <= hi)
        if m.any():
            print(f"   tiles {lo:3d}..{hi:3d}: n {int(m.sum()):7d}  HIP max {eh[m].max():.3e} mean {eh[m].mean():.3e} | f32 oracle max {eo[m].max():.3e} mean {eo[m].mean():.3e}")
    top = np.argsort(-eh)[:8]
    for i in top:
        print(f"   g {i:8d} tiles {int(tt[i]):4d} err HIP {eh[i]:.3e} f32 {eo[i]:.3e}  |grad| {np.abs(w[i]).max() / scale:.3e} opacity {float(cloud['opacities'][i]):.3f}")

# ---- the worst Gaussian in detail, and the same backward with float atomics instead of partial slots + gather
i = int(np.argsort(-(np.abs(got["cov3D"] - np.asarray(g64["cov3D"], np.float64) * fold["cov3D"]).max(1)))[0])
print("worst", i, "mean", cloud["means"][i], "opacity", cloud["opacities"][i], "tiles", tt[i], "radius", int(st.tensors().get("radii", torch.zeros(1, 1))[0].reshape(-1)[i]) if "radii" in st.tensors() else "?")
vm = np.asarray(S["viewmatrix"], np.float64).reshape(4, 4)
p = np.append(cloud["means"][i].astype(np.float64) * sc, 1.0) @ vm
print("view-space", p[:3], "eig(cov)*scale^2", np.linalg.eigvalsh(cloud["covariances"][i].astype(np.float64)) * sc * sc)
for k in got:
    print(k, "HIP", got[k][i], "\n   f64", np.asarray(g64[k], np.float64).reshape(got[k].shape)[i] * fold[k], "\n   f32", np.asarray(g32[k], np.float64).reshape(got[k].shape)[i] * fold[k])
rasterizer.ATOMIC_GRADS = True
out2, st2, ps2 = _single_face_call(params, face, 256, dev, grad_image=gimg)
print("atomic-grads mode: cov3D", ps2[1].grad.cpu().numpy()[i][r, c], "means3D", ps2[0].grad.cpu().numpy()[i])
rasterizer.ATOMIC_GRADS = False
rasterizer.LEAN_LISTS = True
out3, st3, ps3 = _single_face_call(params, face, 256, dev, grad_image=gimg)
print("lean lists:        cov3D", ps3[1].grad.cpu().numpy()[i][r, c], "means3D", ps3[0].grad.cpu().numpy()[i], "tiles", int(st3.tensors()["tiles_touched"][0][i]))
