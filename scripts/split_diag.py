#!/usr/bin/env python3
"""Diagnostic of S360_FLAG_SPLIT_LISTS on one cloud: the fused six-face training step with splitting on and off — how many quadrants
split, how far the images / final_T / n_contrib / gradients are apart, and the per-kernel times of both (HIP events).
usage: split_diag.py [surface_like|encoder_like|uniform] [steps]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import _lib, decoder, rasterizer, synthetic

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "surface_like"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
FW = 1024 if name.endswith("_16m") else 512 if name.endswith("_4m") else 256
cloud = {"encoder_like": lambda: synthetic.encoder_like_cloud(512, 1024), "surface_like": lambda: synthetic.surface_like_cloud(512, 1024),
         "surface_like_4m": lambda: synthetic.surface_like_cloud(1024, 2048), "surface_like_16m": lambda: synthetic.surface_like_cloud(2048, 4096), "encoder_like_4m": lambda: synthetic.encoder_like_cloud(1024, 2048),
         "uniform": lambda: synthetic.uniform_cloud(1 << 20, seed=0, extent=5.0)}[name]()
params = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
gt = torch.full((6, 3, FW, FW), 0.5, device=dev)


def run(split, lean=True):
    ps = [p.clone().requires_grad_(True) for p in params]
    faces, fm = decoder.render_views_fused(ext, K, near, far, (FW, FW), bg, *ps, shared_campos=True, mse_target=gt, split_lists=split, lean=lean)
    st = rasterizer.last_state()
    fm.loss.backward()
    torch.cuda.synchronize()
    t = st.tensors()
    return dict(img=faces.detach(), T=t["final_T"].clone(), nc=t["n_contrib"].clone(), grads=[p.grad for p in ps], loss=fm.loss.detach().clone(),
                nsplit=int(st.header()[5].item()), L=st.num_rendered(), flag=t["seg_flag"].clone())


def timed(split):
    for _ in range(2):
        run(split)
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        run(split)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = {n: round(ms / c * 1e3, 1) for n, (ms, c) in _lib.profile_collect().items() if c}
    _lib.profile_enable(False)
    return dt, k


a, b = run(False), run(True)
b2 = run(True)
print(name, "instances", a["L"], "split quadrants", b["nsplit"], "(off:", a["nsplit"], ")")
d = (a["img"] - b["img"]).abs()
print("image |diff| max %.3e mean %.3e  pixels > 1e-5: %d of %d" % (d.max().item(), d.mean().item(), int((d.amax(1) > 1e-5).sum()), d[:, 0].numel()))
print("final_T |diff| max %.3e" % (a["T"] - b["T"]).abs().max().item(), " n_contrib mismatches", int((a["nc"] != b["nc"]).sum()))
print("loss", a["loss"].item(), b["loss"].item())
for n, x, y in zip(("means", "cov", "sh", "opac"), a["grads"], b["grads"]):
    print("grad", n, "rel max diff %.3e" % ((x - y).abs().max() / (x.abs().max() + 1e-30)).item(), "nan", bool(torch.isnan(y).any()))
print("deterministic:", all(torch.equal(x, y) for x, y in zip([b["img"], b["T"]] + b["grads"], [b2["img"], b2["T"]] + b2["grads"])))
if b["nsplit"]:
    fl = b["flag"].view(6, -1, 4)
    print("segment work items", int(rasterizer.last_state().header()[6].item()), "split errors", rasterizer.last_state().split_errors())
    print("split quadrants per face:", [int((fl[f] == 1).sum()) for f in range(6)])
    bad = (a["nc"] != b["nc"]).nonzero()
    print("first n_contrib mismatches (view,y,x):", bad[:5].tolist(), [(int(a["nc"][tuple(i)]), int(b["nc"][tuple(i)])) for i in bad[:5]])
for split in (False, True):
    dt, k = timed(split)
    print("split", split, "step wall ms %.3f" % (dt * 1e3), "kernels us", k)
