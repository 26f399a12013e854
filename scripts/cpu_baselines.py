"""CPU baselines of SURVEY.md 8(d): the PyTorch-CPU restatement (oracle/torch_ref.py) and the C oracle with
OpenMP (oracle/s360_oracle.c) on BASELINE config 0 (10 000 Gaussians, 256x128 ERP = six 64x64 faces), plus the
C oracle on the headline cloud (1 048 576 Gaussians, six 256x256 faces).  fwd+bwd, L2 seed.  Test infrastructure
timing only — nothing here is on the product path."""
import os, sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(R), str(R / "tests")]
import numpy as np, torch
from helpers import boundary_tensors, face_settings
from oracle import oracle, torch_ref
from splatter360_amd import synthetic

cores = os.cpu_count() or 1
torch.set_num_threads(cores)
oracle.set_parallel_backward(True)


def run_oracle(cloud, fw):
    t0 = time.time()
    for face in range(6):
        S = face_settings(face, fw, fw)
        means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
        o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs)
        f = o.forward()
        o.backward((2.0 / f["image"].size) * (f["image"] - 0.5))
    return time.time() - t0


def run_torch(cloud, fw):
    t0 = time.time()
    for face in range(6):
        S = face_settings(face, fw, fw)
        means, cov6, shs, opac = [torch.tensor(a, requires_grad=True) for a in boundary_tensors(cloud, S["scale"])]
        img = torch_ref.render(S, means, cov6, opac, shs=shs)
        ((img - 0.5) ** 2).mean().backward()
    return time.time() - t0


small = synthetic.uniform_cloud(10_000, seed=0, extent=3.0, scale_range=(0.02, 0.3))
g = 10_000
dt = run_oracle(small, 64); print(f"C oracle (OpenMP, {cores} cores), config 0: {dt:.2f} s/ERP view fwd+bwd = {g / dt / 1e6:.4f} Msplats/s")
dt = run_torch(small, 64); print(f"PyTorch-CPU restatement ({cores} threads), config 0: {dt:.2f} s/ERP view fwd+bwd = {g / dt / 1e6:.4f} Msplats/s")
if "--big" in sys.argv:
    big = synthetic.encoder_like_cloud(512, 1024, seed=0)
    g = big["means"].shape[0]
    dt = run_oracle(big, 256); print(f"C oracle (OpenMP, {cores} cores), 1M Gaussians / six 256x256 faces: {dt:.1f} s = {g / dt / 1e6:.4f} Msplats/s")
