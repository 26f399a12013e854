"""Developer micro-benchmark (not the contract bench): fused six-face render of the encoder-like
1M cloud, forward and forward+backward, timed with HIP events."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from splatter360_amd import decoder, rasterizer, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--pano-h", type=int, default=512)
ap.add_argument("--face", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bwd", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
t0 = time.time()
cloud = synthetic.encoder_like_cloud(a.pano_h, a.pano_h * 2)
print("cloud gen %.1fs G=%d" % (time.time() - t0, cloud["means"].shape[0]), flush=True)
g = {k: torch.tensor(v, device=dev) for k, v in cloud.items()}
pose = torch.eye(4, device=dev)
near, far = torch.tensor(0.1, device=dev), torch.tensor(10.0, device=dev)
bg = torch.zeros(3, device=dev)
G = g["means"].shape[0]


def fwd(req=False):
    ins = [g["means"], g["covariances"], g["harmonics"], g["opacities"]]
    if req:
        ins = [x.clone().requires_grad_(True) for x in ins]
    faces = decoder.render_cube_faces(pose, near, far, a.face, bg, *ins, check="lazy")
    return faces, ins


faces, _ = fwd()
torch.cuda.synchronize()
st = rasterizer.last_state()
print("num_rendered", st.num_rendered(), "overflow", st.overflowed(), "max tile list", int(st.header()[2]), flush=True)
tt = st.tensors()["tiles_touched"]
print("visible pairs per face", (tt > 0).sum(1).tolist())
print("faces mean", faces.mean().item(), "min", faces.min().item(), "max", faces.max().item())
for label, do_bwd in (("fwd", False),) + ((("fwd+bwd", True),) if a.bwd else ()):
    for _ in range(2):
        f, ins = fwd(do_bwd)
        if do_bwd:
            (f * f).mean().backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        f, ins = fwd(do_bwd)
        if do_bwd:
            (f * f).mean().backward()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{label}: {ms:.3f} ms/view  {G / ms / 1e3:.1f} Msplats/s", flush=True)
