#!/usr/bin/env python3
"""Soak test of the one-launch split composite (round 6: phase-2 workers waiting on device-coherent words inside k_render's launch):
N training steps on the 1 M surface-like cloud interleaved with the encoder-like one, every split step compared bit for bit with the
first one (images, final_T, n_contrib, all gradients), split_errors() == 0 throughout, and the mop-up kernel's share counted.
usage: soak_split.py [steps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, rasterizer, synthetic

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
clouds = {n: [torch.tensor(c[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
          for n, c in (("surface_like", synthetic.surface_like_cloud(512, 1024)), ("encoder_like", synthetic.encoder_like_cloud(512, 1024)))}
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
gt = torch.full((6, 3, 256, 256), 0.5, device=dev)


def run(name, split):
    ps = [p.clone().requires_grad_(True) for p in clouds[name]]
    faces, fm = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *ps, shared_campos=True, mse_target=gt, split_lists=split)
    st = rasterizer.last_state()
    fm.loss.backward()
    t = st.tensors()
    return [faces.detach().clone(), t["final_T"].clone(), t["n_contrib"].clone()] + [p.grad for p in ps], st


ref, st = run("surface_like", True)
torch.cuda.synchronize()
assert st.split_errors() == 0 and int(st.header()[5].item()) > 0, "the surface-like cloud must split"
bad = 0
t0 = time.time()
for i in range(steps):
    out, st = run("surface_like", True)
    if not all(torch.equal(a, b) for a, b in zip(ref, out)):
        bad += 1
    if st.split_errors() != 0:
        raise SystemExit(f"split error word {st.split_errors():#x} at step {i}")
    if i % 3 == 0:
        o2, _ = run("encoder_like", i % 2 == 0)      # another cloud in between, with and without the SPLIT kernel instances
        assert all(torch.isfinite(x.float()).all() for x in o2)
torch.cuda.synchronize()
print(f"{steps} split steps in {time.time() - t0:.1f} s: {bad} differed from the first one; split quadrants {int(st.header()[5].item())}, items {int(st.header()[6].item())}, error word 0")
sys.exit(1 if bad else 0)
