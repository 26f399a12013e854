import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder
dev = torch.device("cuda:0")
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
eager = decoder.pack_camera_views(ext, K, near, far, bg)
s_in = [t.clone() for t in (ext, K, near, far, bg)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        out = decoder.pack_camera_views(*s_in)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = decoder.pack_camera_views(*s_in)
    print("capture ok")
except Exception as e:
    print("capture failed:", repr(e)[:400]); sys.exit(0)
# new inputs
pose = torch.eye(4, device=dev); pose[:3, 3] = torch.tensor([0.3, -0.2, 0.1], device=dev)
ext2, K2, near2, far2 = decoder.cube_cameras(pose, 0.1, 10.0)
want = decoder.pack_camera_views(ext2, K2, near2, far2, bg)
for d, s in zip(s_in, (ext2, K2, near2, far2, bg)):
    d.copy_(s)
g.replay()
torch.cuda.synchronize()
print("bit-equal to eager:", torch.equal(out, want), (out - want).abs().max().item())
def run():
    torch._foreach_copy_(s_in, [ext2, K2, near2, far2, bg])
    g.replay()
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): run()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"graph glue: host {1e3*(t1-t0)/50:.3f} ms, total {1e3*(t2-t0)/50:.3f} ms")
