import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
g = [torch.tensor(cloud[k], device=dev)[None] for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(1, 3, device=dev)
def run():
    for f in range(6):
        decoder.render_cuda(ext[f:f+1], K[f:f+1], near[f:f+1], far[f:f+1], (256, 256), bg, *g)
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); print("drop-in render_cuda, 6 faces: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
