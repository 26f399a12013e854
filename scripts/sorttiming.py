import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
g = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
for _ in range(3):
    faces = decoder.render_cube_faces(torch.eye(4, device=dev), torch.tensor(0.1, device=dev), torch.tensor(10.0, device=dev), 256, torch.zeros(3, device=dev), *g)
torch.cuda.synchronize()
stt = rasterizer.last_state()
st = stt.tensors()
lay = stt.layout
cur = stt._arr(lay.tile_cursor, 1536, torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
n = np.diff(st["tile_start"].cpu().numpy().astype(np.int64))
dur = cur / 100.0
for lo, hi in ((0, 1024), (1024, 2048), (2048, 4096), (4096, 8192), (8192, 16384)):
    sel = (n > lo) & (n <= hi)
    if sel.any():
        print(f"n in ({lo},{hi}]: tiles {sel.sum()}, block dur mean {dur[sel].mean():.1f} max {dur[sel].max():.1f} us")
