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
st = rasterizer.last_state().tensors()
d = st["keys"].view(torch.int32)[: 2 * 1536].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
t0, t1 = d[0::2], d[1::2]
base = t0.min()
dur = (t1 - t0) / 100.0  # wall_clock64 = 100 MHz -> us
start = (t0 - base) / 100.0
end = (t1 - base) / 100.0
n = np.diff(st["tile_start"].cpu().numpy().astype(np.int64))
maxc = st["tile_max_contrib"].cpu().numpy()
print("kernel span us", end.max(), "block dur mean/median/max", dur.mean(), np.median(dur), dur.max())
print("start time pct [50,90,99,max]", np.percentile(start, [50, 90, 99, 100]))
o = np.argsort(-dur)[:12]
for i in o:
    print("tile", i, "face", i // 256, "dur", dur[i], "start", start[i], "n", n[i], "maxc", maxc[i])
print("corr(dur, n)", np.corrcoef(dur, n)[0, 1], "corr(dur,maxc)", np.corrcoef(dur, maxc)[0, 1])
h, e = np.histogram(dur, bins=10)
print("hist", h, e)
