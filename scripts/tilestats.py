import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
g = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
faces = decoder.render_cube_faces(torch.eye(4, device=dev), torch.tensor(0.1, device=dev), torch.tensor(10.0, device=dev), 256, torch.zeros(3, device=dev), *g)
st = rasterizer.last_state().tensors()
ts = st["tile_start"].cpu().numpy().astype(np.int64)
n = np.diff(ts)
maxc = st["tile_max_contrib"].cpu().numpy().astype(np.int64)
nc = st["n_contrib"].cpu().numpy().astype(np.int64)
print("tiles", len(n), "sum n", n.sum(), "max n", n.max(), "mean n", n.mean())
print("sum maxc", maxc.sum(), "max maxc", maxc.max(), "mean", maxc.mean(), "ratio", maxc.sum() / n.sum())
for v in range(6):
    sl = slice(v * 256, (v + 1) * 256)
    print("face", v, "n mean/max", n[sl].mean(), n[sl].max(), "maxc mean/max", maxc[sl].mean(), maxc[sl].max(), "pixel n_contrib mean", nc[v].mean())
order = np.argsort(-maxc)
print("top maxc", maxc[order[:10]], "their n", n[order[:10]])
print("pixel n_contrib mean overall", nc.mean(), "sum", nc.sum())
T = st["final_T"].cpu().numpy()
print("final_T < 1e-3 fraction", (T < 1e-3).mean(), "mean T", T.mean())
