import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
g = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
faces = decoder.render_cube_faces(torch.eye(4, device=dev), torch.tensor(0.1, device=dev), torch.tensor(10.0, device=dev), 256, torch.zeros(3, device=dev), *g)
torch.cuda.synchronize()
st = rasterizer.last_state()
k = st.header()[8:12].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
print("walked (entry,wave) pairs", k[3], "past cull (dense mode)", k[0], "with >=1 valid lane", k[1], "valid lanes", k[2], "lanes/valid-entry", k[2] / max(k[1], 1))
