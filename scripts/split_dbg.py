import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.surface_like_cloud(512, 1024)
params = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for it in range(3):
        f = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *params, shared_campos=True, split_lists=True)
        st = rasterizer.last_state()
        torch.cuda.synchronize()
        h = st.header().cpu().tolist()
        nt = 1536
        q = st._arr(st.layout.seg_arrive2, nt * 4 + 1, torch.int32).cpu()
        a1 = st._arr(st.layout.seg_arrive, nt * 4, torch.int32).cpu()
        fl = st._arr(st.layout.seg_flag, nt * 4, torch.int32).cpu()
        print("iter", it, "header[0:14]", h[:14], "queue", int(q[-1]), "max_segments", st.prm.max_segments, "split quads", int((fl == 1).sum()),
              "arrive1 sum", int(a1.sum()), "arrive2 sum", int(q[:-1].sum()), "finite", bool(torch.isfinite(f).all()), flush=True)
