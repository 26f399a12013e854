import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import _lib, decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
g = [torch.tensor(cloud[k], device=dev)[None] for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(1, 3, device=dev)
_lib.lib()
for f in range(6):
    for rep in range(2):
        _lib.profile_enable(True)
        decoder.render_cuda(ext[f:f+1], K[f:f+1], near[f:f+1], far[f:f+1], (256, 256), bg, *g)
        torch.cuda.synchronize()
        prof = {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.profile_collect().items() if n}
        _lib.profile_enable(False)
    st = rasterizer.last_state()
    ts = st.tensors()["tile_start"].cpu().numpy().astype(np.int64)
    n = np.diff(ts)
    print("face", f, "L", st.num_rendered(), "cap", st.prm.max_instances, "tile len max", n.max(), ">2048:", (n > 2048).sum(), ">4096:", (n > 4096).sum(), ">16384:", (n > 16384).sum(), prof)
