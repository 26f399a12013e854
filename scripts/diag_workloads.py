"""Diagnostics of the bench workloads (gpurun): instance / survivor / contributing-pair counts and list statistics per cloud."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from splatter360_amd import decoder, rasterizer, synthetic  # noqa: E402

dev = torch.device("cuda:0")
G = 1 << 20
clouds = {"encoder_like": lambda: synthetic.encoder_like_cloud(512, 1024, seed=0),
          "uniform": lambda: synthetic.uniform_cloud(G, seed=0, extent=5.0),
          "surface_like": lambda: synthetic.surface_like_cloud(512, 1024, seed=0)}
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
for name, mk in clouds.items():
    c = mk()
    ps = [torch.tensor(c[k], device=dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
    faces = decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=dev), *ps, shared_campos=True)
    st = rasterizer.last_state()
    t = st.tensors()
    L = st.num_rendered()
    nc, nb = st.count_contributions()
    ts = t["tile_start"].to(torch.int64)
    tl = (ts[1:] - ts[:-1]).float()
    ncon = t["n_contrib"].float()
    lay = st.layout
    sc = st._arr(lay.surv_count, 6 * 256 * 4, torch.int32).float()
    tt = t["tiles_touched"].float()
    print(name, dict(L=L, contributing_pairs=nc, bwd_pairs=nb, surv_in_front=int(sc.sum()), tile_len_mean=float(tl.mean()), tile_len_max=float(tl.max()),
                     n_contrib_mean=float(ncon.mean()), n_contrib_p99=float(ncon.flatten().kthvalue(int(0.99 * ncon.numel())).values),
                     final_T_mean=float(t["final_T"].mean()), frac_T_below_1e3=float((t["final_T"] < 1e-3).float().mean()),
                     tt_max=float(tt.max()), tt_gt32=int((tt > 32).sum()), inst_in_gt32=float(tt[tt > 32].sum())))
