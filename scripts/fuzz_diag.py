import sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(R), str(R / "tests")]
import numpy as np, torch
from oracle import oracle
from test_gpu_fuzz import _case
from test_gpu_parity import run_hip
seed = int(sys.argv[1])
S, means, cov6, shs, opac, colors, gimg, (n, h, w) = _case(seed)
print("case", n, h, w, "deg", S["sh_degree"], "sh" if shs is not None else "rgb")
res = {}
for dt in (np.float32, np.float64):
    o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, colors_precomp=colors, dtype=dt)
    o.forward(); res[dt] = o.backward(gimg)
hh = run_hip(S, means, cov6, shs, opac, torch.device("cuda:0"), colors=colors, grad_image=gimg)["grads"]
for k in ("means3D", "means2D", "cov3D", "opacities", "shs", "colors_precomp"):
    if res[np.float64].get(k) is None: continue
    ref = np.asarray(res[np.float64][k], np.float64).reshape(-1)
    a = np.asarray(res[np.float32][k], np.float64).reshape(-1); b = hh[k].astype(np.float64).reshape(-1)
    sc = np.abs(ref).max() + 1e-30
    i = np.abs(b - ref).argmax()
    print(f"{k:14s} oracle32-vs-64 {np.abs(a-ref).max()/sc:.2e}   hip-vs-64 {np.abs(b-ref).max()/sc:.2e}   hip-vs-oracle32 {np.abs(a-b).max()/sc:.2e}  worst idx {i} ref {ref[i]:.4e} hip {b[i]:.4e} o32 {a[i]:.4e}")
