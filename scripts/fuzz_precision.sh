cd $GRAFT_REPO_ROOT
for leg in 0 1; do
  if [ $leg = 1 ]; then export S360_BWD_LEGACY=1; fi
  python - <<'P'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import test_gpu_fuzz as T
from helpers import settings_from_views, boundary_tensors
from oracle import oracle
from splatter360_amd import cameras, decoder, synthetic
gpu = torch.device("cuda:0")
for shared in (True, False):
  for seed in range(6):
    cloud, faces, pos, nears, bg, gimg, (n, h, w, v) = T._views_case(seed, shared)
    ext = torch.stack([cameras.cube_face_extrinsics(torch.from_numpy(synthetic.target_pano_pose(pos[i]))[None])[0, faces[i]] for i in range(v)]).to(gpu)
    K = cameras.cube_face_intrinsics(1)[0, :1].repeat(v, 1, 1).to(gpu)
    near = torch.tensor(nears, device=gpu); far = near * 100.0
    ps = [torch.tensor(cloud[k], device=gpu, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
    views = decoder.pack_camera_views(ext, K, near, far, torch.tensor(bg, device=gpu))
    imgs = decoder.render_views_fused(ext, K, near, far, (h, w), torch.tensor(bg, device=gpu), *ps, shared_campos=shared, views=views)
    imgs.backward(torch.tensor(gimg, device=gpu))
    want = [np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3, 25)), np.zeros((n,))]
    want32 = [np.zeros_like(x) for x in want]
    r, c = np.triu_indices(3)
    for i in range(v):
        S = settings_from_views(views, i, h, w)
        means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
        for dt, acc in ((np.float32, want32), (np.float64, want)):
            o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=dt)
            o.forward(); g = o.backward(gimg[i])
            acc[0] += S["scale"] * np.asarray(g["means3D"], np.float64)
            acc[1][:, r, c] += S["scale"] ** 2 * np.asarray(g["cov3D"], np.float64)
            acc[2] += np.asarray(g["shs"], np.float64).transpose(0, 2, 1)
            acc[3] += np.asarray(g["opacities"], np.float64).reshape(-1)
    out = []
    for p, ref, o32 in zip(ps, want, want32):
        scale = np.abs(ref).max() + 1e-30
        out.append("%.1e/%.1e" % (np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / scale, np.abs(o32 - ref).max() / scale))
    print(os.environ.get("S360_BWD_LEGACY", "0"), shared, seed, n, v, " ".join(out), flush=True)
P
done
