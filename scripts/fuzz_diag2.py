"""Diagnostic: where does the gradient error of one fuzz case sit? (python scripts/fuzz_diag2.py SEED SHARED)"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import test_gpu_fuzz as T
from helpers import settings_from_views, boundary_tensors
from oracle import oracle
from splatter360_amd import cameras, decoder, synthetic, rasterizer
seed, shared = int(sys.argv[1]), bool(int(sys.argv[2]))
gpu = torch.device("cuda:0")
cloud, faces, pos, nears, bg, gimg, (n, h, w, v) = T._views_case(seed, shared)
ext = torch.stack([cameras.cube_face_extrinsics(torch.from_numpy(synthetic.target_pano_pose(pos[i]))[None])[0, faces[i]] for i in range(v)]).to(gpu)
K = cameras.cube_face_intrinsics(1)[0, :1].repeat(v, 1, 1).to(gpu)
near = torch.tensor(nears, device=gpu); far = near * 100.0
ps = [torch.tensor(cloud[k], device=gpu, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
views = decoder.pack_camera_views(ext, K, near, far, torch.tensor(bg, device=gpu))
imgs = decoder.render_views_fused(ext, K, near, far, (h, w), torch.tensor(bg, device=gpu), *ps, shared_campos=shared, views=views)
st = rasterizer.last_state().tensors()
imgs.backward(torch.tensor(gimg, device=gpu))
want = np.zeros((n, 3)); want32 = np.zeros((n, 3))
print("n", n, "hw", h, w, "v", v)
for i in range(v):
    S = settings_from_views(views, i, h, w)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    for dt, acc in ((np.float32, want32), (np.float64, want)):
        o = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=dt)
        f = o.forward(); g = o.backward(gimg[i])
        acc += S["scale"] * np.asarray(g["means3D"], np.float64)
        if dt == np.float32:
            d = np.abs(imgs[i].detach().cpu().numpy() - f["image"])
            nc = st["n_contrib"][i].cpu().numpy().astype(np.uint32)
            print("view", i, "img mean %.2e max %.2e" % (d.mean(), d.max()), "n_contrib mismatches", int((nc != f["n_contrib"]).sum()), "of", nc.size,
                  "tiles_touched equal", np.array_equal(st["tiles_touched"][i].cpu().numpy().astype(np.uint32), f["tiles_touched"]))
got = ps[0].grad.cpu().numpy().astype(np.float64)
scale = np.abs(want).max()
eh = np.abs(got - want).max(1) / scale; e32 = np.abs(want32 - want).max(1) / scale
idx = np.argsort(-eh)[:6]
for j in idx:
    ev = np.linalg.eigvalsh(cloud["covariances"][j].astype(np.float64))
    print("g", j, "e_hip %.2e e_o32 %.2e" % (eh[j], e32[j]), "|grad| %.2e" % np.abs(want[j]).max(), "cov eig", ev, "op %.3f" % cloud["opacities"][j], "|mean|", np.linalg.norm(cloud["means"][j]))
print("max |want| at", int(np.abs(want).max(1).argmax()), "scale %.3e" % scale, "median |grad| %.2e" % np.median(np.abs(want).max(1)))
# raster-level view: per-view screen-space gradient of the worst Gaussian (HIP single-view calls vs oracle)
j = int(idx[0])
for i in range(v):
    S = settings_from_views(views, i, h, w)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, dtype=np.float64); o64.forward(); g64 = o64.backward(gimg[i])
    o32 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs); f32 = o32.forward(); g32 = o32.backward(gimg[i])
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=gpu, requires_grad=True)
    m, c, s_, op_ = t(means), t(cov6), t(shs), t(opac)
    m2 = torch.zeros_like(m, requires_grad=True)
    img, _ = rasterizer.rasterize_views(m, c, op_, s_, None, views=torch.cat([views[i:i+1, :40], torch.ones(1, 1, device=gpu), views[i:i+1, 41:]], 1).contiguous(),
                                        image_height=h, image_width=w, sh_degree=4, shared_campos=True, means2D=m2)
    img.backward(torch.tensor(gimg[i], device=gpu)[None])
    for name, hv in (("means2D", m2.grad), ("means3D", m.grad), ("cov3D", c.grad), ("opac", op_.grad)):
        key = {"means2D": "means2D", "means3D": "means3D", "cov3D": "cov3D", "opac": "opacities"}[name]
        a = hv[j].cpu().numpy().astype(np.float64).reshape(-1); b64 = np.asarray(g64[key][j], np.float64).reshape(-1); b32 = np.asarray(g32[key][j], np.float64).reshape(-1)
        sc = np.abs(b64).max() + 1e-30
        print("view", i, name, "rad", int(f32["radii"][j]), "hip_rel %.2e o32_rel %.2e |g| %.2e" % (np.abs(a - b64).max() / sc, np.abs(b32 - b64).max() / sc, sc))
